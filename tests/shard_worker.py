"""One rank of a two-process run of ONE alignment on one GPU (tests/test_gpu_align.py::test_sharded_contexts): both ranks hold
the same genomes and make the same calls; mauve_set_shard deals out the independent units; the result must equal a single
context's.  usage: python -m tests.shard_worker <rank> <world> <port> <out.npz>"""
import os
import sys

import numpy as np


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from mauvealigner_amd import _lib, synth, parallel
    ctx = _lib.Context(0)
    res = {}
    calls = {"n": 0, "bytes": 0}
    inner = parallel.make_allgather(dist)

    def counted(payload):
        calls["n"] += 1; calls["bytes"] += len(payload)
        return inner(payload)
    # recursion gaps (a C5-shaped pair), with the chains coming back from the device
    gs = synth.make_config("C5", scale=0.04)
    ctx.set_genomes(gs)
    ctx.set_shard(rank, world, counted)
    r = ctx.align(_lib.default_params())
    res["c5_exchanges"] = np.array([calls["n"], calls["bytes"]]); calls["n"] = calls["bytes"] = 0
    for k in ("anchor_start", "anchor_length", "cols", "col_off", "dp_score", "left", "right"):
        res["c5_" + k] = r[k]
    # guide tree pairs + node intervals (progressive path)
    gs = synth.make_config("C4", scale=0.08)
    ctx.set_genomes(gs)
    ctx.set_shard(rank, world, counted)
    r = ctx.progressive_align(_lib.default_params())
    res["c4_exchanges"] = np.array([calls["n"], calls["bytes"]])
    for k in ("cols", "col_off", "dp_score", "left", "right", "reverse", "dist"):
        res["c4_" + k] = r[k]
    res["c4_tree"] = np.stack(r["tree"])
    ctx.set_shard(0, 1, None)
    np.savez(out, **res)
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
