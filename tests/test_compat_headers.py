"""The libMems-shaped C++ mirror (include/libMems) compiles with plain g++ against the C-ABI, its host-only
classes behave like the in-tree call sites expect, and -- on the GPU box -- the example call site
(examples/mauve_hip_align.cpp, shaped like mauveAligner.cpp:453-760) produces the same XMFA as mauve_align."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from mauvealigner_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MATCH_UNIT = r'''
#include <cassert>
#include <sstream>
#include "libMems/Match.h"
#include "libMems/MatchList.h"
#include "libMems/MemHash.h"
#include "libMems/HipFinders.h"
#include "libMems/PairwiseScoringScheme.h"
#include "libMems/SortedMerList.h"
#include "libGenome/gnSequence.h"
using namespace mems;
int main() {
    Match m(3);
    m.SetLength(100); m.SetStart(0, 11); m.SetStart(1, -501); m.SetStart(2, NO_MATCH);
    assert(m.SeqCount() == 3 && m.Multiplicity() == 2 && m.FirstStart() == 0);
    assert(m.LeftEnd(1) == 501 && m.RightEnd(1) == 600 && m.Orientation(1) == AbstractMatch::reverse);
    assert(m.Orientation(2) == AbstractMatch::undefined && m[0] == 11);
    m.CropStart(10);            /* first columns: forward start moves, reverse left end stays */
    assert(m.Length() == 90 && m.Start(0) == 21 && m.Start(1) == -501 && m.RightEnd(1) == 590);
    m.CropEnd(5);               /* last columns: reverse left end moves */
    assert(m.Length() == 85 && m.Start(0) == 21 && m.Start(1) == -506);
    m.CropLeft(5, 1);           /* left side of a reverse component = last columns */
    assert(m.Length() == 80 && m.Start(1) == -511 && m.Start(0) == 21);
    m.CropRight(5, 0);
    assert(m.Length() == 75 && m.Start(0) == 21 && m.Start(1) == -516);
    Match *c = m.Copy(); c->Invert(); assert(c->Start(0) == -21 && c->Start(1) == 516); c->Free();
    std::ostringstream os; os << m; assert(os.str() == "75\t21\t-516\t0");
    MatchList ml; ml.push_back(m.Copy()); Match full(3); full.SetLength(5); full.SetStart(0,1); full.SetStart(1,2); full.SetStart(2,3);
    ml.push_back(full.Copy()); ml.MultiplicityFilter(3); assert(ml.size() == 1 && ml[0]->Length() == 5); ml.Clear();
    assert(getSeedLength(getSeed(15, 0)) == 21 && getDefaultSeedWeight(5000000) == 15 && getSeed(15, SOLID_SEED) == 0x7fff);
    genome::gnSequence s("ACGTACGT"); assert(s.length() == 8 && s.ToString(3, 2) == "CGT" && s.ToString() == "ACGTACGT");
    PairwiseScoringScheme pss; assert(pss.gap_open == -400 && pss.gap_extend == -30 && pss.matrix[0][0] == 91);
    HipUniqueMatchFinder umf; MatchFinder *cl = umf.Clone(); delete cl;
    /* progressiveMauve.cpp:199-224: pattern text and default .sslist names */
    assert(getPatternText(getSeed(15, 0)) == "111011010111010110111" && getPatternText(getSeed(5, SOLID_SEED)) == "11111");
    std::vector<std::string> fn; fn.push_back("a.fa"); fn.push_back("b.gbk"); std::vector<std::string> sn;
    getDefaultSmlFileNames(fn, sn, 15, 0);
    assert(sn.size() == 2 && sn[0] == "a.fa.111011010111010110111.sslist" && sn[1] == "b.gbk.111011010111010110111.sslist");
    return 0;
}
'''


IO_UNIT = r'''
// usage: io_unit in.xmfa out.xmfa out.mln out2.xmfa seq0.fa seq1.fa ...
#include <cassert>
#include <fstream>
#include <sstream>
#include "libMems/IntervalList.h"
#include "libMems/MatchList.h"
using namespace mems;
int main(int argc, char **argv) {
    IntervalList il;
    { std::ifstream in(argv[1]); il.ReadStandardAlignment(in); }
    for (int i = 5; i < argc; i++) { genome::gnSequence *s = new genome::gnSequence(); s->LoadSource(argv[i]); il.seq_table.push_back(s); }
    assert(il.seq_filename.size() == il.seq_table.size());
    { std::ofstream out(argv[2]); il.WriteStandardAlignment(out); }           // XMFA -> XMFA
    { std::ofstream out(argv[3]); il.WriteList(out); }                        // -> .mln
    IntervalList il2;
    { std::ifstream in(argv[3]); il2.ReadList(in); }
    il2.seq_table = il.seq_table;
    assert(il2.size() == il.size() && il2.seq_filename == il.seq_filename);
    for (size_t i = 0; i < il.size(); i++) {
        assert(il2[i].Columns() == il[i].Columns());
        for (uint g = 0; g < il[i].SeqCount(); g++) assert(il2[i].Start(g) == il[i].Start(g) && il2[i].Length(g) == il[i].Length(g));
    }
    { std::ofstream out(argv[4]); il2.WriteStandardAlignment(out); }          // .mln -> XMFA
    // .mums round trip
    MatchList ml; ml.seq_table = il.seq_table; ml.seq_filename = il.seq_filename;
    for (size_t i = 0; i < il.size(); i++) {
        Match m((uint)il.seq_table.size()); m.SetLength(il[i].AlignmentLength());
        for (uint g = 0; g < il[i].SeqCount(); g++) m.SetStart(g, il[i].Start(g));
        ml.push_back(m.Copy());
    }
    std::stringstream ss; WriteList(ml, ss);
    MatchList back; ReadList(back, ss);
    assert(back.size() == ml.size() && back.seq_filename == ml.seq_filename);
    for (size_t i = 0; i < ml.size(); i++) {
        assert(back[i]->Length() == ml[i]->Length());
        for (uint g = 0; g < ml[i]->SeqCount(); g++) assert(back[i]->Start(g) == ml[i]->Start(g));
    }
    back.seq_table = ml.seq_table;            // ReadList leaves loading the sequences to the caller
    std::stringstream s2; WriteList(back, s2);
    assert(s2.str() == ss.str());
    ml.Clear(); back.Clear();
    return 0;
}
'''


SML_UNIT = r'''
// usage: sml_unit seq.fa  -- device-built sorted mer list -> .sslist -> reload (uniqueMerCount.cpp:30-39)
#include <cassert>
#include <iostream>
#include "libMems/MatchList.h"
#include "libMems/DNAFileSML.h"
using namespace mems;
int main(int argc, char **argv) {
    MatchList ml;
    genome::gnSequence *s = new genome::gnSequence(); s->LoadSource(argv[1]);
    ml.seq_table.push_back(s); ml.seq_filename.push_back(argv[1]);
    ml.CreateMemorySMLs(11, nullptr, 0);
    std::vector<std::string> names; getDefaultSmlFileNames(ml.seq_filename, names, 11, 0);
    ml.sml_table[0]->WriteFile(names[0]);
    DNAFileSML back; back.LoadFile(names[0]);
    assert(back.Seed() == ml.sml_table[0]->Seed() && back.Length() == ml.sml_table[0]->Length());
    for (gnSeqI i = 0; i < back.Length(); i++) {
        assert(back.SortedMer(i) == ml.sml_table[0]->SortedMer(i) && back.SortedPosition(i) == ml.sml_table[0]->SortedPosition(i));
        assert(back.GetMer(i) == ml.sml_table[0]->GetMer(i));
    }
    std::cout << back.UniqueMerCount() << std::endl;
    return 0;
}
'''


def _compile(src_text, out, extra=()):
    src = out + ".cpp"
    with open(src, "w") as f:
        f.write(src_text)
    cmd = ["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", out, src,
           "-L" + os.path.join(ROOT, "mauvealigner_amd"), "-lmauve_hip",
           "-Wl,-rpath," + os.path.join(ROOT, "mauvealigner_amd")] + list(extra)
    subprocess.check_call(cmd)


def test_mirror_compiles_and_host_classes_work():
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "match_unit")
        _compile(MATCH_UNIT, exe)
        subprocess.check_call([exe])


@pytest.mark.parametrize("name", ["g2x2k", "g3x5k_inv", "g5x3k_unique", "g4x3k_tree"])
def test_stage_seam_formats_round_trip(name):
    """the text formats at the stage seams, host only.  A committed golden XMFA goes through
    ReadStandardAlignment -> WriteStandardAlignment and through WriteList(.mln) -> ReadList -> WriteStandardAlignment
    and must come back byte for byte; the match list goes through WriteList/ReadList(.mums)."""
    golden = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(golden, name + ".npz"))
    want = open(os.path.join(golden, name + ".xmfa")).read()
    names = [ln.split("\t", 1)[1] for ln in want.splitlines() if ln.startswith("#Sequence") and "File\t" in ln]
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "io_unit")
        _compile(IO_UNIT, exe)
        fas = []
        for g in range(int(z["nseq"])):
            p = os.path.join(td, "s%d.fa" % g)
            with open(p, "w") as f:
                f.write(">s%d\n%s\n" % (g, synth.to_ascii(z["genome%d" % g]).decode()))
            fas.append(p)
        assert len(names) == len(fas)
        out1, mln, out2 = (os.path.join(td, x) for x in ("a.xmfa", "a.mln", "b.xmfa"))
        subprocess.check_call([exe, os.path.join(golden, name + ".xmfa"), out1, mln, out2] + fas)
        assert open(out1).read() == want
        assert open(out2).read() == want
        assert open(mln).read().startswith("FormatVersion\tmauve_hip_mln_1\nSequenceCount\t%d\n" % len(fas))


REFERENCE = "/root/reference/src"

REF_UNIT = r'''
// The in-tree plug-ins, included from where they lie, compiled against the mirror -- unmodified.
#include "%(ref)s/UniqueMatchFinder.h"
#include "%(ref)s/SeedMatchEnumerator.h"
int main() {
    UniqueMatchFinder umf; mems::MatchFinder *c = umf.Clone(); delete c;
    SeedMatchEnumerator sme; mems::MatchFinder *d = sme.Clone(); delete d;
    return 0;
}
'''


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference tree is only present in the build container")
def test_reference_plugins_compile_unmodified():
    """src/UniqueMatchFinder.cpp and src/SeedMatchEnumerator.h compile and link, as they
    are, against -I include (virtual EnumerateMatches(IdmerList&), HashMatch, FindMatchSeeds, idmer and its
    comparators, Match / MatchProjectionAdapter, GetSar, seq_count, sar_table)."""
    inc = os.path.join(ROOT, "include")
    with tempfile.TemporaryDirectory() as td:
        obj = os.path.join(td, "umf.o")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + inc, "-c", os.path.join(REFERENCE, "UniqueMatchFinder.cpp"), "-o", obj])
        src = os.path.join(td, "ref_unit.cpp")
        with open(src, "w") as f:
            f.write(REF_UNIT % {"ref": REFERENCE})
        exe = os.path.join(td, "ref_unit")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + inc, src, obj, "-o", exe, "-L" + os.path.join(ROOT, "mauvealigner_amd"),
                               "-lmauve_hip", "-Wl,-rpath," + os.path.join(ROOT, "mauvealigner_amd")])
        subprocess.check_call([exe])        # construction and Clone only: nothing here needs a GPU


REFERENCE_TOOLS = [
    "UniqueMatchFinder.cpp", "addUnalignedIntervals.cpp", "backbone_global_to_local.cpp", "bbFilter.cpp", "calculateBackboneCoverage.cpp",
    "calculateCoverage.cpp", "coordinateTranslate.cpp", "countInPlaceInversions.cpp", "createBackboneMFA.cpp", "extractBackbone.cpp",
    "extractBackbone2.cpp", "extractSubalignments.cpp", "gappiness.cpp", "makeBadgerMatrix.cpp", "makeMc4Matrix.cpp", "mauveToXMFA.cpp",
    "mfa2xmfa.cpp", "scoreAlignment.cpp", "sortContigs.cpp", "stripGapColumns.cpp", "toEvoHighwayFormat.cpp", "toGrimmFormat.cpp",
    "toMultiFastA.cpp", "toRawSequence.cpp", "transposeCoordinates.cpp", "uniqueMerCount.cpp",
]


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree not present")
@pytest.mark.parametrize("tool", REFERENCE_TOOLS)
def test_reference_tools_compile_unmodified(tool):
    """The in-tree programs that need nothing but libMems / libGenome (no boost, no tree or annotation classes) compile
    unmodified, where they lie, against -I include: the data model, the LCB helpers, the stage-seam readers and writers,
    the backbone files and the libGenome sequence / FastA surface they use are all there under the names they use."""
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-w", "-I" + os.path.join(ROOT, "include"), os.path.join(REFERENCE, tool)])


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree not present")
def test_reference_mfa2xmfa_runs_on_the_mirror():
    """src/mfa2xmfa.cpp -- the in-tree twin of the XMFA writer -- built from its own source against the mirror's
    libGenome and RUN: the XMFA it writes is read by IntervalList::ReadStandardAlignment and written back byte for byte
    by WriteStandardAlignment (80-column wrap, deflines, header, terminator), and the FastA it writes through
    gnFASSource::Write is the ungapped input."""
    rng = np.random.default_rng(3)
    rows = []
    L = 437                                                   # several wrapped lines and a ragged last one
    for g in range(4):
        r = rng.choice(list("ACGT"), L)
        gaps = rng.random(L) < (0.05 + 0.1 * g)
        r[gaps] = "-"
        rows.append("".join(r)[:L - 7 * g])                   # shorter entries: the tool pads them with gaps
    with tempfile.TemporaryDirectory() as td:
        mfa = os.path.join(td, "in.mfa")
        with open(mfa, "w") as f:
            for g, r in enumerate(rows):
                f.write(">seq_%d some words\n" % g)
                for p in range(0, len(r), 61):
                    f.write(r[p:p + 61] + "\n")
        inc = os.path.join(ROOT, "include")
        tool = os.path.join(td, "mfa2xmfa")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", "-I" + inc, os.path.join(REFERENCE, "mfa2xmfa.cpp"), "-o", tool])
        xmfa, fa = os.path.join(td, "out.xmfa"), os.path.join(td, "un.fa")
        subprocess.check_call([tool, mfa, xmfa, fa])
        text = open(xmfa).read()
        assert text.startswith("#FormatVersion Mauve1\n#Sequence1File\t" + fa + "\n#Sequence1Entry\t1\n#Sequence1Format\tFastA\n")
        for g, r in enumerate(rows):
            assert "> %d:1-%d + seq_%d some words\n" % (g + 1, len(r.replace("-", "")), g) in text
        assert text.endswith("=\n")
        got = open(fa).read().split(">")[1:]
        assert ["".join(x.split("\n")[1:]) for x in got] == [r.replace("-", "") for r in rows]
        rt = os.path.join(td, "rt")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + inc, os.path.join(ROOT, "tests", "cpp", "xmfa_roundtrip.cpp"), "-o", rt,
                               "-L" + os.path.join(ROOT, "mauvealigner_amd"), "-lmauve_hip", "-Wl,-rpath," + os.path.join(ROOT, "mauvealigner_amd")])
        r = subprocess.run([rt, xmfa], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout == text, r.stderr


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree not present")
def test_oracle_xmfa_blocks_equal_what_mfa2xmfa_writes():
    """The oracle's XMFA writer against the reference's own formatting code: every all-forward block of an oracle
    alignment, handed to src/mfa2xmfa.cpp (built on the mirror's libGenome) as a multi-FastA of its gapped rows, comes
    back as the same block text -- deflines, 80-column wrap, terminator -- up to the coordinates, which mfa2xmfa counts
    from 1."""
    import re
    from oracle import pyoracle as O
    from mauvealigner_amd import synth
    gs = synth.make_config("C1", scale=0.05)
    names = ["genomeA.fa", "genomeB.fa"]
    text = O.align(gs, O.default_params(), names=names, want_xmfa=True)["xmfa"]
    body = text[text.index("> "):]
    blocks = [b for b in body.split("=\n") if b.strip()]
    checked = 0
    with tempfile.TemporaryDirectory() as td:
        tool = os.path.join(td, "mfa2xmfa")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", "-I" + os.path.join(ROOT, "include"), os.path.join(REFERENCE, "mfa2xmfa.cpp"), "-o", tool])
        for b in blocks:
            heads = re.findall(r"^> (\d+):(\d+)-(\d+) ([+-]) (.*)$", b, re.M)
            if len(heads) < 2 or any(h[3] == "-" for h in heads):
                continue
            parts = re.split(r"^> .*$", b, flags=re.M)[1:]
            rows = ["".join(p.split()) for p in parts]
            mfa = os.path.join(td, "b.mfa")
            with open(mfa, "w") as f:
                for h, r in zip(heads, rows):
                    f.write(">%s\n%s\n" % (h[4], r))
            out = os.path.join(td, "b.xmfa")
            subprocess.check_call([tool, mfa, out])
            want = re.sub(r"^> (\d+):(\d+)-(\d+) ", lambda m: "> %s:1-%d " % (m.group(1), int(m.group(3)) - int(m.group(2)) + 1), b, flags=re.M) + "=\n"
            # mfa2xmfa numbers the rows 1..k in file order; the oracle's block may skip absent genomes
            got = open(out).read()
            for k, h in enumerate(heads):
                got = got.replace("> %d:1-" % (k + 1), "> %s:1-" % h[0], 1) if str(k + 1) != h[0] else got
            assert got == want
            checked += 1
    assert checked >= 1


def test_data_model_host_classes():
    """GappedAlignment, CompactGappedAlignment, MatchProjectionAdapter, Interval::SetMatches / GetColumn, the LCB
    helpers (struct LCB, IdentifyBreakpoints, ComputeLCBs_v2, computeLCBAdjacencies_v2, EliminateOverlaps,
    transposeMatches, addUnalignedIntervals, readSubstitutionMatrix): host only, plain g++."""
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "model_test")
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "model_test.cpp"),
                               "-o", exe, "-L" + os.path.join(ROOT, "mauvealigner_amd"), "-lmauve_hip",
                               "-Wl,-rpath," + os.path.join(ROOT, "mauvealigner_amd")])
        r = subprocess.run([exe], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.strip() == "OK", r.stdout + r.stderr


@pytest.mark.gpu
def test_plugin_seams_on_the_device():
    """The two plugs of the reference (mauveAligner.cpp:585,674,698) honoured: a MatchFinder subclass with an
    EnumerateMatches of its own runs through the host callback path and gives what the in-kernel rule gives; a
    GappedAligner of the caller's own is called for every inter-anchor interval; Aligner::align chains the list it
    is handed (tests/cpp/plug_test.cpp)."""
    gs = synth.make_config("C3", scale=0.01)
    with tempfile.TemporaryDirectory() as td:
        paths = []
        for i, g in enumerate(gs[:3]):
            p = os.path.join(td, "g%d.fa" % i)
            with open(p, "w") as f:
                a = synth.to_ascii(g).decode()
                f.write(">g%d\n%s\n" % (i, a))
            paths.append(p)
        exe = os.path.join(td, "plug_test")
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "plug_test.cpp"),
                               "-o", exe, "-L" + os.path.join(ROOT, "mauvealigner_amd"), "-lmauve_hip",
                               "-Wl,-rpath," + os.path.join(ROOT, "mauvealigner_amd")])
        r = subprocess.run([exe] + paths, capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr


def test_example_call_site_compiles():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-B"], stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(ROOT, "examples", "mauve_hip_align"))


@pytest.mark.gpu
@pytest.mark.parametrize("flag", [None, "-u", "-p"])
def test_example_matches_c_abi(flag):
    from mauvealigner_amd import _lib
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")], stdout=subprocess.DEVNULL)
    gs = synth.make_config("C3", scale=0.01)
    with tempfile.TemporaryDirectory() as td:
        paths = []
        for i, g in enumerate(gs):
            p = os.path.join(td, "g%d.fa" % i)
            with open(p, "w") as f:
                f.write(">g%d\n" % i)
                a = synth.to_ascii(g).decode()
                for k in range(0, len(a), 70):
                    f.write(a[k:k + 70] + "\n")
            paths.append(p)
        cmd = [os.path.join(ROOT, "examples", "mauve_hip_align")] + ([flag] if flag else []) + paths
        env = dict(os.environ, MAUVE_MUMS_OUT=os.path.join(td, "o.mums"), MAUVE_MLN_OUT=os.path.join(td, "o.mln"),
                   MAUVE_BACKBONE_OUT=os.path.join(td, "o.backbone"))
        out = subprocess.run(cmd, check=True, capture_output=True, text=True, env=env).stdout
        if flag != "-p":        # the seam files of the mauveAligner path: match list and interval list
            mums = open(env["MAUVE_MUMS_OUT"]).read().splitlines()
            assert mums[0] == "FormatVersion\t3" and mums[1] == "SequenceCount\t%d" % len(gs)
            n_m = int(mums[2 + 2 * len(gs)].split("\t")[1])
            assert n_m > 0 and len(mums) == 3 + 2 * len(gs) + n_m
            assert open(env["MAUVE_MLN_OUT"]).read().startswith("FormatVersion\tmauve_hip_mln_1\n")
        ctx = _lib.Context(0)
        try:
            ctx.set_genomes(gs)
            if flag == "-p":
                # the progressiveMauve call site's defaults: SP scoring, weight scaling on, both scales 0.5, refinement on
                r = ctx.progressive_align(_lib.default_progressive_params(), names=paths, want_xmfa=True)
            else:
                r = ctx.align(_lib.default_params(extend_lcbs=1), names=paths, want_xmfa=True)    # the call site passes lcb_extension = true
            if flag == "-p":        # applyBackbone in the example: homology pass (it rewrites the intervals that are written afterwards), then the
                                    # .bbcols rows are the segments of mauve_backbone on that alignment
                r = ctx.apply_homology(names=paths, want_xmfa=True)
                bb = ctx.backbone(island_gap=20)
                rows = [ln.split("\t") for ln in open(env["MAUVE_BACKBONE_OUT"] + ".bbcols").read().splitlines()]
                assert len(rows) == len(bb["seg_iv"]) > 0
                for row, iv, col, ln_, mask in zip(rows, bb["seg_iv"], bb["seg_col"], bb["seg_len"], bb["seg_mask"]):
                    assert [int(x) for x in row[:3]] == [int(iv), int(col), int(ln_)]
                    assert [int(x) for x in row[3:]] == [g for g in range(len(gs)) if int(mask) >> g & 1]
                head = open(env["MAUVE_BACKBONE_OUT"]).readline().rstrip("\n").split("\t")
                assert head == sum((["seq%d_leftend" % g, "seq%d_rightend" % g] for g in range(len(gs))), [])
        finally:
            ctx.close()
        assert out == r["xmfa"]


@pytest.mark.gpu
def test_sslist_cache_and_unique_mer_count():
    """a device-built sorted mer list written as <seq>.<pattern>.sslist, reloaded, and counted
    (uniqueMerCount.cpp:39) -- the count equals the oracle's number of distinct canonical mers."""
    from oracle import pyoracle as O
    g = synth.make_config("C1", scale=0.05)[0]
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "g.fa")
        with open(fa, "w") as f:
            f.write(">g\n%s\n" % synth.to_ascii(g).decode())
        exe = os.path.join(td, "sml_unit")
        _compile(SML_UNIT, exe)
        out = subprocess.run([exe, fa], check=True, capture_output=True, text=True).stdout
        assert os.path.exists(fa + "." + bin(O.get_seed(11, 0))[2:] + ".sslist")
        mer, _ = O.sorted_mer_list(g, O.get_seed(11, 0))
        assert int(out) == len(np.unique(mer >> 1))


def test_host_spin_pool():
    """csrc/workers.hpp (the helpers behind the host loops of mauve_align): every index exactly once, over many
    jobs, armed and disarmed, 1/2/4 threads.  Host-only C++, plain g++."""
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, "workers_test")
        subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "mauvealigner_amd", "csrc"),
                        os.path.join(ROOT, "tests", "cpp", "workers_test.cpp"), "-o", exe], check=True)
        r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and r.stdout.strip() == "OK", r.stdout + r.stderr
