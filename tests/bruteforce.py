"""Independent pure-Python statements of the frozen definitions for tiny inputs.

Nothing here shares code with oracle/ or with the product; it works on strings."""
import itertools

COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def pattern_offsets(pattern):
    s = bin(pattern)[2:]
    return [i for i, ch in enumerate(s) if ch == "1"], len(s)


def masked(seq, p, offs):
    return "".join(seq[p + t] for t in offs)


def rc(s):
    return "".join(COMP[c] for c in reversed(s))


def brute_matches(seqs, pattern, mode="mem", mask=0, extend=True, valid=None):
    """seqs: list of ACGT strings.  Returns a set of (length, (signed 1-based starts...)).
    valid: optional per-genome list of (lo, hi) 1-based inclusive intervals; a window counts only if it lies
    inside one of them."""
    offs, span = pattern_offsets(pattern)
    N = len(seqs)
    fw = [[masked(s, p, offs) for p in range(len(s) - span + 1)] for s in seqs]

    def wvalid(g, p):
        return valid is None or any(lo - 1 <= p and p + span <= hi for lo, hi in valid[g])
    groups = {}
    for g in range(N):
        for p, f in enumerate(fw[g]):
            if not wvalid(g, p):
                continue
            r = rc(f)
            canon, strand = (r, 1) if r < f else (f, 0)
            groups.setdefault(canon, []).append((g, p, strand))
    out = set()
    hit_at = {}
    hits = []
    for canon, occ in groups.items():
        cnt = {}
        for g, p, s in occ:
            cnt[g] = cnt.get(g, 0) + 1
        if mode == "mem" and any(c > 1 for c in cnt.values()):
            continue
        comps = sorted(g for g, c in cnt.items() if c == 1)
        if len(comps) < 2:
            continue
        m = sum(1 << g for g in comps)
        if mask and m != mask:
            continue
        pos = {g: (p, s) for g, p, s in occ if cnt[g] == 1}
        hits.append((m, comps, pos))
        hit_at[(comps[0], pos[comps[0]][0])] = m

    def agree(comps, pos, k):
        a = comps[0]
        pa, sa = pos[a]
        qa = pa + k
        if qa < 0 or qa >= len(fw[a]) or not wvalid(a, qa):
            return False
        fa = fw[a][qa]
        for g in comps[1:]:
            pg, sg = pos[g]
            o = sg ^ sa
            qg = pg - k if o else pg + k
            if qg < 0 or qg >= len(fw[g]) or not wvalid(g, qg):
                return False
            f = fw[g][qg]
            if o:
                f = rc(f)
            if f != fa:
                return False
        return True

    for m, comps, pos in hits:
        a = comps[0]
        if not extend:
            klo = khi = 0
        else:
            ks = {0}
            cur = 0
            while True:
                nxt = [cur - d for d in range(1, span + 1) if agree(comps, pos, cur - d)]
                if not nxt:
                    break
                cur = nxt[0]
                ks.add(cur)
            klo = cur
            cur = 0
            while True:
                nxt = [cur + d for d in range(1, span + 1) if agree(comps, pos, cur + d)]
                if not nxt:
                    break
                cur = nxt[0]
            khi = cur
        starts = []
        for g in range(N):
            if g not in pos or g not in comps:
                starts.append(0)
                continue
            pg, sg = pos[g]
            o = sg ^ pos[a][1]
            starts.append(-(pg - khi + 1) if o else (pg + klo + 1))
        out.add((khi - klo + span, tuple(starts)))
    return out


def score_path(ops, cnt, k_rows, seq, matrix, go, ge):
    """Score an ops string (1 = column vs gap, 2 = base vs gap, 3 = aligned) under DESIGN.md S7."""
    i = j = 0
    total = 0
    prev = 0
    for op in ops:
        if op == 3:
            c = cnt[i]
            total += sum(c[a] * matrix[a][seq[j]] for a in range(4))
            i += 1
            j += 1
        elif op == 1:
            r = sum(cnt[i])
            total += (ge if prev == 1 else go) * r
            i += 1
        else:
            total += (ge if prev == 2 else go) * k_rows
            j += 1
        prev = op
    assert i == len(cnt) and j == len(seq)
    return total


def brute_best_score(cnt, k_rows, seq, matrix, go, ge):
    m, n = len(cnt), len(seq)
    best = None

    def rec(i, j, ops):
        nonlocal best
        if i == m and j == n:
            s = score_path(ops, cnt, k_rows, seq, matrix, go, ge)
            if best is None or s > best:
                best = s
            return
        if i < m and j < n:
            rec(i + 1, j + 1, ops + [3])
        if i < m:
            rec(i + 1, j, ops + [1])
        if j < n:
            rec(i, j + 1, ops + [2])
    rec(0, 0, [])
    return best
