// The plug-in seams of the reference, exercised on the device (usage: plug_test a.fa b.fa c.fa):
//  1. a MatchFinder subclass with its own EnumerateMatches (the shape of src/UniqueMatchFinder.cpp, written here
//     independently) -> host callback path -> same matches as the in-kernel rule (HipUniqueMatchFinder);
//  2. a MatchFinder subclass that builds its matches itself in HashMatch, no extension (the shape of
//     src/SeedMatchEnumerator.h) -> same matches as the device enumeration (HipSeedMatchEnumerator);
//  3. a GappedAligner of the caller's own installed with Aligner::SetGappedAligner is called once per interval and,
//     delegating to the built-in DP, reproduces the batched result;
//  4. Aligner::align chains the match list it is given: with half the matches removed the anchors change.
//  5. Aligner::SetPermutationOutput: a signed permutation per LCB set between the two weights, the last the aligned one;
//  6. ProgressiveAligner with an output / input guide tree file: the written tree read back gives the same alignment.
//  8. a seed family through one finder (progressiveMauve.cpp:510-546) and ProgressiveAligner::setUseSeedFamilies;
//  7. the backbone stage as applyBackbone calls it (detectBackbone with a BigGapsDetector, the .backbone / .bbcols writers and
//     readers) and simpleFindBackbone / simpleFindIslands / findIslandsBetweenLCBs: segment ends agree with the columns.
#include <cassert>
#include <fstream>
#include <iostream>
#include <map>
#include <set>
#include <sstream>
#include "libMems/mems_hip.h"
using namespace mems;

// rule: every sequence that holds the mer exactly once takes part; two or more must remain
class OnceOnlyFinder : public MemHash {
public:
    virtual OnceOnlyFinder *Clone() const { return new OnceOnlyFinder(*this); }
    size_t calls = 0;
protected:
    virtual boolean EnumerateMatches(IdmerList &match_list)
    {
        calls++;
        std::map<uint32, int> seen;
        for (const idmer &e : match_list) seen[e.id]++;
        IdmerList keep;
        for (const idmer &e : match_list) if (seen[e.id] == 1) keep.push_back(e);
        return keep.size() >= 2 ? HashMatch(keep) : true;
    }
};

// every repeated mer of ONE sequence becomes a match of seed length, components in position order, strands relative to the first
class RepeatLister : public MatchFinder {
public:
    virtual RepeatLister *Clone() const { return new RepeatLister(*this); }
    MatchList found;
protected:
    virtual boolean EnumerateMatches(IdmerList &l) { return HashMatch(l); }
    virtual boolean HashMatch(IdmerList &l)
    {
        l.sort(&idmer_position_lessthan);
        if (l.size() < 2 || l.size() > 1000) return true;
        Match m((uint)l.size());
        m.SetLength(GetSar(0)->SeedLength());
        const uint64 first_strand = l.front().mer & 1;
        uint k = 0;
        for (const idmer &e : l) { const int64 p1 = (int64)e.position + 1; m.SetStart(k++, (e.mer & 1) == first_strand ? p1 : -p1); }
        found.push_back(m.Copy());
        return true;
    }
};

class CountingAligner : public GappedAligner {
public:
    size_t aligned = 0, fast = 0;
    virtual bool CallMuscleFast(std::vector<std::string> &out, const std::vector<std::string> &in, int go, int ge)
    {
        fast++;
        return HipGappedAligner::getInterface().CallMuscleFast(out, in, go, ge);
    }
    virtual boolean Align(GappedAlignment &cr, AbstractMatch *l, AbstractMatch *r, std::vector<genome::gnSequence *> &seq_table)
    {
        aligned++;
        return GappedAligner::Align(cr, l, r, seq_table);
    }
};

static std::string xmfa(const IntervalList &il) { std::ostringstream os; il.WriteStandardAlignment(os); return os.str(); }
static std::multiset<std::string> rows(const MatchList &ml) { std::multiset<std::string> s; for (const Match *m : ml) { std::ostringstream os; os << *m; s.insert(os.str()); } return s; }

int main(int argc, char **argv)
{
    try {
        MatchList ml;
        for (int a = 1; a < argc; a++) { genome::gnSequence *s = new genome::gnSequence(); s->LoadSource(argv[a]); ml.seq_table.push_back(s); ml.seq_filename.push_back(argv[a]); }
        const uint N = (uint)ml.seq_table.size();
        ml.CreateMemorySMLs(11, nullptr, 0);
        // 1. host callback path == in-kernel rule
        MatchList dev; dev.seq_table = ml.seq_table; dev.sml_table = ml.sml_table; dev.seq_filename = ml.seq_filename;
        HipUniqueMatchFinder huf; huf.FindMatches(dev);
        MatchList cb; cb.seq_table = ml.seq_table; cb.sml_table = ml.sml_table; cb.seq_filename = ml.seq_filename;
        OnceOnlyFinder oof; oof.FindMatches(cb);
        assert(oof.calls > 100 && dev.size() > 50 && dev.size() == cb.size());
        for (size_t i = 0; i < dev.size(); i++) { std::ostringstream x, y; x << *dev[i]; y << *cb[i]; assert(x.str() == y.str()); }
        // ... and with the N-way mask of MaskedMemHash
        MatchList devm = dev; devm.clear(); MatchList cbm = cb; cbm.clear();
        { MaskedMemHash mmh; mmh.SetMask((1ull << N) - 1); mmh.FindMatches(devm); OnceOnlyFinder o2; o2.SetMask((1ull << N) - 1); o2.FindMatches(cbm); }
        assert(devm.size() > 10 && rows(devm) == rows(cbm));
        // 2. matches built on the host from the enumerated runs == the device enumeration
        MatchList one; one.seq_table.push_back(ml.seq_table[0]); one.sml_table.push_back(ml.sml_table[0]); one.seq_filename.push_back(ml.seq_filename[0]);
        MatchList devrep = one; HipSeedMatchEnumerator hse; hse.FindMatches(devrep);
        RepeatLister rl; rl.AddSequence(one.sml_table[0], one.seq_table[0]); rl.CreateMatches();
        assert(rows(devrep) == rows(rl.found));
        RepeatHash rh; MatchList rhl = one; rh.FindMatches(rhl); assert(rows(rhl) == rows(devrep));
        // 3. a GappedAligner of the caller's own
        const uint w = MatchList::GetDefaultMerSize(ml.seq_table);
        MatchList nway = ml; nway.clear(); nway.seed_pattern = getSeed((int)w, 0);
        { MatchList tmp; tmp.seq_table = ml.seq_table; tmp.seq_filename = ml.seq_filename; tmp.CreateMemorySMLs(w, nullptr, 0);
          MaskedMemHash f; f.SetMask((1ull << N) - 1); f.FindMatches(tmp); nway.insert(nway.end(), tmp.begin(), tmp.end()); nway.sml_table = tmp.sml_table; nway.seed_pattern = tmp.seed_pattern; }
        Aligner builtin(N); IntervalList il1; builtin.SetGappedAligner(HipGappedAligner::getInterface());
        builtin.align(nway, il1, 0, (int64)w * 3 * N, true, false, true, "");
        CountingAligner ca; Aligner plugged(N); plugged.SetGappedAligner(ca); IntervalList il2;
        plugged.align(nway, il2, 0, (int64)w * 3 * N, true, false, true, "");
        assert(il1.sizes.n_gap_dp > 10 && ca.aligned == (size_t)il1.sizes.n_gap_dp && ca.fast == ca.aligned);
        assert(xmfa(il1) == xmfa(il2));
        // 4. the list that is handed in is what gets chained
        MatchList half = nway; half.clear();
        for (size_t i = 0; i < nway.size(); i += 2) half.push_back(nway[i]);
        IntervalList il3; builtin.align(half, il3, 0, (int64)w * 3 * N, false, false, true, "");
        assert(il3.sizes.n_mums == (int64_t)half.size() && il3.sizes.n_anchor <= (int64_t)half.size() && il3.sizes.n_anchor > 0);
        // 5. signed permutations of the LCB sets between two weights (mauveAligner.cpp:678-686)
        {
            const std::string pf = std::string(argv[1]) + ".perm";
            Aligner pa(N); pa.SetGappedAligner(HipGappedAligner::getInterface()); pa.SetPermutationOutput(pf, (int64)w * N);
            IntervalList il4; pa.align(nway, il4, 0, (int64)w * 3 * N, false, false, true, "");
            std::ifstream in(pf.c_str()); std::string line; std::vector<std::vector<std::string>> sets(1);
            while (std::getline(in, line)) { if (line.empty()) sets.emplace_back(); else sets.back().push_back(line); }
            assert(sets.back().empty()); sets.pop_back();
            assert(!sets.empty());
            size_t prev = (size_t)-1;
            for (const auto &set : sets) {
                assert(set.size() == N);
                std::vector<size_t> cnt;
                for (const std::string &row : set) {                        // every row a signed permutation of 1..K
                    std::istringstream is(row); long v; std::set<long> ids; while (is >> v) { assert(v != 0); ids.insert(std::labs(v)); }
                    cnt.push_back(ids.size()); assert(!ids.empty() && *ids.begin() == 1 && *ids.rbegin() == (long)ids.size());
                }
                for (size_t g = 1; g < N; g++) assert(cnt[g] == cnt[0]);
                { std::istringstream is(set[0]); long v, want = 1; while (is >> v) assert(v == want++); }    // sequence 0 is the identity
                assert(cnt[0] <= prev); prev = cnt[0];                      // heavier minimum weight, fewer LCBs
            }
            assert(prev == (size_t)il4.sizes.n_lcb);                        // the last set is the one that was aligned
        }
        // 6. progressive alignment along the caller's guide tree (progressiveMauve.cpp:689-692)
        if (N >= 3) {
            const std::string tf = std::string(argv[1]) + ".tree";
            ProgressiveAligner p1(N); p1.setOutputGuideTreeFileName(tf); IntervalList a1; p1.align(ml.seq_table, a1);
            ProgressiveAligner p2(N); p2.setInputGuideTreeFileName(tf); IntervalList a2; p2.align(ml.seq_table, a2);
            assert(a1.size() > 0 && xmfa(a1) == xmfa(a2) && p1.treeLeft() == p2.treeLeft() && p1.treeRight() == p2.treeRight());
            { std::ofstream f(tf.c_str()); f << "(seq" << N << ",(seq1,seq2)"; for (uint g = 2; g + 1 < N; g++) f << ",seq" << g + 1; f << ");\n"; }
            ProgressiveAligner p3(N); p3.setInputGuideTreeFileName(tf); IntervalList a3; p3.align(ml.seq_table, a3);
            assert(a3.size() > 0 && p3.treeLeft()[N] == 0 && p3.treeRight()[N] == 1 && p3.treeLeft()[N + 1] == (int32_t)N - 1 && p3.treeRight()[N + 1] == (int32_t)N);
            { std::ofstream f(tf.c_str()); f << "((seq1,seq2),seq1);\n"; }
            bool threw = false; try { ProgressiveAligner p4(N); p4.setInputGuideTreeFileName(tf); IntervalList a4; p4.align(ml.seq_table, a4); } catch (genome::gnException &) { threw = true; }
            assert(threw);
        }
        // 7. applyBackbone as progressiveMauve.cpp:226-260 writes it, and mauveAligner's backbone / island outputs (:807-847)
        size_t n_bb = 0;
        {
            IntervalList &iv_list = il1;
            backbone_list_t bb_list;
            Params hmm_params = getAdaptedHoxdMatrixParameters(computeGC(iv_list.seq_table));
            hmm_params.iGoHomologous = 0.00001; hmm_params.iGoUnrelated = 0.000000001;
            adaptToPercentIdentity(hmm_params, 0.7);
            detectAndApplyBackbone(iv_list, bb_list, hmm_params);
            bb_list.clear();
            BigGapsDetector bgd(20);
            detectBackbone(iv_list, bb_list, &bgd);
            assert(bb_list.size() == iv_list.size());
            for (size_t i = 0; i < bb_list.size(); i++)
                for (const BackboneSegment &b : bb_list[i]) {
                    n_bb++;
                    assert(b.iv == i && b.Multiplicity() >= 2 && b.left_col + b.length <= iv_list[i].AlignmentLength());
                    const std::vector<uint32_t> &cols = iv_list[i].Columns();
                    for (uint g = 0; g < N; g++) {                      // the ends are what the columns of the segment hold
                        gnSeqI inside = 0, before = 0;
                        for (gnSeqI cI = 0; cI < b.left_col + b.length; cI++) { const bool r = cols[(size_t)cI] >> g & 1; if (cI < b.left_col) before += r; else inside += r; }
                        if (!(b.genomes >> g & 1)) { assert(b.Start(g) == 0); continue; }
                        assert(inside == b.Length(g) && inside > 0);
                        if (iv_list[i].Orientation(g) == AbstractMatch::forward) assert(b.Start(g) > 0 && b.LeftEnd(g) == iv_list[i].LeftEnd(g) + before);
                        else assert(b.Start(g) < 0 && b.RightEnd(g) == iv_list[i].RightEnd(g) - before);
                    }
                }
            assert(n_bb >= (size_t)iv_list.sizes.n_lcb && n_bb > 0);      // every LCB holds backbone
            std::ostringstream bb_out; writeBackboneSeqCoordinates(bb_list, iv_list, bb_out);
            std::vector<bb_seqentry_t> bb_seq_list; std::istringstream bbseq_input(bb_out.str()); readBackboneSeqFile(bbseq_input, bb_seq_list);
            assert(bb_seq_list.size() == n_bb);
            mergeAdjacentSegments(bb_seq_list); addUniqueSegments(bb_seq_list);
            std::ostringstream bb_final; writeBackboneSeqFile(bb_final, bb_seq_list);
            std::ostringstream bbcols; writeBackboneColumns(bbcols, bb_list);
            std::istringstream bbcols_in(bbcols.str()); std::vector<bb_colentry_t> colrows; readBackboneColsFile(bbcols_in, colrows);
            assert(colrows.size() == n_bb);
            iv_list.backbone_filename = "x.bbcols";
            // the other detector plug is refused, not ignored
            HssDetector other; bool threw = false; try { detectBackbone(iv_list, bb_list, &other); } catch (genome::gnException &) { threw = true; }
            assert(threw);
            std::vector<GappedAlignment> backbone_data; simpleFindBackbone(iv_list, 50, 20, backbone_data);
            assert(!backbone_data.empty());
            for (const GappedAlignment &ga : backbone_data)
                for (uint g = 0; g < N; g++) {
                    size_t bases = 0; for (char ch : ga.GetAlignment()[g]) bases += ch != '-';
                    assert(ga.Length(g) >= 50 && bases == ga.Length(g) && ga.Start(g) != 0);
                }
            std::ostringstream bbtxt; outputBackbone(backbone_data, bbtxt); assert(!bbtxt.str().empty());
            std::ostringstream isl; simpleFindIslands(iv_list, 1, isl);
            std::istringstream isl_in(isl.str()); std::string ln; size_t rows_isl = 0; while (std::getline(isl_in, ln)) rows_isl++;
            size_t runs = 0;                                          // one-sided runs of every pair, counted column by column
            for (const Interval &iv : iv_list)
                for (uint x = 0; x < N; x++) for (uint y = x + 1; y < N; y++) {
                    if (!iv.LeftEnd(x) || !iv.LeftEnd(y)) continue;
                    int prev = 0;
                    for (uint32_t m : iv.Columns()) { const int t = (int)(m >> x & 1) | (int)(m >> y & 1) << 1; if (!t) continue; if (t != 3 && t != prev) runs++; prev = t; }
                }
            assert(rows_isl == runs);
            std::ostringstream between; findIslandsBetweenLCBs(iv_list, 1, between);
            size_t singles = 0; for (const Interval &iv : iv_list) singles += iv.Multiplicity() == 1;
            std::istringstream btw_in(between.str()); size_t rows_btw = 0; while (std::getline(btw_in, ln)) rows_btw++;
            assert(rows_btw <= singles && (singles == 0) == (rows_btw == 0));   // leftovers between LCBs; adjacent ones print as one
        }
        // 8. a seed family through ONE finder (progressiveMauve.cpp:510-546): three searches, longest seed first; the finder keeps
        //    what it found, takes nothing an earlier match contains, and GetMatchList hands out the union
        {
            const uint wf = MatchList::GetDefaultMerSize(ml.seq_table);
            std::vector<std::pair<int, int>> length_ranks(3);
            for (int r = 0; r < 3; r++) length_ranks[(size_t)r] = std::make_pair((int)getSeedLength(getSeed((int)wf, r)), r);
            std::sort(length_ranks.begin(), length_ranks.end());
            HipUniqueMatchFinder umf; size_t first = 0, sum = 0;
            for (int seedI = 2; seedI >= 0; seedI--) {
                MatchList cur_list; cur_list.seq_filename = ml.seq_filename; cur_list.seq_table = ml.seq_table;
                cur_list.CreateMemorySMLs(wf, nullptr, length_ranks[(size_t)seedI].second);
                umf.FindMatches(cur_list);
                umf.ClearSequences();
                if (seedI == 2) first = cur_list.size();
                for (size_t i = 0; i < cur_list.size(); i++) cur_list[i]->Free();
                MatchList alone; alone.seq_filename = ml.seq_filename; alone.seq_table = ml.seq_table; alone.sml_table = cur_list.sml_table;
                HipUniqueMatchFinder single; single.FindMatches(alone); sum += alone.size();
                for (size_t i = 0; i < alone.size(); i++) alone[i]->Free();
                for (size_t i = 0; i < cur_list.sml_table.size(); i++) delete cur_list.sml_table[i];
            }
            MatchList family; family.seq_table = ml.seq_table; umf.GetMatchList(family);
            assert(first > 0 && family.size() >= first && family.size() < sum);      // the union, minus what was found twice
            umf.Clear();
            MatchList none; none.seq_table = ml.seq_table; umf.GetMatchList(none); assert(none.empty());
            for (size_t i = 0; i < family.size(); i++) family[i]->Free();
            if (N >= 3) {
                ProgressiveAligner pf(N); pf.setUseSeedFamilies(true); IntervalList af; pf.align(ml.seq_table, af);
                ProgressiveAligner ps(N); IntervalList as; ps.align(ml.seq_table, as);
                assert(af.size() > 0 && as.size() > 0);
            }
        }
        std::cout << "backbone segments " << n_bb << "\n";
        std::cout << "callbacks " << oof.calls << ", matches " << dev.size() << ", repeats " << devrep.size() << ", plug calls " << ca.aligned << "\nOK" << std::endl;
        return 0;
    } catch (std::exception &e) {
        std::cerr << "exception: " << e.what() << std::endl;
        return 2;
    }
}
