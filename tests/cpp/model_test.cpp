// Host-side data model of the mirror (include/libMems): the shapes the in-tree tools use, checked on small hand cases.
#include <cassert>
#include <fstream>
#include <iostream>
#include <sstream>
#include "libMems/mems_hip.h"
using namespace mems;

int main()
{
    // ---- MatchProjectionAdapter (SeedMatchEnumerator.h:98; MatchRecord.h:242) ----
    Match m(4); m.SetLength(20); m.SetStart(0, 101); m.SetStart(1, -301); m.SetStart(3, 501);
    std::vector<size_t> map; map.push_back(3); map.push_back(1);
    MatchProjectionAdapter mpa(m.Copy(), map);
    assert(mpa.SeqCount() == 2 && mpa.Multiplicity() == 2 && mpa.Start(0) == 501 && mpa.Start(1) == -301 && mpa.LeftEnd(1) == 301 && mpa.RightEnd(1) == 320);
    AbstractMatch *cp = mpa.Copy(); cp->CropLeft(5, 0); assert(cp->Start(0) == 506 && cp->Length(0) == 15 && mpa.Start(0) == 501); cp->Free();
    // ---- GappedAlignment (repeatoire.cpp:1238,1264-1265) ----
    GappedAlignment ga(3, 0);
    std::vector<std::string> rows; rows.push_back("ACG-TA"); rows.push_back("A-GGTA"); rows.push_back("------");
    ga.SetAlignment(rows); ga.SetStart(0, 11); ga.SetStart(1, -21);
    assert(ga.AlignmentLength() == 6 && ga.Length(0) == 5 && ga.Length(1) == 5 && ga.Length(2) == 0 && ga.Multiplicity() == 2);
    assert(ga.LeftEnd(0) == 11 && ga.RightEnd(0) == 15 && ga.RightEnd(1) == 25);
    std::vector<gnSeqI> pos; std::vector<bool> col;
    ga.GetColumn(4, pos, col); assert(col[0] && col[1] && !col[2] && pos[0] == 14 && pos[1] == 22);   // reverse row counts down from its right end
    ga.GetColumn(1, pos, col); assert(col[0] && !col[1] && pos[0] == 12);
    GappedAlignment g2 = ga; g2.CropStart(2); assert(g2.AlignmentLength() == 4 && g2.Start(0) == 13 && g2.Length(0) == 3 && g2.Start(1) == -21 && g2.Length(1) == 4);
    g2 = ga; g2.CropEnd(2); assert(g2.Start(0) == 11 && g2.Length(0) == 3 && g2.Start(1) == -23 && g2.Length(1) == 3);
    g2 = ga; g2.Invert(); assert(g2.Start(0) == -11 && g2.Start(1) == 21 && g2.GetAlignment()[0] == "TA-CGT");
    // ---- CompactGappedAlignment (repeatoire.cpp:1316-1318,1347; bbBreakOnGenes.cpp:154-155) ----
    CompactGappedAlignment<> cga(ga);
    assert(cga.AlignmentLength() == 6 && cga.Length(0) == 5 && cga.GetAlignment()[1][1] == false && cga.GetAlignment()[0][3] == false);
    assert(cga.SeqPosToColumn(0, 11) == 0 && cga.SeqPosToColumn(0, 14) == 4 && cga.SeqPosToColumn(1, 25) == 0 && cga.SeqPosToColumn(1, 21) == 5);
    CompactGappedAlignment<> part; cga.copyRange(part, 2, 3);
    assert(part.AlignmentLength() == 3 && part.Start(0) == 13 && part.Length(0) == 2 && part.Length(1) == 3 && part.Start(1) == -22);
    // ---- Interval over matches: SetMatches steals, GetColumn, StealMatches (MatchRecord.h:338-343) ----
    std::vector<AbstractMatch *> chain;
    Match a(2); a.SetLength(10); a.SetStart(0, 1); a.SetStart(1, 101);
    GappedAlignment mid(2, 0); std::vector<std::string> mr; mr.push_back("AC-"); mr.push_back("A-G"); mid.SetAlignment(mr); mid.SetStart(0, 11); mid.SetStart(1, 111);
    Match b(2); b.SetLength(5); b.SetStart(0, 13); b.SetStart(1, 113);
    chain.push_back(a.Copy()); chain.push_back(mid.Copy()); chain.push_back(b.Copy());
    Interval iv; iv.SetMatches(chain);
    assert(chain.empty() && iv.GetMatches().size() == 3 && iv.AlignmentLength() == 18 && iv.LeftEnd(0) == 1 && iv.RightEnd(0) == 17 && iv.RightEnd(1) == 117);
    iv.GetColumn(11, pos, col); assert(col[0] && !col[1] && pos[0] == 12);
    iv.GetColumn(12, pos, col); assert(!col[0] && col[1] && pos[1] == 112);
    CompactGappedAlignment<> civ(iv); assert(civ.AlignmentLength() == 18 && civ.SeqPosToColumn(1, 113) == 13);
    std::vector<AbstractMatch *> back; iv.StealMatches(back); assert(back.size() == 3 && iv.GetMatches().empty());
    for (AbstractMatch *x : back) x->Free();
    // ---- LCB helpers (toGrimmFormat.cpp:51-79; projectAndStrip.cpp:110-112; sortContigs.cpp:55-84) ----
    MatchList ml;
    const int64 st[5][2] = {{1, 1001}, {101, 1101}, {201, -2201}, {301, -2101}, {401, 1301}};      // 2 forward, 2 in an inversion, 1 forward
    for (int i = 0; i < 5; i++) { Match x(2); x.SetLength(50); x.SetStart(0, st[i][0]); x.SetStart(1, st[i][1]); ml.push_back(x.Copy()); }
    std::vector<gnSeqI> bps; IdentifyBreakpoints(ml, bps);
    assert(bps.size() == 3 && bps[0] == 1 && bps[1] == 3 && bps[2] == 4);
    std::vector<MatchList> lcbs; std::vector<int64> w; ComputeLCBs_v2(ml, bps, lcbs, w);
    assert(lcbs.size() == 3 && lcbs[0].size() == 2 && lcbs[1].size() == 2 && w[0] == 200 && w[2] == 100);
    std::vector<LCB> adj; computeLCBAdjacencies_v2(lcbs, w, adj);
    assert(adj[1].left_end[1] == -2101 && adj[1].right_end[1] == -2250 && adj[0].right_end[0] == 150 && adj[1].lcb_id == 1);
    assert(adj[0].left_adjacency[0] == NO_ADJACENCY && adj[0].right_adjacency[0] == 1 && adj[1].right_adjacency[0] == 2);
    assert(adj[0].right_adjacency[1] == 2 && adj[2].right_adjacency[1] == 1 && adj[1].right_adjacency[1] == NO_ADJACENCY);      // genome 1 order: 0, 2, 1
    std::vector<int64> tr; transposeMatches(ml, 1, tr); assert(tr.size() == 5 && tr[2] == -2201);
    // EliminateOverlaps: the shorter of two overlapping matches gives way
    MatchList ov; { Match x(2); x.SetLength(100); x.SetStart(0, 1); x.SetStart(1, 1); ov.push_back(x.Copy()); Match y(2); y.SetLength(30); y.SetStart(0, 91); y.SetStart(1, 91); ov.push_back(y.Copy()); }
    EliminateOverlaps(ov); assert(ov.size() == 2 && ov[1]->Start(0) == 101 && ov[1]->Length() == 20 && ov[0]->Length() == 100);
    ov.Clear(); ml.Clear();
    // addUnalignedIntervals (mauveAligner.cpp:748)
    IntervalList il; genome::gnSequence s0(std::string(30, 'A')), s1(std::string(20, 'C'));
    il.seq_table.push_back(&s0); il.seq_table.push_back(&s1);
    { std::vector<int64> l(2), r(2); std::vector<char> rv(2, 0); l[0] = 5; r[0] = 14; l[1] = 1; r[1] = 10; il.push_back(Interval(l, r, rv, std::vector<uint32_t>(10, 3u))); }
    addUnalignedIntervals(il);
    assert(il.size() == 4 && il[1].LeftEnd(0) == 1 && il[1].RightEnd(0) == 4 && il[2].LeftEnd(0) == 15 && il[2].RightEnd(0) == 30 && il[3].LeftEnd(1) == 11 && il[3].RightEnd(1) == 20);
    // readSubstitutionMatrix (progressiveMauve.cpp:684)
    std::istringstream mat("# HOXD70\n  A C G T\nA 91 -114 -31 -123\nC -114 100 -125 -31\nG -31 -125 100 -114\nT -123 -31 -114 91\n");
    score_t M[4][4]; readSubstitutionMatrix(mat, M);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) assert(M[i][j] == hoxd_matrix[i][j]);
    // .mln of another library is refused, not misparsed
    std::istringstream foreign("FormatVersion\t4\nSequenceCount\t2\n");
    bool threw = false; try { IntervalList x; x.ReadList(foreign); } catch (genome::gnException &) { threw = true; }
    assert(threw);
    std::istringstream bigdef("> 40:1-5 + x\nACGTA\n=\n");
    threw = false; try { IntervalList x; x.ReadStandardAlignment(bigdef); } catch (genome::gnException &) { threw = true; }
    assert(threw);
    std::cout << "OK" << std::endl;
    // ---- computeSPScore (repeatoire.cpp:2527), LoadSequences / LoadMFASequences (addUnalignedIntervals.cpp:23, evd.cpp:97) ----
    {
        PairwiseScoringScheme pss;                      // HOXD70, -400 / -30
        std::vector<std::string> aln; aln.push_back("AC--GT"); aln.push_back("ACTTGT"); aln.push_back("A---GA");
        std::vector<score_t> colscore; score_t total = 0;
        computeSPScore(aln, pss, colscore, total);
        // pairs (0,1): A/A C/C open ext G/G T/T; (0,2): A/A, C vs gap opens, two gap-gap columns, G/G, T/A; (1,2): A/A, open ext ext, G/G, T/A
        const score_t want = (91 + 100 - 400 - 30 + 100 + 91) + (91 - 400 + 100 - 123) + (91 - 400 - 30 - 30 + 100 - 123);
        assert(total == want && colscore.size() == 6 && colscore[0] == 3 * 91 && colscore[2] == -400 - 30);
        score_t s2 = 0; for (size_t c = 0; c < colscore.size(); c++) s2 += colscore[c];
        assert(s2 == total);
        const char *mfa = "/tmp/mauve_hip_model_test.mfa";
        { std::ofstream f(mfa); f << ">one\nACGT\nAC\n>two\nGGTT\n"; }
        MatchList ml2; LoadMFASequences(ml2, mfa, nullptr);
        assert(ml2.seq_table.size() == 2 && ml2.seq_table[0]->length() == 6 && ml2.seq_table[1]->ToString() == "GGTT" && ml2.seq_filename.size() == 2);
        IntervalList il2; il2.seq_filename.push_back(mfa); LoadSequences(il2, nullptr);
        assert(il2.seq_table.size() == 1 && il2.seq_table[0]->length() == 10 && il2.seq_table[0]->contigListSize() == 2);
        for (size_t i2 = 0; i2 < ml2.seq_table.size(); i2++) delete ml2.seq_table[i2];
        delete il2.seq_table[0];
    }
    // ---- guide tree files (progressiveMauve.cpp:689-692) and signed permutations (mauveAligner.cpp:678-686) ----
    {
        // ((seq1,seq3),(seq2,seq4)) in merge order: 4 = (0,2), 5 = (1,3), 6 = (4,5)
        std::vector<int32_t> L(7, -1), R(7, -1); L[4] = 0; R[4] = 2; L[5] = 1; R[5] = 3; L[6] = 4; R[6] = 5;
        assert(guideTreeToNewick(4, L, R) == "((seq1,seq3),(seq2,seq4));\n");
        std::vector<int64_t> D(16, 400000); for (int i = 0; i < 4; i++) D[(size_t)i * 5] = 0;
        D[0 * 4 + 2] = D[2 * 4 + 0] = 100000; D[1 * 4 + 3] = D[3 * 4 + 1] = 200000;
        const std::string nw = guideTreeToNewick(4, L, R, D);
        assert(nw == "((seq1:0.050000,seq3:0.050000):0.150000,(seq2:0.100000,seq4:0.100000):0.100000);\n");
        std::vector<int32_t> L2, R2; std::string why;
        assert(guideTreeFromNewick(nw, 4, L2, R2, &why) && L2 == L && R2 == R);
        // labels, comments, quoted names, bare numbers, an unrooted (trifurcating) root
        assert(guideTreeFromNewick(" ( 'seq2':1e-3 , (3:0.1,seq1)x[c]:2 , SEQ4 ) root ;", 4, L2, R2, &why));
        assert(L2[4] == 2 && R2[4] == 0 && L2[5] == 1 && R2[5] == 4 && L2[6] == 5 && R2[6] == 3);
        assert(!guideTreeFromNewick("((seq1,seq2),seq3);", 4, L2, R2, &why) && why.find("missing") != std::string::npos);
        assert(!guideTreeFromNewick("((seq1,seq2),(seq3,seq1));", 4, L2, R2, &why) && why.find("twice") != std::string::npos);
        assert(!guideTreeFromNewick("((seq1,seq2),(seq3,seq4)", 4, L2, R2, &why));
        assert(!guideTreeFromNewick("((seq1,genomeB),(seq3,seq4));", 4, L2, R2, &why));
        assert(!guideTreeFromNewick("((seq1,seq2),(seq3,seq5));", 4, L2, R2, &why));
        assert(!guideTreeFromNewick("(seq1,seq2));", 2, L2, R2, &why));
        // three LCBs, the middle one inverted in sequence 1 and the order 3 -2 1 there
        std::vector<int64_t> le = {10, 900, 400, -500, 800, 100}, la = {-1, 1, 0, 2, 1, -1}, ra = {1, -1, 2, 0, -1, 1};
        std::ostringstream po; WritePermutation(po, 2, 3, le, la, ra);
        assert(po.str() == "1\t2\t3\n3\t-2\t1\n\n");
    }
    // ---- .backbone / .bbcols files and the list helpers (progressiveMauve.cpp:245-258; bbFilter.cpp:75-90) ----
    {
        std::vector<bb_seqentry_t> rows;
        auto row = [](int64 a, int64 b, int64 c, int64 d) { bb_seqentry_t r; r.push_back(std::make_pair(a, b)); r.push_back(std::make_pair(c, d)); return r; };
        rows.push_back(row(101, 200, -900, -999));        // continues in both: forward 201.., reverse ..-899
        rows.push_back(row(201, 260, -840, -899));
        rows.push_back(row(300, 400, 0, 0));              // only sequence 0
        rows.push_back(row(500, 600, 100, 200));
        rows.push_back(row(601, 700, 250, 349));          // abuts in sequence 0 only: stays
        std::ostringstream os; writeBackboneSeqFile(os, rows);
        assert(os.str().find("seq0_leftend\tseq0_rightend\tseq1_leftend\tseq1_rightend\n101\t200\t-900\t-999\n") == 0);
        std::istringstream is(os.str()); std::vector<bb_seqentry_t> back; readBackboneSeqFile(is, back);
        assert(back == rows);
        mergeAdjacentSegments(back);
        assert(back.size() == 4 && back[0] == row(101, 260, -840, -999) && back[1] == row(300, 400, 0, 0));
        addUniqueSegments(back, 20);
        // sequence 0: 1-100, 261-299, 401-499; sequence 1: 1-99, 201-249, 350-839
        size_t uniq0 = 0, uniq1 = 0;
        for (const bb_seqentry_t &r : back) { if (r[0].first && !r[1].first && r != row(300, 400, 0, 0)) uniq0++; if (!r[0].first && r[1].first) uniq1++; }
        assert(uniq0 == 3 && uniq1 == 3);
        bool found = false; for (const bb_seqentry_t &r : back) found = found || r == row(0, 0, 350, 839);
        assert(found);
        std::istringstream bad("1\t2\t3\t4\n"); bool threw2 = false;
        try { readBackboneSeqFile(bad, back); } catch (genome::gnException &) { threw2 = true; }
        assert(threw2);
        backbone_list_t bl(2);
        BackboneSegment b; b.iv = 1; b.left_col = 7; b.length = 30; b.genomes = 5; b.ends.resize(3); b.ends[0] = std::make_pair(10, 39); b.ends[2] = std::make_pair(-61, -90);
        bl[1].push_back(b);
        assert(b.Multiplicity() == 2 && b.LeftEnd(2) == 61 && b.RightEnd(2) == 90 && b.Length(2) == 30 && b.Length(1) == 0);
        std::ostringstream oc; writeBackboneColumns(oc, bl);
        assert(oc.str() == "1\t7\t30\t0\t2\n");
        std::istringstream ic(oc.str()); std::vector<bb_colentry_t> cl; readBackboneColsFile(ic, cl);
        assert(cl.size() == 1 && cl[0].first == 1 && cl[0].second.size() == 4 && cl[0].second[1] == 30 && cl[0].second[3] == 2);
        bb_entry_t e; e.bb_seq = b.ends; e.bb_cols = cl[0].second; e.iv = cl[0].first;       // bbAnalyze.cpp:1007-1012
        assert(e.iv == 1 && e.bb_seq.size() == 3);
        IntervalList il; std::ostringstream o2; writeBackboneSeqCoordinates(bl, il, o2);
        assert(o2.str().find("10\t39\t0\t0\t-61\t-90\n") != std::string::npos);
        genome::gnSequence sa("GGCCAT"), sb("ATATGC"); std::vector<genome::gnSequence *> st; st.push_back(&sa); st.push_back(&sb);
        assert(computeGC(st) == 0.5);
        Params hp = getAdaptedHoxdMatrixParameters(0.5); hp.iGoHomologous = 1e-5; hp.iGoUnrelated = 1e-9; adaptToPercentIdentity(hp, 0.7);
        assert(hp.identity == 0.7);
    }
    return 0;
}
