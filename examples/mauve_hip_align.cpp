// mauve_hip_align.cpp -- the hot section of doAlignment (src/mauveAligner.cpp:453-466,523-531,585,648-698,
// 746-760) written against the libMems-shaped headers in include/libMems, i.e. what a reference maintainer's
// call site looks like after switching the path to libmauve_hip.so.  Not a CLI: positional FastA files in, XMFA
// on stdout.  Optional first argument "-u" uses UniqueMatchFinder as progressiveMauve.cpp:490-495 does; "-p" runs
// the progressiveMauve alignment stage instead (ProgressiveAligner, progressiveMauve.cpp:575-722).  With
// MAUVE_MUMS_OUT / MAUVE_MLN_OUT set, the match list and the interval list are also written at the stage seams
// (--mums / --output of the original, mauveAligner.cpp:603,702); with MAUVE_BACKBONE_OUT set, "-p" also runs applyBackbone
// as progressiveMauve.cpp:226-260 writes it and leaves <name> (.backbone rows) and <name>.bbcols behind.
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>

#include "libGenome/gnSequence.h"
#include "libMems/Aligner.h"
#include "libMems/HipFinders.h"
#include "libMems/MaskedMemHash.h"
#include "libMems/MatchList.h"
#include "libMems/ProgressiveAligner.h"
#include "libMems/Backbone.h"

using namespace mems;
using namespace genome;

int main(int argc, char **argv)
{
    try {
        int a = 1;
        bool unique = argc > 1 && std::string(argv[1]) == "-u";
        bool progressive = argc > 1 && std::string(argv[1]) == "-p";
        if (unique || progressive) a++;
        if (argc - a < 2) { std::cerr << "usage: mauve_hip_align [-u] <seq1.fa> <seq2.fa> [...]\n"; return -1; }
        MatchList match_list;
        for (; a < argc; a++) {
            gnSequence *s = new gnSequence();
            s->LoadSource(argv[a]);                                  // LoadSequences, mauveAligner.cpp:463
            match_list.seq_table.push_back(s);
            match_list.seq_filename.push_back(argv[a]);
        }
        if (progressive) {
            const uint n = (uint)match_list.seq_table.size();
            ProgressiveAligner aligner(n);                               // progressiveMauve.cpp:575
            PairwiseScoringScheme pss;                                   // :666-687 (hoxd_matrix, -400, -30)
            aligner.setPairwiseScoringScheme(pss);
            aligner.setLcbScoringScheme(ProgressiveAligner::ExtantSumOfPairsScoring);   // :624-625 "default to extant sp"
            IntervalList interval_list;
            interval_list.seq_filename = match_list.seq_filename;
            aligner.align(match_list.seq_table, interval_list);          // :710
            if (const char *bp = getenv("MAUVE_BACKBONE_OUT")) {         // applyBackbone, :226-260 (:712-719)
                backbone_list_t bb_list;
                interval_list.seq_table = match_list.seq_table;
                const double gc_content = computeGC(interval_list.seq_table);                     // :231
                Params hmm_params = getAdaptedHoxdMatrixParameters(gc_content);                   // :234
                hmm_params.iGoHomologous = 0.00001; hmm_params.iGoUnrelated = 0.000000001;        // pgh, pgu: :319-320
                adaptToPercentIdentity(hmm_params, 0.7);                                          // hmm_identity, :321
                detectAndApplyBackbone(interval_list, bb_list, hmm_params);                       // :239 (the homology pass rewrites the intervals)
                bb_list.clear();                                         // :240
                BigGapsDetector bgd(20);                                 // island_gap_size, :322
                detectBackbone(interval_list, bb_list, &bgd);            // :242-243
                std::ofstream bb_out(bp);
                writeBackboneSeqCoordinates(bb_list, interval_list, bb_out);                      // :245
                bb_out.close();
                std::vector<bb_seqentry_t> bb_seq_list;
                std::ifstream bbseq_input(bp);
                readBackboneSeqFile(bbseq_input, bb_seq_list);           // :249
                mergeAdjacentSegments(bb_seq_list);                      // :251
                addUniqueSegments(bb_seq_list);                          // :252
                bbseq_input.close();
                bb_out.open(bp);
                writeBackboneSeqFile(bb_out, bb_seq_list);               // :255
                const std::string bbcols_fname = std::string(bp) + ".bbcols";
                std::ofstream bbcols_out(bbcols_fname.c_str());
                writeBackboneColumns(bbcols_out, bb_list);               // :257-258
                interval_list.backbone_filename = bbcols_fname;          // :259
            }
            interval_list.WriteStandardAlignment(std::cout);             // :722
            for (auto *s : match_list.seq_table) delete s;
            return 0;
        }
        uint seed_size = 0, seed_rank = 0;
        match_list.CreateMemorySMLs(seed_size, &std::cerr, seed_rank);   // :456
        const uint N = (uint)match_list.seq_table.size();

        std::unique_ptr<MatchFinder> finder;
        if (unique) finder.reset(new HipUniqueMatchFinder());            // progressiveMauve.cpp:490-495 (UniqueMatchFinder's rule in the join kernel)
        else { finder.reset(new MaskedMemHash()); finder->SetMask((1ull << N) - 1); }   // mauveAligner.cpp:523-531
        finder->LogProgress(&std::cerr);
        finder->FindMatches(match_list);                                 // :585
        std::cerr << match_list.size() << " multi-MUMs\n";
        if (const char *mp = getenv("MAUVE_MUMS_OUT")) { std::ofstream mo(mp); WriteList(match_list, mo); }   // :603

        seed_size = MatchList::GetDefaultMerSize(match_list.seq_table);  // :650-651
        int64 LCB_size = (int64)seed_size * 3 * N;                       // :652
        Aligner aligner(N);                                              // :668
        aligner.SetGappedAligner(HipGappedAligner::getInterface());      // :674
        IntervalList interval_list;
        aligner.align(match_list, interval_list, 0, LCB_size, true, true, true, "");   // :698
        if (const char *lp = getenv("MAUVE_MLN_OUT")) { std::ofstream lo(lp); interval_list.WriteList(lo); }  // :702
        interval_list.WriteStandardAlignment(std::cout);                 // :746-760
        match_list.Clear();
        for (auto *s : match_list.sml_table) delete s;
        for (auto *s : match_list.seq_table) delete s;
        return 0;
    } catch (gnException &gne) {
        std::cerr << gne << std::endl;                                   // :852-864
        return -10;
    }
}
