// libMems/ProgressiveAligner.h -- the progressiveMauve alignment stage: setters and align() as called at
// progressiveMauve.cpp:575-710.
#ifndef MAUVE_HIP_PROGRESSIVEALIGNER_H
#define MAUVE_HIP_PROGRESSIVEALIGNER_H

#include "Aligner.h"
#include "GuideTree.h"
#include <fstream>
#include <iostream>
#include <sstream>

namespace mems {

// ---- ProgressiveAligner: setters and align() as called at progressiveMauve.cpp:575-710 -------------------
// The guide tree and the progressive anchoring run on the device (mauve_progressive_align); the extant
// sum-of-pairs LCB scoring is DESIGN.md S11, the penalty scaling by conservation and breakpoint distance S11b / S11c, the
// refinement S13 (frozen forms of library-internal stages).  Not reproducible without libMems: the two ancestral scoring
// schemes (they need its ancestral sequence reconstruction) -- they fall back to the extant scheme -- and the cache database.
class ProgressiveAligner {
public:
    explicit ProgressiveAligner(uint seq_count) : seq_count_(seq_count), tree_left_(2 * seq_count - 1, -1), tree_right_(2 * seq_count - 1, -1)
    {
        // the call site's defaults: extant sum-of-pairs scoring (progressiveMauve.cpp:624-625), scaling on with both scales 0.5
        // (:285-287), refinement on unless --skip-refinement (:578-579)
        mauve_default_progressive_params(&p_);
    }
    // --weight, :584-593: a length (x seq_count) under LengthScoring, a score under the sum-of-pairs scheme
    void setBreakpointPenalty(double w) { if (w >= 0) bp_penalty_ = w; }
    void setMinimumBreakpointPenalty(double w) { if (w >= 0) p_.min_scaled_penalty = (int64_t)w; }   // :649-652: floor of the scaled weight
    void setCollinear(boolean c) { p_.collinear = c; }                        // :594-597
    void setGappedAlignment(boolean g) { p_.gapped = g; }                     // --skip-gapped-alignment
    void setRefinement(boolean r) { p_.refine_rounds = r ? 2 : 0; }           // :578-579; DESIGN.md S13: two rotated orders per interval, best sum-of-pairs score kept
    void setRecursion(boolean r) { p_.recursive = r; }
    void SetRecursive(boolean r) { p_.recursive = r; }                        // :661-664
    void SetMaxGappedAlignmentLength(gnSeqI n) { p_.max_gapped_len = (int64_t)n; }
    void setPairwiseScoringScheme(const PairwiseScoringScheme &pss)           // :666-687
    {
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) p_.scoring.matrix[i][j] = pss.matrix[i][j];
        p_.scoring.gap_open = pss.gap_open; p_.scoring.gap_extend = pss.gap_extend;
    }
    // :611-625.  ExtantSumOfPairsScoring (the call site's default, :625) scores LCBs by the sum-of-pairs score of their
    // anchors (MAUVE_LCB_SCORE_SP); the two ancestral schemes need libMems' ancestral sequence
    // reconstruction and fall back to it as well.  LengthScoring (not in libMems) keeps the Aligner::align weights.
    enum LcbScoringScheme { AncestralScoring, AncestralSumOfPairsScoring, ExtantSumOfPairsScoring, LengthScoring };
    void setLcbScoringScheme(int s)
    {
        if (s == AncestralScoring || s == AncestralSumOfPairsScoring) {      // said once where the caller sees it, not only in DESIGN.md
            static bool told = false;
            if (!told) {
                told = true;
                std::cerr << "mauve_hip: the ancestral LCB scoring schemes need libMems' ancestral sequence reconstruction, which this "
                             "library does not reproduce; scoring LCBs with the extant sum-of-pairs scheme instead\n";
            }
        }
        p_.lcb_scoring = s == LengthScoring ? MAUVE_LCB_SCORE_LENGTH : MAUVE_LCB_SCORE_SP; score_set_ = true;
    }
    // :626-642.  A node's minimum LCB weight shrinks with the mean pairwise conservation distance between its two subtrees
    //  and with their mean breakpoint distance (S11c: broken adjacencies between the pairwise matches of at least
    // the given length, relative to the most rearranged pair).
    void setUseLcbWeightScaling(boolean b) { p_.weight_scaling = b ? 1 : 0; }
    void setBreakpointDistanceScale(double d) { if (d >= 0 && d <= 1) p_.bp_dist_scale_ppm = (int32_t)(d * 1e6 + 0.5); }      // :628-632
    void setConservationDistanceScale(double d) { if (d >= 0 && d <= 1) p_.conservation_scale_ppm = (int32_t)(d * 1e6 + 0.5); }
    void setBpDistEstimateMinScore(double d) { if (d >= 0) p_.bp_dist_min_score = (int64_t)d; }                               // :638-642
    // :689-692.  The input tree replaces the UPGMA one (mauve_progressive_align_tree); the output file receives the
    // tree the alignment used, NEWICK both ways with leaves seq1..seqN (GuideTree.h).
    void setInputGuideTreeFileName(const std::string &fn) { input_tree_fn_ = fn; }
    void setOutputGuideTreeFileName(const std::string &fn) { output_tree_fn_ = fn; }
    void setUseSeedFamilies(boolean b) { p_.seed_family = b ? 1 : 0; }        // :604-605: every node searches with the family of three seeds
    void SetUseCacheDb(boolean) {}                                            // :643-646
    // progressiveMauve.cpp:652-655 hands the pairwise matches over; the device path finds them itself
    // (PairwiseMatchFinder rule on the resident genomes), so only the seed pattern is taken from the list.
    void setPairwiseMatches(MatchList &pairwise) { if (pairwise.seed_pattern) p_.seed_pattern = (uint64_t)pairwise.seed_pattern; }
    void setSeedWeight(uint w) { p_.seed_weight = (int32_t)w; }
    // aligner.align(interval_list.seq_table, interval_list)  (:710)
    void align(std::vector<genome::gnSequence *> &seq_table, IntervalList &il)
    {
        if (seq_table.size() != seq_count_) throw genome::gnException("ProgressiveAligner::align: sequence count mismatch");
        if (p_.seed_family) {                                 // the family comes from the weight; a pattern taken from the pairwise list only names it
            if (!p_.seed_weight && p_.seed_pattern) p_.seed_weight = mauve_seed_weight(p_.seed_pattern);
            p_.seed_pattern = 0;
        }
        if (bp_penalty_ >= 0) p_.lcb_weight = p_.lcb_scoring == MAUVE_LCB_SCORE_SP ? (int64_t)bp_penalty_ : (int64_t)bp_penalty_ * (int64_t)seq_count_;
        HipContext &hc = HipContext::global();
        MatchList tmp; tmp.seq_table = seq_table;
        tmp.upload(hc);
        std::vector<int64_t> dist;
        if (!input_tree_fn_.empty()) {
            std::ifstream in(input_tree_fn_.c_str());
            if (!in) throw genome::gnException("ProgressiveAligner::align: cannot read the input guide tree " + input_tree_fn_);
            std::stringstream ss; ss << in.rdbuf();
            std::string why;
            if (!guideTreeFromNewick(ss.str(), (int)seq_count_, tree_left_, tree_right_, &why))
                throw genome::gnException("ProgressiveAligner::align: input guide tree: " + why);
            hc.check(mauve_progressive_align_tree(hc.get(), &p_, &il.sizes, tree_left_.data(), tree_right_.data()), "mauve_progressive_align_tree");
        } else {
            dist.assign((size_t)seq_count_ * seq_count_, 0);
            hc.check(mauve_progressive_align(hc.get(), &p_, &il.sizes, tree_left_.data(), tree_right_.data(), dist.data()), "mauve_progressive_align");
        }
        if (!output_tree_fn_.empty()) {
            std::ofstream out(output_tree_fn_.c_str());
            if (!out) throw genome::gnException("ProgressiveAligner::align: cannot write the guide tree " + output_tree_fn_);
            out << guideTreeToNewick((int)seq_count_, tree_left_, tree_right_, dist);
        }
        il.seq_table = seq_table;
        il.fetch(hc, seq_count_);
    }
    // guide tree of the last align(): child ids per node (leaves -1), nodes seq_count.. in merge order
    const std::vector<int32_t> &treeLeft() const { return tree_left_; }
    const std::vector<int32_t> &treeRight() const { return tree_right_; }
private:
    uint seq_count_;
    mauve_params p_;
    double bp_penalty_ = -1;
    bool score_set_ = false;
    std::vector<int32_t> tree_left_, tree_right_;
    std::string input_tree_fn_, output_tree_fn_;
};

}  // namespace mems
#endif
