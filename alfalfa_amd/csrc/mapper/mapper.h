// mapper.h -- host-side caller of the hot path: candidate regions, chaining, extension bookkeeping,
// CIGAR/MAPQ and SAM records, with the reference's semantics (src/mapper.cpp, src/mapper.h, src/options.cpp).
// This is the *caller* of libalfa_hot, not the accelerated path: it decides which seeds/DP calls are issued
// and turns their results into SAM.  Written from scratch; every function cites the reference lines whose
// behaviour it reproduces.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include <atomic>

#include "../../../include/alfa_index.h"
#include "../map_core.h"
#include "hot_sched.h"

namespace alfa {

enum SeedType { SEED_SMAM = 0, SEED_MEM = 1, SEED_SMEM = 2, SEED_SMEMP = 3 };
enum Orientation { PAIR_FR, PAIR_RF, PAIR_FF };

struct AlignOpt {                 // align_opt, defaults of mapOptions_t::initOptions (src/options.cpp:43-92)
    int match = 1, mismatch = -4, openGap = -6, extendGap = -1;
    int8_t mat[25];
    bool global = true;
    double errorPercent = 0.08;
    int minMemLength = 40;
    bool fixedMinLength = false;
    int alignmentCount = 1;
    int maxTrial = 10, minTrial = 5000;
    bool maxTrialBest = true;
    bool alwaysUnique = true;
    double minCoverage = 0.25;
    bool rescue = true, noSparseQ = false, fullDP = false;
    int maxSeedCandidates = 10000, maxSmemMemStart = 10, maxRegionMems = 20, minMemRegLength = 50;
    bool noFW = false, noRC = false;
    int memType = SEED_SMAM;
    int fixedBandSize = 100;
    int maxAlnCount = 5000;
    bool deviceRegions = false;   // --device-regions: identifyAndFilter + coveredBases run on the hot path's side
    bool coreTrials = false;      // --core-trials (checker): the trial loop runs through csrc/map_core.h -- the code the mapping kernel is
                                  // compiled from -- with the DP calls still going through the scheduler
    bool deviceMap = true;        // reads are mapped by the hot path's whole-read entry point (alfa_map_reads_batch) when the binding has
                                  // one; --host-trials switches it off
    void fill_m();                // align_opt::fill_m src/options.cpp:33-41
};

struct PairedOpt {
    int minInsert = 0, maxInsert = 1000;
    bool mixed = true, discordant = true, dovetail = false, contain = true, overlap = true;
    int mode = 1;
    bool pairedRescue = false;
    Orientation orientation = PAIR_FR;
};

struct MateInfo {                 // mate_t src/mapper.h:67-78
    unsigned flag = 0;
    std::string rnext = "*";
    int64_t pnext = 0, pnextGlob = 0;
    int tLength = 0;
    bool concordant = false;
    int alignmentScore = 0;
};

struct Alignment {                // alignment_t src/mapper.h:80-124
    int64_t refB = 0, refE = 0;
    int qB = 0, qE = 0;
    int64_t pos = 0;
    std::string cigar = "*";
    unsigned flag = 0;
    std::string rname = "*";
    int mapq = 0, editDist = 0, alignmentScore = 0;
    std::string NMtag = "*";
    int refLength = 0;
    MateInfo mate;
    std::vector<Match> chain;
    // extended CIGAR made with the alignment (whole reads on the device / --core-trials): constructCIGAR's output as operations
    std::vector<mapcore::CigOp> ops;
    bool haveOps = false;
    int opsEdit = 0;
    bool paired() const { return flag & 1u; }
    bool concordant() const { return paired() && mate.concordant; }
    bool rc() const { return (flag >> 4) & 1u; }
};

struct Read {                     // read_t src/mapper.h:126-219
    std::string qname, sequence, qual = "*", rcSequence, rQual;
    std::vector<uint8_t> readP, rcReadP;
    std::vector<Alignment*> alignments;
    int pairedAlignmentCount = 0;
    ~Read();
    Read() = default;
    Read(const Read&) = delete;
    Read& operator=(const Read&) = delete;
    Read(Read&& o) noexcept;
    void init();                  // read_t::init src/mapper.h:155-166
    void release();               // drops the alignments once the read's SAM lines are rendered (strings keep their capacity:
                                  // the object is reused for a later read of the input)
    std::string emptyAlignment(bool paired, bool mateFailed, bool upstream) const;
    std::string printUnpaired(int i) const;
    std::string printPaired(int i, bool firstMate);
};

// one input record as the reader split it: pointers into the input's memory, nothing copied yet (name = up to the first white
// space, '/1' '/2' stripped; seq and qual are `len` bytes each)
struct RawRead { const char* name; int nameLen; const char* seq; const char* qual; int len; };

// the results of one whole-read device call (alfa_map_reads_batch), in the call's own dense form; freed with the object
struct MapCallResult {
    std::vector<int32_t> st;
    std::vector<int64_t> aoff, soff, ooff;
    alfa_map_alignment* al = nullptr;
    alfa_match* sd = nullptr;
    alfa_map_cigar_op* op = nullptr;
    void (*free_fn)(void*) = nullptr;
    MapCallResult() = default;
    MapCallResult(const MapCallResult&) = delete;
    MapCallResult& operator=(const MapCallResult&) = delete;
    ~MapCallResult() { if (free_fn) { free_fn(al); free_fn(sd); free_fn(op); } }
};

// where a read's final result lies after map_raw's rounds: entry j of one of the calls, or nowhere (unmapped after the last round)
struct RawResult { const MapCallResult* R = nullptr; int j = 0; bool unmapped = false; };
// the lower-cased reads and their reverse complements map_raw made (the calls' P / Prc point into it): stays with whoever
// renders the results, which need not be the thread that made the calls
struct RawStrands {
    std::vector<char> bytes;
    std::vector<size_t> off;
    std::vector<int> len;
    const char* strand(int i, bool rc) const { return bytes.data() + off[(size_t)i] + (rc ? len[(size_t)i] : 0); }
};

struct RefIndex {                 // the parts of `sparseSA` the mapper itself touches
    const alfa_host_index* h = nullptr;
    const alfa_index_view* v = nullptr;
    const char* S() const { return (const char*)v->S; }
    int64_t K() const { return v->K; }
    void from_set(int64_t hit, int64_t& seq, int64_t& seqpos) const { alfa_host_index_from_set(h, hit, &seq, &seqpos); }
    void chrom_bounds(int64_t hit, int64_t& cs, int64_t& ce) const { alfa_host_index_chrom_bounds(h, hit, &cs, &ce); }
    const char* descr(int64_t i) const { return alfa_host_index_descr(h, (int)i); }
};

class Mapper {
   public:
    Mapper(const RefIndex& idx, const AlignOpt& a, const PairedOpt& p, HotSched& hot) : idx_(idx), opt_(a), popt_(p), hot_(hot) {}
    // map_unpaired: unpairedMatch + postprocess (src/main.cpp:87-104)
    void map_unpaired(Read& read);
    // map_paired: pairedMatch (modes 1..6) + postprocess of both mates (src/main.cpp:179-195)
    void map_paired(Read& m1, Read& m2);
    // ---- whole reads on the hot path's side (HotOps::map_reads_batch): usable() says whether this option set can go there;
    // describe() fills the per-read record (the reference's length-dependent arithmetic stays here, on the host); map_batch()
    // sends reads[0..n) (init() done) and adopts the results: done[i] = 1 when read i has its alignments (as unpaired_match would
    // have left them), 0 when it still has to be mapped here
    bool device_map_usable() const;
    void map_batch(Read* const* reads, int n, std::vector<uint8_t>& done);
    // ---- the same for records the reader only split: map_raw() lower-cases every read and its reverse complement into one
    // buffer (strand(i, rc)) and makes the call -- and then the reference's rescue rounds as further calls for the reads that
    // need them (unpairedMatch src/mapper.cpp:1290-1300 RESCUE2, calculateSeedsBothStrands :542-555 MEMRESCUE): res[i] says
    // where read i's final result lies.  A read the device finished with at most one alignment, CIGAR included, -- nearly all
    // reads -- then goes from there straight to its SAM line (render_direct: read_t::postprocess src/mapper.cpp:322-358 and
    // printUnpairedAlignment src/mapper.h:181-193 for that case, no read_t / alignment_t objects in between); every other
    // read is materialised and takes the general path (adopt() gives it the call's alignments when it has them)
    void map_raw(const RawRead* raw, int n, RawStrands& S, std::vector<std::unique_ptr<MapCallResult>>& calls, std::vector<RawResult>& res);
    bool direct_ok(const RawResult& rs) const;
    bool device_done(const RawResult& rs) const { return rs.unmapped || rs.R->st[(size_t)rs.j] == 0; }
    void render_direct(const RawStrands& S, const RawRead& rr, int i, const RawResult& rs, std::string& out, long* mappedPrinted) const;
    void adopt(Read& r, const MapCallResult& R, int i);
    void map_unpaired_rest(Read& read, bool done);      // what is left of map_unpaired after map_batch: (unpaired_match,) postprocess
    bool finish_unpaired_direct(Read& read);            // ... for a read whose CIGARs came with its alignments: no fiber needed; false = needs one
    void map_mate_rest(Read& read, bool done);          // ... of map_mate
    // paired mode 1 split into its independent parts: map_mate on each mate, pair_mates on the pair, finish_mate on each mate
    bool splittable() const { return popt_.mode == 1; }
    void map_mate(Read& r);
    void pair_mates(Read& m1, Read& m2);
    void finish_mate(Read& r);

    struct PairInfo { bool first; int pos; int score; };       // alnScoreInfo_t src/mapper.cpp:1512-1518

   private:
    struct Lis {                  // lis_t src/mapper.h:42-65
        int score = 0; bool fw = true; int64_t refBegin = 0, refEnd = 0; bool extended = false; std::vector<Match> matches;
        // --device-regions: the chain extendChain builds first, computed with the region (empty: not provided)
        std::vector<int32_t> chain0; int chain0Anchor = -1, chain0Edit = 0, chain0Coverage = 0;
    };
    void unpaired_match(Read& read);
    void seeds_both_strands(const std::string& P, const std::string& Prc, int min_len, std::vector<Match>& fw, std::vector<Match>& rc);
    void identify_and_filter(std::vector<Match>& matches, bool fw, int qLength, int editDist, std::vector<Lis>& lis);
    void cand_regions(const Read& read, std::vector<Lis>& lis, double errorPercent, bool sortByScore = true);
    bool core_trials(Read& read, const std::vector<Lis>& lis);
    void process_lis(Read& read, const std::vector<Lis>& lis, int editDist, int maxAlnCount, int minCoverage, int minScore);
    int extend_chain(const std::string& P, const std::vector<Match>& matches, int editDist, int minScore, int minCoverage,
                     int64_t chrStart, int64_t chrEnd, Read& read, bool fw, const Lis* first = nullptr);
    void left_ext(const std::string& P, int& tmpQB, int64_t& tmpRB, int64_t chrStart, int& curScore, int bandSize, int editDist);
    void right_ext(const std::string& P, int& tmpQE, int64_t& tmpRE, int64_t chrEnd, int& curScore, int bandSize, int editDist);
    // mayWait = false: returns false, with nothing submitted, when a ksw_global call would be needed (the caller is not on a fiber)
    bool postprocess(Read& read, bool paired, bool mayWait = true);
    void set_local_pos(Alignment& a);
    void paired_match1(Read& m1, Read& m2);
    void pair_mates1(Read& m1, Read& m2);
    void paired_match2(Read& m1, Read& m2);
    void paired_match3(Read& m1, Read& m2);
    void paired_match45(Read& m1, Read& m2, bool andPair);
    void paired_match6(Read& m1, Read& m2);
    struct LisCouple { int first, second, score; };            // LIScouple_t src/mapper.cpp:1845-1851
    void pair_alignment_to_window(Read& m1, Read& m2, const std::vector<PairInfo>& order, std::vector<PairInfo>& windowAlns,
                                  int& alnCount, int maxAlnCount);
    void pair_alignments_ordered(Read& m1, Read& m2, const std::vector<PairInfo>& order, int& alnCount, int maxPaired);
    void discordant_alignments(Read& m1, Read& m2);
    static std::vector<PairInfo> score_order(const Read& m1, const Read& m2);
    void pair_and_finish(Read& m1, Read& m2, bool withWindowRescue);
    int window_align(Read& windowMate, int64_t refB, int64_t refE, bool fw, int minScore);
    bool is_concordant_lis(const Lis& a, const Lis& b, int editDist1, int editDist2) const;
    void process_lis_simultaneous(Read& m1, Read& m2, const std::vector<Lis>& lis1, const std::vector<Lis>& lis2,
                                  const std::vector<LisCouple>& order, int maxAlnCount, double minCoveragePercent, int minScore1, int minScore2);
    void process_lis_both(Read& m1, Read& m2, std::vector<Lis>& lis1, std::vector<Lis>& lis2, const std::vector<PairInfo>& order,
                          int maxAlnCount, double minCoveragePercent, int minScore1, int minScore2, bool andPair);
    void find_or_align_mate(Read& alignmentMate, Read& windowMate, bool alnMateFirst, std::vector<Lis>& lisWindow, int minScore,
                            std::vector<PairInfo>& windowAlns, int alignmentIndex);
    void set_paired(Read& m1, Read& m2, int i, int j, bool concordant);
    bool is_concordant(const Alignment& a, const Alignment& b) const;
    void map_options(alfa_map_options& mo) const;
    // the per-read record of a call; errorPercent and memStage select the round (calculateCandRegions' parameter, :1224-1231; the
    // seed rescue's stage 0 / 1 / 2, :542-555)
    void describe(const char* P, const char* Prc, int Plength, alfa_map_read& x, double errorPercent, int memStage = 0) const;
    void map_call(const alfa_map_options& mo, int n, MapCallResult& R);      // rd_[0..n) described
    const RefIndex& idx_;
    const AlignOpt& opt_;
    const PairedOpt& popt_;
    HotSched& hot_;
    std::vector<alfa_map_read> rd_;       // reused from call to call
};

std::string sam_header(const RefIndex& idx, int argc, char** argv);

// the trial loop's options and per-read parameters in map_core's form (the float arithmetic of src/mapper.cpp:1224,1280-1287
// is done here, on the host)
mapcore::Options core_options(const AlignOpt& o);
mapcore::ReadParams core_read_params(const AlignOpt& o, int Plength);
extern std::atomic<long> g_map_us[5];
extern std::atomic<long> g_core_done, g_core_fallback;     // --core-trials / device mapping: reads done there, reads handed back

}  // namespace alfa
