// hot_sched.h -- fiber scheduler that turns the mapper's per-read, synchronous-looking hot-path calls
// (sa.SMAM(...), ksw_extend(...), ksw_nw(...), ksw_global(...), ksw_align(...)) into large batches for libalfa_hot.
//
// The reference maps one read per kt_for work item and calls the hot path synchronously
// (src/main.cpp:87-104,126).  A GPU needs >=10^4 independent calls in flight, so every read is mapped
// inside its own stackful fiber: a hot-path call records a request and yields; when all
// fibers of a worker thread are parked the scheduler issues ONE batch call per request kind through the
// C ABI (include/alfa_hot.h), scatters the results and resumes the fibers.  The mapper code therefore
// keeps the reference's call structure (and its data-dependent control flow) unchanged.
#pragma once
#include <cstdint>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "../../../include/alfa_hot.h"

namespace alfa {

// The one seam between the host mapper and the hot path: the batch entry points of include/alfa_hot.h.
// The product binds them to libalfa_hot (HIP); tests/ may bind a checker with the same signatures.
struct HotOps {
    void* index = nullptr;   // alfa_index* (what seed_batch expects as its first argument)
    int (*seed_batch)(void*, int, const alfa_seed_query*, int, alfa_match**, int64_t*) = nullptr;
    // optional: seeds and the candidate regions built from them in one call; nullptr = the mapper builds them
    int (*seed_regions_batch)(void*, int, const alfa_seed_query*, const int32_t*, int, const alfa_region_params*, alfa_match**, int64_t*,
                              alfa_region**, int64_t*, alfa_match**, int64_t*, int32_t**, int32_t*) = nullptr;
    // optional (whole reads): seeds, regions and the trial loop with its DP in one call; nullptr = the mapper
    // runs the trial loop itself through the batch calls below
    int (*map_reads_batch)(void*, const alfa_map_options*, const alfa_dp_params*, const alfa_map_read*, int, int32_t*, int64_t*,
                           alfa_map_alignment**, int64_t*, alfa_match**, int64_t*, alfa_map_cigar_op**) = nullptr;
    int (*ksw_extend_batch)(const alfa_dp_params*, const alfa_dp_task*, int, alfa_extend_result*) = nullptr;
    int (*ksw_nw_batch)(const alfa_dp_params*, const alfa_dp_task*, int, int32_t*) = nullptr;
    int (*ksw_global_batch)(const alfa_dp_params*, const alfa_dp_task*, int, int32_t*, uint32_t**, int32_t*, int64_t*) = nullptr;
    int (*ksw_align_batch)(const alfa_dp_params*, const alfa_align_task*, int, alfa_kswr*) = nullptr;
    void (*free_fn)(void*) = nullptr;
    const char* (*last_error)() = nullptr;
};

struct Match { int64_t ref; int32_t query; int32_t len; };   // == match_t / alfa_match

struct SeedReq {
    const std::string* P; int min_len, max_count, sparse_q, mem_type, max_smem; std::vector<Match>* out;
};
// a region as identifyAndFilter + coveredBases leave it (lis_t without the bookkeeping fields)
// chain: the chain extendChain builds first in the region (member indices, nextChain's edit estimate and coverage, the anchor)
struct RegionOut { int64_t refBegin, refEnd; int score; std::vector<Match> matches; std::vector<int32_t> chain; int chainAnchor, chainEdit, chainCoverage; };
// one strand: seeds and, when the hot path could build them (*status == 0), its candidate regions in creation order
struct SeedRegionReq {
    const std::string* P; int min_len, max_count, sparse_q, mem_type, max_smem, edit_dist; alfa_region_params prm;
    std::vector<Match>* out; std::vector<RegionOut>* regions; int* status;
};
struct ExtendReq { alfa_dp_task t; alfa_extend_result* out; };
struct NwReq { alfa_dp_task t; int32_t* out; };
struct GlobalReq { alfa_dp_task t; int32_t* score; std::vector<uint32_t>* cigar; };
struct AlignReq { alfa_align_task t; alfa_kswr* out; };

struct SchedStats {
    int64_t rounds = 0, seed_calls = 0, seed_queries = 0, ext_calls = 0, ext_tasks = 0, nw_calls = 0, nw_tasks = 0, glob_calls = 0,
            glob_tasks = 0, align_calls = 0, align_tasks = 0;
    double hot_seconds = 0;   // wall time spent inside the batch entry points
};

class HotSched {
   public:
    HotSched(const HotOps& ops, const alfa_dp_params& prm, size_t stack_bytes = 256 * 1024);
    ~HotSched();
    // runs job(i) for i in [0,n) with at most `max_fibers` in flight; jobs call the submit_*/wait API below
    void run(int n, int max_fibers, const std::function<void(int)>& job);

    // ---- called from inside a job
    void submit_seed(const SeedReq& r) { seed_.push_back(r); }
    void submit_seed_regions(const SeedRegionReq& r) { seedreg_.push_back(r); }
    bool has_seed_regions() const { return ops_.seed_regions_batch != nullptr; }
    void submit_extend(const ExtendReq& r) { ext_.push_back(r); }
    void submit_nw(const NwReq& r) { nw_.push_back(r); }
    void submit_global(const GlobalReq& r) { glob_.push_back(r); }
    void submit_align(const AlignReq& r) { align_.push_back(r); }
    void wait();              // returns when every request submitted by this fiber has been answered
    const SchedStats& stats() const { return stats_; }
    const HotOps& ops() const { return ops_; }
    const alfa_dp_params& dp_params() const { return prm_; }
    void add_hot_seconds(double s) { stats_.hot_seconds += s; }
    // when set, every DP task issued is appended to <prefix>.hdr (5 x int32: kind 0 nw|1 global|2 extend, qlen, tlen, w, h0)
    // and <prefix>.seq (query bytes then target bytes); used by bench.py to replay the mapper's real task stream
    void set_task_log(const std::string& prefix);

   private:
    // Fibers switch with a few instructions (callee-saved registers + stack pointer, x86-64 SysV); ucontext's swapcontext
    // costs a signal-mask system call per switch, ~14 of them per read.
    struct Fiber { void* sp = nullptr; char* stack = nullptr; bool done = true; int job = -1; };
    static void fiber_entry();
    void prepare(Fiber& f);
    void flush();
    HotOps ops_;
    alfa_dp_params prm_;
    size_t stack_bytes_;
    std::vector<Fiber> fibers_;
    std::vector<std::pair<char*, size_t>> arenas_;   // mappings the fiber stacks were cut from
    void* main_sp_ = nullptr;
    int current_ = -1;
    const std::function<void(int)>* job_ = nullptr;
    std::vector<SeedReq> seed_;
    std::vector<SeedRegionReq> seedreg_;
    std::vector<ExtendReq> ext_;
    std::vector<NwReq> nw_;
    std::vector<GlobalReq> glob_;
    std::vector<AlignReq> align_;
    SchedStats stats_;
    FILE* log_hdr_ = nullptr;
    FILE* log_seq_ = nullptr;
    void log_task(int kind, const alfa_dp_task& t);
};

}  // namespace alfa
