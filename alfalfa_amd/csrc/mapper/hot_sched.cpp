// hot_sched.cpp -- see hot_sched.h
#include "hot_sched.h"

#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

#if !defined(__x86_64__)
#error "hot_sched.cpp: the fiber switch is written for x86-64 (System V ABI)"
#endif

// void alfa_ctx_switch(void** save_sp, void* load_sp): saves the callee-saved registers on the current stack, stores the
// stack pointer, loads the other stack pointer and restores that context's registers; returns on the other stack.
extern "C" void alfa_ctx_switch(void** save_sp, void* load_sp);
asm(R"(
    .text
    .globl alfa_ctx_switch
    .type alfa_ctx_switch,@function
alfa_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size alfa_ctx_switch, .-alfa_ctx_switch
)");

namespace alfa {

static thread_local HotSched* tl_sched = nullptr;      // the scheduler whose fibers run on this thread
// Fiber stacks come out of arenas (one mapping per batch of new fibers) so that the kernel can back them with huge pages -- a
// guard page below EVERY stack split the mapping into 4 KiB-paged islands and cost 17 % of the map phase (1.40 -> 1.65 s per
// 10^6 reads).  Layout: groups of GROUP stacks, each group preceded by an inaccessible gap; gaps and groups are multiples of
// 2 MiB and the arena is 2 MiB-aligned, so every group is a VMA of whole huge pages (the gaps cost address space only).  An
// overflow therefore runs at most GROUP - 1 stacks down before it faults, and each stack also carries a poisoned zone of
// CANARY_WORDS words at its bottom that is checked every time its fiber gives control back: an overflow into a neighbour is
// reported with the read's number, not silent (a frame would have to skip all 64 bytes to escape).
// Depth: the deepest mapper path (map_paired -> paired_match45 -> process_lis_both -> extend_chain -> left_ext -> wait) holds
// < 4 KiB of frames; the vectors live on the heap.  256 KiB per stack is 60x that.
static constexpr size_t HUGE = (size_t)2 << 20;
static constexpr int GROUP = 16;
static constexpr int CANARY_WORDS = 8;
static constexpr uint64_t CANARY = 0x616c66615f667372ull;
static inline void poison(char* stack) { for (int k = 0; k < CANARY_WORDS; ++k) ((uint64_t*)stack)[k] = CANARY + (uint64_t)k; }
static inline bool intact(const char* stack) {
    uint64_t bad = 0;
    for (int k = 0; k < CANARY_WORDS; ++k) bad |= ((const uint64_t*)stack)[k] ^ (CANARY + (uint64_t)k);
    return bad == 0;
}

HotSched::HotSched(const HotOps& ops, const alfa_dp_params& prm, size_t stack_bytes) : ops_(ops), prm_(prm), stack_bytes_(stack_bytes) {}

void HotSched::set_task_log(const std::string& prefix) {
    log_hdr_ = fopen((prefix + ".hdr").c_str(), "wb");
    log_seq_ = fopen((prefix + ".seq").c_str(), "wb");
    if (!log_hdr_ || !log_seq_) {
        fprintf(stderr, "ERROR: cannot write the task log %s.{hdr,seq}\n", prefix.c_str());
        exit(1);
    }
}

void HotSched::log_task(int kind, const alfa_dp_task& t) {
    int32_t h[5] = {kind, t.qlen, t.tlen, t.w, t.h0};
    fwrite(h, 4, 5, log_hdr_);
    fwrite(t.query, 1, (size_t)t.qlen, log_seq_);
    fwrite(t.target, 1, (size_t)t.tlen, log_seq_);
}

HotSched::~HotSched() {
    if (log_hdr_) fclose(log_hdr_);
    if (log_seq_) fclose(log_seq_);
    for (auto& a : arenas_) munmap(a.first, a.second);
}

void HotSched::fiber_entry() {
    HotSched* self = tl_sched;
    {
        Fiber& f = self->fibers_[self->current_];
        // an exception cannot unwind past this frame (its return address is a dummy): report it and stop
        try {
            (*self->job_)(f.job);
        } catch (const std::exception& e) {
            fprintf(stderr, "ERROR: exception while mapping read %d: %s\n", f.job, e.what());
            exit(1);
        } catch (...) {
            fprintf(stderr, "ERROR: unknown exception while mapping read %d\n", f.job);
            exit(1);
        }
        f.done = true;
    }
    Fiber& f = self->fibers_[self->current_];
    alfa_ctx_switch(&f.sp, self->main_sp_);     // never resumed: a finished fiber is prepared afresh for its next job
    abort();
}

// initial frame: six zeroed callee-saved registers, then the entry point as the return address of alfa_ctx_switch; the
// entry function starts with the stack 8 bytes off 16-byte alignment, as after a call
void HotSched::prepare(Fiber& f) {
    uintptr_t top = ((uintptr_t)f.stack + stack_bytes_) & ~(uintptr_t)15;
    void** p = (void**)(top - 64);
    for (int k = 0; k < 6; ++k) p[k] = nullptr;
    p[6] = (void*)&HotSched::fiber_entry;
    p[7] = nullptr;                              // fake return address of the entry function
    f.sp = (void*)p;
}

void HotSched::wait() {
    Fiber& f = fibers_[current_];
    alfa_ctx_switch(&f.sp, main_sp_);
}

static void die(const HotOps& ops, const char* what, int rc) {
    fprintf(stderr, "ERROR: %s failed (%d): %s\n", what, rc, ops.last_error ? ops.last_error() : "");
    exit(1);
}

void HotSched::flush() {
    auto t0 = std::chrono::steady_clock::now();
    stats_.rounds++;
    if (!seed_.empty()) {
        // group by seed type (one batch call per type)
        for (int type = 0; type < 4; ++type) {
            std::vector<alfa_seed_query> q;
            std::vector<size_t> who;
            for (size_t i = 0; i < seed_.size(); ++i)
                if (seed_[i].mem_type == type) {
                    const SeedReq& r = seed_[i];
                    q.push_back(alfa_seed_query{r.P->data(), (int32_t)r.P->size(), r.min_len, r.max_count, r.sparse_q, r.max_smem});
                    who.push_back(i);
                }
            if (q.empty()) continue;
            alfa_match* m = nullptr;
            std::vector<int64_t> off(q.size() + 1);
            int rc = ops_.seed_batch(ops_.index, type, q.data(), (int)q.size(), &m, off.data());
            if (rc) die(ops_, "alfa_seed_batch", rc);
            for (size_t k = 0; k < who.size(); ++k) {
                std::vector<Match>* out = seed_[who[k]].out;
                static_assert(sizeof(Match) == sizeof(alfa_match), "Match == alfa_match");
                out->insert(out->end(), (const Match*)(m + off[k]), (const Match*)(m + off[k + 1]));
            }
            ops_.free_fn(m);
            stats_.seed_calls++;
            stats_.seed_queries += (int64_t)q.size();
        }
        seed_.clear();
    }
    if (!seedreg_.empty()) {
        // one call per (seed type, region parameters): the parameters are per mapper, so per type in practice
        for (int type = 0; type < 4; ++type) {
            std::vector<alfa_seed_query> q;
            std::vector<int32_t> ed;
            std::vector<size_t> who;
            for (size_t i = 0; i < seedreg_.size(); ++i)
                if (seedreg_[i].mem_type == type) {
                    const SeedRegionReq& r = seedreg_[i];
                    q.push_back(alfa_seed_query{r.P->data(), (int32_t)r.P->size(), r.min_len, r.max_count, r.sparse_q, r.max_smem});
                    ed.push_back(r.edit_dist);
                    who.push_back(i);
                }
            if (q.empty()) continue;
            const size_t nq = q.size();
            alfa_match *m = nullptr, *mem = nullptr;
            alfa_region* reg = nullptr;
            int32_t* chn = nullptr;
            std::vector<int64_t> off(nq + 1), roff(nq + 1), boff(nq + 1);
            std::vector<int32_t> st(nq);
            int rc = ops_.seed_regions_batch(ops_.index, type, q.data(), ed.data(), (int)nq, &seedreg_[who[0]].prm, &m, off.data(), &reg, roff.data(),
                                             &mem, boff.data(), &chn, st.data());
            if (rc) die(ops_, "alfa_seed_regions_batch", rc);
            for (size_t k = 0; k < nq; ++k) {
                const SeedRegionReq& r = seedreg_[who[k]];
                static_assert(sizeof(Match) == sizeof(alfa_match), "Match == alfa_match");
                r.out->resize((size_t)(off[k + 1] - off[k]));
                if (off[k + 1] > off[k]) memcpy((void*)r.out->data(), m + off[k], sizeof(Match) * (size_t)(off[k + 1] - off[k]));
                *r.status = st[k];
                r.regions->clear();
                if (st[k] == 0) {
                    r.regions->resize((size_t)(roff[k + 1] - roff[k]));
                    for (int64_t j = roff[k]; j < roff[k + 1]; ++j) {
                        RegionOut& o = (*r.regions)[(size_t)(j - roff[k])];
                        o.refBegin = reg[j].ref_begin; o.refEnd = reg[j].ref_end; o.score = reg[j].score;
                        o.matches.resize((size_t)reg[j].count);
                        memcpy((void*)o.matches.data(), mem + boff[k] + reg[j].first, sizeof(Match) * (size_t)reg[j].count);
                        o.chain.assign(chn + boff[k] + reg[j].first, chn + boff[k] + reg[j].first + reg[j].chain_count);
                        o.chainAnchor = reg[j].chain_anchor; o.chainEdit = reg[j].chain_edit; o.chainCoverage = reg[j].chain_coverage;
                    }
                }
            }
            ops_.free_fn(m); ops_.free_fn(reg); ops_.free_fn(mem); ops_.free_fn(chn);
            stats_.seed_calls++;
            stats_.seed_queries += (int64_t)nq;
        }
        seedreg_.clear();
    }
    if (!nw_.empty()) {
        std::vector<alfa_dp_task> t(nw_.size());
        std::vector<int32_t> s(nw_.size());
        for (size_t i = 0; i < nw_.size(); ++i) { t[i] = nw_[i].t; if (log_hdr_) log_task(0, t[i]); }
        int rc = ops_.ksw_nw_batch(&prm_, t.data(), (int)t.size(), s.data());
        if (rc) die(ops_, "alfa_ksw_nw_batch", rc);
        for (size_t i = 0; i < nw_.size(); ++i) *nw_[i].out = s[i];
        stats_.nw_calls++; stats_.nw_tasks += (int64_t)nw_.size();
        nw_.clear();
    }
    if (!ext_.empty()) {
        std::vector<alfa_dp_task> t(ext_.size());
        std::vector<alfa_extend_result> r(ext_.size());
        for (size_t i = 0; i < ext_.size(); ++i) { t[i] = ext_[i].t; if (log_hdr_) log_task(2, t[i]); }
        int rc = ops_.ksw_extend_batch(&prm_, t.data(), (int)t.size(), r.data());
        if (rc) die(ops_, "alfa_ksw_extend_batch", rc);
        for (size_t i = 0; i < ext_.size(); ++i) *ext_[i].out = r[i];
        stats_.ext_calls++; stats_.ext_tasks += (int64_t)ext_.size();
        ext_.clear();
    }
    if (!glob_.empty()) {
        std::vector<alfa_dp_task> t(glob_.size());
        std::vector<int32_t> s(glob_.size()), nc(glob_.size());
        std::vector<int64_t> off(glob_.size() + 1);
        for (size_t i = 0; i < glob_.size(); ++i) { t[i] = glob_[i].t; if (log_hdr_) log_task(1, t[i]); }
        uint32_t* cig = nullptr;
        int rc = ops_.ksw_global_batch(&prm_, t.data(), (int)t.size(), s.data(), &cig, nc.data(), off.data());
        if (rc) die(ops_, "alfa_ksw_global_batch", rc);
        for (size_t i = 0; i < glob_.size(); ++i) {
            *glob_[i].score = s[i];
            glob_[i].cigar->clear();
            if (nc[i] > 0) glob_[i].cigar->assign(cig + off[i], cig + off[i] + nc[i]);
        }
        ops_.free_fn(cig);
        stats_.glob_calls++; stats_.glob_tasks += (int64_t)glob_.size();
        glob_.clear();
    }
    if (!align_.empty()) {
        // ksw_align is called with alnOptions.openGap / extendGap as they are (src/mapper.cpp:1572,2197): negative numbers,
        // the opposite sign of what the other three calls get
        alfa_dp_params ap = prm_;
        ap.gapo = -prm_.gapo; ap.gape = -prm_.gape;
        std::vector<alfa_align_task> t(align_.size());
        std::vector<alfa_kswr> r(align_.size());
        for (size_t i = 0; i < align_.size(); ++i) t[i] = align_[i].t;
        int rc = ops_.ksw_align_batch(&ap, t.data(), (int)t.size(), r.data());
        if (rc) die(ops_, "alfa_ksw_align_batch", rc);
        for (size_t i = 0; i < align_.size(); ++i) *align_[i].out = r[i];
        stats_.align_calls++; stats_.align_tasks += (int64_t)align_.size();
        align_.clear();
    }
    stats_.hot_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void HotSched::run(int n, int max_fibers, const std::function<void(int)>& job) {
    if (n <= 0) return;
    job_ = &job;
    const int nf = std::min(n, std::max(1, max_fibers));
    if ((int)fibers_.size() < nf) {
        const size_t add = (size_t)nf - fibers_.size();
        const size_t groups = (add + GROUP - 1) / GROUP;
        const size_t gbytes = ((size_t)GROUP * stack_bytes_ + HUGE - 1) / HUGE * HUGE;     // one group of stacks, whole huge pages
        const size_t bytes = HUGE + groups * (HUGE + gbytes);                             // alignment slack + (gap, group) pairs
        char* raw = (char*)mmap(nullptr, bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (raw == (char*)MAP_FAILED) throw std::runtime_error("mmap of the fiber stacks failed");
        char* base = (char*)(((uintptr_t)raw + HUGE - 1) / HUGE * HUGE);
        arenas_.push_back({raw, bytes});
        for (size_t g = 0; g < groups; ++g) {
            char* grp = base + g * (HUGE + gbytes) + HUGE;      // the gap in front of it stays PROT_NONE
            if (mprotect(grp, gbytes, PROT_READ | PROT_WRITE) != 0) throw std::runtime_error("mprotect of the fiber stacks failed");
            (void)madvise(grp, gbytes, MADV_HUGEPAGE);
            for (int k = 0; k < GROUP && fibers_.size() < (size_t)nf; ++k) {
                Fiber f;
                f.stack = grp + (size_t)k * stack_bytes_;
                poison(f.stack);
                fibers_.push_back(f);
            }
        }
    }
    tl_sched = this;
    int next = 0, live = 0;
    auto start = [&](int fi) {
        Fiber& f = fibers_[fi];
        f.done = false;
        f.job = next++;
        prepare(f);
        live++;
    };
    for (int fi = 0; fi < nf; ++fi) start(fi);
    while (live > 0) {
        // resume every live fiber once: each either finishes or parks in wait() after submitting requests
        for (int fi = 0; fi < nf; ++fi) {
            Fiber& f = fibers_[fi];
            if (f.done) {
                if (next >= n) continue;
                start(fi);               // recycle the fiber for the next read
            }
            for (;;) {
                current_ = fi;
                alfa_ctx_switch(&main_sp_, f.sp);
                if (!intact(f.stack)) {
                    fprintf(stderr, "ERROR: the stack of a mapping fiber overflowed (read %d; stack %zu bytes)\n", f.job, stack_bytes_);
                    abort();
                }
                if (!f.done) break;      // parked in wait()
                live--;
                if (next >= n) break;
                start(fi);               // finished: take the next read right away
            }
        }
        flush();
    }
    current_ = -1;
    job_ = nullptr;
}

}  // namespace alfa
