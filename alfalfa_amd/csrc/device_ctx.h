// device_ctx.h -- per-host-thread HIP context of libalfa_hot: one stream, timing events and
// growable device / pinned-host staging buffers that are reused from call to call.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "errors.h"

namespace alfa {

#define ALFA_HIP(expr)                                                                              \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return ::alfa::fail(ALFA_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));      \
    } while (0)

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);  // keeps contents undefined; returns ALFA_OK / error
    void release();
    template <class T> T* as() const { return (T*)p; }
};
// grow-only pinned host buffer
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <class T> T* as() const { return (T*)p; }
};

struct Timing {
    double kernel_ms = 0, h2d_ms = 0, d2h_ms = 0;
    int64_t work_units = 0;
    int32_t launches = 0;
};

struct Ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipStream_t up = nullptr;        // host-to-device staging copies: waited for by the host alone, never behind running kernels
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // side streams: independent kernel chains of one batch (one per kernel bin) fork from `stream` and join it again,
    // so that a bin holding a handful of long tasks does not serialise behind (or in front of) the others
    static constexpr int NSIDE = 14;
    hipStream_t side[NSIDE] = {};    // created at first use: side_stream(k)
    hipStream_t side_stream(int k);
    hipEvent_t side_ev[NSIDE] = {};
    hipEvent_t l2_ev[6] = {};        // spare
    hipEvent_t fill_ev[3] = {};      // a group bin's fill kernel is done: its two backtrack classes may run side by side
    hipEvent_t fork_ev = nullptr;
    // lazy join (alfa_set_lazy_join): launches leave their side chains unjoined so that successive batches overlap; the
    // main stream waits for them before anything that consumes results (fetch, sync, timer stop, free)
    bool lazy_join = false, sides_pending = false;
    int join_sides();
    // staging
    DevBuf d_in, d_in2, d_tasks, d_out, d_out2, d_scratch, d_z, d_cnt, d_off, d_ivl;
    PinBuf h_in, h_in2, h_tasks, h_out, h_out2;
    Timing last;
    int ensure(int dev);  // binds the calling thread to `dev` (creates stream/events on first use)
    // A context dies with its host thread: streams, events, device and pinned staging buffers are given back (a host
    // application that calls the batch API from short-lived threads must not leak HBM per thread).  Not from the dying
    // thread's destructors, though -- HIP interposers (rocprofv3's tool library) keep thread-local state of their own that may
    // already be gone there: the thread only BURIES its objects (bury()), and the next library call on any live thread
    // releases them (thread_ctx() drains the graveyard).
    ~Ctx();
    Ctx() = default;
    Ctx(const Ctx&) = delete;
    Ctx& operator=(const Ctx&) = delete;
};

// hands an object of a dying thread to the graveyard: fn(p) runs on a live thread at the next thread_ctx() call
void bury(void (*fn)(void*), void* p);

// timing of the calling thread's most recent batch call (alfa_last_timing)
extern thread_local Timing g_last_timing;

// the calling thread's context for `device`; nullptr + last error when HIP is unusable
Ctx* thread_ctx(int device);
// number of usable devices (0 when the runtime cannot be initialised)
int device_count();

}  // namespace alfa
