// device_ctx.hip -- see device_ctx.h
#include "device_ctx.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

// Hardware queues.  A DP call forks its kernel bins onto side streams (ksw_kernels.hip: launch_all) and a mapper thread adds
// its seed and upload streams; HIP multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (4 unless the
// variable says otherwise), and streams that share a queue run one after the other.  Measured on the C4 stream
// (tools/gpu_queue_ab.sh, driven through ALFA_HW_QUEUES): 12.5 ms per step with 4 queues, 10.7 with 8, 7.9 with 16, 8.1-8.2
// with 24-32; the C3 step is unchanged.
// The HOST chooses the queue count: a GPU_MAX_HW_QUEUES that is already in the environment is never touched.  Only when the
// variable is absent does loading the library install a default of 16; ALFA_HW_QUEUES=<n> is the operator's explicit request
// to this library and wins (0 = hands off altogether).  The HIP runtime reads the variable when it initialises (its first
// call), so a host that has used HIP before loading the library keeps whatever it had.  Queues of all processes on a GPU
// share its hardware slots: beyond them the driver time-slices (two 16-queue processes on one GPU: 1.6 -> 2.4 s per million
// reads end to end), so hosts that run several processes per GPU choose a lower number; the command line tool does
// (mapper/alfalfa_main.cpp, through alfa_hw_queues_defaulted()).
static int g_hw_queues_defaulted = 0;
__attribute__((constructor)) static void alfa_default_hw_queues() {
    const char* e = getenv("ALFA_HW_QUEUES");
    if (e) {
        int n = atoi(e);
        if (n <= 0) return;
        if (n > 32) n = 32;
        char buf[8];
        snprintf(buf, sizeof buf, "%d", n);
        (void)setenv("GPU_MAX_HW_QUEUES", buf, 1);
        return;
    }
    if (getenv("GPU_MAX_HW_QUEUES")) return;      // the host application or the operator has chosen
    (void)setenv("GPU_MAX_HW_QUEUES", "16", 0);
    g_hw_queues_defaulted = 1;
}
extern "C" int alfa_hw_queues_defaulted(void) { return g_hw_queues_defaulted; }

namespace alfa {

int DevBuf::reserve(size_t bytes) {
    if (bytes <= cap) return ALFA_OK;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    // growing is expensive beyond the allocation itself: hipFree waits for the whole device, i.e. for the kernels of every
    // other host thread (milliseconds each on long reads), so a buffer doubles (up to 256 MiB of headroom) rather than creep
    size_t want = bytes + std::min<size_t>(bytes, (size_t)256 << 20) + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        p = nullptr;
        return fail(ALFA_ENOMEM, std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
    }
    cap = want;
    return ALFA_OK;
}
void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
}
int PinBuf::reserve(size_t bytes) {
    if (bytes <= cap) return ALFA_OK;
    if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
    size_t want = bytes + std::min<size_t>(bytes, (size_t)64 << 20) + 256;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        p = nullptr;
        return fail(ALFA_ENOMEM, std::string("hipHostMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
    }
    cap = want;
    return ALFA_OK;
}
void PinBuf::release() {
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
}

int device_count() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

namespace {
struct Dead { void (*fn)(void*); void* p; };
std::mutex g_grave_mu;
std::vector<Dead> g_grave;
void drain_graveyard() {
    std::vector<Dead> dead;
    {
        std::lock_guard<std::mutex> g(g_grave_mu);
        dead.swap(g_grave);
    }
    for (const Dead& d : dead) d.fn(d.p);
}
}  // namespace

void bury(void (*fn)(void*), void* p) {
    std::lock_guard<std::mutex> g(g_grave_mu);
    g_grave.push_back(Dead{fn, p});
}

Ctx::~Ctx() {
    if (!stream) return;
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return; }
    (void)hipStreamSynchronize(stream);
    for (auto& st : side) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : side_ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : fill_ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : l2_ev) if (e) (void)hipEventDestroy(e);
    if (fork_ev) (void)hipEventDestroy(fork_ev);
    for (DevBuf* b : {&d_in, &d_in2, &d_tasks, &d_out, &d_out2, &d_scratch, &d_z, &d_cnt, &d_off, &d_ivl}) b->release();
    for (PinBuf* b : {&h_in, &h_in2, &h_tasks, &h_out, &h_out2}) b->release();
    if (up) (void)hipStreamDestroy(up);
    (void)hipStreamDestroy(stream);
    stream = nullptr;
}

// STREAM PRIORITIES ARE PROHIBITED HERE: every stream of a context is created with the default priority, and must stay so.
// Round 2 tried a higher priority for `stream` (the seed kernels) than for the DP side chains and the C4 bench stopped making
// progress ("the stream-priority hang").  The library's kernels never wait on each other (no spin loops,
// no inter-workgroup flags; the only device-side waits are hipCUB's look-back scans, inside one in-order dispatch), so what can
// fail to be satisfied is a CROSS-STREAM wait: join_sides() and the fill_ev / fork_ev edges put barrier packets on one stream
// that wait for events of another.  With one priority level all those queues are scheduled round-robin and every wait is
// eventually satisfied.  With two levels the waiting packet sits at the head of a HIGH-priority hardware queue, which the
// scheduler keeps runnable in its strict upper band, while the signal has to come from a NORMAL-priority queue: HIP keeps a
// pool of GPU_MAX_HW_QUEUES queues PER priority level (16 + 16 here, plus torch's), which oversubscribes the hardware queue
// slots, and under oversubscription a lower band is only mapped when the upper band is idle -- it never is: priority inversion.
// alfa_stream_priority_span() lets a test check the rule (tests/test_gpu_ksw.py).
// The context is usable only when EVERY stream and event exists: a failure part-way tears down what was created, so the
// next call starts from scratch instead of running on a half-built context.
int Ctx::ensure(int dev) {
    int n = device_count();
    if (n <= 0) return fail(ALFA_ENODEV, "no HIP device is usable; libalfa_hot has no CPU fallback");
    if (dev < 0 || dev >= n) return fail(ALFA_EINVAL, "device ordinal out of range");
    ALFA_HIP(hipSetDevice(dev));
    if (!stream) {
        device = dev;
        hipStream_t main_stream = nullptr;
        hipError_t err = hipStreamCreateWithFlags(&main_stream, hipStreamNonBlocking);
        auto ok = [&](hipError_t e) { if (err == hipSuccess) err = e; return err == hipSuccess; };
        (void)ok(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
        for (auto& e : ev) if (!ok(hipEventCreate(&e))) break;
        for (auto& e : side_ev) if (!ok(hipEventCreateWithFlags(&e, hipEventDisableTiming))) break;
        for (auto& e : fill_ev) if (!ok(hipEventCreateWithFlags(&e, hipEventDisableTiming))) break;
        for (auto& e : l2_ev) if (!ok(hipEventCreateWithFlags(&e, hipEventDisableTiming))) break;
        (void)ok(hipEventCreateWithFlags(&fork_ev, hipEventDisableTiming));
        if (err != hipSuccess) {
            for (auto& st : side) if (st) { (void)hipStreamDestroy(st); st = nullptr; }
            for (auto& e : ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
            for (auto& e : side_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
            for (auto& e : fill_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
            for (auto& e : l2_ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
            if (fork_ev) { (void)hipEventDestroy(fork_ev); fork_ev = nullptr; }
            if (up) { (void)hipStreamDestroy(up); up = nullptr; }
            if (main_stream) (void)hipStreamDestroy(main_stream);
            return fail(ALFA_EHIP, std::string("creating the context's streams and events: ") + hipGetErrorString(err));
        }
        stream = main_stream;          // set last: a context with a stream is complete
    }
    return ALFA_OK;
}

// Side streams are made at the first use of any of them, all NSIDE at once and in index order: a thread that only maps whole
// reads (alfa_map_reads_batch: one stream) never needs them -- 16 such threads would otherwise hold 224 idle streams, which
// HIP's shutdown then takes 0.1 s to destroy -- and for a thread that does, creation order decides which streams share a
// hardware queue (HIP deals them out round-robin), which the assignment of kernel bins to side streams was tuned for
// (ksw_kernels.hip launch_all).
hipStream_t Ctx::side_stream(int k) {
    if (!side[0]) {
        for (int j = 0; j < NSIDE; ++j) {
            if (hipStreamCreateWithFlags(&side[j], hipStreamNonBlocking) != hipSuccess) {
                (void)hipGetLastError();
                for (int i = 0; i <= j; ++i) { if (side[i]) (void)hipStreamDestroy(side[i]); side[i] = nullptr; }
                return stream;       // (out of streams: the chain runs on the main stream, in order -- slower, never wrong)
            }
        }
    }
    return side[k];
}

int Ctx::join_sides() {
    if (!sides_pending) return ALFA_OK;
    for (int k = 0; k < NSIDE; ++k) ALFA_HIP(hipStreamWaitEvent(stream, side_ev[k], 0));
    sides_pending = false;
    return ALFA_OK;
}

Ctx* thread_ctx(int device) {
    // one context per (thread, device); contexts live until the thread exits, then wait in the graveyard for a live thread
    struct Mine {
        std::map<int, Ctx*> by_device;
        ~Mine() { for (auto& kv : by_device) bury([](void* p) { delete (Ctx*)p; }, kv.second); }
    };
    thread_local Mine mine;
    drain_graveyard();
    Ctx*& slot = mine.by_device[device];
    if (!slot) slot = new Ctx();
    if (slot->ensure(device) != ALFA_OK) return nullptr;
    return slot;
}

}  // namespace alfa

extern "C" int alfa_device_count(void) { return alfa::device_count(); }

namespace alfa { thread_local Timing g_last_timing; }

extern "C" int alfa_last_timing(alfa_timing* out) {
    if (!out) return alfa::fail(ALFA_EINVAL, "null argument");
    out->kernel_ms = alfa::g_last_timing.kernel_ms;
    out->h2d_ms = alfa::g_last_timing.h2d_ms;
    out->d2h_ms = alfa::g_last_timing.d2h_ms;
    out->work_units = alfa::g_last_timing.work_units;
    out->launches = alfa::g_last_timing.launches;
    return ALFA_OK;
}

extern "C" void alfa_free(void* p) { free(p); }

// lowest and highest HIP priority among the streams of the calling thread's context for `device` (they must be equal, see
// Ctx::ensure)
extern "C" int alfa_stream_priority_span(int device, int* lo, int* hi) {
    if (!lo || !hi) return alfa::fail(ALFA_EINVAL, "null argument");
    alfa::Ctx* cx = alfa::thread_ctx(device);
    if (!cx) return ALFA_ENODEV;
    int mn = 0, mx = 0;
    bool first = true;
    auto see = [&](hipStream_t st) -> hipError_t {
        int p = 0;
        hipError_t e = hipStreamGetPriority(st, &p);
        if (e != hipSuccess) return e;
        if (first) { mn = mx = p; first = false; }
        mn = std::min(mn, p); mx = std::max(mx, p);
        return hipSuccess;
    };
    ALFA_HIP(see(cx->stream));
    ALFA_HIP(see(cx->up));
    for (int k = 0; k < alfa::Ctx::NSIDE; ++k) ALFA_HIP(see(cx->side_stream(k)));      // (makes the ones not used yet: the rule covers them all)
    *lo = mn; *hi = mx;
    return ALFA_OK;
}
