// alfalfa_main.cpp -- `alfalfa_amd`: the reference's `alfalfa index|align` commands on top of libalfa_hot (HIP only).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <malloc.h>
#include <string>
#include <thread>
#include <unistd.h>

#include "mapper_cli.h"

// ALFA_ALIAS_DEVICES=1 maps device ordinals beyond the visible ones back onto them (ordinal % count): `--gpus G` then runs G
// worker groups, each with its own index copy and contexts, on fewer cards -- how the multi-GPU path is exercised on a one-GPU box
static int physical(int device) {
    static const bool alias = getenv("ALFA_ALIAS_DEVICES") != nullptr;
    const int n = alfa_device_count();
    return (alias && n > 0) ? device % n : device;
}

int main(int argc, char** argv) {
    // When nobody has chosen a hardware-queue count, loading libalfa_hot installs 16 (device_ctx.hip), which pays when one
    // caller overlaps large batches of every ksw kind.  The mapper's threads issue small synchronous batches, and several
    // mapper processes may share a GPU -- once the queues of all processes on a GPU exceed its hardware slots the driver
    // time-slices them -- so the tool replaces that default (and only that: a GPU_MAX_HW_QUEUES from the environment or an
    // ALFA_HW_QUEUES request stands) with HIP's own default of 4.  Nothing has initialised HIP yet at this point, and the
    // process is still single-threaded.  --stats prints the effective value.
    if (alfa_hw_queues_defaulted()) (void)setenv("GPU_MAX_HW_QUEUES", "4", 1);
    // 16 workers build and drop reads, task copies and CIGAR vectors of 10-100 KiB all the time: keep such blocks in the heap
    // (glibc would mmap/munmap each one above 128 KiB and trim the heap top after frees; every fresh page then faults under the
    // address-space lock all threads share).  Long reads: map phase 1.7 -> 1.37 s per 50 000 reads.
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_TOP_PAD, 64 << 20);
    if (const char* m = getenv("ALFA_SYNC")) {            // how a host thread waits for the GPU: block (sleep), yield, spin; default: HIP's choice
        const unsigned f = m[0] == 'b' ? hipDeviceScheduleBlockingSync : m[0] == 'y' ? hipDeviceScheduleYield : m[0] == 's' ? hipDeviceScheduleSpin : hipDeviceScheduleAuto;
        const int nd = alfa_device_count();
        for (int d = 0; d < nd; ++d) { (void)hipSetDevice(d); (void)hipSetDeviceFlags(f); }
        if (nd > 0) (void)hipSetDevice(0);
    }
    // HIP's start-up (50 ms: runtime, device, code objects) has nothing to wait for: it runs beside the index load (C4: 60 ms of
    // file reads) on a thread of its own; make_ops joins it before the first upload
    static std::thread hip_start;
    if (argc > 1 && std::string(argv[1]) == "align") hip_start = std::thread([] { (void)hipFree(nullptr); });
    alfa::HotBinding b;
    static auto fill_ops = [](alfa_index* dix, alfa::HotOps* ops) {
        ops->index = dix;
        ops->seed_batch = [](void* ix, int type, const alfa_seed_query* q, int nq, alfa_match** m, int64_t* off) {
            return alfa_seed_batch((alfa_index*)ix, type, q, nq, m, off);
        };
        ops->seed_regions_batch = [](void* ix, int type, const alfa_seed_query* q, const int32_t* ed, int nq, const alfa_region_params* p, alfa_match** m,
                                     int64_t* off, alfa_region** reg, int64_t* roff, alfa_match** mem, int64_t* boff, int32_t** chn, int32_t* st) {
            return alfa_seed_regions_batch((alfa_index*)ix, type, q, ed, nq, p, m, off, reg, roff, mem, boff, chn, st);
        };
        ops->map_reads_batch = [](void* ix, const alfa_map_options* o, const alfa_dp_params* dp, const alfa_map_read* r, int n, int32_t* st, int64_t* aoff,
                                  alfa_map_alignment** al, int64_t* soff, alfa_match** sd, int64_t* ooff, alfa_map_cigar_op** op) {
            return alfa_map_reads_batch((alfa_index*)ix, o, dp, r, n, st, aoff, al, soff, sd, ooff, op);
        };
        ops->ksw_extend_batch = alfa_ksw_extend_batch;
        ops->ksw_nw_batch = alfa_ksw_nw_batch;
        ops->ksw_global_batch = alfa_ksw_global_batch;
        ops->ksw_align_batch = alfa_ksw_align_batch;
        ops->free_fn = alfa_free;
        ops->last_error = alfa_last_error;
    };
    b.make_ops = [](int device, const alfa_host_index* hidx, alfa::HotOps* ops) {
        if (hip_start.joinable()) hip_start.join();
        alfa_index* dix = nullptr;
        int rc = alfa_index_upload(alfa_host_index_view(hidx), physical(device), &dix);
        if (rc) return rc;
        fill_ops(dix, ops);
        return 0;
    };
    b.load_ops = [](const char* prefix, int device, alfa_host_index** hidx, alfa::HotOps* ops) {
        if (hip_start.joinable()) hip_start.join();
        alfa_index* dix = nullptr;
        int rc = alfa_index_load_to_device(prefix, physical(device), hidx, &dix);
        if (rc) return rc;
        fill_ops(dix, ops);
        return 0;
    };
    b.release_ops = [](alfa::HotOps* ops) { alfa_index_free((alfa_index*)ops->index); };
    b.bind_thread = [](int device) { (void)hipSetDevice(physical(device)); };
    b.last_error = alfa_last_error;
    b.index_build_on_device = [](int device) { return alfa_index_build_on_device(physical(device)); };
    const int rc = alfa::mapper_main(argc, argv, b);
    if (hip_start.joinable()) hip_start.join();
    if (getenv("ALFA_TEARDOWN_CLOCK")) {      // (with --stats' #CLOCK line: how long the mapper's own teardown took, and what is left for HIP's)
        timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        fprintf(stderr, "#CLOCK main_returns %.3f\n", (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec);
    }
    return rc;
}
