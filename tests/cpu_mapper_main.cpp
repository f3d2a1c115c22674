// tests/cpu_mapper_main.cpp -- TEST INFRASTRUCTURE ONLY.
// Builds the host mapper (alfalfa_amd/csrc/mapper/*.cpp, the caller of the hot path) against the CPU oracle instead of
// libalfa_hot's HIP kernels, so that the host logic (regions, chaining, CIGAR, MAPQ, SAM) can be checked byte-for-byte
// against the reference's SAM goldens in this GPU-less container.  The product binary is alfalfa_main.cpp (HIP only).
#include "../alfalfa_amd/csrc/mapper/mapper_cli.h"
#include "../oracle/alfa_oracle.h"

#include <cstdio>
#include <cstdlib>
#include <string>

// `_cpu_mapper --fiber-overflow-test <bytes>`: one fiber whose frames use <bytes> of stack and then yield once, next to
// 31 well-behaved ones; a frame chain beyond the stack size must be REPORTED (the canary zone, exit by abort) or FAULT on the
// gap in front of its group of stacks -- never corrupt a neighbour silently (tests/test_mapper_cpu.py)
static int burn(alfa::HotSched& hs, size_t bytes) {
    volatile char pad[1024];
    for (size_t i = 0; i < sizeof(pad); i += 64) pad[i] = (char)i;
    int r = bytes > sizeof(pad) ? burn(hs, bytes - sizeof(pad)) : (hs.wait(), 0);
    return r + pad[64];
}
static int overflow_test(size_t bytes) {
    alfa::HotOps ops;
    alfa_dp_params prm{};
    alfa::HotSched hs(ops, prm, 64 * 1024);
    hs.run(32, 32, [&](int i) { if (i == 17) (void)burn(hs, bytes); else hs.wait(); });
    printf("no overflow detected\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc == 3 && std::string(argv[1]) == "--fiber-overflow-test") return overflow_test((size_t)atoll(argv[2]));
    alfa::HotBinding b;
    b.make_ops = [](int, const alfa_host_index* hidx, alfa::HotOps* ops) {
        ops->index = (void*)alfa_host_index_view(hidx);
        ops->seed_batch = [](void* v, int type, const alfa_seed_query* q, int nq, alfa_match** m, int64_t* off) {
            return orc_seed_batch((const alfa_index_view*)v, type, q, nq, m, off);
        };
        ops->ksw_extend_batch = orc_ksw_extend_batch;
        ops->ksw_nw_batch = orc_ksw_nw_batch;
        ops->ksw_global_batch = orc_ksw_global_batch;
        ops->ksw_align_batch = orc_ksw_align_batch;
        ops->free_fn = orc_free;
        ops->last_error = [] { return "oracle"; };
        return 0;
    };
    b.release_ops = [](alfa::HotOps*) {};
    b.bind_thread = [](int) {};
    b.last_error = [] { return "oracle"; };
    return alfa::mapper_main(argc, argv, b);
}
