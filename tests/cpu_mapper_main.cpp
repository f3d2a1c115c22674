// tests/cpu_mapper_main.cpp -- TEST INFRASTRUCTURE ONLY.
// Builds the host mapper (alfalfa_amd/csrc/mapper/*.cpp, the caller of the hot path) against the CPU oracle instead of
// libalfa_hot's HIP kernels, so that the host logic (regions, chaining, CIGAR, MAPQ, SAM) can be checked byte-for-byte
// against the reference's SAM goldens in this GPU-less container.  The product binary is alfalfa_main.cpp (HIP only).
#include "../alfalfa_amd/csrc/mapper/mapper_cli.h"
#include "../oracle/alfa_oracle.h"

#include <cstdio>
#include <cstdlib>
#include <string>
#include <algorithm>
#include <vector>

#include "../alfalfa_amd/csrc/map_core.h"

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

// `_cpu_mapper --sort-selftest`: map_core's restatement of libstdc++'s std::sort against std::sort itself, on index arrays
// whose keys are full of ties (the order of equal elements is what the restatement is for)
static int sort_selftest() {
    uint64_t x = 88172645463325252ull;
    auto rnd = [&] { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    long cases = 0;
    for (int n = 0; n <= 128; ++n)
        for (int rep = 0; rep < 400; ++rep) {
            const int kinds = 1 + (int)(rnd() % (rep % 7 == 0 ? 128 : (rep % 3 ? 4 : 20)));
            std::vector<int32_t> key((size_t)n);
            for (auto& k : key) k = (int32_t)(rnd() % (uint64_t)kinds);
            if (rep % 11 == 0) std::sort(key.begin(), key.end());
            if (rep % 13 == 0) std::sort(key.rbegin(), key.rend());
            std::vector<uint16_t> a((size_t)n), b((size_t)n);
            for (int i = 0; i < n; ++i) a[i] = b[i] = (uint16_t)i;
            const int32_t* kp = key.data();
            auto before = [kp](uint16_t u, uint16_t v) { return kp[u] > kp[v]; };
            std::sort(a.begin(), a.end(), before);
            uint32_t stk[alfa::mapcore::BAM_CAP];                  // (the stack map_read hands over: BAM_CAP / 3 entries)
            if (!alfa::mapcore::sort_like_libstdcxx(b.data(), n, stk, alfa::mapcore::BAM_CAP / 3, before)) { printf("depth limit hit at n=%d\n", n); return 1; }
            if (a != b) { printf("order differs at n=%d rep=%d\n", n, rep); return 1; }
            ++cases;
        }
    printf("sort selftest OK: %ld cases\n", cases);
    return 0;
}

int main(int argc, char** argv) {
    if (argc == 3 && std::string(argv[1]) == "--fiber-overflow-test") return overflow_test((size_t)atoll(argv[2]));
    if (argc == 2 && std::string(argv[1]) == "--sort-selftest") return sort_selftest();
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
