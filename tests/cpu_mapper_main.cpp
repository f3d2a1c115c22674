// tests/cpu_mapper_main.cpp -- TEST INFRASTRUCTURE ONLY.
// Builds the host mapper (alfalfa_amd/csrc/mapper/*.cpp, the caller of the hot path) against the CPU oracle instead of
// libalfa_hot's HIP kernels, so that the host logic (regions, chaining, CIGAR, MAPQ, SAM) can be checked byte-for-byte
// against the reference's SAM goldens in this GPU-less container.  The product binary is alfalfa_main.cpp (HIP only).
#include "../alfalfa_amd/csrc/mapper/mapper_cli.h"
#include "../oracle/alfa_oracle.h"

int main(int argc, char** argv) {
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
