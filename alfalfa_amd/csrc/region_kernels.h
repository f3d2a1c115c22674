// region_kernels.h -- launcher of the candidate-region kernel (region_kernels.hip) for seed_kernels.hip's entry point
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/alfa_hot.h"

namespace alfa {

constexpr int REGION_MATCH_CAP = 512;     // matches of one read strand the kernel handles (more: the host builds the regions)
constexpr int REGION_CAP = 128;           // regions of one read strand

struct RegionParams { int64_t max_smem_occ, max_region_mems, min_mem_reg_length; double error_percent; };

// strand i: matches d_m[d_base[i] .. d_base[i+1]); its regions go to d_reg[d_base[i] ..], their members to d_rm[2*d_base[i] + 8*i ..]
// and every region's first chain (member indices) to the same positions of d_chain
void launch_regions(hipStream_t st, const alfa_match* d_m, const int64_t* d_base, const int32_t* d_qlen, const int32_t* d_edit, int nq,
                    const RegionParams& rp, alfa_region* d_reg, alfa_match* d_rm, int32_t* d_chain, int32_t* d_nreg, int32_t* d_nrm, int32_t* d_status);

}  // namespace alfa
