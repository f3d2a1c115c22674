// dpband_kernels.hip -- dynProg::dpBand* batch entry point (src/dp.cpp:648-1040; dead code in the reference's
// release build).  Kernels not written yet: the entry point fails loudly.
#include "device_ctx.h"

extern "C" int alfa_dp_band_batch(int variant, const alfa_dpband_scores* s, const alfa_dpband_task* t, int n, alfa_dpband_result* r,
                                  char** cig_chars, int32_t** cig_lens) {
    (void)variant; (void)s; (void)t; (void)n; (void)r; (void)cig_chars; (void)cig_lens;
    return alfa::fail(ALFA_ENOTSUP, "alfa_dp_band_batch: the dynProg (dp.cpp) kernels are not implemented yet");
}
