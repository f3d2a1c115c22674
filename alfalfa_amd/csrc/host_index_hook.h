// host_index_hook.h -- how the host index builder (host_index.cpp, also linked into CPU-only tools) hands the construction
// of its arrays to the GPU builder (index_build.hip, libalfa_hot only): a function pointer that stays null without HIP.
#pragma once
#include <cstdint>
#include <vector>

namespace alfa {

struct DeviceBuildIO {
    const uint8_t* S;            // padded text (init(): '$' to a multiple of K plus K more), N bytes, 16 readable bytes behind
    int64_t N, K;
    uint32_t* SA;                // n = N / K entries each
    uint8_t* LCP;
    int32_t* CHILD;              // null: not wanted
    uint32_t* KMR;               // 2 * 4^10 words; null: not wanted
    int32_t* ISA;                // null: not wanted
    std::vector<int64_t>* Midx;  // overflow table of LCP, sorted by rank
    std::vector<int32_t>* Mval;
};

extern int (*g_device_construct)(int device, DeviceBuildIO* io);
extern int g_build_device;

}  // namespace alfa
