// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE for the seed kernels' access pattern (MI355X_MICROARCH.md: the
// counter is calibrated only for wide streaming reads; "calibrate on a known byte count in your own access pattern").
// Kernel gather<W>: every lane reads ONE W-byte word at a pseudo-random W-aligned offset of a 2 GiB table (far larger than
// the 256 MiB Infinity Cache, so every request misses on die), N = 2^24 requests per launch.  FETCH_SIZE / N = bytes the
// memory side moves per isolated narrow read.  A streaming kernel over 1 GiB gives the reference point of the guide.
// build: hipcc -O3 --offload-arch=gfx950 tools/fetch_calib.hip -o tools/_fetch_calib ; run under rocprofv3 --pmc FETCH_SIZE
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

template <int W>
__global__ void gather(const uint8_t* __restrict__ tab, uint64_t mask, uint32_t* __restrict__ out) {
    const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t off = (mix(t + 12345) & mask) & ~(uint64_t)(W - 1);
    uint32_t acc = 0;
    if (W == 4) acc = *(const uint32_t*)(tab + off);
    else if (W == 8) { const uint2 v = *(const uint2*)(tab + off); acc = v.x ^ v.y; }
    else { const uint4 v = *(const uint4*)(tab + off); acc = v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;      // keeps the load alive
}

__global__ void stream16(const uint4* __restrict__ tab, uint64_t n, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = tab[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const uint64_t bytes = 2ull << 30;
    uint8_t* tab; uint32_t* out;
    if (hipMalloc(&tab, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(tab, 1, bytes);
    const int N = 1 << 24;
    for (int rep = 0; rep < 3; ++rep) {
        gather<4><<<N / 256, 256>>>(tab, bytes - 1, out);
        gather<8><<<N / 256, 256>>>(tab, bytes - 1, out);
        gather<16><<<N / 256, 256>>>(tab, bytes - 1, out);
        stream16<<<4096, 256>>>((const uint4*)tab, (1ull << 30) / 16, out);
    }
    hipDeviceSynchronize();
    printf("requests per gather launch: %d ; streamed bytes per stream16 launch: %llu\n", N, 1ull << 30);
    return 0;
}
