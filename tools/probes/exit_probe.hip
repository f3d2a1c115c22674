// exit_probe: what ending a HIP process costs on this box (kernel-side teardown of the GPU context), by what it holds.
// usage: exit_probe <device MiB> <pinned MiB> <streams> <0|1 free everything before exit>; prints the epoch time of its last line
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <unistd.h>
#include <vector>
__global__ void touch(char* p, size_t n) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = 1; }
int main(int argc, char** argv) {
    const size_t dmb = argc > 1 ? atol(argv[1]) : 0, pmb = argc > 2 ? atol(argv[2]) : 0;
    const int ns = argc > 3 ? atoi(argv[3]) : 0, fr = argc > 4 ? atoi(argv[4]) : 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    if (hipSetDevice(0) != hipSuccess) return 1;
    char *d = nullptr, *h = nullptr;
    if (dmb && hipMalloc(&d, dmb << 20) != hipSuccess) return 2;
    if (pmb && hipHostMalloc(&h, pmb << 20) != hipSuccess) return 3;
    std::vector<hipStream_t> st((size_t)ns);
    for (auto& s : st) if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 4;
    if (d) touch<<<(unsigned)(((dmb << 20) + 255) / 256), 256, 0, ns ? st[0] : 0>>>(d, dmb << 20);
    for (auto& s : st) touch<<<1, 64, 0, s>>>(d ? d : h, d || h ? 64 : 0);
    if (hipDeviceSynchronize() != hipSuccess) return 5;
    const double t1 = now();
    if (fr) { for (auto& s : st) (void)hipStreamDestroy(s); (void)hipFree(d); (void)hipHostFree(h); }
    printf("init+alloc %.3f free %.3f last_line %.6f\n", t1 - t0, now() - t1, now());
    fflush(nullptr);
    _exit(0);
}
