// index_build.hip -- construction of the enhanced sparse suffix array on the GPU.
//
// Replaces, array for array and bit for bit, what the host builder (host_index.cpp) and the reference compute on one core:
//   sparseSA::construct   src/sparseSA.cpp:478-558 (+ src/qsufsort.c)   suffix array of the K-sampled suffixes
//   sparseSA::computeLCP  src/sparseSA.cpp:161-173                       LCP bytes + the overflow table for values >= 255
//   sparseSA::computeChild src/sparseSA.cpp:176-219                      up / down / next-l-index values merged in one array
//   sparseSA::computeKmer src/sparseSA.cpp:222-313                       SA interval of every ACGT 10-mer
// The results are functions of the text alone (a suffix order is unique), so the methods are free:
//   * suffix array: prefix doubling over the sampled suffixes.  Round 0 orders them by their first K bytes (stable 64-bit
//     radix sorts, last 8-byte chunk first); every further round sorts (rank, rank of the suffix h bytes further on) with
//     h = K, 2K, 4K ... -- h stays a multiple of K, so the second rank is again that of a sampled suffix -- until all
//     ranks differ.  hipCUB DeviceRadixSort / DeviceScan do the sorting and the rank renumbering.
//   * LCP: one lane per suffix array entry compares its suffix with its predecessor's, 8 bytes per step (sparse Kasai is a
//     sequential amortisation of exactly these comparisons).
//   * CHILD: the reference's two stack passes are rewritten as nearest-smaller-value queries.  With pse/pss(k) the nearest
//     previous index whose lcp is <= / < lcp[k], nss(k) the next index whose lcp is < lcp[k] (n with lcp 0 behind the array):
//       down:  CHILD[pss(k)] = k      if pse(k) == pss(k) and lcp[nss(k)] <= lcp[pss(k)]   (k = leftmost minimum above pss(k))
//       up:    CHILD[nss(k) - 1] = k  if nss(k) < n and lcp[pse(k)] <= lcp[nss(k)]
//       next:  CHILD[pse(i)] = i      if lcp[pse(i)] == lcp[i]                              (written last: it overrides)
//     (derivation in DESIGN.md; each target receives at most one value per rule).  A query scans its own 256-entry block,
//     then the per-block minima, then the block that holds the answer.
//   * KMR: atomicMin / atomicMax of the SA rank per 10-mer code.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstring>
#include <vector>

#include "device_ctx.h"
#include "host_index_hook.h"

namespace {

constexpr int KMER = 10;
constexpr uint32_t TABLE = 1u << (2 * KMER);
constexpr int BLK = 256;

// ---- suffix array
// the chunk_len (<= 8) bytes of suffix p from byte offset `off` as a big-endian number in the LOW bits of the key (rocPRIM's
// merge path mis-masks a bit range that ends at bit 64 but starts above 0, so the sorted bits always start at 0); bytes
// behind the text count as 0 (the end is smaller than any symbol)
__global__ void sa_chunk_keys(const uint8_t* __restrict__ S, int64_t N, int64_t K, const uint32_t* __restrict__ idx, int64_t n, int off,
                              int chunk_len, uint64_t* __restrict__ keys) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int64_t p = (int64_t)idx[t] * K + off;
    uint64_t k = 0;
    for (int b = 0; b < chunk_len; ++b) {
        const uint64_t c = p + b < N ? S[p + b] : 0u;
        k = (k << 8) | c;
    }
    keys[t] = k;
}
__global__ void iota_kernel(uint32_t* __restrict__ v, int64_t n) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) v[t] = (uint32_t)t;
}
// flag[t] = t when entry t starts a new group of equal K-byte prefixes, else 0
__global__ void prefix_group_starts(const uint8_t* __restrict__ S, int64_t N, int64_t K, const uint32_t* __restrict__ idx, int64_t n,
                                    uint32_t* __restrict__ flag) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    bool start = t == 0;
    if (!start) {
        const int64_t a = (int64_t)idx[t] * K, b = (int64_t)idx[t - 1] * K;
        for (int64_t j = 0; j < K && !start; ++j) {
            const uint32_t ca = a + j < N ? S[a + j] : 0u, cb = b + j < N ? S[b + j] : 0u;
            start = ca != cb;
        }
    }
    flag[t] = start ? (uint32_t)t : 0u;
}
__global__ void scatter_rank(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ gstart, int64_t n, uint32_t* __restrict__ rank) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) rank[idx[t]] = gstart[t];
}
// key = (rank of the suffix) << 32 | (1 + rank of the suffix hb blocks further on, 0 behind the text)
__global__ void doubling_keys(const uint32_t* __restrict__ idx, const uint32_t* __restrict__ rank, int64_t n, int64_t hb, uint64_t* __restrict__ keys) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int64_t i = idx[t];
    const uint64_t r2 = i + hb < n ? (uint64_t)rank[i + hb] + 1u : 0u;
    keys[t] = ((uint64_t)rank[i] << 32) | r2;
}
__global__ void key_group_starts(const uint64_t* __restrict__ keys, int64_t n, uint32_t* __restrict__ flag, unsigned long long* __restrict__ ngroups) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const bool start = t == 0 || keys[t] != keys[t - 1];
    flag[t] = start ? (uint32_t)t : 0u;
    if (start) atomicAdd(ngroups, 1ull);
}
__global__ void finish_sa(const uint32_t* __restrict__ idx, int64_t n, int64_t K, uint32_t* __restrict__ SA, int32_t* __restrict__ ISA) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    SA[t] = (uint32_t)((int64_t)idx[t] * K);
    if (ISA) ISA[idx[t]] = (int32_t)t;
}

// ---- LCP
__device__ __forceinline__ uint64_t ld8u(const uint8_t* p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
__global__ void lcp_kernel(const uint8_t* __restrict__ S, int64_t N, const uint32_t* __restrict__ SA, int64_t n, int32_t* __restrict__ L,
                           uint8_t* __restrict__ LCP, uint8_t* __restrict__ big) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= n) return;
    int64_t v = 0;
    if (m > 0) {
        const int64_t i = SA[m], j = SA[m - 1];
        const int64_t lim = N - (i > j ? i : j);           // comparisons stop at the end of the text (S carries 16 bytes of slack)
        while (v < lim) {
            const uint64_t x = ld8u(S + i + v) ^ ld8u(S + j + v);
            if (x) { v += __builtin_ctzll(x) >> 3; break; }
            v += 8;
        }
        v = v < lim ? v : lim;
    }
    L[m] = (int32_t)v;
    LCP[m] = v >= 255 ? (uint8_t)255 : (uint8_t)v;
    big[m] = v >= 255;
}

// ---- CHILD: nearest smaller values through 256-entry blocks
__global__ void block_min_kernel(const int32_t* __restrict__ L, int64_t n, int32_t* __restrict__ bmin) {
    __shared__ int32_t sh[BLK];
    const int64_t t = (int64_t)blockIdx.x * BLK + threadIdx.x;
    sh[threadIdx.x] = t < n ? L[t] : 0x7fffffff;
    __syncthreads();
    for (int o = BLK / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] = min(sh[threadIdx.x], sh[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) bmin[blockIdx.x] = sh[0];
}
// nearest j < i with L[j] < v (STRICT) or L[j] <= v; -1 when there is none
template <bool STRICT>
__device__ int64_t prev_smaller(const int32_t* __restrict__ L, const int32_t* __restrict__ bmin, int64_t i, int32_t v) {
    auto hit = [&](int32_t x) { return STRICT ? x < v : x <= v; };
    const int64_t b0 = i / BLK;
    for (int64_t j = i - 1; j >= b0 * BLK; --j) if (hit(L[j])) return j;
    for (int64_t b = b0 - 1; b >= 0; --b)
        if (hit(bmin[b]))
            for (int64_t j = b * BLK + BLK - 1;; --j) if (hit(L[j])) return j;
    return -1;
}
// nearest j > i with L[j] < v; n when there is none (the reference's final flush compares with 0 behind the array)
__device__ int64_t next_strictly_smaller(const int32_t* __restrict__ L, const int32_t* __restrict__ bmin, int64_t n, int64_t i, int32_t v) {
    const int64_t b0 = i / BLK, nb = (n + BLK - 1) / BLK;
    const int64_t e0 = (b0 + 1) * BLK < n ? (b0 + 1) * BLK : n;
    for (int64_t j = i + 1; j < e0; ++j) if (L[j] < v) return j;
    for (int64_t b = b0 + 1; b < nb; ++b)
        if (bmin[b] < v)
            for (int64_t j = b * BLK;; ++j) if (L[j] < v) return j;
    return n;
}
__global__ void child_updown_kernel(const int32_t* __restrict__ L, const int32_t* __restrict__ bmin, int64_t n, int32_t* __restrict__ CHILD) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < 1 || k >= n) return;
    const int32_t v = L[k];
    const int64_t pse = prev_smaller<false>(L, bmin, k, v);      // exists: L[0] = 0 is a minimum
    const int64_t pss = prev_smaller<true>(L, bmin, k, v);
    const int64_t nss = next_strictly_smaller(L, bmin, n, k, v);
    const int32_t lnss = nss < n ? L[nss] : 0;
    if (pss >= 0 && pse == pss && lnss <= L[pss]) CHILD[pss] = (int32_t)k;          // down value of pss
    if (nss < n && L[pse] <= lnss) CHILD[nss - 1] = (int32_t)k;                     // up value of nss - 1
}
__global__ void child_next_kernel(const int32_t* __restrict__ L, const int32_t* __restrict__ bmin, int64_t n, int32_t* __restrict__ CHILD) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 1 || i >= n) return;
    const int32_t v = L[i];
    const int64_t pse = prev_smaller<false>(L, bmin, i, v);
    if (L[pse] == v) CHILD[pse] = (int32_t)i;
}

// ---- KMR
__global__ void kmer_kernel(const uint8_t* __restrict__ S, int64_t N, const uint32_t* __restrict__ SA, int64_t n, uint32_t* __restrict__ KMR) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t p = SA[i];
    if (p + KMER > N) return;
    uint32_t code = 0;
    for (int j = 0; j < KMER; ++j) {
        const uint32_t c = S[p + j] | 0x20u;
        const uint32_t b = c == 'a' ? 0u : c == 'c' ? 1u : c == 'g' ? 2u : c == 't' ? 3u : 0xffffffffu;
        if (b > 3u) return;
        code = (code << 2) | b;
    }
    atomicMin(&KMR[2 * (size_t)code], (uint32_t)i);
    atomicMax(&KMR[2 * (size_t)code + 1], (uint32_t)i);
}
__global__ void kmer_init_kernel(uint32_t* __restrict__ KMR) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < TABLE) { KMR[2 * (size_t)t] = 0xffffffffu; KMR[2 * (size_t)t + 1] = 0u; }
}
__global__ void kmer_fix_kernel(uint32_t* __restrict__ KMR) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < TABLE && KMR[2 * (size_t)t] == 0xffffffffu) KMR[2 * (size_t)t] = 0u;       // untouched entries stay (0, 0)
}

inline dim3 grid_for(int64_t n, int b = 256) { return dim3((unsigned)((n + b - 1) / b)); }

int device_construct(int device, alfa::DeviceBuildIO* io) {
    using alfa::DevBuf;
    alfa::Ctx* cx = alfa::thread_ctx(device);
    if (!cx) return ALFA_ENODEV;
    hipStream_t st = cx->stream;
    const int64_t N = io->N, K = io->K, n = N / K;
    if (n <= 0 || n >= 0x7fffffffLL || N >= 0xffffffffLL) return alfa::fail(ALFA_EINVAL, "text outside the 32-bit range of the index arrays");
    DevBuf dS, idxA, idxB, keyA, keyB, flag, gstart, rank, dSA, dISA, dL, dLCP, dbig, dbmin, dCHILD, dKMR, tmp, dcnt, selIdx, selVal, d64;
    auto free_all = [&] { for (DevBuf* b : {&dS, &idxA, &idxB, &keyA, &keyB, &flag, &gstart, &rank, &dSA, &dISA, &dL, &dLCP, &dbig, &dbmin, &dCHILD, &dKMR, &tmp,
                                              &dcnt, &selIdx, &selVal, &d64}) b->release(); };
    int rc;
#define TRY(x) do { if ((rc = (x))) { free_all(); return rc; } } while (0)
#define TRYHIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { free_all(); return alfa::fail(ALFA_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } } while (0)
    TRY(dS.reserve((size_t)N + 64));
    TRYHIP(hipMemsetAsync(dS.p, 0, (size_t)N + 64, st));
    TRYHIP(hipMemcpyAsync(dS.p, io->S, (size_t)N, hipMemcpyHostToDevice, st));
    const uint8_t* S = dS.as<uint8_t>();
    TRY(idxA.reserve(4 * (size_t)n)); TRY(idxB.reserve(4 * (size_t)n)); TRY(keyA.reserve(8 * (size_t)n)); TRY(keyB.reserve(8 * (size_t)n));
    TRY(flag.reserve(4 * (size_t)n)); TRY(gstart.reserve(4 * (size_t)n)); TRY(rank.reserve(4 * (size_t)n)); TRY(dcnt.reserve(64));
    size_t sort_bytes = 0, scan_bytes = 0;
    {
        TRYHIP(hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, keyA.as<uint64_t>(), keyB.as<uint64_t>(), idxA.as<uint32_t>(), idxB.as<uint32_t>(),
                                                  (int)n, 0, 64, st));
        TRYHIP(hipcub::DeviceScan::InclusiveScan(nullptr, scan_bytes, flag.as<uint32_t>(), gstart.as<uint32_t>(), hipcub::Max(), (int)n, st));
    }
    TRY(tmp.reserve(std::max(sort_bytes, scan_bytes) + 256));
    uint32_t* cur = idxA.as<uint32_t>();
    uint32_t* alt = idxB.as<uint32_t>();
    hipLaunchKernelGGL(iota_kernel, grid_for(n), dim3(256), 0, st, cur, n);
    // ---- round 0: the first K bytes, least significant 8-byte chunk first (the sorts are stable)
    const int nchunk = (int)((K + 7) / 8);
    for (int c = nchunk - 1; c >= 0; --c) {
        const int len = (int)std::min<int64_t>(8, K - 8 * (int64_t)c);
        hipLaunchKernelGGL(sa_chunk_keys, grid_for(n), dim3(256), 0, st, S, N, K, cur, n, 8 * c, len, keyA.as<uint64_t>());
        size_t tb = tmp.cap;
        TRYHIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, keyA.as<uint64_t>(), keyB.as<uint64_t>(), cur, alt, (int)n, 0, 8 * len, st));
        std::swap(cur, alt);
    }
    hipLaunchKernelGGL(prefix_group_starts, grid_for(n), dim3(256), 0, st, S, N, K, cur, n, flag.as<uint32_t>());
    {
        size_t tb = tmp.cap;
        TRYHIP(hipcub::DeviceScan::InclusiveScan(tmp.p, tb, flag.as<uint32_t>(), gstart.as<uint32_t>(), hipcub::Max(), (int)n, st));
    }
    hipLaunchKernelGGL(scatter_rank, grid_for(n), dim3(256), 0, st, cur, gstart.as<uint32_t>(), n, rank.as<uint32_t>());
    // ---- doubling rounds
    for (int64_t hb = 1;; hb <<= 1) {
        hipLaunchKernelGGL(doubling_keys, grid_for(n), dim3(256), 0, st, cur, rank.as<uint32_t>(), n, hb, keyA.as<uint64_t>());
        size_t tb = tmp.cap;
        TRYHIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tb, keyA.as<uint64_t>(), keyB.as<uint64_t>(), cur, alt, (int)n, 0, 64, st));
        std::swap(cur, alt);
        TRYHIP(hipMemsetAsync(dcnt.p, 0, 8, st));
        hipLaunchKernelGGL(key_group_starts, grid_for(n), dim3(256), 0, st, keyB.as<uint64_t>(), n, flag.as<uint32_t>(), dcnt.as<unsigned long long>());
        tb = tmp.cap;
        TRYHIP(hipcub::DeviceScan::InclusiveScan(tmp.p, tb, flag.as<uint32_t>(), gstart.as<uint32_t>(), hipcub::Max(), (int)n, st));
        hipLaunchKernelGGL(scatter_rank, grid_for(n), dim3(256), 0, st, cur, gstart.as<uint32_t>(), n, rank.as<uint32_t>());
        unsigned long long groups = 0;
        TRYHIP(hipMemcpyAsync(&groups, dcnt.p, 8, hipMemcpyDeviceToHost, st));
        TRYHIP(hipStreamSynchronize(st));
        if ((int64_t)groups == n) break;
        if (hb > n) { free_all(); return alfa::fail(ALFA_EINVAL, "suffix sort did not converge (two sampled suffixes compare equal)"); }
    }
    TRY(dSA.reserve(4 * (size_t)n));
    if (io->ISA) TRY(dISA.reserve(4 * (size_t)n));
    hipLaunchKernelGGL(finish_sa, grid_for(n), dim3(256), 0, st, cur, n, K, dSA.as<uint32_t>(), io->ISA ? dISA.as<int32_t>() : (int32_t*)nullptr);
    // ---- LCP and its overflow table
    TRY(dL.reserve(4 * (size_t)n)); TRY(dLCP.reserve((size_t)n)); TRY(dbig.reserve((size_t)n));
    hipLaunchKernelGGL(lcp_kernel, grid_for(n), dim3(256), 0, st, S, N, dSA.as<uint32_t>(), n, dL.as<int32_t>(), dLCP.as<uint8_t>(), dbig.as<uint8_t>());
    {
        // ranks with lcp >= 255, in rank order, and their values
        hipLaunchKernelGGL(iota_kernel, grid_for(n), dim3(256), 0, st, flag.as<uint32_t>(), n);
        size_t sb = 0;
        TRYHIP(hipcub::DeviceSelect::Flagged(nullptr, sb, flag.as<uint32_t>(), dbig.as<uint8_t>(), gstart.as<uint32_t>(), dcnt.as<int>(), (int)n, st));
        if (sb + 256 > tmp.cap) TRY(tmp.reserve(sb + 256));
        sb = tmp.cap;
        TRYHIP(hipcub::DeviceSelect::Flagged(tmp.p, sb, flag.as<uint32_t>(), dbig.as<uint8_t>(), gstart.as<uint32_t>(), dcnt.as<int>(), (int)n, st));
        int nM = 0;
        TRYHIP(hipMemcpyAsync(&nM, dcnt.p, 4, hipMemcpyDeviceToHost, st));
        TRYHIP(hipStreamSynchronize(st));
        std::vector<uint32_t> mi((size_t)nM);
        if (nM) TRYHIP(hipMemcpy(mi.data(), gstart.p, 4 * (size_t)nM, hipMemcpyDeviceToHost));
        sb = tmp.cap;
        TRYHIP(hipcub::DeviceSelect::Flagged(tmp.p, sb, dL.as<int32_t>(), dbig.as<uint8_t>(), (int32_t*)gstart.p, dcnt.as<int>(), (int)n, st));
        TRYHIP(hipStreamSynchronize(st));
        io->Midx->resize((size_t)nM); io->Mval->resize((size_t)nM);
        if (nM) TRYHIP(hipMemcpy(io->Mval->data(), gstart.p, 4 * (size_t)nM, hipMemcpyDeviceToHost));
        for (int k = 0; k < nM; ++k) (*io->Midx)[k] = (int64_t)mi[k];
    }
    // ---- CHILD
    if (io->CHILD) {
        const int64_t nb = (n + BLK - 1) / BLK;
        TRY(dbmin.reserve(4 * (size_t)nb)); TRY(dCHILD.reserve(4 * (size_t)n));
        TRYHIP(hipMemsetAsync(dCHILD.p, 0xff, 4 * (size_t)n, st));
        hipLaunchKernelGGL(block_min_kernel, dim3((unsigned)nb), dim3(BLK), 0, st, dL.as<int32_t>(), n, dbmin.as<int32_t>());
        hipLaunchKernelGGL(child_updown_kernel, grid_for(n), dim3(256), 0, st, dL.as<int32_t>(), dbmin.as<int32_t>(), n, dCHILD.as<int32_t>());
        hipLaunchKernelGGL(child_next_kernel, grid_for(n), dim3(256), 0, st, dL.as<int32_t>(), dbmin.as<int32_t>(), n, dCHILD.as<int32_t>());
        TRYHIP(hipMemcpyAsync(io->CHILD, dCHILD.p, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    }
    // ---- KMR
    if (io->KMR) {
        TRY(dKMR.reserve(8 * (size_t)TABLE));
        hipLaunchKernelGGL(kmer_init_kernel, grid_for(TABLE), dim3(256), 0, st, dKMR.as<uint32_t>());
        hipLaunchKernelGGL(kmer_kernel, grid_for(n), dim3(256), 0, st, S, N, dSA.as<uint32_t>(), n, dKMR.as<uint32_t>());
        hipLaunchKernelGGL(kmer_fix_kernel, grid_for(TABLE), dim3(256), 0, st, dKMR.as<uint32_t>());
        TRYHIP(hipMemcpyAsync(io->KMR, dKMR.p, 8 * (size_t)TABLE, hipMemcpyDeviceToHost, st));
    }
    TRYHIP(hipMemcpyAsync(io->SA, dSA.p, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    TRYHIP(hipMemcpyAsync(io->LCP, dLCP.p, (size_t)n, hipMemcpyDeviceToHost, st));
    if (io->ISA) TRYHIP(hipMemcpyAsync(io->ISA, dISA.p, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    TRYHIP(hipGetLastError());
    TRYHIP(hipStreamSynchronize(st));
    free_all();
#undef TRY
#undef TRYHIP
    return ALFA_OK;
}

}  // namespace

// from now on alfa_host_index_build_fasta / _build_mem construct their arrays on `device` (device < 0: on the host again)
extern "C" int alfa_index_build_on_device(int device) {
    if (device < 0) { alfa::g_device_construct = nullptr; alfa::g_build_device = -1; return ALFA_OK; }
    if (device >= alfa::device_count()) return alfa::fail(alfa::device_count() <= 0 ? ALFA_ENODEV : ALFA_EINVAL, "no such HIP device");
    alfa::g_device_construct = device_construct;
    alfa::g_build_device = device;
    return ALFA_OK;
}
