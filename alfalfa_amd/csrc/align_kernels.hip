// align_kernels.hip -- batch form of ksw_align (src/ksw.c:330-352): local Smith-Waterman with the best score, both end
// positions, the second-best score (KSW_XSUBO) and, by a second pass over the reversed prefixes, the start positions
// (KSW_XSTART).  SURVEY 8f item 1: the mate-rescue call of src/mapper.cpp:1571 and :2196.
//
// The reference is Farrar's striped SSE2 kernel (ksw_u8 :106, ksw_i16 :223) and its results depend on the striping:
// E is taken from H before the lazy-F corrections, the lazy-F loop stops at the first vector after which no lane can
// improve, padded query positions take part in the row maximum, and the query end is the first maximum in vector-major
// memory order.  The striping is also exactly what a GPU wants, so the kernel keeps it: a task owns P adjacent lanes
// (P = 16 for the byte kernel, 8 for the 16-bit one; a DPP row holds one or two tasks, a wavefront four or eight), lane k
// owns query positions k*slen .. (k+1)*slen-1, and every SSE2 instruction becomes one VALU instruction on int32 registers
// holding the same lane values (all intermediate values of both kernels are in [0, 32767], so the saturating unsigned
// arithmetic is max(0, a - b) with b the 8- or 16-bit pattern of the penalty -- the reference's callers pass negative
// penalties, which makes b huge and a gap never pay; reproduced).  The vectors H(prev), H(cur), H(best row) and E and the
// striped query profile live in LDS ([vector j][lane k], 16-bit entries; one byte per profile entry); the three H buffers
// rotate roles instead of being copied (:195-196).  Queries too long for LDS use the same code on an HBM arena.
//
// Bound: VALU issue + LDS (about 14 VALU and 5 LDS operations per cell); no HBM traffic in the inner loop.
#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

#include "device_ctx.h"

namespace {

constexpr int XBYTE = 0x10000, XSTOP = 0x20000, XSUBO = 0x40000, XSTART = 0x80000;    // src/ksw.h:6-9

struct ATask {
    uint64_t goff;          // arena offset (bytes) of the task's vectors, HBM variant only
    uint32_t qoff, toff;    // sequences in the packed byte buffer
    int32_t qlen, tlen, xtra;
    uint32_t boff;          // first entry of the task's (score, row) run list (KSW_XSUBO)
    int32_t orig;           // position in the caller's array
    int32_t pad;
};

struct AParams {
    int32_t m, shift, max;
    int32_t gapoe8, gape8, gapoe16, gape16;     // the penalties as the 8- and 16-bit lanes hold them (:127-128, :243-244)
    int8_t mat[64];
};

struct PassOut { int score, te, qe, score2, te2; };

// _mm_subs_epu8 / _mm_subs_epu16 on values that are all in [0, 65535]: one v_sub_u32 with the clamp bit
__device__ __forceinline__ int subs(int a, int b) { return (int)__builtin_elementwise_sub_sat((unsigned)a, (unsigned)b); }
__device__ __forceinline__ int lane_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false); }   // row_shr:1

template <int P>
__device__ __forceinline__ int grp_max(int v) {
#pragma unroll
    for (int o = 1; o < P; o <<= 1) v = max(v, __shfl_xor(v, o));
    return v;
}

// One ksw_u8 / ksw_i16 call for every task group of the wavefront.  All lanes enter; `run` is false for groups without a
// call.  qrev / trev >= 0 select the reversed prefixes of the second pass (:344-346): query'[p] = query[qrev - p],
// target'[i] = target[trev - i] for i <= trev and target[i] behind it (the reference keeps running over all tlen rows).
template <int P, bool GLOB>
__device__ __forceinline__ PassOut striped_pass(const AParams& pr, const int8_t* smat, bool run, const uint8_t* q, int qlen, int qrev,
                                                const uint8_t* tg, int tlen, int trev, int xtra, uint16_t* HB, uint16_t* E, uint8_t* prof,
                                                int Lcap, int2* bruns, int k, int gshift) {
    constexpr uint64_t GM = P == 16 ? 0xffffull : 0xffull;
    const int slen = run ? (qlen + P - 1) / P : 0;                                                  // :65
    const int gapoe = P == 16 ? pr.gapoe8 : pr.gapoe16, gape = P == 16 ? pr.gape8 : pr.gape16;
    const int shift = pr.shift, m = pr.m;
    const int minsc = (xtra & XSUBO) ? (xtra & 0xffff) : 0x10000;                                   // :123-124
    const int endsc = (xtra & XSTOP) ? (xtra & 0xffff) : 0x10000;
    int oPrev = 0, oCur = Lcap, oBest = 2 * Lcap;
    // ksw_qinit's profile (:84-102) and the zeroed vectors (:132-136)
    for (int j = 0; j < slen; ++j) {
        const int x = j * P + k, p = j + k * slen;
        HB[oPrev + x] = 0; HB[oBest + x] = 0; E[x] = 0;
        const int code = p < qlen ? q[qrev >= 0 ? qrev - p : p] : -1;
        for (int a = 0; a < m; ++a) {
            const int s = code < 0 ? 0 : smat[a * m + code];
            prof[a * Lcap + x] = (uint8_t)(P == 16 ? s + shift : s);
        }
    }
    int gmax = 0, te = -1, nb = 0, lastS = 0, lastRow = -2;
    bool active = run && tlen > 0;      // slen == 0 still walks the rows: the run list can fill (:184)
    int tc_next = active ? tg[trev >= 0 ? trev : 0] : 0;
    for (int i = 0; __ballot(active) != 0; ++i) {
        if (GLOB) __threadfence_block();
        __builtin_amdgcn_wave_barrier();
        int f = 0, mx = 0;
        if (active) {
            const int tc = tc_next;
            if (i + 1 < tlen) tc_next = tg[i + 1 <= trev ? trev - (i + 1) : i + 1];
            const uint8_t* S = prof + tc * Lcap;
            int h = k > 0 && slen > 0 ? HB[oPrev + (slen - 1) * P + k - 1] : 0;                                 // :141-142
            for (int j = 0; j < slen; ++j) {                                                        // :143-167
                const int x = j * P + k;
                if (P == 16) h = subs(min(h + (int)S[x], 255), shift);                              // adds_epu8 then subs_epu8
                else h = h + (int)(int8_t)S[x];                                                     // adds_epi16 (cannot saturate: host check)
                int e = E[x];
                h = max(max(h, e), f);
                mx = max(mx, h);
                HB[oCur + x] = (uint16_t)h;
                h = subs(h, gapoe);
                e = max(subs(e, gape), h);
                E[x] = (uint16_t)e;
                f = max(subs(f, gape), h);
                h = HB[oPrev + x];
            }
        }
        // lazy F (:169-180): one step per trip for every group still in it; a group leaves at the first vector after which
        // none of its lanes can improve, or after 16 shifts
        {
            bool lz = active && slen > 0;
            int it = 0, j = 0;
            const int f0 = lane_shr1(f);
            f = k == 0 ? 0 : f0;
            while (__ballot(lz) != 0) {
                int hh = 0;
                if (lz) {
                    const int x = j * P + k;
                    hh = max((int)HB[oCur + x], f);
                    HB[oCur + x] = (uint16_t)hh;
                    hh = subs(hh, gapoe);
                    f = subs(f, gape);
                }
                const uint64_t bal = __ballot(lz && f > hh);
                const bool gdone = ((bal >> gshift) & GM) == 0;
                bool wrap = false;
                if (lz) {
                    if (gdone) lz = false;
                    else if (++j == slen) { j = 0; wrap = true; if (++it == 16) lz = false; }
                }
                const int fs = lane_shr1(f);
                if (wrap) f = k == 0 ? 0 : fs;
            }
        }
        const int imax = grp_max<P>(mx);                                                            // :183
        if (active) {
            if (imax >= minsc) {                                                                    // :184-192
                if (lastRow == -2 || lastRow + 1 != i) { ++nb; lastS = imax; lastRow = i; if (k == 0) bruns[nb - 1] = make_int2(lastS, lastRow); }
                else if (lastS < imax) { lastS = imax; lastRow = i; if (k == 0) bruns[nb - 1] = make_int2(lastS, lastRow); }
            }
            if (imax > gmax) {                                                                      // :193-198; cur becomes the kept row
                gmax = imax; te = i;
                const int op = oPrev;
                oBest = oCur; oPrev = oCur; oCur = op;
                if (P == 16 ? (gmax + shift >= 255 || gmax >= endsc) : (gmax >= endsc)) active = false;
            } else {                                                                                // :199 without touching the kept row
                const int nc = oPrev != oBest ? oPrev : 3 * Lcap - oPrev - oCur;
                oPrev = oCur; oCur = nc;
            }
            if (i + 1 >= tlen) active = false;
        }
    }
    PassOut r;
    r.score = P == 16 ? (gmax + shift < 255 ? gmax : 255) : gmax;                                   // :201, :303
    r.te = te; r.qe = -1; r.score2 = -1; r.te2 = -1;
    const bool fin = run && (P == 8 || r.score != 255);
    if (GLOB) __threadfence_block();
    int key = -1;                                                                                   // :204-207: first maximum in memory order
    if (fin)
        for (int j = 0; j < slen; ++j) {
            const int x = j * P + k;
            key = max(key, ((int)HB[oBest + x] << 16) | (0xffff - x));
        }
    key = grp_max<P>(key);
    if (fin && key >= 0) { const int x = 0xffff - (key & 0xffff); r.qe = x / P + x % P * slen; }
    if (fin && nb > 0 && k == 0) {                                                                  // :209-217
        const int d = (r.score + pr.max - 1) / pr.max, low = te - d, high = te + d;
        for (int e = 0; e < nb; ++e) {
            const int2 b = bruns[e];
            if ((b.y < low || b.y > high) && b.x > r.score2) r.score2 = b.x, r.te2 = b.y;
        }
    }
    return r;
}

template <int P, bool GLOB>
__global__ __launch_bounds__(64) void ksw_align_kernel(AParams pr, const ATask* tasks, int ntask, const uint8_t* seq, int Lcap, int task_bytes,
                                                       uint8_t* arena, int2* bruns, alfa_kswr* out) {
    extern __shared__ __align__(16) uint8_t dyn_lds[];
    __shared__ int8_t smat[64];
    constexpr int G = 64 / P;
    const int lane = threadIdx.x, g = lane / P, k = lane % P;
    if (lane < pr.m * pr.m) smat[lane] = pr.mat[lane];
    __syncthreads();
    const int ti = blockIdx.x * G + g;
    const bool have = ti < ntask;
    ATask t = tasks[have ? ti : ntask - 1];
    uint8_t* base;
    int lcap = Lcap;
    if (GLOB) {
        base = arena + t.goff;
        lcap = ((((t.qlen + P - 1) / P) * P + 63) / 64) * 64;
    } else {
        base = dyn_lds + (size_t)g * task_bytes;
    }
    uint16_t* HB = (uint16_t*)base;
    uint16_t* E = HB + 3 * lcap;
    uint8_t* prof = base + 8 * lcap;
    const uint8_t* q = seq + t.qoff;
    const uint8_t* tg = seq + t.toff;
    int2* br = bruns + t.boff;
    PassOut a = striped_pass<P, GLOB>(pr, smat, have, q, t.qlen, -1, tg, t.tlen, -1, t.xtra, HB, E, prof, lcap, br, k, g * P);
    int tb = -1, qb = -1;
    // :343 -- no second pass without KSW_XSTART or below the KSW_XSUBO threshold; a qe on a padded lane is undefined in the
    // reference (it reverses memory behind the query) and is left without start positions
    const bool second = have && (t.xtra & XSTART) && !((t.xtra & XSUBO) && a.score < (t.xtra & 0xffff)) && a.qe < t.qlen;
    PassOut b = striped_pass<P, GLOB>(pr, smat, second, q, a.qe + 1, a.qe, tg, t.tlen, a.te, XSTOP | a.score, HB, E, prof, lcap, br, k, g * P);
    if (second && a.score == b.score) { tb = a.te - b.te; qb = a.qe - b.qe; }                       // :349-350
    if (have && k == 0) {
        alfa_kswr o;
        o.score = a.score; o.te = a.te; o.qe = a.qe; o.score2 = a.score2; o.te2 = a.te2; o.tb = tb; o.qb = qb;
        out[t.orig] = o;
    }
}

struct Bin { int P; bool glob; int Lcap; int first, count; };

}  // namespace

extern "C" int alfa_ksw_align_batch(const alfa_dp_params* p, const alfa_align_task* t, int n, alfa_kswr* r) {
    alfa::g_last_timing = alfa::Timing();
    if (!p || n < 0 || (n > 0 && (!t || !r))) return alfa::fail(ALFA_EINVAL, "null argument");
    if (p->m < 1 || p->m > 8) return alfa::fail(ALFA_EINVAL, "ksw_align: m must be in 1..8");
    int dev = 0;
    (void)hipGetDevice(&dev);
    alfa::Ctx* cx = alfa::thread_ctx(dev);
    if (!cx) return ALFA_ENODEV;
    if (n == 0) return ALFA_OK;
    AParams pr;
    memset(&pr, 0, sizeof pr);
    pr.m = p->m;
    {   // ksw_qinit :74-81 with its uint8_t / int8_t conversions
        uint8_t sh = 127, md = 0;
        for (int a = 0; a < p->m * p->m; ++a) {
            if (p->mat[a] < (int8_t)sh) sh = (uint8_t)p->mat[a];
            if (p->mat[a] > (int8_t)md) md = (uint8_t)p->mat[a];
        }
        pr.max = md;
        pr.shift = (uint8_t)(256 - sh);
        memcpy(pr.mat, p->mat, 64);
    }
    pr.gapoe8 = (uint8_t)(p->gapo + p->gape); pr.gape8 = (uint8_t)p->gape;
    pr.gapoe16 = (uint16_t)(p->gapo + p->gape); pr.gape16 = (uint16_t)p->gape;
    const int m = p->m;
    size_t seq_bytes = 0, nruns = 0;
    for (int k = 0; k < n; ++k) {
        const alfa_align_task& a = t[k];
        if (a.qlen < 0 || a.tlen < 0 || (a.qlen > 0 && !a.query) || (a.tlen > 0 && !a.target)) return alfa::fail(ALFA_EINVAL, "ksw_align: bad sequence");
        if (a.qlen > 32767 || (int64_t)a.qlen * pr.max > 32767) return alfa::fail(ALFA_EINVAL, "ksw_align: qlen * max(mat) exceeds the 16-bit score range");
        if ((a.xtra & XSUBO) && pr.max <= 0) return alfa::fail(ALFA_EINVAL, "ksw_align: KSW_XSUBO needs a positive score in mat (the reference divides by it)");
        seq_bytes += (size_t)a.qlen + a.tlen;
        nruns += (a.xtra & XSUBO) ? (size_t)a.tlen / 2 + 1 : 0;
    }
    if (seq_bytes > 0xfff00000ull || nruns > 0xfff00000ull) return alfa::fail(ALFA_ENOMEM, "ksw_align: batch too large");
    // kernel bins: lanes per task x vector capacity; tasks of a wavefront should be alike, so order by work inside a bin
    auto lanes = [&](int k) { return (t[k].xtra & XBYTE) ? 16 : 8; };
    auto lneed = [&](int k) { const int P = lanes(k); return std::max(1, (t[k].qlen + P - 1) / P) * P; };
    const int LDS_MAX = 64 * 1024 - 128;
    auto bin_of = [&](int k) {          // 0..15: byte kernel, vector capacity 64..1024 in steps of 64; 16..31: 16-bit kernel; 32/33: HBM variants
        const int P = lanes(k), G = 64 / P, L = lneed(k);
        const int c = (L + 63) / 64 - 1, cap = (c + 1) * 64;
        if (c > 15 || G * (8 + m) * cap > LDS_MAX) return P == 16 ? 32 : 33;
        return (P == 16 ? 0 : 16) + c;
    };
    std::vector<int> order(n), bins(n);
    std::iota(order.begin(), order.end(), 0);
    for (int k = 0; k < n; ++k) bins[k] = bin_of(k);
    std::sort(order.begin(), order.end(), [&](int a, int b) {
        if (bins[a] != bins[b]) return bins[a] < bins[b];
        const int64_t wa = (int64_t)lneed(a) * t[a].tlen, wb = (int64_t)lneed(b) * t[b].tlen;
        if (wa != wb) return wa > wb;
        return a < b;
    });
    int rc;
    if ((rc = cx->h_in.reserve(seq_bytes + 64)) || (rc = cx->d_in.reserve(seq_bytes + 64)) || (rc = cx->h_tasks.reserve(sizeof(ATask) * (size_t)n)) ||
        (rc = cx->d_tasks.reserve(sizeof(ATask) * (size_t)n)) || (rc = cx->d_out.reserve(sizeof(alfa_kswr) * (size_t)n)) ||
        (rc = cx->h_out.reserve(sizeof(alfa_kswr) * (size_t)n)) || (rc = cx->d_scratch.reserve(sizeof(int2) * (nruns + 1))))
        return rc;
    uint8_t* hs = cx->h_in.as<uint8_t>();
    ATask* ht = cx->h_tasks.as<ATask>();
    std::vector<Bin> launches;
    size_t so = 0, bo = 0, go = 0;
    for (int i = 0; i < n; ++i) {
        const int k = order[i];
        const alfa_align_task& a = t[k];
        for (int x = 0; x < a.qlen; ++x) if (a.query[x] >= m) return alfa::fail(ALFA_EINVAL, "ksw_align: residue code >= m");
        for (int x = 0; x < a.tlen; ++x) if (a.target[x] >= m) return alfa::fail(ALFA_EINVAL, "ksw_align: residue code >= m");
        ATask d;
        memset(&d, 0, sizeof d);
        d.qoff = (uint32_t)so; if (a.qlen) memcpy(hs + so, a.query, (size_t)a.qlen); so += (size_t)a.qlen;
        d.toff = (uint32_t)so; if (a.tlen) memcpy(hs + so, a.target, (size_t)a.tlen); so += (size_t)a.tlen;
        d.qlen = a.qlen; d.tlen = a.tlen; d.xtra = a.xtra; d.orig = k;
        d.boff = (uint32_t)bo;
        if (a.xtra & XSUBO) bo += (size_t)a.tlen / 2 + 1;
        const int b = bins[k], P = lanes(k);
        if (b >= 32) { d.goff = go; go += (size_t)(8 + m) * ((lneed(k) + 63) / 64 * 64); go = (go + 255) & ~(size_t)255; }
        ht[i] = d;
        if (launches.empty() || bins[order[launches.back().first]] != b) {
            Bin L;
            L.P = P; L.glob = b >= 32; L.Lcap = b >= 32 ? 0 : 64 * (b % 16 + 1); L.first = i; L.count = 0;
            launches.push_back(L);
        }
        launches.back().count++;
    }
    if (go && (rc = cx->d_z.reserve(go + 256))) return rc;
    hipStream_t st = cx->stream;
    ALFA_HIP(hipMemcpyAsync(cx->d_in.p, hs, so, hipMemcpyHostToDevice, st));
    ALFA_HIP(hipMemcpyAsync(cx->d_tasks.p, ht, sizeof(ATask) * (size_t)n, hipMemcpyHostToDevice, st));
    ALFA_HIP(hipEventRecord(cx->ev[0], st));
    for (const Bin& L : launches) {
        const int G = 64 / L.P, tb = (8 + m) * L.Lcap;
        const dim3 grid((L.count + G - 1) / G), block(64);
        const size_t lds = L.glob ? 0 : (size_t)G * tb;
        const ATask* dt = cx->d_tasks.as<ATask>() + L.first;
        uint8_t* ds = cx->d_in.as<uint8_t>();
        int2* br = cx->d_scratch.as<int2>();
        alfa_kswr* o = cx->d_out.as<alfa_kswr>();
        if (L.P == 16 && !L.glob) hipLaunchKernelGGL((ksw_align_kernel<16, false>), grid, block, lds, st, pr, dt, L.count, ds, L.Lcap, tb, (uint8_t*)nullptr, br, o);
        else if (L.P == 8 && !L.glob) hipLaunchKernelGGL((ksw_align_kernel<8, false>), grid, block, lds, st, pr, dt, L.count, ds, L.Lcap, tb, (uint8_t*)nullptr, br, o);
        else if (L.P == 16) hipLaunchKernelGGL((ksw_align_kernel<16, true>), grid, block, 0, st, pr, dt, L.count, ds, 0, 0, cx->d_z.as<uint8_t>(), br, o);
        else hipLaunchKernelGGL((ksw_align_kernel<8, true>), grid, block, 0, st, pr, dt, L.count, ds, 0, 0, cx->d_z.as<uint8_t>(), br, o);
        ALFA_HIP(hipGetLastError());
        alfa::g_last_timing.launches++;
    }
    ALFA_HIP(hipEventRecord(cx->ev[1], st));
    ALFA_HIP(hipMemcpyAsync(cx->h_out.p, cx->d_out.p, sizeof(alfa_kswr) * (size_t)n, hipMemcpyDeviceToHost, st));
    ALFA_HIP(hipStreamSynchronize(st));
    float ms = 0;
    if (hipEventElapsedTime(&ms, cx->ev[0], cx->ev[1]) == hipSuccess) alfa::g_last_timing.kernel_ms = ms;
    memcpy(r, cx->h_out.p, sizeof(alfa_kswr) * (size_t)n);
    alfa::g_last_timing.work_units = n;
    return ALFA_OK;
}
