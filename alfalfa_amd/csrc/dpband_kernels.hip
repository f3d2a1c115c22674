// dpband_kernels.hip -- batch form of the reference's dynProg class (src/dp.cpp): dpBandStatic :648-759, dpBandFull
// :761-837, dpBandScore :872-924, dpBasic :926-1040 (dead code in the reference's release build, so
// this path is built for parity and completeness, not tuned: one lane per task, matrices in an HBM scratch arena).
//
// Layout: task k owns up to three (L2+1) x (L1+2) int32 matrices M, UP, LEFT in the arena (UP/LEFT only with affine gaps),
// its two sequence windows in a packed byte buffer and an edit-string area of L1+L2+2 bytes that is run-length encoded in
// place into (char, length) pairs.  The arena is filled with a sentinel before every launch: the reference's matrices are
// uninitialised `new int[]`, and wherever it would read a cell it never wrote (e.g. the start-cell search of
// findTraceBackPosStatic :164-181 under an asymmetric band) the result is undefined -- the kernel reports ret = 2 for such
// tasks instead of a value, and ret = 3 where the reference prints "Error in dynamic programming" and exits (:402 etc.).
#include <algorithm>
#include <cstring>
#include <vector>

#include "device_ctx.h"

namespace {

constexpr int32_t SENT = (int32_t)0x80808080;      // arena fill: no reachable score has this value (|score| < 2^27)

struct DTask {
    int64_t moff;            // arena offset (ints) of M; UP and LEFT follow
    int64_t soff_r, soff_q;  // offsets of the two windows in the sequence buffer
    int64_t eoff;            // offset of the edit-string / cigar area
    int64_t refB, refE;
    int32_t queryB, queryE;
    int32_t L1, L2, band1, band2, out_type, query_len;
    int32_t flags;           // bit0 freeRefB, bit1 freeRefE, bit2 freeQueryB, bit3 freeQueryE, bit4 local
    int32_t pad;
};

struct DScores { int32_t match, mismatch, openGap, extendGap; };

struct DRes {
    int64_t refB, refE;
    int32_t ret, dpScore, editDist, queryB, queryE, n_cigar;
};

struct Mat {
    int32_t* p;
    int W;
    __device__ __forceinline__ void set(int i, int j, int v) { p[(int64_t)i * W + j] = v; }
    __device__ __forceinline__ int get(int i, int j, bool& undef) const {
        const int v = p[(int64_t)i * W + j];
        if (v == SENT) undef = true;
        return v;
    }
};

// dp_scores::updateScoreMatrixDna src/dp.cpp:83-94
__device__ __forceinline__ int sc_of(const DScores& s, uint32_t a, uint32_t b) {
    const bool up = a - 'A' < 26u && b - 'A' < 26u;
    const bool lo = a - 'a' < 26u && b - 'a' < 26u;
    return (up || lo) ? (a == b ? s.match : s.mismatch) : -200000;
}

__global__ void dpband_kernel(int variant, DScores S, const DTask* __restrict__ tasks, int n, int32_t* __restrict__ arena,
                              const uint8_t* __restrict__ seq, char* __restrict__ ebuf, int32_t* __restrict__ elen, DRes* __restrict__ res) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const DTask T = tasks[k];
    const int L1 = T.L1, L2 = T.L2;
    const bool fRB = T.flags & 1, fRE = T.flags & 2, fQB = T.flags & 4, fQE = T.flags & 8, local = T.flags & 16;
    const bool aff = S.openGap != 0;
    const int e = S.extendGap, o = S.openGap;
    const uint8_t* R = seq + T.soff_r;      // R[j-1] = ref[refB + j - 1]
    const uint8_t* Q = seq + T.soff_q;
    int64_t refB = T.refB, refE = T.refE;
    int queryB = T.queryB, queryE = T.queryE;
    DRes out;
    out.ret = 0; out.dpScore = 0; out.editDist = 0; out.n_cigar = 0;
    bool undef = false, failed = false;
    int bandSize = 0, bandLeft = 0, bandRight = 0;
    const bool banded = variant != ALFA_DPBAND_BASIC;
    bool backward = false;
    if (variant == ALFA_DPBAND_STATIC) {
        bandSize = T.band1;
        if ((fQB && !fRB && L2 + bandSize < L1) || (fRB && !fQB && L2 - bandSize > L1) || (!(fQB || fRB) && !fRE && L2 + bandSize < L1) ||
            (!(fQB || fRB) && !fQE && L2 - bandSize > L1)) {                       // :671-685
            const int diff = max(L2, L1) - min(L2, L1);
            if (diff < T.band2) bandSize = diff;
            else {
                out.ret = 1; out.dpScore = S.mismatch * T.query_len; out.editDist = T.query_len;
                out.refB = refB; out.refE = refE; out.queryB = queryB; out.queryE = queryE;
                res[k] = out;
                return;
            }
        }
        backward = (fQB || fRB) && !local;
    } else if (banded) {
        bandLeft = T.band1; bandRight = T.band2;
    }
    const int bandL = max(bandLeft, bandSize), bandR = max(bandRight, bandSize);
    const int W = L1 + 2;
    Mat M, UP, LEFT;
    M.W = UP.W = LEFT.W = W;
    M.p = arena + T.moff;
    UP.p = M.p + (int64_t)(L2 + 1) * W;
    LEFT.p = UP.p + (int64_t)(L2 + 1) * W;

    // updateMatrix :96-129
    const bool freeQ0 = fQB && !backward, freeR0 = fRB && !backward;
    M.set(0, 0, 0);
    for (int i = 1; i <= L2; ++i) M.set(i, 0, freeQ0 ? 0 : o + i * e);
    for (int j = 1; j <= L1; ++j) M.set(0, j, freeR0 ? 0 : o + j * e);
    if (aff) {
        UP.set(0, 0, 0); LEFT.set(0, 0, 0);
        for (int j = 1; j <= L1; ++j) { const int v = (freeR0 ? 0 : o + j * e) + o; UP.set(0, j, v); LEFT.set(0, j, v); }
        for (int i = 1; i <= L2; ++i) { const int v = (freeQ0 ? 0 : o + i * e) + o; UP.set(i, 0, v); LEFT.set(i, 0, v); }
    }

    // fills :209-355 (banded) and :943-951 (dpBasic)
    const bool forward = !backward;
    const bool clamp = banded && forward && (local || (fQB && fRB));
    auto d_at = [&](int i, int j) -> int {
        return forward ? sc_of(S, R[j - 1], Q[i - 1]) : sc_of(S, R[L1 - j], Q[L2 - i]);      // backward: ref[refE-j+1], query[queryE-i+1]
    };
    if (!banded) {
        for (int i = 1; i <= L2; ++i)
            for (int j = 1; j <= L1; ++j) {
                int m = max(M.get(i - 1, j - 1, undef) + d_at(i, j), M.get(i, j - 1, undef) + e);
                m = max(m, M.get(i - 1, j, undef) + e);
                M.set(i, j, m);
            }
    } else {
        for (int i = 1; i <= L2; ++i) {
            int lo = i - bandL;
            const int hi = i + bandR;
            if (lo > 0) {
                if (lo <= L1) {          // left band edge: no LEFT predecessor, generic LEFT value (:226)
                    const int d = d_at(i, lo);
                    int m;
                    if (aff) {
                        m = M.get(i - 1, lo - 1, undef) + d;
                        const int up = max(M.get(i - 1, lo, undef) + e + o, UP.get(i - 1, lo, undef) + e);
                        UP.set(i, lo, up);
                        m = max(m, up);
                        LEFT.set(i, lo, m + e + 2 * o);
                    } else {
                        m = max(M.get(i - 1, lo - 1, undef) + d, M.get(i - 1, lo, undef) + e);
                    }
                    M.set(i, lo, clamp ? max(m, 0) : m);
                }
                ++lo;
            }
            const int jend = min(L1, hi - 1);
            for (int j = max(1, lo); j <= jend; ++j) {
                const int d = d_at(i, j);
                int m;
                if (aff) {
                    const int up = max(M.get(i - 1, j, undef) + e + o, UP.get(i - 1, j, undef) + e);
                    UP.set(i, j, up);
                    m = M.get(i - 1, j - 1, undef) + d;
                    const int left = max(M.get(i, j - 1, undef) + e + o, LEFT.get(i, j - 1, undef) + e);
                    LEFT.set(i, j, left);
                    m = max(max(m, left), up);
                } else {
                    m = max(M.get(i - 1, j - 1, undef) + d, M.get(i, j - 1, undef) + e);
                    m = max(m, M.get(i - 1, j, undef) + e);
                }
                M.set(i, j, clamp ? max(m, 0) : m);
            }
            if (hi <= L1) {              // right band edge: no UP predecessor, generic UP value (:247)
                const int d = d_at(i, hi);
                int m;
                if (aff) {
                    m = M.get(i - 1, hi - 1, undef) + d;
                    const int left = max(M.get(i, hi - 1, undef) + e + o, LEFT.get(i, hi - 1, undef) + e);
                    LEFT.set(i, hi, left);
                    m = max(m, left);
                    UP.set(i, hi, m + e + 2 * o);
                } else {
                    m = max(M.get(i - 1, hi - 1, undef) + d, M.get(i, hi - 1, undef) + e);
                }
                M.set(i, hi, clamp ? max(m, 0) : m);
            }
        }
    }

    // start cell: findTraceBackPosStatic :133-184 / the last-row scan of dpBasic :953-961
    int i, j;
    if (variant == ALFA_DPBAND_STATIC) { i = min(L2, L1 + bandSize); j = min(L1, L2 + bandSize); }
    else if (banded) { i = min(L2, L1 + bandLeft); j = min(L1, L2 + bandRight); }
    else { i = L2; j = L1; }
    int best = M.get(i, j, undef);
    if (!banded) {
        for (int c = L1; c >= 0; --c) { const int v = M.get(i, c, undef); if (v > best) { best = v; j = c; } }
    } else if ((fQE && fRE) || (fQB && fRB) || local) {
        for (int r = L2; r >= 0; --r)
            for (int c = min(r + bandR, L1); c >= max(0, r - bandL); --c) {
                const int v = M.get(r, c, undef);
                if (v > best) { best = v; i = r; j = c; }
            }
    } else if (fQE || fQB) {
        for (int r = min(L2, L1 + bandR); r >= min(L2, max(0, L1 - bandL)); --r) {
            const int v = M.get(r, j, undef);
            if (v > best) { best = v; i = r; }
        }
    } else if (fRE || fRB) {
        for (int c = min(L1, L2 + bandR); c >= min(L1, max(0, L2 - bandL)); --c) {
            const int v = M.get(i, c, undef);
            if (v > best) { best = v; j = c; }
        }
    }
    out.dpScore = M.get(i, j, undef);

    int editDist = 0, nes = 0;
    char* es = ebuf + T.eoff;
    int32_t* el = elen + T.eoff;
    const bool want_cigar = T.out_type == 3 || T.out_type == 4;
    if (T.out_type > 0) {
        if (variant == ALFA_DPBAND_STATIC) {
            if (fQE && i < L2) queryE = queryB + i - 1;
            if (fRE && j < L1) refE = refB + j - 1;
            if (fQB && i < L2) queryB = queryE - i + 1;
            if (fRB && j < L1) refB = refE - j + 1;
        } else if (banded) {
            if (fQE && i < L2) queryE = queryB + i - 1;
            if (fRE && j < L1) refE = refB + j - 1;
        } else {
            if (fRE && j < L1) refE = refB + j - 1;
        }
        // tracebacks :357-561, :839-870, :970-1008.  Tie order: '=' then 'X' then 'D' then 'I'.
        const bool loc = local || (fQB && fRB);
        const bool is_static = variant == ALFA_DPBAND_STATIC, is_score = variant == ALFA_DPBAND_SCORE;
        const int cap = L1 + L2 + 2;
        for (;;) {
            bool go;
            if (!banded) go = i > 0;
            else if (is_score) go = i > 0 && j > 0;
            else go = (is_static ? (i > 0 || j > 0) : (i > 0 && j > 0)) || (!aff && loc);
            if (!go) break;
            const bool in = i > 0 && j > 0;
            bool equal = false;
            if (in) equal = forward ? R[j - 1] == Q[i - 1] : R[L1 - j] == Q[L2 - i];
            const int cur = M.get(i, j, undef);
            char op = 0;
            if (in && cur == M.get(i - 1, j - 1, undef) + S.match && equal) op = '=';
            else if (in && cur == M.get(i - 1, j - 1, undef) + S.mismatch) op = 'X';
            else {
                const bool use_gap_mats = banded && aff;     // dpBasic always follows the linear recurrence
                const bool guardD = !banded || is_score || j > i - bandL;
                const bool guardI = !banded || is_score || j < i + bandR;
                bool isD = false, isI = false;
                if (j > 0 && guardD) {
                    const bool eq = use_gap_mats ? cur == LEFT.get(i, j, undef) : cur == M.get(i, j - 1, undef) + e;
                    isD = eq || (!is_score && i == 0);
                }
                if (!isD && i > 0 && guardI) {
                    const bool eq = use_gap_mats ? cur == UP.get(i, j, undef) : cur == M.get(i - 1, j, undef) + e;
                    isI = eq || (!is_score && j == 0);
                }
                op = isD ? 'D' : isI ? 'I' : 0;
            }
            if (!op) { failed = true; break; }
            if (op != '=') ++editDist;
            if (op == '=' || op == 'X') { --i; --j; }
            else if (op == 'D') --j;
            else --i;
            if (!is_score && nes < cap) es[nes++] = op;
            if (undef) break;
        }
        if (!failed && !undef && variant != ALFA_DPBAND_STATIC) {
            if (i > 0 && !fQB) editDist += i;
            if (j > 0 && !fRB) editDist += j;
            if (want_cigar && !is_score) {
                if (i > 0 && !fQB) for (int c = 0; c < i && nes < cap; ++c) es[nes++] = 'I';
                else if (j > 0 && !fRB) for (int c = 0; c < j && nes < cap; ++c) es[nes++] = 'D';
            }
        }
        if (!failed && !undef && want_cigar && !is_score) {
            // the string was produced end-to-start: reverse it unless the fill ran backward (:738-739), then run-length encode
            if (forward) for (int a = 0, b = nes - 1; a < b; ++a, --b) { const char t = es[a]; es[a] = es[b]; es[b] = t; }
            int nrun = 0, p = 0;
            while (p < nes) {
                const char c = es[p];
                int q = p;
                while (q < nes && es[q] == c) ++q;
                es[nrun] = c;            // nrun <= p: never overwrites unread input
                el[nrun] = q - p;
                ++nrun;
                p = q;
            }
            out.n_cigar = nrun;
        }
    }
    if (!failed && !undef && variant != ALFA_DPBAND_STATIC) {
        if (banded) { if (fQB && i > 0) queryB = queryB + i; }
        if (fRB && j > 0) refB = refB + j;
    }
    if (undef || failed) {
        out.ret = undef ? 2 : 3; out.dpScore = 0; out.editDist = 0; out.n_cigar = 0;
        out.refB = out.refE = 0; out.queryB = out.queryE = 0;
    } else {
        out.editDist = editDist;
        out.refB = refB; out.refE = refE; out.queryB = queryB; out.queryE = queryE;
    }
    res[k] = out;
}

// the reference asserts on these (:662-664,:776,:887,:934) or indexes scoreMatrix[127][127] / its strings out of bounds
int check_task(int variant, const alfa_dpband_scores* s, const alfa_dpband_task& t) {
    if ((s->affine != 0) != (s->openGap != 0)) return alfa::fail(ALFA_EINVAL, "dpband: scores.affine must equal (openGap != 0)");
    if (variant == ALFA_DPBAND_SCORE && s->openGap == 0)
        return alfa::fail(ALFA_EINVAL, "dpBandScore needs affine gaps (its traceback reads the UP/LEFT matrices)");
    if (!t.ref || !t.query) return alfa::fail(ALFA_EINVAL, "dpband: null sequence");
    const int64_t L1 = t.refE - t.refB + 1, L2 = (int64_t)t.queryE - t.queryB + 1;
    if (L1 < 0 || L2 < 0 || t.refB < 0 || t.queryB < 0 || t.refE >= t.ref_len || t.queryE >= t.query_len || L1 > 100000 || L2 > 100000)
        return alfa::fail(ALFA_EINVAL, "dpband: boundaries outside the sequences");
    if (variant == ALFA_DPBAND_STATIC) {
        if (t.band1 <= 0 || t.local) return alfa::fail(ALFA_EINVAL, "dpBandStatic: bandSize > 0 and a non-local type required");
        if ((t.freeQueryB || t.freeRefB) && (t.freeQueryE || t.freeRefE))
            return alfa::fail(ALFA_EINVAL, "dpBandStatic: free begin and free end cannot be combined");
    } else if (variant != ALFA_DPBAND_BASIC) {
        if (t.band1 <= 0 || t.band2 <= 0) return alfa::fail(ALFA_EINVAL, "dpBandFull/dpBandScore: bandLeft and bandRight must be > 0");
    }
    for (int64_t k = t.refB; k <= t.refE; ++k)
        if ((unsigned char)t.ref[k] >= 127) return alfa::fail(ALFA_EINVAL, "dpband: sequence byte >= 127");
    for (int64_t k = t.queryB; k <= t.queryE; ++k)
        if ((unsigned char)t.query[k] >= 127) return alfa::fail(ALFA_EINVAL, "dpband: sequence byte >= 127");
    return ALFA_OK;
}

}  // namespace

extern "C" int alfa_dp_band_batch(int variant, const alfa_dpband_scores* s, const alfa_dpband_task* t, int n, alfa_dpband_result* r,
                                  char** cig_chars, int32_t** cig_lens) {
    alfa::g_last_timing = alfa::Timing();
    if (variant < 0 || variant > 3 || !s || n < 0 || (n > 0 && (!t || !r)) || !cig_chars || !cig_lens) return alfa::fail(ALFA_EINVAL, "null argument");
    int dev = 0;
    (void)hipGetDevice(&dev);
    alfa::Ctx* cx = alfa::thread_ctx(dev);
    if (!cx) return ALFA_ENODEV;
    int rc;
    for (int k = 0; k < n; ++k)
        if ((rc = check_task(variant, s, t[k]))) return rc;
    const int nmat = s->openGap != 0 ? 3 : 1;
    const DScores S = {s->match, s->mismatch, s->openGap, s->extendGap};
    std::vector<char> allc;
    std::vector<int32_t> alll;
    hipStream_t st = cx->stream;
    const int64_t ARENA_MAX = (int64_t)1 << 30;     // ints per launch (4 GiB)
    int k0 = 0;
    while (k0 < n) {
        // chunk [k0,k1): as many tasks as fit the arena
        std::vector<DTask> dt;
        int64_t ar = 0, sq = 0, eo = 0;
        int k1 = k0;
        while (k1 < n) {
            const alfa_dpband_task& a = t[k1];
            const int64_t L1 = a.refE - a.refB + 1, L2 = (int64_t)a.queryE - a.queryB + 1;
            const int64_t need = (int64_t)nmat * (L2 + 1) * (L1 + 2);
            if (need > ARENA_MAX) return alfa::fail(ALFA_ENOMEM, "dpband: one task exceeds the matrix arena");
            if (k1 > k0 && ar + need > ARENA_MAX) break;
            DTask d;
            memset(&d, 0, sizeof(d));
            d.moff = ar; d.soff_r = sq; d.soff_q = sq + L1; d.eoff = eo;
            d.refB = a.refB; d.refE = a.refE; d.queryB = a.queryB; d.queryE = a.queryE;
            d.L1 = (int32_t)L1; d.L2 = (int32_t)L2; d.band1 = a.band1; d.band2 = a.band2; d.out_type = a.out_type;
            d.query_len = (int32_t)a.query_len;
            d.flags = (a.freeRefB ? 1 : 0) | (a.freeRefE ? 2 : 0) | (a.freeQueryB ? 4 : 0) | (a.freeQueryE ? 8 : 0) | (a.local ? 16 : 0);
            dt.push_back(d);
            ar += need; sq += L1 + L2; eo += L1 + L2 + 2;
            ++k1;
        }
        const int m = k1 - k0;
        if ((rc = cx->d_scratch.reserve(4 * (size_t)ar + 64)) || (rc = cx->d_in.reserve((size_t)sq + 64)) || (rc = cx->h_in.reserve((size_t)sq + 64)) ||
            (rc = cx->d_tasks.reserve(sizeof(DTask) * (size_t)m)) || (rc = cx->d_out.reserve(sizeof(DRes) * (size_t)m)) ||
            (rc = cx->d_out2.reserve(5 * (size_t)eo + 64)) || (rc = cx->h_out.reserve(sizeof(DRes) * (size_t)m)) ||
            (rc = cx->h_out2.reserve(5 * (size_t)eo + 64)))
            return rc;
        uint8_t* hs = cx->h_in.as<uint8_t>();
        for (int k = 0; k < m; ++k) {
            const alfa_dpband_task& a = t[k0 + k];
            if (dt[k].L1) memcpy(hs + dt[k].soff_r, a.ref + a.refB, (size_t)dt[k].L1);
            if (dt[k].L2) memcpy(hs + dt[k].soff_q, a.query + a.queryB, (size_t)dt[k].L2);
        }
        char* d_es = cx->d_out2.as<char>();
        int32_t* d_el = (int32_t*)(d_es + ((eo + 63) & ~(int64_t)63));
        if ((rc = cx->d_out2.reserve(((size_t)eo + 64) + 4 * (size_t)eo + 64)) || (rc = cx->h_out2.reserve(((size_t)eo + 64) + 4 * (size_t)eo + 64))) return rc;
        d_es = cx->d_out2.as<char>();
        d_el = (int32_t*)(d_es + ((eo + 63) & ~(int64_t)63));
        ALFA_HIP(hipMemsetAsync(cx->d_scratch.p, 0x80, 4 * (size_t)ar, st));
        ALFA_HIP(hipMemcpyAsync(cx->d_in.p, hs, (size_t)sq, hipMemcpyHostToDevice, st));
        ALFA_HIP(hipMemcpyAsync(cx->d_tasks.p, dt.data(), sizeof(DTask) * (size_t)m, hipMemcpyHostToDevice, st));
        ALFA_HIP(hipEventRecord(cx->ev[0], st));
        hipLaunchKernelGGL(dpband_kernel, dim3((m + 63) / 64), dim3(64), 0, st, variant, S, cx->d_tasks.as<DTask>(), m, cx->d_scratch.as<int32_t>(),
                           cx->d_in.as<uint8_t>(), d_es, d_el, cx->d_out.as<DRes>());
        ALFA_HIP(hipGetLastError());
        ALFA_HIP(hipEventRecord(cx->ev[1], st));
        const size_t ebytes = ((size_t)eo + 63 & ~(size_t)63) + 4 * (size_t)eo;
        ALFA_HIP(hipMemcpyAsync(cx->h_out.p, cx->d_out.p, sizeof(DRes) * (size_t)m, hipMemcpyDeviceToHost, st));
        ALFA_HIP(hipMemcpyAsync(cx->h_out2.p, cx->d_out2.p, ebytes, hipMemcpyDeviceToHost, st));
        ALFA_HIP(hipStreamSynchronize(st));
        float ms = 0;
        if (hipEventElapsedTime(&ms, cx->ev[0], cx->ev[1]) == hipSuccess) alfa::g_last_timing.kernel_ms += ms;
        alfa::g_last_timing.launches++;
        const DRes* hr = cx->h_out.as<DRes>();
        const char* h_es = cx->h_out2.as<char>();
        const int32_t* h_el = (const int32_t*)(h_es + ((eo + 63) & ~(int64_t)63));
        for (int k = 0; k < m; ++k) {
            alfa_dpband_result& o = r[k0 + k];
            o.ret = hr[k].ret; o.dpScore = hr[k].dpScore; o.editDist = hr[k].editDist;
            o.refB = hr[k].refB; o.refE = hr[k].refE; o.queryB = hr[k].queryB; o.queryE = hr[k].queryE;
            o.n_cigar = hr[k].n_cigar;
            o.cig_off = (int64_t)allc.size();
            allc.insert(allc.end(), h_es + dt[k].eoff, h_es + dt[k].eoff + hr[k].n_cigar);
            alll.insert(alll.end(), h_el + dt[k].eoff, h_el + dt[k].eoff + hr[k].n_cigar);
        }
        k0 = k1;
    }
    *cig_chars = (char*)malloc(allc.size() + 1);
    *cig_lens = (int32_t*)malloc(sizeof(int32_t) * (alll.size() + 1));
    if (!*cig_chars || !*cig_lens) return alfa::fail(ALFA_ENOMEM, "out of host memory");
    if (!allc.empty()) { memcpy(*cig_chars, allc.data(), allc.size()); memcpy(*cig_lens, alll.data(), sizeof(int32_t) * alll.size()); }
    alfa::g_last_timing.work_units = n;
    return ALFA_OK;
}
