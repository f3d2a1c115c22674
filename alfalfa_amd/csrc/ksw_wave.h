// ksw_wave.h -- device-side building blocks of the banded-DP kernels that more than one translation unit uses: the task and
// parameter records, the DPP wave primitives and wave_task, the wave-per-task form of ksw_extend / ksw_nw / ksw_global
// (one task on the 64 lanes of the calling wavefront, moving band followed exactly).  ksw_kernels.hip launches it as
// ksw_wave_kernel; map_kernels.hip calls it from inside the per-read mapping kernel.  Everything is TU-local (anonymous
// namespace, forceinline device functions).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "../../include/alfa_hot.h"

namespace {

constexpr int NEG_INF = -0x40000000;  // MINUS_INF, ksw.c:454
constexpr int WPB = 4;                // wavefronts (tasks) per workgroup
enum { MODE_NW = 0, MODE_GLOBAL = 1, MODE_EXTEND = 2 };

struct DevTask {
    int64_t qoff, toff;   // byte offsets into the packed sequence buffer
    int64_t zoff;         // GLOBAL: byte offset of this task's direction matrix
    int64_t cigoff;       // GLOBAL: offset (u32 units) of this task's CIGAR slot
    int64_t ehoff;        // scalar kernel: offset (int2 units) of this task's eh[] scratch
    int32_t qlen, tlen, w, h0;
    int32_t orig;         // index in the caller's task array
    int32_t cigcap;       // GLOBAL: capacity of the CIGAR slot
    int32_t cigcap2;      // GLOBAL, lane kernel with LDS: row stride (cells) of the wave's interleaved direction bytes
    int32_t pad;
};

struct DevParams {
    uint64_t matrow[8];   // matrow[t] byte q = mat[t*m+q]
    int32_t m, gapo, gape, zdrop;
    int32_t maxs, pad_;   // largest matrix entry (>= 0): bounds a task's scores by h0 + qlen * maxs
};

struct ExtOut { int32_t score, qle, tle, gtle, gscore, max_off; };

// the device form of validated scoring parameters (1 <= m <= 8, non-negative penalties)
inline void dev_params_from(const alfa_dp_params& p, DevParams& dp) {
    memset(&dp, 0, sizeof(dp));
    for (int t = 0; t < p.m; ++t)
        for (int q = 0; q < p.m; ++q) dp.matrow[t] |= (uint64_t)(uint8_t)p.mat[t * p.m + q] << (8 * q);
    dp.m = p.m; dp.gapo = p.gapo; dp.gape = p.gape; dp.zdrop = p.zdrop;
    for (int i = 0; i < p.m * p.m; ++i) dp.maxs = dp.maxs > p.mat[i] ? dp.maxs : p.mat[i];
}

// ---------------------------------------------------------------- wave primitives
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Cross-lane steps are DPP moves (ALU latency), not ds_bpermute round trips through the LDS crossbar: a row of the DP needs
// ~30 of them back to back, and they are the critical path of the wave-per-task kernel.
//   row_shr:n  lane i <- lane i-n inside its row of 16 (0x110+n)     wave_shr:1  lane i <- lane i-1 (0x138)
//   row_bcast:15 / :31  last lane of the previous row / half -> next (0x142 / 0x143)   wave_shl:1  lane i <- lane i+1 (0x130)
// Lanes without a source keep `old`.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false); }

// inclusive prefix max across the 64 lanes
__device__ __forceinline__ int wave_incl_max_scan(int v, int identity) {
    v = max(v, dpp<0x111, 0xf>(identity, v));
    v = max(v, dpp<0x112, 0xf>(identity, v));
    v = max(v, dpp<0x114, 0xf>(identity, v));
    v = max(v, dpp<0x118, 0xf>(identity, v));
    v = max(v, dpp<0x142, 0xa>(identity, v));
    v = max(v, dpp<0x143, 0xc>(identity, v));
    return v;
}
// exclusive prefix max across the 64 lanes; lane 0 receives `identity`
__device__ __forceinline__ int wave_excl_max_scan(int v, int identity) {
    return dpp<0x138, 0xf>(identity, wave_incl_max_scan(v, (int)0x80000000));
}
__device__ __forceinline__ int wave_max(int v) {
    return __builtin_amdgcn_readlane(wave_incl_max_scan(v, (int)0x80000000), 63);
}
__device__ __forceinline__ int wave_min(int v) { return -wave_max(-v); }     // callers pass values > INT_MIN
// value of lane `src` (wave-uniform index)
__device__ __forceinline__ int lane_bcast(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
// lane i <- lane i+1; lane 63 <- fill
__device__ __forceinline__ int wave_shift_down1(int v, int fill) { return dpp<0x130, 0xf>(fill, v); }

// max(a - b, 0) for 0 <= a, b: one v_sub_u32 with the clamp bit
__device__ __forceinline__ int sub_floor0(int a, int b) { return (int)__builtin_elementwise_sub_sat((unsigned)a, (unsigned)b); }

__device__ __forceinline__ int score_of(uint64_t mr, uint32_t q) { return (int)(int8_t)(mr >> (8 * q)); }
// the same with the residue kept as its shift amount (8 x code): the band kernels carry query residues in that form
__device__ __forceinline__ int score_at(uint64_t mr, uint32_t q8) { return (int)(int8_t)(mr >> q8); }

// A wavefront's rows are a serial chain of 100-200 instructions each at ~6 cycles per instruction: a 10 000-row task takes
// milliseconds even with a SIMD to itself (tools/dp_long_probe.py: 232 / 360 / 491 ns per row for ksw_nw / ksw_global /
// ksw_extend at 201 diagonals), and a call ends when its longest chain does.  Wavefronts with long chains therefore raise
// their issue priority (s_setprio) over the crowd of short tasks that shares their SIMD.
__device__ __forceinline__ void chain_priority(int rows) {
    if (rows >= 4096) __builtin_amdgcn_s_setprio(3);
    else if (rows >= 2048) __builtin_amdgcn_s_setprio(2);
    else if (rows >= 1024) __builtin_amdgcn_s_setprio(1);
}

// ---------------------------------------------------------------- wave-per-task kernel
// one task on the 64 lanes of the calling wavefront (bands of up to 64 x C diagonals); all lanes call it together
// PACK (ksw_extend): the row maximum and its last column come out of one reduction over (H << 8 | diagonal); the caller
// passes true only when the task's scores stay below 2^22 (every H <= h0 + qlen * maxs: wave_task_packs)
// results are wave-uniform and come back through `eo` (ksw_extend) / `score_out` (ksw_nw, ksw_global)
// ZCOMPACT (ksw_global): rows of direction bytes are `zstride` bytes apart (a multiple of C, > 2w + 1) instead of 64 * C, and only
// the lanes inside a row store -- a short task's matrix then fits a few KB of LDS (map_kernels.hip)
// ZEQ (ksw_global): bit 6 of a cell's direction byte says whether the cell's two residues are equal (same nt4 code) -- the
// walk of map_kernels.hip then writes '=' / 'X' runs directly instead of comparing read and text again (the walk's two-bit
// fields are bits 0-1, 2-3 and 4-5; bits 3 and 4 stay clear)
template <int MODE, int C, bool PACK = false, bool ZCOMPACT = false, bool ZEQ = false>
__device__ __forceinline__ void wave_task_run(const DevTask T, const uint8_t* __restrict__ seq, const DevParams* __restrict__ prm,
                                              uint8_t* __restrict__ zbuf, ExtOut& eo, int& score_out, int zstride = 64 * C) {
    const int lane = lane_id();
    // the task is the same for all 64 lanes: its fields in scalar registers make the row loop and its bounds wave-uniform
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto uni64 = [&](int64_t v) { return (int64_t)(((uint64_t)(uint32_t)uni((int)((uint64_t)v >> 32)) << 32) | (uint32_t)uni((int)v)); };
    const uint8_t* __restrict__ q = seq + uni64(T.qoff);
    const uint8_t* __restrict__ tg = seq + uni64(T.toff);
    const int qlen = uni(T.qlen), tlen = uni(T.tlen), w = uni(T.w);
    const int gapo = prm->gapo, gape = prm->gape, gapoe = gapo + gape;
    const int zdrop = prm->zdrop;
    constexpr int FLOOR = (MODE == MODE_EXTEND) ? 0 : NEG_INF;
    constexpr int ROWB = 64 * C;  // direction bytes per row
    uint8_t* const zrow0 = MODE == MODE_GLOBAL ? zbuf + uni64(T.zoff) : nullptr;

    int Hd[C], E[C];
    uint32_t qc[C];
    int h0 = uni(T.h0);
    if (MODE == MODE_EXTEND && h0 < 0) h0 = 0;
    const int eh1 = h0 > gapoe ? h0 - gapoe : 0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int d = lane * C + c;
        const int j = d - w;  // column of this diagonal on row 0
        qc[c] = (j >= 0 && j < qlen) ? 8u * q[j] : 0u;
        if (MODE == MODE_EXTEND) {
            // first row of eh[] (ksw.c:377-379): h0, max(h0-gapoe,0), then -gape per column while > 0
            int v = 0;
            if (j == 0) v = h0;
            else if (j > 0) { v = eh1 - (j - 1) * gape; v = v > 0 ? v : 0; }
            Hd[c] = v;
            E[c] = 0;
        } else {
            // ksw.c:486-489 / 558-561
            int v = NEG_INF;
            if (j == 0) v = 0;
            else if (j > 0 && j <= w && j <= qlen) v = -(gapo + gape * j);
            Hd[c] = v;
            E[c] = NEG_INF;
        }
    }

    // ksw_nw / ksw_global: the band is static in diagonal coordinates, so no cell needs a per-row mask.  Cells left of column
    // 0 start at MINUS_INF and stay hugely negative (they only see each other, a score per row and the exact first-column value
    // that is injected below), cells right of the query's end or beyond diagonal 2w hold junk that only flows rightwards
    // (every dependency of (i,j) has a column <= j), and the one value that crosses back into the band -- E of diagonal
    // 2w+1, read by diagonal 2w one row later -- is pinned to MINUS_INF, which is what the reference's untouched eh[j].e holds
    constexpr bool STATIC_BAND = MODE != MODE_EXTEND;
    bool dead[C];
#pragma unroll
    for (int c = 0; c < C; ++c) dead[c] = lane * C + c == 2 * w + 1;
    int beg = 0, end = qlen;                                    // EXTEND state
    int mx = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
    // the row's target residue and the query residue that enters the top diagonal come out of 8-byte windows that are loaded
    // eight rows ahead, and the matrix rows sit in registers (read by lane index): no row waits for a memory round trip
    const uint64_t mat_l = prm->matrow[lane & 7];                 // lane k (k < 8) holds row k of the scoring matrix
    const int mat_lo = (int)(uint32_t)mat_l, mat_hi = (int)(uint32_t)(mat_l >> 32);
    auto load8 = [](const uint8_t* __restrict__ p, int from, int len) -> uint64_t {      // bytes from..from+7 of p[0..len), 0 outside
        uint64_t v = 0;
        if (from >= 0 && from + 8 <= len) __builtin_memcpy(&v, p + from, 8);
        else
            for (int k = 0; k < 8; ++k)
                if (from + k >= 0 && from + k < len) v |= (uint64_t)p[from + k] << (8 * k);
        return v;
    };
    const int xq0 = 1 - w + ROWB - 1;                           // query position fed in after row 0
    uint64_t tcur = load8(tg, 0, tlen), tnxt = load8(tg, 8, tlen);
    uint64_t qcur = load8(q, xq0, qlen), qnxt = load8(q, xq0 + 8, qlen);

    int dcol[C];
#pragma unroll
    for (int c = 0; c < C; ++c) dcol[c] = lane * C + c;
    int base = -w;                       // column of diagonal 0 on row i
    for (int i = 0; i < tlen; ++i, ++base) {
        const int ph = 8 * (i & 7);
        const int tch = __builtin_amdgcn_readfirstlane((int)((tcur >> ph) & 0xffu));
        const int xq = i + xq0;
        const uint32_t qfeed = (xq >= 0 && xq < qlen) ? 8u * (uint32_t)((qcur >> ph) & 0xffu) : 0u;
        if ((i & 7) == 7) {
            tcur = tnxt; tnxt = load8(tg, i + 9, tlen);
            qcur = qnxt; qnxt = load8(q, xq + 9, qlen);
        }
        const uint64_t mr = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(mat_hi, tch) << 32) | (uint32_t)__builtin_amdgcn_readlane(mat_lo, tch);

        if (MODE == MODE_EXTEND) {
            beg = max(beg, base);
            end = min(min(end, base + 2 * w + 1), qlen);
        }
        const int db = beg - base, de = end - base;
        const uint32_t dwidth = (uint32_t)max(de - db, 0);

        int A[C], Bv[C];
        bool act[C];
        uint32_t dirH[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            act[c] = STATIC_BAND || (uint32_t)(dcol[c] - db) < dwidth;
            const int hh = Hd[c] + score_at(mr, qc[c]);
            const int e = E[c];
            if (MODE == MODE_EXTEND) { A[c] = hh > e ? hh : e; dirH[c] = 0; }
            else { dirH[c] = hh >= e ? 0u : 1u; A[c] = hh >= e ? hh : e; }
            const int bv = MODE == MODE_EXTEND ? sub_floor0(A[c], gapoe) : A[c] - gapoe;      // ksw_extend clamps at 0 (A >= 0 there)
            Bv[c] = act[c] ? bv : FLOOR;
        }
        // lane-local contribution to the F chain, then the cross-lane max-plus scan.  P[c] = what the lane's cells before c
        // hand to cell c: F(c) = max(f_in - c*gape, P[c]) for every cell at once instead of a chain through the cells
        // (F(c+1) = max(F(c) - gape, max(A(c), F(c)) - gapoe) = max(F(c) - gape, Bv(c)) because gapoe >= gape)
        int P[C + 1];
        P[1] = Bv[0];
#pragma unroll
        for (int c = 1; c < C; ++c) P[c + 1] = max(P[c] - gape, Bv[c]);
        const int L = P[C];
        const int Cg = C * gape;
        const int gx = wave_excl_max_scan(L + lane * Cg, 0);
        const int f_in = lane == 0 ? FLOOR : gx - (lane - 1) * Cg;

        int Hrow[C], Eo[C];
        uint32_t zpack = 0;
        int m_l = -1, mj_l = -1;       // EXTEND: lane-local row max and its last column (or their packed form in m_l)
        uint32_t hmin_l = 0xffffffffu;  // EXTEND: smallest H of the lane's in-band cells
        int zl_l = -1, zr_l = 0x7fffffff;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int F = c == 0 ? f_in : max(f_in - c * gape, P[c]);
            int h = A[c];
            uint32_t dd = dirH[c];
            if (MODE == MODE_EXTEND) { h = h > F ? h : F; }
            else { if (!(h >= F)) { dd = 2u; h = F; } }
            Hrow[c] = h;
            const int t = MODE == MODE_EXTEND ? sub_floor0(h, gapoe) : h - gapoe;
            const int e2 = E[c] - gape;
            const int f2 = F - gape;
            if (MODE == MODE_GLOBAL) {
                dd |= (e2 > t) ? 4u : 0u;
                dd |= (f2 > t) ? 32u : 0u;
                if (ZEQ) dd |= (qc[c] == 8u * (uint32_t)tch) ? 64u : 0u;
                zpack |= (act[c] ? dd : 0u) << (8 * c);
            }
            Eo[c] = (STATIC_BAND ? !dead[c] : act[c]) ? (e2 > t ? e2 : t) : FLOOR;
            if (act[c]) Hd[c] = h;
            if (MODE == MODE_EXTEND) {
                const int hx = act[c] ? h : -1;
                hmin_l = min(hmin_l, (uint32_t)hx);
                if (PACK) m_l = max(m_l, (int)(((uint32_t)hx << 8) | (uint32_t)dcol[c]));      // out of band: negative
                else if (act[c] && h >= m_l) { m_l = h; mj_l = base + dcol[c]; }               // ties -> later column (ksw.c:414)
            }
        }
        if (MODE == MODE_GLOBAL) {
            uint8_t* zr = zrow0 + (int64_t)i * (ZCOMPACT ? zstride : ROWB) + lane * C;
            if (!ZCOMPACT || lane * C < zstride) {
                if (C == 4) *(uint32_t*)zr = zpack;
                else if (C == 2) *(uint16_t*)zr = (uint16_t)zpack;
                else *zr = (uint8_t)zpack;
            }
        }

        if (MODE == MODE_EXTEND) {
            // ---- row maximum m and its last column mj (ksw.c:414-415)
            int m, mj;                                      // -1 when the row is empty
            if (PACK) {
                const int key = wave_max(m_l);
                m = key >> 8;
                mj = key < 0 ? -1 : base + (key & 255);
            } else {
                m = wave_max(m_l);
                mj = wave_max(m_l == m ? mj_l : -1);
            }
            const bool empty = !(beg < end);
            const int jfin = empty ? beg : end;             // value of the loop variable after the j loop
            int h1_init = h0 - (gapo + gape * (i + 1));     // first-column value (ksw.c:395-396)
            if (h1_init < 0) h1_init = 0;
            if (jfin == qlen) {                             // ksw.c:425-428
                int h1 = h1_init;
                if (!empty) {
                    // H(i,end-1): the value the reference leaves in h1
                    const int dl = end - 1 - base;
                    int hv = Hrow[0];
#pragma unroll
                    for (int c = 1; c < C; ++c) hv = (dl % C == c) ? Hrow[c] : hv;
                    h1 = lane_bcast(hv, dl / C);
                }
                max_ie = gscore > h1 ? max_ie : i;
                gscore = gscore > h1 ? gscore : h1;
            }
            const int mrow = m < 0 ? 0 : m;
            const int mjrow = m < 0 ? -1 : mj;
            // (straight-line scalar code: every instruction of the row is on the task's critical path, branches most of all)
            {
                int dlt = (i - max_i) - (jfin - max_j);
                dlt = dlt < 0 ? -dlt : dlt;
                const int stop = (int)(mrow == 0) | ((int)(zdrop > 0) & (int)(mx - mrow - dlt * gape > zdrop));   // ksw.c:429
                if (stop) break;
            }
            {
                const bool up = mrow > mx;
                int off = mjrow - i; off = off < 0 ? -off : off;
                off = max_off > off ? max_off : off;
                mx = up ? mrow : mx; max_i = up ? i : max_i; max_j = up ? mjrow : max_j; max_off = up ? off : max_off;
            }
            // ---- band shrink (ksw.c:435-438) in terms of this row's H values:
            //   eh[j].h == H(i,j-1) for beg<j<=end, eh[beg].h == h1_init
            // without an in-band zero the band keeps its left edge (or loses one column once the first-column value is 0) and
            // grows by one column on the right; a zero is rare while the extension is alive, and only then are the two
            // reductions run
            int nbeg = beg + (int)(h1_init == 0), nend = end + 1;
            if (__ballot(hmin_l == 0)) {        // wave-uniform
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const int j = base + lane * C + c;
                    if (act[c] && Hrow[c] == 0) {
                        if (j <= mjrow - 1) zl_l = j;                       // slots ascend with c: keeps the largest
                        if (j >= mjrow + 1 && j < zr_l) zr_l = j;
                    }
                }
                const int zl = wave_max(zl_l);
                const int zr = wave_min(zr_l);
                if (zl >= 0) nbeg = zl + 2;
                if (zr != 0x7fffffff) nend = zr + 1;
            }
            // inject H(i, beg-1) := h1_init on the diagonal just left of the band (read by row i+1 if beg stays)
            {
                const int di = db - 1;
                if (di >= 0) {
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        if (lane * C + c == di) Hd[c] = h1_init;
                }
            }
            beg = nbeg;
            end = nend;
        } else {
            // first-column value H(i,-1) for the next row while the band still touches column 0 (ksw.c:497,568)
            const int di = w - i - 1;
            if (di >= 0) {
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (lane * C + c == di) Hd[c] = -(gapo + gape * (i + 1));
            }
        }

        // ---- shift E and the query characters one diagonal down for the next row
        {
            const int e_up = wave_shift_down1(Eo[0], FLOOR);
            const uint32_t q_up = (uint32_t)wave_shift_down1((int)qc[0], (int)qfeed);
#pragma unroll
            for (int c = 0; c < C - 1; ++c) { E[c] = Eo[c + 1]; qc[c] = qc[c + 1]; }
            E[C - 1] = e_up;
            qc[C - 1] = q_up;
        }
    }

    if (MODE == MODE_EXTEND) {
        eo.score = mx; eo.qle = max_j + 1; eo.tle = max_i + 1; eo.gtle = max_ie + 1; eo.gscore = gscore; eo.max_off = max_off;
    } else {
        // score = eh[qlen].h = H(tlen-1, qlen-1) when that cell lies inside the band, else MINUS_INF
        const int dl = qlen - tlen + w;
        int score = NEG_INF;
        if (dl >= 0 && dl <= 2 * w) {
            int hv = Hd[0];
#pragma unroll
            for (int c = 1; c < C; ++c) hv = (dl % C == c) ? Hd[c] : hv;
            score = lane_bcast(hv, dl / C);
        }
        score_out = score;
    }
}

// the same with the result stored by lane 0 at the task's slot of `out` (the batch kernels)
template <int MODE, int C, bool PACK = false>
__device__ __forceinline__ void wave_task(const DevTask T, const uint8_t* __restrict__ seq, const DevParams* __restrict__ prm,
                                          int32_t* __restrict__ out, uint8_t* __restrict__ zbuf) {
    ExtOut eo;
    int score;
    wave_task_run<MODE, C, PACK>(T, seq, prm, zbuf, eo, score);
    if (lane_id() == 0) {
        if (MODE == MODE_EXTEND) ((ExtOut*)out)[T.orig] = eo;
        else out[T.orig] = score;
    }
}

__device__ __forceinline__ bool wave_task_packs(const DevTask& T, const DevParams* __restrict__ prm) {
    const int h0 = T.h0 < 0 ? 0 : T.h0;
    return __builtin_amdgcn_readfirstlane((int64_t)h0 + (int64_t)T.qlen * prm->maxs < (1 << 22) ? 1 : 0) != 0;
}

}  // namespace
