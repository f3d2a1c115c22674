// ksw_kernels.hip -- inter-task batched banded integer DP for gfx950 (wave64).
//
// Batch forms of the reference's three live DP routines (src/ksw.c):
//   ksw_extend  :362-448   semi-global extension from h0, H/E clamped at 0, shrinking band, Z-drop
//   ksw_nw      :544-591   banded global score
//   ksw_global  :468-542   banded global + direction bytes + backtrack -> BAM CIGAR
//
// Design (no MFMA -- integer max-plus; registers only, no barriers in the row loop):
//   * Band-diagonal register layout.  A lane owns C consecutive band diagonals
//     d = l*C+c, d = j - i + w.  H(i-1,j-1) -> H(i,j) stays in the same register,
//     E(i,j) arrives from diagonal d+1 (one cross-lane shift per row) and the
//     horizontal F recurrence, the only true intra-row dependency, becomes an exact
//     max-plus prefix scan across the lanes:
//        F(d+1) = max(F(d) - gape, B(d)),  B(d) = A(d) - gapoe  (A = max(Hdiag+s, E))
//     which is lane-local over C slots plus one exclusive max-scan of
//     G(l) = L(l) + l*C*gape.  Integer max/plus is associative, so the scan is
//     bit-identical to the reference's left-to-right loop.
//   * Row maxima (m, mj), the Z-drop test, the gscore cell and the band-shrink scans
//     of ksw_extend are reductions over the same registers.
//   * Two widths: ksw_group_kernel gives a task one DPP row (16 lanes, bands of up to 16/32/64
//     diagonals, four tasks per wavefront: the short-read case); ksw_wave_kernel gives it the whole
//     wavefront (bands of up to 128/256 diagonals).  All cross-lane steps are DPP moves.
//   * ksw_global packs one direction byte per cell (reference encoding, ksw.c:506-519)
//     into coalesced per-row stores to HBM; a second kernel walks them back (from LDS copies
//     when the matrix is small).
//   * Tasks whose band does not fit 256 diagonals, and degenerate 0-length tasks, run in a
//     lane-per-task kernel that follows the reference loop literally (the rare tail).
//   * The kernel bins of a batch run as independent chains on side streams (launch_all).
#include <chrono>
#include <algorithm>
#include <memory>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "device_ctx.h"
#include "ksw_wave.h"

namespace {

template <int MODE, int C>
__global__ __launch_bounds__(64 * WPB) void ksw_wave_kernel(const DevTask* __restrict__ tasks, int ntask,
                                                           const uint8_t* __restrict__ seq,
                                                           const DevParams* __restrict__ prm,
                                                           int32_t* __restrict__ out, uint8_t* __restrict__ zbuf) {
    const int wave = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (wave >= ntask) return;
    const DevTask T = tasks[wave];
    chain_priority(__builtin_amdgcn_readfirstlane(T.tlen));
    if (MODE == MODE_EXTEND && wave_task_packs(T, prm)) wave_task<MODE, C, true>(T, seq, prm, out, zbuf);
    else wave_task<MODE, C>(T, seq, prm, out, zbuf);
}

// ---------------------------------------------------------------- group kernel: 16 lanes per task, 4 tasks per wavefront
// Short-read tasks have bands of 10-60 diagonals and 20-50 columns: one task per 64-lane wave leaves most lanes idle and
// the kernel is bound by the instructions issued per row.  Here a DPP row (16 lanes) owns a task -- same band-diagonal
// layout and exact max-plus scan as above with C = 1, 2 or 4 diagonals per lane (bands up to 16, 32, 64 diagonals) -- so
// one instruction stream advances four tasks, and every cross-lane step stays inside a row (row_shr / row_shl / row_ror).
// What was wave-uniform above (band limits, maxima, the Z-drop exit) is row-uniform here and kept per lane.
constexpr int GL16 = 16;              // lanes per task of the 16-lane form: the direction rows hold 16 x C bytes either way

__device__ __forceinline__ int row_excl_max_scan(int v, int identity) {
    constexpr int NEG = (int)0x80000000;
    v = max(v, dpp<0x111, 0xf>(NEG, v));
    v = max(v, dpp<0x112, 0xf>(NEG, v));
    v = max(v, dpp<0x114, 0xf>(NEG, v));
    v = max(v, dpp<0x118, 0xf>(NEG, v));
    return dpp<0x111, 0xf>(identity, v);
}
// the same for values >= 0 with identity 0 (ksw_extend: every H, E, F is clamped at 0): lanes shifted in from outside the row
// read 0 (bound_ctrl), which lets the compiler fuse every step into one v_max_i32_dpp
template <int CTRL>
__device__ __forceinline__ int dpp_zero(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ int row_excl_max_scan_nonneg(int v) {
    v = max(v, dpp_zero<0x111>(v));
    v = max(v, dpp_zero<0x112>(v));
    v = max(v, dpp_zero<0x114>(v));
    v = max(v, dpp_zero<0x118>(v));
    return dpp_zero<0x111>(v);
}
// maximum over the 16 lanes of the row, in every lane (rotations: max is idempotent)
__device__ __forceinline__ int row_max(int v) {
    // every lane of a rotation has a source, so `old` is never used: the identity of max there lets each step fuse into one
    // v_max_i32_dpp (with old = v the compiler keeps a separate v_mov_b32_dpp)
    constexpr int NEG = (int)0x80000000;
    v = max(v, dpp<0x128, 0xf>(NEG, v));
    v = max(v, dpp<0x124, 0xf>(NEG, v));
    v = max(v, dpp<0x122, 0xf>(NEG, v));
    v = max(v, dpp<0x121, 0xf>(NEG, v));
    return v;
}
// lane i <- lane i+1 of its row; the row's last lane <- fill
__device__ __forceinline__ int row_shift_down1(int v, int fill) { return dpp<0x101, 0xf>(fill, v); }

// The same three primitives for a task group of G = 16 lanes (a DPP row) or 8 lanes (half a row; gl = lane within the group).
// Half rows: the shifts cross the middle of the row and are masked per lane; the maximum is a butterfly of two quad
// permutations and the half-row mirror.
template <int G>
__device__ __forceinline__ int grp_excl_max_scan(int v, int identity, int gl) {
    if (G == 16) return row_excl_max_scan(v, identity);
    constexpr int NEG = (int)0x80000000;
    int t;
    t = dpp<0x111, 0xf>(NEG, v); v = max(v, gl >= 1 ? t : NEG);
    t = dpp<0x112, 0xf>(NEG, v); v = max(v, gl >= 2 ? t : NEG);
    t = dpp<0x114, 0xf>(NEG, v); v = max(v, gl >= 4 ? t : NEG);
    t = dpp<0x111, 0xf>(identity, v);
    return gl >= 1 ? t : identity;
}
template <int G>
__device__ __forceinline__ int grp_max(int v) {
    if (G == 16) return row_max(v);
    v = max(v, dpp<0xB1, 0xf>(v, v));       // quad_perm [1,0,3,2]
    v = max(v, dpp<0x4E, 0xf>(v, v));       // quad_perm [2,3,0,1]
    v = max(v, dpp<0x141, 0xf>(v, v));      // row_half_mirror
    return v;
}
template <int G>
__device__ __forceinline__ int grp_min(int v) { return -grp_max<G>(-v); }
template <int G>
__device__ __forceinline__ int grp_shift_down1(int v, int fill, int gl) {
    const int t = row_shift_down1(v, fill);
    return (G == 16 || gl != G - 1) ? t : fill;
}

template <int MODE, int C, int GL>
__global__ __launch_bounds__(64 * WPB) void ksw_group_kernel(const DevTask* __restrict__ tasks, int ntask,
                                                            const uint8_t* __restrict__ seq,
                                                            const DevParams* __restrict__ prm,
                                                            int32_t* __restrict__ out, uint8_t* __restrict__ zbuf) {
    __shared__ uint64_t s_mat[8];
    if (threadIdx.x < 8) s_mat[threadIdx.x] = prm->matrow[threadIdx.x];
    __syncthreads();
    const int lane = lane_id();
    const int gl = lane & (GL - 1);
    const int task = (blockIdx.x * WPB + (threadIdx.x >> 6)) * (64 / GL) + lane / GL;
    const bool valid = task < ntask;
    const DevTask T = tasks[valid ? task : ntask - 1];
    const uint8_t* __restrict__ q = seq + T.qoff;
    const uint8_t* __restrict__ tg = seq + T.toff;
    const int qlen = T.qlen, tlen = valid ? T.tlen : 0, w = T.w;
    const int gapo = prm->gapo, gape = prm->gape, gapoe = gapo + gape;
    const int zdrop = prm->zdrop;
    constexpr int FLOOR = (MODE == MODE_EXTEND) ? 0 : NEG_INF;
    constexpr int ROWB = GL * C;      // direction bytes per row of one task

    int Hd[C], E[C];
    uint32_t qc[C];
    int h0 = T.h0;
    if (MODE == MODE_EXTEND && h0 < 0) h0 = 0;
    const int eh1 = h0 > gapoe ? h0 - gapoe : 0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int d = gl * C + c;
        const int j = d - w;
        qc[c] = (j >= 0 && j < qlen) ? 8u * q[j] : 0u;
        if (MODE == MODE_EXTEND) {
            int v = 0;
            if (j == 0) v = h0;
            else if (j > 0) { v = eh1 - (j - 1) * gape; v = v > 0 ? v : 0; }
            Hd[c] = v;
            E[c] = 0;
        } else {
            int v = NEG_INF;
            if (j == 0) v = 0;
            else if (j > 0 && j <= w && j <= qlen) v = -(gapo + gape * j);
            Hd[c] = v;
            E[c] = NEG_INF;
        }
    }

    // ksw_nw / ksw_global: static band, no per-row cell masks (see wave_task); a task whose rows are over only keeps its H
    constexpr bool STATIC_BAND = MODE != MODE_EXTEND;
    bool dead[C];
#pragma unroll
    for (int c = 0; c < C; ++c) dead[c] = gl * C + c == 2 * w + 1;
    int beg = 0, end = qlen;
    int mx = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
    bool done = false;
    // rows run to the longest of the wave's targets
    int tmax = tlen;
    if (GL == 8) tmax = max(tmax, __shfl_xor(tmax, 8, 64));
    tmax = max(tmax, __shfl_xor(tmax, 16, 64));
    tmax = max(tmax, __shfl_xor(tmax, 32, 64));
    chain_priority(__builtin_amdgcn_readfirstlane(tmax));
    uint32_t tnext = tlen > 0 ? tg[0] : 0u;           // target residue of row i, fetched one row ahead
    uint64_t mr = s_mat[tnext & 7];
    tnext = tlen > 1 ? tg[1] : 0u;

    for (int i = 0; i < tmax; ++i) {
        const bool live = !done && i < tlen;
        const uint64_t mr_next = s_mat[tnext & 7];
        const uint32_t tnext2 = i + 2 < tlen ? tg[i + 2] : 0u;
        const int xq = i + 1 - w + ROWB - 1;
        const uint32_t qfeed = (xq >= 0 && xq < qlen) ? 8u * q[xq] : 0u;

        if (MODE == MODE_EXTEND) {
            if (beg < i - w) beg = i - w;
            if (end > i + w + 1) end = i + w + 1;
            if (end > qlen) end = qlen;
        } else {
            beg = i > w ? i - w : 0;
            end = i + w + 1 < qlen ? i + w + 1 : qlen;
        }
        const int base = i - w;
        const int db = beg - base, de = end - base;

        int A[C], Bv[C];
        bool act[C];
        uint32_t dirH[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int d = gl * C + c;
            act[c] = STATIC_BAND || (live && d >= db && d < de);
            const int hh = Hd[c] + score_at(mr, qc[c]);
            const int e = E[c];
            if (MODE == MODE_EXTEND) { A[c] = hh > e ? hh : e; dirH[c] = 0; }
            else { dirH[c] = hh >= e ? 0u : 1u; A[c] = hh >= e ? hh : e; }
            const int bv = MODE == MODE_EXTEND ? sub_floor0(A[c], gapoe) : A[c] - gapoe;      // ksw_extend clamps at 0 (A >= 0 there)
            Bv[c] = act[c] ? bv : FLOOR;
        }
        int P[C + 1];                  // see wave_task: F(c) = max(f_in - c*gape, P[c])
        P[1] = Bv[0];
#pragma unroll
        for (int c = 1; c < C; ++c) P[c + 1] = max(P[c] - gape, Bv[c]);
        const int L = P[C];
        const int Cg = C * gape;
        const int gx = (MODE == MODE_EXTEND && GL == 16) ? row_excl_max_scan_nonneg(L + gl * Cg) : grp_excl_max_scan<GL>(L + gl * Cg, 0, gl);
        const int f_in = gl == 0 ? FLOOR : gx - (gl - 1) * Cg;

        int Hrow[C], Eo[C];
        uint64_t zpack = 0;
        int m_l = -1, mj_l = -1;
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
                zpack |= (uint64_t)(act[c] ? dd : 0u) << (8 * c);
            }
            if (STATIC_BAND) Eo[c] = dead[c] ? FLOOR : (e2 > t ? e2 : t);
            else Eo[c] = act[c] ? (e2 > t ? e2 : t) : (live ? FLOOR : E[c]);
            if (STATIC_BAND ? live : act[c]) Hd[c] = h;
            if (MODE == MODE_EXTEND && act[c]) {
                if (h >= m_l) { m_l = h; mj_l = base + gl * C + c; }
            }
        }
        if (MODE == MODE_GLOBAL && live) {
            uint8_t* zr = zbuf + T.zoff + (int64_t)i * ROWB + gl * C;
            if (C == 8) *(uint64_t*)zr = zpack;
            else if (C == 4) *(uint32_t*)zr = (uint32_t)zpack;
            else if (C == 2) *(uint16_t*)zr = (uint16_t)zpack;
            else *zr = (uint8_t)zpack;
        }

        if (MODE == MODE_EXTEND) {
            const int m = grp_max<GL>(m_l);
            const int mj = grp_max<GL>(m_l == m ? mj_l : -1);
            const bool empty = !(beg < end);
            const int jfin = empty ? beg : end;
            int h1 = h0 - (gapo + gape * (i + 1));
            if (h1 < 0) h1 = 0;
            const int h1_init = h1;
            if (!empty) {
                // H(i,end-1): the value the reference leaves in h1 (all values are >= 0 here)
                const int dl = end - 1 - base;
                int hv = -1;
#pragma unroll
                for (int c = 0; c < C; ++c) hv = (gl * C + c == dl) ? Hrow[c] : hv;
                h1 = grp_max<GL>(hv);
            }
            int zl = -1, zr = 0x7fffffff;
            bool lane_zero = false;
#pragma unroll
            for (int c = 0; c < C; ++c) lane_zero |= act[c] & (Hrow[c] == 0);
            if (__ballot(lane_zero & live)) {   // wave-uniform; in-band zeros are rare while an extension is alive
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const int j = base + gl * C + c;
                    const int mjrow_l = m < 0 ? -1 : mj;
                    if (act[c] && Hrow[c] == 0) {
                        if (j <= mjrow_l - 1) zl_l = j;
                        if (j >= mjrow_l + 1 && j < zr_l) zr_l = j;
                    }
                }
                zl = grp_max<GL>(zl_l);
                zr = grp_min<GL>(zr_l);
            }
            if (live) {
                if (jfin == qlen) {
                    max_ie = gscore > h1 ? max_ie : i;
                    gscore = gscore > h1 ? gscore : h1;
                }
                const int mrow = m < 0 ? 0 : m;
                const int mjrow = m < 0 ? -1 : mj;
                int dlt = (i - max_i) - (jfin - max_j);
                dlt = dlt < 0 ? -dlt : dlt;
                if (mrow == 0 || (zdrop > 0 && mx - mrow - dlt * gape > zdrop)) done = true;     // ksw.c:429
                else {
                    if (mrow > mx) {
                        mx = mrow; max_i = i; max_j = mjrow;
                        int off = mjrow - i; off = off < 0 ? -off : off;
                        max_off = max_off > off ? max_off : off;
                    }
                    const int di = db - 1;
                    if (di >= 0) {
#pragma unroll
                        for (int c = 0; c < C; ++c)
                            if (gl * C + c == di) Hd[c] = h1_init;
                    }
                    int nbeg;
                    if (zl >= 0) nbeg = zl + 2;
                    else if (h1_init == 0) nbeg = beg + 1;
                    else nbeg = beg;
                    beg = nbeg;
                    end = (zr != 0x7fffffff) ? zr + 1 : end + 1;
                }
            }
        } else if (live) {
            const int di = w - i - 1;
            if (di >= 0) {
#pragma unroll
                for (int c = 0; c < C; ++c)
                    if (gl * C + c == di) Hd[c] = -(gapo + gape * (i + 1));
            }
        }

        // shift E and the query characters one diagonal down for the next row (rows that are over keep their state)
        {
            const int e_up = grp_shift_down1<GL>(Eo[0], FLOOR, gl);
            const uint32_t q_up = (uint32_t)grp_shift_down1<GL>((int)qc[0], (int)qfeed, gl);
            if (STATIC_BAND || live) {
#pragma unroll
                for (int c = 0; c < C - 1; ++c) { E[c] = Eo[c + 1]; qc[c] = qc[c + 1]; }
                E[C - 1] = e_up;
                qc[C - 1] = q_up;
            }
        }
        mr = mr_next;
        tnext = tnext2;
    }

    if (MODE == MODE_EXTEND) {
        if (valid && gl == 0) {
            ExtOut* o = (ExtOut*)out + T.orig;
            o->score = mx; o->qle = max_j + 1; o->tle = max_i + 1; o->gtle = max_ie + 1; o->gscore = gscore; o->max_off = max_off;
        }
    } else {
        const int dl = qlen - tlen + w;
        int hv = (int)0x80000000;
#pragma unroll
        for (int c = 0; c < C; ++c) hv = (gl * C + c == dl) ? Hd[c] : hv;
        const int score = (dl >= 0 && dl <= 2 * w) ? grp_max<GL>(hv) : NEG_INF;
        if (valid && gl == 0) out[T.orig] = score;
    }
}

// ---------------------------------------------------------------- backtrack of the wave kernel's direction bytes
// one lane per task; follows ksw.c:524-538 in band-diagonal coordinates.  CIGAR runs are written from the
// end of the task's slot towards its start, so the slot tail holds the CIGAR already in forward order.
__global__ void ksw_backtrack_kernel(const DevTask* __restrict__ tasks, int ntask, const uint8_t* __restrict__ zbuf,
                                     uint32_t* __restrict__ cig, int32_t* __restrict__ ncig, int ROWB) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntask) return;
    const DevTask T = tasks[t];
    const int qlen = T.qlen, tlen = T.tlen, w = T.w;
    uint32_t* slot = cig + T.cigoff;
    int pos = T.cigcap;   // next free position is pos-1
    int n = 0;
    uint32_t cur_op = 0xffffffffu, cur_len = 0;
    auto push = [&](uint32_t op, uint32_t len) {
        if (op == cur_op) { cur_len += len; return; }
        if (cur_op != 0xffffffffu) { slot[--pos] = cur_len << 4 | cur_op; ++n; }
        cur_op = op; cur_len = len;
    };
    if (tlen > qlen + w) { ncig[T.orig] = -1; return; }   // reference indexes outside the matrix: unspecified
    int i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
    int which = 0;
    const uint8_t* z = zbuf + T.zoff;
    while (i >= 0 && k >= 0) {
        const int d = k - i + w;
        const uint32_t zv = z[(int64_t)i * ROWB + d];
        which = (zv >> (which << 1)) & 3;
        if (which == 0) { push(0, 1); --i; --k; }
        else if (which == 1) { push(2, 1); --i; }
        else { push(1, 1); --k; }
    }
    if (i >= 0) push(2, (uint32_t)(i + 1));
    if (k >= 0) push(1, (uint32_t)(k + 1));
    if (cur_op != 0xffffffffu) { slot[--pos] = cur_len << 4 | cur_op; ++n; }
    ncig[T.orig] = n;
}

// Same walk for long tasks, one wavefront per task: the direction matrix is pulled through LDS in tiles of TILE_ROWS rows with
// coalesced 16-byte loads by all lanes -- one HBM round trip per tile instead of one per step (a 5 kbp x 5 kbp alignment is ~10^4
// dependent steps: ~10 ms walking HBM, the critical path of a long-read batch) -- and the walk inside a tile takes a whole
// DIAGONAL RUN per step: in the match state lane t looks at the cell t steps up the diagonal, a ballot finds the first cell
// that leaves it, and the run is pushed as one CIGAR piece (gaps are ~25 residues apart at 10 % error; the walk was one
// dependent LDS read per residue).  Every lane carries the same walk state; lane 0 writes.  Tile rows are padded by 16 bytes
// so that a diagonal (stride one row) spreads over the LDS banks.
constexpr int TILE_ROWS = 64;
constexpr int LONG_TLEN = 512;        // from this many rows on, the tiled walk beats one lane per task walking HBM
__global__ __launch_bounds__(64) void ksw_backtrack_tile_kernel(const DevTask* __restrict__ tasks, int ntask, const uint8_t* __restrict__ zbuf,
                                                               uint32_t* __restrict__ cig, int32_t* __restrict__ ncig, int ROWB) {
    __shared__ uint4 tile[TILE_ROWS * (256 + 16) / 16];          // ROWB <= 256
    const int t = blockIdx.x;
    if (t >= ntask) return;
    const int lane = threadIdx.x;
    const DevTask T = tasks[t];
    const int qlen = T.qlen, tlen = T.tlen, w = T.w;
    if (tlen > qlen + w) { if (lane == 0) ncig[T.orig] = -1; return; }   // reference indexes outside the matrix: unspecified
    chain_priority(__builtin_amdgcn_readfirstlane(tlen));
    uint32_t* slot = cig + T.cigoff;
    int pos = T.cigcap, n = 0;
    uint32_t cur_op = 0xffffffffu, cur_len = 0;
    auto push = [&](uint32_t op, uint32_t len) {
        if (op == cur_op) { cur_len += len; return; }
        if (cur_op != 0xffffffffu) { --pos; if (lane == 0) slot[pos] = cur_len << 4 | cur_op; ++n; }
        cur_op = op; cur_len = len;
    };
    int i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
    int which = 0;
    const uint8_t* z = zbuf + T.zoff;                            // 16-byte aligned, rows of ROWB (a multiple of 16) bytes
    const int v_per_row = ROWB >> 4;
    const int TROW = ROWB + 16;                                  // bytes of a tile row in LDS
    const uint8_t* zt = (const uint8_t*)tile;
    while (i >= 0 && k >= 0) {                                   // i, k, which are wave-uniform throughout
        const int lo = i - TILE_ROWS + 1 > 0 ? i - TILE_ROWS + 1 : 0;
        const int nv = (i - lo + 1) * v_per_row;
        const uint4* src = (const uint4*)(z + (int64_t)lo * ROWB);
        for (int x = lane; x < nv; x += 64) {
            const int r = x / v_per_row, c = x - r * v_per_row;
            tile[r * (v_per_row + 1) + c] = src[x];
        }
        __syncthreads();
        while (i >= lo && k >= 0) {
            const int d = k - i + w;
            if (which == 0) {
                int avail = i - lo + 1;
                avail = avail < k + 1 ? avail : k + 1;
                avail = avail < 64 ? avail : 64;
                const uint32_t zv = lane < avail ? zt[(i - lane - lo) * TROW + d] : 1u;
                const uint64_t stay = __ballot((zv & 3u) == 0u);
                const int run = stay == ~0ull ? 64 : __builtin_ctzll(~stay);       // <= avail: the lanes beyond it report "leaves"
                if (run > 0) { push(0, (uint32_t)run); i -= run; k -= run; }
                if (run < avail) {
                    which = (int)(__shfl((int)zv, run, 64) & 3);
                    if (which == 1) { push(2, 1); --i; }
                    else { push(1, 1); --k; }
                }
            } else {
                const uint32_t zv = zt[(i - lo) * TROW + d];
                which = (zv >> (which << 1)) & 3;
                if (which == 0) { push(0, 1); --i; --k; }
                else if (which == 1) { push(2, 1); --i; }
                else { push(1, 1); --k; }
            }
        }
        __syncthreads();
    }
    if (i >= 0) push(2, (uint32_t)(i + 1));
    if (k >= 0) push(1, (uint32_t)(k + 1));
    if (cur_op != 0xffffffffu) { --pos; if (lane == 0) slot[pos] = cur_len << 4 | cur_op; ++n; }
    if (lane == 0) ncig[T.orig] = n;
}

// Same walk for small tasks (direction matrix of at most ZC bytes): every lane first copies its task's matrix into its own
// LDS slice with independent 16-byte loads, then walks it there -- ~50 dependent LDS reads instead of ~50 dependent HBM reads.
constexpr int ZC = 512;        // 64 tasks per workgroup
template <int ZC, int TPB>
__global__ __launch_bounds__(TPB) void ksw_backtrack_lds_kernel(const DevTask* __restrict__ tasks, int ntask, const uint8_t* __restrict__ zbuf,
                                                               uint32_t* __restrict__ cig, int32_t* __restrict__ ncig, int ROWB) {
    __shared__ uint4 zl[TPB * ZC / 16];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntask) return;
    const DevTask T = tasks[t];
    const int qlen = T.qlen, tlen = T.tlen, w = T.w;
    const int nv = (tlen * ROWB + 15) >> 4;
    const uint4* src = (const uint4*)(zbuf + T.zoff);
    uint4* mine = zl + threadIdx.x * (ZC / 16);
    for (int k = 0; k < nv; ++k) mine[k] = src[k];
    const uint8_t* z = (const uint8_t*)mine;
    uint32_t* slot = cig + T.cigoff;
    int pos = T.cigcap;
    int n = 0;
    uint32_t cur_op = 0xffffffffu, cur_len = 0;
    auto push = [&](uint32_t op, uint32_t len) {
        if (op == cur_op) { cur_len += len; return; }
        if (cur_op != 0xffffffffu) { slot[--pos] = cur_len << 4 | cur_op; ++n; }
        cur_op = op; cur_len = len;
    };
    if (tlen > qlen + w) { ncig[T.orig] = -1; return; }   // reference indexes outside the matrix: unspecified
    int i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
    int which = 0;
    while (i >= 0 && k >= 0) {
        const uint32_t zv = z[i * ROWB + (k - i + w)];
        which = (zv >> (which << 1)) & 3;
        if (which == 0) { push(0, 1); --i; --k; }
        else if (which == 1) { push(2, 1); --i; }
        else { push(1, 1); --k; }
    }
    if (i >= 0) push(2, (uint32_t)(i + 1));
    if (k >= 0) push(1, (uint32_t)(k + 1));
    if (cur_op != 0xffffffffu) { slot[--pos] = cur_len << 4 | cur_op; ++n; }
    ncig[T.orig] = n;
}

// ---------------------------------------------------------------- lane-per-task kernels
// The reference loops (ksw.c:391-443, 491-522, 524-538) run literally, one task per lane.  Two storage policies:
//   LDS=true   the throughput path for short tasks (the bulk of a short-read workload: queries of 20-100 residues):
//              eh[] and the query of every lane live in LDS, interleaved by lane (entry j of lane l at [j*64+l]: one
//              conflict-free ds_read_b64/ds_write_b64 per cell), direction bytes go to HBM interleaved the same way
//              (coalesced per wave), and 64 independent tasks keep all lanes of the wave busy -- unlike the
//              wave-per-task kernel, whose band of 2w+1 diagonals leaves most lanes idle on queries this short.
//   LDS=false  generic fallback for tasks outside every other kernel's envelope (global-memory eh[], contiguous z).
template <bool LDS>
struct LaneStore {
    int2* eh;            // LDS: base of the wave's interleaved array (+lane); global: this task's array
    uint8_t* q;          // LDS: base of the wave's interleaved query bytes (+lane); global: unused
    const uint8_t* qg;
    __device__ __forceinline__ int2 get(int j) const { return LDS ? eh[j * 64] : eh[j]; }
    __device__ __forceinline__ void set(int j, int2 v) { if (LDS) eh[j * 64] = v; else eh[j] = v; }
    __device__ __forceinline__ int geth(int j) const { return LDS ? eh[j * 64].x : eh[j].x; }
    __device__ __forceinline__ uint32_t qat(int j) const { return LDS ? q[j * 64] : qg[j]; }
};

template <int MODE, bool LDS>
__global__ __launch_bounds__(64) void ksw_lane_kernel(const DevTask* __restrict__ tasks, int ntask, const uint8_t* __restrict__ seq,
                                                     const DevParams* __restrict__ prm, int32_t* __restrict__ out, int2* __restrict__ ehbuf,
                                                     uint8_t* __restrict__ zbuf, uint32_t* __restrict__ cig, int32_t* __restrict__ ncig,
                                                     int qcap) {
    extern __shared__ int2 lds_eh[];        // LDS: (qcap+2)*64 int2, then (qcap+1)*64 query bytes
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntask) return;
    const DevTask T = tasks[t];
    const uint8_t* q = seq + T.qoff;
    const uint8_t* tg = seq + T.toff;
    const int qlen = T.qlen, tlen = T.tlen;
    int w = T.w;
    const int gapo = prm->gapo, gape = prm->gape, gapoe = gapo + gape;
    LaneStore<LDS> st;
    st.qg = q;
    if (LDS) {
        st.eh = lds_eh + threadIdx.x;
        st.q = (uint8_t*)(lds_eh + (size_t)(qcap + 2) * 64) + threadIdx.x;
        for (int j = 0; j < qlen; ++j) st.q[j * 64] = q[j];
    } else {
        st.eh = ehbuf + T.ehoff;   // .x = h, .y = e ; qlen+2 entries
        st.q = nullptr;
    }
    // direction bytes: LDS policy -> cell c of this lane at z[c*64] (T.zoff already includes the lane), row stride T.cigcap2 cells
    const int64_t zs = LDS ? 64 : 1;
    if (MODE == MODE_EXTEND) {
        int h0 = T.h0 < 0 ? 0 : T.h0;
        for (int j = 0; j <= qlen + 1; ++j) st.set(j, make_int2(0, 0));
        st.set(0, make_int2(h0, 0));
        { int prev = h0 > gapoe ? h0 - gapoe : 0; if (qlen + 1 >= 1) st.set(1, make_int2(prev, 0));
          for (int j = 2; j <= qlen && prev > gape; ++j) { prev -= gape; st.set(j, make_int2(prev, 0)); } }
        int mx = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
        int beg = 0, end = qlen;
        uint32_t tnext = tlen > 0 ? tg[0] : 0u;
        for (int i = 0; i < tlen; ++i) {
            int f = 0, h1, m = 0, mj = -1, j;
            const uint64_t mr = prm->matrow[tnext];
            if (i + 1 < tlen) tnext = tg[i + 1];
            h1 = h0 - (gapo + gape * (i + 1));
            if (h1 < 0) h1 = 0;
            if (beg < i - w) beg = i - w;
            if (end > i + w + 1) end = i + w + 1;
            if (end > qlen) end = qlen;
            int2 p = beg < end ? st.get(beg) : make_int2(0, 0);
            for (j = beg; j < end; ++j) {
                const int2 pn = st.get(j + 1);           // next cell's state: independent of this iteration's store
                int h = p.x, e = p.y;
                p.x = h1;
                h += score_of(mr, st.qat(j));
                h = h > e ? h : e;
                h = h > f ? h : f;
                h1 = h;
                mj = m > h ? mj : j;
                m = m > h ? m : h;
                h -= gapoe;
                h = h > 0 ? h : 0;
                e -= gape;
                e = e > h ? e : h;
                p.y = e;
                st.set(j, p);
                f -= gape;
                f = f > h ? f : h;
                p = pn;
            }
            st.set(end, make_int2(h1, 0));
            if (j == qlen) {
                max_ie = gscore > h1 ? max_ie : i;
                gscore = gscore > h1 ? gscore : h1;
            }
            int dlt = (i - max_i) - (j - max_j);
            dlt = dlt < 0 ? -dlt : dlt;
            if (m == 0 || (prm->zdrop > 0 && mx - m - dlt * gape > prm->zdrop)) break;
            if (m > mx) {
                mx = m; max_i = i; max_j = mj;
                int off = mj - i; off = off < 0 ? -off : off;
                max_off = max_off > off ? max_off : off;
            }
            for (j = mj; j >= beg && st.geth(j); --j) {}
            beg = j + 1;
            for (j = mj + 2; j <= end && st.geth(j); ++j) {}
            end = j;
        }
        ExtOut* o = (ExtOut*)out + T.orig;
        o->score = mx; o->qle = max_j + 1; o->tle = max_i + 1; o->gtle = max_ie + 1; o->gscore = gscore; o->max_off = max_off;
    } else {
        const int n_col = LDS ? T.cigcap2 : (qlen < 2 * w + 1 ? qlen : 2 * w + 1);     // row stride of z, in cells
        uint8_t* z = zbuf + T.zoff;
        st.set(0, make_int2(0, NEG_INF));
        int j;
        for (j = 1; j <= qlen && j <= w; ++j) st.set(j, make_int2(-(gapo + gape * j), NEG_INF));
        for (; j <= qlen; ++j) st.set(j, make_int2(NEG_INF, NEG_INF));
        uint32_t tnext = tlen > 0 ? tg[0] : 0u;
        for (int i = 0; i < tlen; ++i) {
            int f = NEG_INF, h1;
            const uint64_t mr = prm->matrow[tnext];
            if (i + 1 < tlen) tnext = tg[i + 1];
            const int beg = i > w ? i - w : 0;
            const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
            h1 = beg == 0 ? -(gapo + gape * (i + 1)) : NEG_INF;
            int2 p = beg < end ? st.get(beg) : make_int2(0, 0);
            for (j = beg; j < end; ++j) {
                const int2 pn = st.get(j + 1);
                int h = p.x, e = p.y;
                uint32_t d;
                p.x = h1;
                h += score_of(mr, st.qat(j));
                d = h >= e ? 0u : 1u;
                h = h >= e ? h : e;
                d = h >= f ? d : 2u;
                h = h >= f ? h : f;
                h1 = h;
                h -= gapoe;
                e -= gape;
                d |= e > h ? 4u : 0u;
                e = e > h ? e : h;
                p.y = e;
                st.set(j, p);
                f -= gape;
                d |= f > h ? 32u : 0u;
                f = f > h ? f : h;
                if (MODE == MODE_GLOBAL) z[((int64_t)i * n_col + (j - beg)) * zs] = (uint8_t)d;
                p = pn;
            }
            st.set(end, make_int2(h1, NEG_INF));
        }
        out[T.orig] = st.geth(qlen);
        if (MODE == MODE_GLOBAL) {
            uint32_t* slot = cig + T.cigoff;
            int pos = T.cigcap, n = 0;
            uint32_t cur_op = 0xffffffffu, cur_len = 0;
            auto push = [&](uint32_t op, uint32_t len) {
                if (op == cur_op) { cur_len += len; return; }
                if (cur_op != 0xffffffffu) { slot[--pos] = cur_len << 4 | cur_op; ++n; }
                cur_op = op; cur_len = len;
            };
            if (tlen > qlen + w) { ncig[T.orig] = -1; return; }
            int i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1, which = 0;
            while (i >= 0 && k >= 0) {
                const uint32_t zv = z[((int64_t)i * n_col + (k - (i > w ? i - w : 0))) * zs];
                which = (zv >> (which << 1)) & 3;
                if (which == 0) { push(0, 1); --i; --k; }
                else if (which == 1) { push(2, 1); --i; }
                else { push(1, 1); --k; }
            }
            if (i >= 0) push(2, (uint32_t)(i + 1));
            if (k >= 0) push(1, (uint32_t)(k + 1));
            if (cur_op != 0xffffffffu) { slot[--pos] = cur_len << 4 | cur_op; ++n; }
            ncig[T.orig] = n;
        }
    }
}

// ---------------------------------------------------------------- band-in-registers kernel (ksw_lane2_kernel)
// The short-read task stream (0.7-1.5 tasks per read, queries of 1-100 residues, bands of 7-49 diagonals,
// 200-700 cells per task) is inter-task parallel work.  G lanes own a task (G = 1, 2, 4, 8 or 16: 64/G tasks per wavefront),
// each lane C = ND/G consecutive band diagonals in registers, statically indexed and fully unrolled:
//   * H(i-1,j-1) -> H(i,j) stays in its register; E moves one register down per row (written in place: slot c-1 is free
//     once slot c has been read); F is a running register along the row.  No scan, no reduction.
//   * lanes of a task are SKEWED by one row: in iteration k lane g works on row k - g.  F (and the running row maximum
//     of ksw_extend) enter lane g from what lane g-1 finished one iteration earlier (one DPP move); the E that lane g's
//     last slot needs comes from lane g+1's FIRST slot of the same iteration, which is computed C-1 slot steps before it is
//     read.  The critical path of a task is (rows + G - 1) iterations of C slots instead of rows x ND.
//   * the query sits in LDS at byte position j + w of the task's string (PAD elsewhere), so slot s of row r reads byte
//     r + s: one ds_read_b32 per 4 slots, one v_alignbyte + one v_perm_b32 (four scores out of the packed 8-entry matrix
//     row of the lane's target residue) per 4 slots;
//   * band limits are one unsigned compare per slot; a slot outside the band keeps its H register (which is how the
//     first-column values H(i,-1) get in: they are preloaded into the diagonals that reach column 0 later) and feeds
//     FLOOR into E and F;
//   * ksw_extend: row maximum and its LAST column (ksw.c:414) are one packed key per slot, (H - gapoe) << 6 | slot;
//     gscore / max_ie need H(i, qlen-1) of every row: a diagonal that leaves the matrix through the last column is frozen
//     there, so they are read off the registers once, after the last row.  The kernel only follows the STATIC band: the
//     band of ksw_extend narrows only where an in-band H is 0 (ksw.c:435-438), so a task that meets one gives up and is
//     finished at once by its own wavefront with the moving-band code (wave_task).  Tasks whose first-column value reaches 0
//     while the band touches column 0 are kept out by lane2_class; zeros of row -1 are preloaded as DEAD.
//   * ksw_global: four direction BITS per cell (sign bits of the four comparisons, shifted in with v_alignbit) go to LDS,
//     16 bits per 4 slots, and lane 0 of the task walks them back right after the last row: no direction bytes in HBM, no
//     second kernel.  Only "regular" tasks come here (|qlen - tlen| <= w, both lengths >= 1): every in-band cell then has
//     a finite predecessor, so the exact drift of MINUS_INF outside the band (which irregular tasks expose) cannot reach
//     a result.
// The row loop must stay ONE basic block (no break, no short-circuit &&): any control flow in it makes the register
// allocator double-buffer the band.
constexpr int QPAD = 7;               // query code of positions outside the query: needs m <= 7
constexpr int DEAD = -(1 << 28);      // ksw_extend: H of a dead (zero) boundary cell as the diagonal predecessor

__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }           // v_max3_i32
// lane i <- lane i-1 / i+1 inside its DPP row of 16 (groups of G lanes never straddle a row); callers mask the group edges
__device__ __forceinline__ int from_prev_lane(int v) { return dpp<0x111, 0xf>(v, v); }
__device__ __forceinline__ int from_next_lane(int v) { return dpp<0x101, 0xf>(v, v); }

template <int MODE, int ND, int G>
__device__ __forceinline__ void lane2_body(const DevTask* __restrict__ tasks, int ntask, const uint8_t* __restrict__ seq,
                                           const DevParams* __restrict__ prm, int32_t* __restrict__ out,
                                           uint32_t* __restrict__ cig, int32_t* __restrict__ ncig,
                                           int* __restrict__ bail, int task_base, int tcap, int blk) {
    extern __shared__ uint32_t lds2[];
    constexpr int C = ND / G;         // band diagonals per lane
    constexpr int TPW = 64 / G;       // tasks per wavefront
    constexpr int NW4 = C / 4;        // query dwords per lane window
    constexpr int DH = ND / 4;        // direction half-words per row
    constexpr int FLOOR = MODE == MODE_EXTEND ? 0 : NEG_INF;
    static_assert(C % 4 == 0 && C * G == ND && ND <= 64 && (G & (G - 1)) == 0 && G <= 16, "band class");
    const int lane = threadIdx.x;
    const int g = lane & (G - 1), tw = lane / G;
    uint64_t* s_mat = (uint64_t*)lds2;
    uint32_t* ring = lds2 + 16;
    const int RW = ((tcap + ND + G + 8) >> 2) + 2;             // dwords of a task's query string
    const int TW = (tcap + G + 11) >> 2;                       // dwords of a task's target string
    uint32_t* tring = ring + RW * TPW;
    uint16_t* dirs = (uint16_t*)(tring + TW * TPW);
    if (lane < 8) s_mat[lane] = prm->matrow[lane];
    const int tpos = blk * TPW + tw;
    const bool valid = tpos < ntask;
    const DevTask T = tasks[valid ? tpos : ntask - 1];
    const uint8_t* __restrict__ q = seq + T.qoff;
    const uint8_t* __restrict__ tg = seq + T.toff;
    const int qlen = T.qlen, tlen = valid ? T.tlen : 0, w = T.w;
    const int gapo = prm->gapo, gape = prm->gape, gapoe = gapo + gape;
    const int zdrop = prm->zdrop;
    int h0 = T.h0;
    if (MODE == MODE_EXTEND && h0 < 0) h0 = 0;
    // (ksw_extend tasks whose first-column value max(h0 - gapo - gape*(i+1), 0) reaches 0 while the band still touches column 0
    // never come here -- lane2_class: the reference may then move `beg` depending on E at the band's edge, ksw.c:435-436)
    const int tmax = wave_max(tlen);

    // ---- the task's query string: PAD everywhere, query residue j at byte j + w (dword d of task tw at ring[d * TPW + tw]);
    // its target string: residue r at byte r + G (lane g reads byte k + G - g in iteration k), 0 elsewhere
    for (int d = g; d < RW; d += G) ring[d * TPW + tw] = 0x01010101u * QPAD;
    for (int d = g; d < TW; d += G) tring[d * TPW + tw] = 0u;
    __syncthreads();
    {
        // One (unaligned) dword load per four residues, four loads in flight per lane: the load addresses are chosen so that
        // every loaded dword lands on a whole dword of the string -- dword k of the query stream starts at residue 4k - sq,
        // sq = w & 3; the host packs sequences from byte 16 of the buffer on and leaves slack behind them, so the up to three
        // bytes read in front of / behind a sequence exist (they are masked to PAD / 0).  This is what a cold batch waits for.
        const uint32_t PAD4 = 0x01010101u * QPAD;
        auto stage = [&](const uint8_t* __restrict__ src, int len, int shift, int dword0, uint32_t pad4, uint32_t* __restrict__ dst, int ndw, int cap) {
            const int nmax = wave_max(ndw);
            for (int k0 = g; k0 < nmax; k0 += 4 * G) {
                uint32_t x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + u * G;
                    x[u] = 0;
                    if (k < ndw) __builtin_memcpy(&x[u], src + 4 * k - shift, 4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + u * G;
                    const int jf = 4 * k - shift;                                   // residue of the dword's first byte
                    int nlo = -jf, nhi = jf + 4 - len;
                    nlo = nlo < 0 ? 0 : nlo;
                    nhi = nhi < 0 ? 0 : (nhi > 4 ? 4 : nhi);
                    const uint32_t mask = (uint32_t)(0xffffffffull << (8 * nlo)) & (uint32_t)(0xffffffffull >> (8 * nhi));
                    if (k < ndw && k + dword0 < cap) dst[(k + dword0) * TPW + tw] = (x[u] & mask) | (pad4 & ~mask);
                }
            }
        };
        const int sq = w & 3;
        stage(q, qlen, sq, (w - sq) >> 2, PAD4, ring, valid ? (qlen + sq + 3) >> 2 : 0, RW);
        constexpr int ST = G & 3;
        stage(tg, tlen, ST, (G - ST) >> 2, 0u, tring, (tlen + ST + 3) >> 2, TW);
    }
    __syncthreads();

    // ---- row -1 / column -1 values of the lane's diagonals (see the header comment)
    int Hd[C], E[C];
    const int eh1 = h0 > gapoe ? h0 - gapoe : 0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int s = g * C + c;
        const int j0 = s - w;             // column of diagonal s on row 0
        int v;
        if (MODE == MODE_EXTEND) {
            const int a = eh1 - (j0 - 1) * gape;                          // ksw.c:377-379
            const int bq = h0 - (gapo + gape * (w - s));                  // H(w-s-1, -1), ksw.c:395-396
            v = j0 == 0 ? h0 : (j0 > 0 ? a : bq);
            // a zero in row -1 / column -1 is a dead cell: the reference extends no diagonal out of it (M ? M + s : 0,
            // ksw.c:403).  DEAD makes the first cell of such a diagonal max(e, f) as there; if that is 0 the task gives up
            // like on any other in-band zero
            v = v > 0 ? v : DEAD;
            E[c] = 0;
        } else {
            const int a = (j0 <= w && j0 <= qlen) ? -(gapo + gape * j0) : NEG_INF;   // ksw.c:486-489 / 558-561
            const int bq = -(gapo + gape * (w - s));                      // H(w-s-1, -1), ksw.c:497,568
            v = j0 == 0 ? 0 : (j0 > 0 ? a : bq);
            E[c] = NEG_INF;
        }
        Hd[c] = v;
    }

    bool done = !valid, give_up = false;
    int mx = h0, max_i = -1, max_j = -1, max_off = 0, last_row = -1;
    int quirk_h1 = -1, quirk_i = -1;                   // ksw_extend: the gscore update of the first empty row (j == qlen there)
    int Flast = FLOOR, klast = 0;
    uint32_t zlast = 0xffffffffu;
    uint32_t Wq[NW4 + 1], Wn[NW4 + 1];
    const int woff = g * (C - 1);                      // byte offset of the lane's window: slot g*C + c of row k - g is byte k + woff + c
    const int niter = tmax + G - 1;
    // software pipeline of the LDS reads (their latency would otherwise open every iteration): the query window one
    // iteration ahead, the target residue two ahead and its matrix row one ahead
    const uint8_t* tbytes = (const uint8_t*)tring;
    auto tres = [&](int k) -> uint32_t { const int p = k + G - g; return tbytes[((p >> 2) * TPW + tw) * 4 + (p & 3)]; };
    {
        const int d0 = woff >> 2;
#pragma unroll
        for (int j = 0; j <= NW4; ++j) Wn[j] = ring[(d0 + j) * TPW + tw];
    }
    uint64_t mr_n = s_mat[tres(0) & 7];
    uint32_t t_n = tres(1);

    for (int k = 0; k < niter; ++k) {
        const int r = k - g;                           // this lane's row
        const bool rowv = (r >= 0) & (r < tlen);
        const uint64_t mr = mr_n;
        mr_n = s_mat[t_n & 7];
        t_n = tres(k + 2);
        const int pos = k + woff;
#pragma unroll
        for (int j = 0; j <= NW4; ++j) Wq[j] = Wn[j];
        {
            const int d0 = (pos + 1) >> 2;
#pragma unroll
            for (int j = 0; j <= NW4; ++j) Wn[j] = ring[(d0 + j) * TPW + tw];
        }
        const uint32_t ph = (uint32_t)pos & 3u;
        const int rdir = rowv ? r : tcap;              // ksw_global: rows outside the task write to a spare row of the direction bits
        // band of this row in diagonal coordinates: slot s is inside iff db <= s < de
        const int db = w - r > 0 ? w - r : 0;
        int de = qlen - r + w;
        de = de < 2 * w + 1 ? de : 2 * w + 1;
        const bool nonempty = de > db;
        const int sbase = g * C - db;
        const uint32_t bw = (rowv & nonempty) ? (uint32_t)(de - db) : 0u;
        int F = FLOOR, kin = 0;
        uint32_t zin = 0xffffffffu;
        if (G > 1) {
            const int fp = from_prev_lane(Flast), kp = from_prev_lane(klast);
            const uint32_t zp = (uint32_t)from_prev_lane((int)zlast);
            F = g == 0 ? FLOOR : fp;
            kin = g == 0 ? 0 : kp;
            zin = g == 0 ? 0xffffffffu : zp;
        }
        int kown = 0;
        uint32_t zmin = zin;
        uint32_t acc = 0;
        int e_last = FLOOR;                            // E of the lane's last slot: lane g+1's first slot of this iteration
#pragma unroll
        for (int k4 = 0; k4 < NW4; ++k4) {
            const uint32_t sel = __builtin_amdgcn_alignbyte(Wq[k4 + 1], Wq[k4], ph);
            const uint32_t sc4 = __builtin_amdgcn_perm((uint32_t)(mr >> 32), (uint32_t)mr, sel);
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int c = k4 * 4 + b;
                const bool act = (uint32_t)(c + sbase) < bw;
                const int hh = Hd[c] + (int)(int8_t)(sc4 >> (8 * b));
                const int e = (c == C - 1) ? e_last : E[c];
                int t, en;
                if (MODE == MODE_EXTEND) {
                    const int h = max3i(hh, e, F);
                    Hd[c] = act ? h : Hd[c];
                    t = act ? sub_floor0(h, gapoe) : 0;
                    zmin = min(zmin, act ? (uint32_t)h : 0xffffffffu);
                    en = max(sub_floor0(e, gape), t);
                    F = max(sub_floor0(F, gape), t);
                    kown = max(kown, (t << 6) | c);
                } else if (MODE == MODE_NW) {
                    const int h = max3i(hh, e, F);
                    Hd[c] = act ? h : Hd[c];
                    t = act ? h - gapoe : NEG_INF;
                    en = max(e - gape, t);
                    F = max(F - gape, t);
                } else {
                    acc = __builtin_amdgcn_alignbit(acc, (uint32_t)(hh - e), 31);          // hh < e: came from E (ksw.c:506)
                    const int A = max(hh, e);
                    acc = __builtin_amdgcn_alignbit(acc, (uint32_t)(A - F), 31);           // < F: came from F
                    const int h = max(A, F);
                    Hd[c] = act ? h : Hd[c];
                    t = act ? h - gapoe : NEG_INF;
                    const int e2 = e - gape;
                    acc = __builtin_amdgcn_alignbit(acc, (uint32_t)(t - e2), 31);          // e2 > t: E extends
                    en = max(e2, t);
                    const int f2 = F - gape;
                    acc = __builtin_amdgcn_alignbit(acc, (uint32_t)(t - f2), 31);          // f2 > t: F extends
                    F = max(f2, t);
                    if (b == 3) dirs[(rdir * DH + (g * C + c) / 4) * TPW + tw] = (uint16_t)acc;
                }
                if (c > 0) E[c - 1] = en;
                else if (G > 1) { const int nx = from_next_lane(en); e_last = g == G - 1 ? FLOOR : nx; }
            }
            // keep the slots in program order: left to itself the scheduler runs the F chain of the whole row first and
            // keeps every H of the row live (twice the registers, half the waves per SIMD)
            __builtin_amdgcn_sched_barrier(0);
        }
        Flast = F;

        if (MODE == MODE_EXTEND) {
            // the row's packed maximum: slots of this lane, then the lanes before it (ties -> the later column)
            const int tkey = max(kin, kown + g * C);
            klast = tkey;
            zlast = zmin;
            // row epilogue of ksw_extend (ksw.c:424-433), branch-free; only the task's last lane has the whole row
            const bool live = rowv & !done;
            const int beg = r - w + db;                      // = max(r - w, 0)
            int end = r + w + 1;
            end = end < qlen ? end : qlen;
            int h1i = h0 - (gapo + gape * (r + 1));
            h1i = h1i > 0 ? h1i : 0;
            const bool row_empty = live & !nonempty;       // the reference's j loop does not run: m == 0 -> break
            const bool quirk = row_empty & (beg == qlen);  // ... after the gscore update fired with h1_init (j == qlen)
            const bool row_on = live & nonempty;
            const bool narrow = row_on & ((zmin == 0u) | ((tkey >> 6) == 0));   // the band may narrow (ksw.c:435-438)
            const bool row_ok = row_on & !narrow;
            const int m = (tkey >> 6) + gapoe;
            const int mj = r - w + (tkey & 63);
            int dlt = (r - max_i) - (end - max_j);
            dlt = dlt < 0 ? -dlt : dlt;
            const bool zd = row_ok & (zdrop > 0) & (mx - m - dlt * gape > zdrop);         // ksw.c:429 (no short circuit: no branch)
            const bool upd = row_ok & !zd & (m > mx);
            int off = mj - r;
            off = off < 0 ? -off : off;
            last_row = live ? r : last_row;
            quirk_h1 = quirk ? h1i : quirk_h1;
            quirk_i = quirk ? r : quirk_i;
            give_up = give_up | narrow;
            done = done | row_empty | narrow | zd;
            max_off = (upd & (off > max_off)) ? off : max_off;
            max_i = upd ? r : max_i;
            max_j = upd ? mj : max_j;
            mx = upd ? m : mx;
        }
    }

    // ---- results: the state of ksw_extend lives in the task's last lane, H values in whichever lane owns the diagonal
    const int lane_last = lane | (G - 1), lane_first = lane & ~(G - 1);
    if (MODE == MODE_EXTEND) {
        const int lr = __shfl(last_row, lane_last, 64);
        // gscore / max_ie: H(i, qlen-1) of row i = qlen-1+w-s is what diagonal s froze at (ties -> the later row, ksw.c:426)
        int gk = -1;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int s = g * C + c;
            const int is = qlen - 1 + w - s;
            const bool ok = (s <= 2 * w) & (is >= 0) & (is <= lr);
            const int key = (Hd[c] << 6) | (63 - s);
            gk = (ok & (key > gk)) ? key : gk;
        }
#pragma unroll
        for (int o = 1; o < G; o <<= 1) gk = max(gk, __shfl_xor(gk, o, 64));
        const bool mine = valid && g == G - 1;
        if (mine && !give_up) {
            int gscore = -1, max_ie = -1;
            if (gk >= 0) { gscore = gk >> 6; max_ie = qlen - 1 + w - (63 - (gk & 63)); }
            if (quirk_i >= 0 && !(gscore > quirk_h1)) { gscore = quirk_h1; max_ie = quirk_i; }
            ExtOut* o = (ExtOut*)out + T.orig;
            o->score = mx; o->qle = max_j + 1; o->tle = max_i + 1; o->gtle = max_ie + 1; o->gscore = gscore; o->max_off = max_off;
        }
        // tasks that gave up (rare: about one in 10^5 on the C3 stream) run here and now, one at a time on all 64 lanes of this
        // wavefront, with the kernel that follows the reference's moving band (ksw_wave_kernel's body): no list, no second launch
        uint64_t todo = __ballot(mine && give_up);
        if (todo && lane == 0) atomicAdd(bail, (int)__popcll(todo));
        while (todo) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int tp = __shfl(tpos, src, 64);
            wave_task<MODE_EXTEND, 1>(tasks[tp], seq, prm, out, nullptr);
        }
        return;
    }
    // NW / GLOBAL: eh[qlen].h = H(tlen-1, qlen-1), frozen in diagonal qlen - tlen + w
    {
        const int dl = qlen - tlen + w;
        int score = (int)0x80000000;
#pragma unroll
        for (int c = 0; c < C; ++c) score = (g * C + c == dl) ? Hd[c] : score;
#pragma unroll
        for (int o = 1; o < G; o <<= 1) score = max(score, __shfl_xor(score, o, 64));
        if (valid && g == 0) out[T.orig] = score;
    }
    if (MODE == MODE_GLOBAL) {
        __syncthreads();
        if (!valid || lane != lane_first) return;
        uint32_t* slot = cig + T.cigoff;
        int pos = T.cigcap, n = 0;
        uint32_t cur_op = 0xffffffffu, cur_len = 0;
        auto push = [&](uint32_t op, uint32_t len) {
            if (op == cur_op) { cur_len += len; return; }
            if (cur_op != 0xffffffffu) { slot[--pos] = cur_len << 4 | cur_op; ++n; }
            cur_op = op; cur_len = len;
        };
        int i = tlen - 1, kq = qlen - 1, which = 0;         // regular task: min(i+w+1, qlen) == qlen
        while (i >= 0 && kq >= 0) {
            const int s = kq - i + w;
            const uint32_t wd = dirs[(i * DH + (s >> 2)) * TPW + tw];
            const uint32_t nib = (wd >> (12 - 4 * (s & 3))) & 15u;
            if (which == 0) which = (nib & 4u) ? 2 : (int)((nib >> 3) & 1u);
            else if (which == 1) which = (nib & 2u) ? 1 : 0;
            else which = (nib & 1u) ? 2 : 0;
            if (which == 0) { push(0, 1); --i; --kq; }
            else if (which == 1) { push(2, 1); --i; }
            else { push(1, 1); --kq; }
        }
        if (i >= 0) push(2, (uint32_t)(i + 1));
        if (kq >= 0) push(1, (uint32_t)(kq + 1));
        if (cur_op != 0xffffffffu) { slot[--pos] = cur_len << 4 | cur_op; ++n; }
        ncig[T.orig] = n;
    }
}

// One launch per ksw kind: the grid is the concatenation of the classes' blocks (the box gives a process 4 hardware queues:
// a launch per class and chunk would serialise on them), long-target classes first.
constexpr int NL2 = 11;
// grid position k (longest critical path first) holds class cls[k]
struct L2Grid { int blk_end[NL2]; int task_start[NL2]; int task_cnt[NL2]; int tcap[NL2]; int cls[NL2]; };
template <int MODE>
__global__ __launch_bounds__(64) void ksw_lane2_kernel(L2Grid L, const DevTask* __restrict__ tasks, const uint8_t* __restrict__ seq,
                                                      const DevParams* __restrict__ prm, int32_t* __restrict__ out,
                                                      uint32_t* __restrict__ cig, int32_t* __restrict__ ncig, int* __restrict__ bail) {
    const int b = blockIdx.x;
    int k = 0;
    while (k < NL2 - 1 && b >= L.blk_end[k]) ++k;
    const int blk = b - (k ? L.blk_end[k - 1] : 0);
    const DevTask* t = tasks + L.task_start[k];
    const int n = L.task_cnt[k], base = L.task_start[k], tcap = L.tcap[k];
    // the classes at the front of the grid are the launch's critical path: their wavefronts issue ahead of the others on a SIMD
    if (k == 0) __builtin_amdgcn_s_setprio(3);
    else if (k <= 2) __builtin_amdgcn_s_setprio(2);
    switch (L.cls[k]) {
    case 0: lane2_body<MODE, 64, 16>(t, n, seq, prm, out, cig, ncig, bail, base, tcap, blk); break;
    case 1: lane2_body<MODE, 32, 8>(t, n, seq, prm, out, cig, ncig, bail, base, tcap, blk); break;
    case 2: lane2_body<MODE, 16, 4>(t, n, seq, prm, out, cig, ncig, bail, base, tcap, blk); break;
    case 3: lane2_body<MODE, 48, 4>(t, n, seq, prm, out, cig, ncig, bail, base, tcap, blk); break;
    case 4: lane2_body<MODE, 32, 2>(t, n, seq, prm, out, cig, ncig, bail, base, tcap, blk); break;
    case 5: lane2_body<MODE, 16, 1>(t, n, seq, prm, out, cig, ncig, bail, base, tcap, blk); break;
    case 6: lane2_body<MODE, 8, 1>(t, n, seq, prm, out, cig, ncig, bail, base, tcap, blk); break;
    case 7: lane2_body<MODE, 12, 1>(t, n, seq, prm, out, cig, ncig, bail, base, tcap, blk); break;
    case 8: lane2_body<MODE, 40, 2>(t, n, seq, prm, out, cig, ncig, bail, base, tcap, blk); break;
    case 9: lane2_body<MODE, 48, 2>(t, n, seq, prm, out, cig, ncig, bail, base, tcap, blk); break;
    default: lane2_body<MODE, 24, 1>(t, n, seq, prm, out, cig, ncig, bail, base, tcap, blk); break;
    }
}

}  // namespace

// ================================================================= host side
constexpr int NBIN = 9 + NL2 + 1;
constexpr int L2BIN = 9;        // first lane2 bin (NL2 classes, then ksw_global's long targets: class 0 in a launch of its own)
constexpr int L2_TCAP = 512;    // longest target (rows) of the band-in-registers kernel
struct alfa_dp_batch {
    int mode = 0, n = 0, device = 0;
    DevParams hp;
    std::vector<DevTask> ht;        // device order, .orig already remapped to the dense output index
    std::vector<int> dense_of;      // device order -> caller index
    // lane kernel (qlen <= 30, 62, 126), group kernel C=1,2,4, wave kernel C=2,4, generic, lane2 kernel classes (L2_CLASS)
    int bin_cnt[NBIN] = {};
    L2Grid l2grid, l2big;           // ksw_lane2_kernel: blocks, tasks and longest target of each class; l2big = ksw_global's long targets
    int l2blocks = 0, l2big_blocks = 0;
    size_t l2lds = 0, l2big_lds = 0;
    int nbig[3] = {0, 0, 0};        // group bins: tasks whose direction matrix exceeds ZC bytes (a prefix: sorted by tlen)
    int nlong[3] = {0, 0, 0};       // ... of which targets of at least LONG_TLEN rows (a shorter prefix): tiled wave-per-task backtrack
    int64_t cig_words = 0, cells = 0, zbytes = 0, eh_items = 0;
    size_t seq_bytes = 0, out_bytes = 0;
    alfa::DevBuf d_seq, d_tasks, d_out, d_z, d_cig, d_eh, d_bail;
    int launches = 0;                 // d_bail[0] counts the give-ups of all launches so far
    alfa::PinBuf h_out, h_cig;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool launched = false;
    bool ev1_pending = false;           // launched under lazy joins: ev1 is recorded at the fetch-time join
    double kernel_ms = 0;
    const alfa::Ctx* owner = nullptr;   // launch and fetch are ordered on the creating thread's streams: other threads are refused
};

namespace {

using alfa::Ctx;

int pack_params(const alfa_dp_params* p, DevParams* dp) {
    if (!p) return alfa::fail(ALFA_EINVAL, "null scoring parameters");
    if (p->m < 1 || p->m > 8) return alfa::fail(ALFA_EINVAL, "m (residue types) must be in 1..8");
    if (p->gapo < 0 || p->gape < 0) return alfa::fail(ALFA_EINVAL, "gap penalties must be passed as non-negative numbers");
    dev_params_from(*p, *dp);
    return ALFA_OK;
}

// effective band: ksw_extend's max_gap clamp (ksw.c:381-386) then the exact no-op clamp to max(qlen,tlen)
int effective_w(int mode, const alfa_dp_params* p, const alfa_dp_task& x, int maxs) {      // maxs: largest matrix entry (>= 0)
    int w = x.w;
    if (mode == MODE_EXTEND) {
        int max_gap = p->gape > 0 ? (int)((double)(x.qlen * maxs - p->gapo) / p->gape + 1.) : 0x3fffffff;
        max_gap = max_gap > 1 ? max_gap : 1;
        w = w < max_gap ? w : max_gap;
    }
    const int lim = std::max(x.qlen, x.tlen);
    if (w > lim) w = lim;   // the band never binds beyond this: identical loops
    return w;
}

// cells of the static band envelope: sum_i (min(i+w+1,qlen) - max(i-w,0))^+   (SURVEY 8d), in closed form (a loop over the
// rows of every task was a tenth of the host time of a call)
int64_t band_cells(int64_t qlen, int64_t tlen, int64_t w) {
    if (qlen <= 0 || tlen <= 0) return 0;
    const int64_t T = std::min(tlen, qlen + w);                      // rows whose band still meets the query
    const int64_t a = std::min(std::max<int64_t>(qlen - w, 0), T);   // rows whose right edge is i + w + 1
    const int64_t b = std::min(w + 1, T);                            // rows whose left edge is column 0
    const int64_t hi = a * (w + 1) + a * (a - 1) / 2 + (T - a) * qlen;
    const int64_t lo = (T * (T - 1) / 2 - b * (b - 1) / 2) - (T - b) * w;
    return hi - lo;
}

// ksw_lane2_kernel's classes: ND band diagonals over G lanes.  A band width has a narrow form (few lanes per task: least
// work per cell, long critical path) and a wide form (the task's critical path is (rows + G - 1) iterations of ND/G slots).
// The launch is bound by instruction issue (profiles/r02_pmc_sq.csv: the VALU of every SIMD is busy ~87 % of the kernel),
// so a task goes to the class where it costs the fewest wavefront instructions, (rows + G - 1) x L2_IPI / (64/G), among
// those whose critical path (rows + G - 1) x L2_IPI stays below L2_LATCAP and whose LDS stays in budget; class 0 takes
// what no other class can (up to L2_TCAP rows).  L2_IPI = instructions of one row-loop iteration, from the ISA listing.
struct L2Class { int nd, g; };
constexpr L2Class L2_CLASS[NL2] = {{64, 16}, {32, 8}, {16, 4}, {48, 4}, {32, 2}, {16, 1}, {8, 1}, {12, 1}, {40, 2}, {48, 2}, {24, 1}};
// instructions of one row-loop iteration with C = ND/G slots per lane: a fit of the ISA listing (C = 4, 12, 16)
constexpr int L2_IPI_BASE[3] = {58, 41, 110}, L2_IPI_SLOT[3] = {12, 22, 16};      // MODE_NW, MODE_GLOBAL, MODE_EXTEND
inline int lane2_ipi(int mode, int k) { return L2_IPI_BASE[mode] + L2_IPI_SLOT[mode] * (L2_CLASS[k].nd / L2_CLASS[k].g); }
size_t lane2_lds(int mode, int nd, int g, int tcap) {
    const size_t tpw = 64 / g;
    const size_t rw = (size_t)((tcap + nd + g + 8) >> 2) + 2;
    const size_t tw = (size_t)(tcap + g + 11) >> 2;
    return 64 + (rw + tw) * tpw * 4 + (mode == MODE_GLOBAL ? (size_t)(tcap + 1) * (nd / 4) * tpw * 2 : 0);
}
int lane2_class(int mode, const alfa_dp_params* p, const alfa_dp_task& x, int w, int rows, int maxs) {
    static const bool off = getenv("ALFA_DP_NO_LANE2") != nullptr;
    static const int64_t latcap_env = getenv("ALFA_DP_LATCAP") ? atoll(getenv("ALFA_DP_LATCAP")) : 0;
    static const int64_t latcap_mode[3] = {16000, 16000, 24000};       // measured on the C3 stream (tools/dp_replay.py sweeps)
    const int64_t latcap = latcap_env ? latcap_env : latcap_mode[mode];
    // ksw_global: 32 KiB keeps the C3 stream's longest gap fills in the common launch (a second launch for a handful of
    // tasks costs a cross-stream fork/join, ~25 us: 0.137 -> 0.112 ms per batch)
    static const size_t ldscap_env = getenv("ALFA_DP_LDSCAP") ? (size_t)atoll(getenv("ALFA_DP_LDSCAP")) : 0;
    const size_t ldscap = ldscap_env ? ldscap_env : (mode == MODE_GLOBAL ? 32 * 1024 : 24 * 1024);
    if (off || p->m > QPAD || x.qlen < 1 || rows < 1) return -1;
    const int W = 2 * w + 1;
    if (W > 64) return -1;
    if (mode == MODE_EXTEND) {
        // scores stay far below 2^25 (packed row-maximum key)
        const int h0 = x.h0 < 0 ? 0 : x.h0;
        if ((int64_t)h0 + (int64_t)x.qlen * maxs >= (1 << 24)) return -1;
        // the first-column value h0 - gapo - gape*(i+1) must stay positive for as long as the band touches column 0 (rows
        // 0..w): once it is 0 the reference narrows the band from the left or not depending on E at the edge (ksw.c:435-436),
        // which the static band of the fast kernel does not follow
        if ((int64_t)h0 - p->gapo - (int64_t)p->gape * (w + 1) <= 0) return -1;
    } else if (std::abs(x.qlen - x.tlen) > w) return -1;         // "regular" tasks only (see the kernel's header comment)
    int best = -1;
    int64_t best_cost = 0;
    for (int k = 1; k < NL2; ++k) {
        const int g = L2_CLASS[k].g;
        const int64_t path = (int64_t)(rows + g - 1) * lane2_ipi(mode, k);
        if (W > L2_CLASS[k].nd || path > latcap || lane2_lds(mode, L2_CLASS[k].nd, g, rows) > ldscap) continue;
        const int64_t cost = path * g;           // x 64 / (64/g) tasks per wavefront
        if (best < 0 || cost < best_cost) { best = k; best_cost = cost; }
    }
    if (best >= 0) return best;
    if (rows > L2_TCAP) return -1;
    if (lane2_lds(mode, L2_CLASS[0].nd, L2_CLASS[0].g, rows) <= ldscap) return 0;
    return mode == MODE_GLOBAL ? NL2 : -1;       // class 0 in a launch of its own, with more LDS than the common launch is given
}

template <int MODE, int C>
void launch_wave(hipStream_t st, const DevTask* d_tasks, int n, const uint8_t* d_seq, const DevParams* d_prm, int32_t* d_out, uint8_t* d_z) {
    if (n <= 0) return;
    dim3 grid((n + WPB - 1) / WPB), block(64 * WPB);
    hipLaunchKernelGGL((ksw_wave_kernel<MODE, C>), grid, block, 0, st, d_tasks, n, d_seq, d_prm, d_out, d_z);
}

// a band class of 16 x C diagonals runs as 16 lanes x C diagonals (4 tasks per wavefront) or, with ALFA_DP_GROUP8 set, as 8 lanes
// x 2C diagonals (8 tasks per wavefront, same direction-byte layout).  Measured on the C3 task stream the 8-lane form is 7-15 %
// slower (extend 142 vs 153, nw 296 vs 347, global 66 vs 75 GCUPS): the cell work per lane doubles and the masked half-row
// shifts cost more than sharing the per-row reductions saves; it stays as a parity-tested A/B switch.
template <int MODE, int C>
void launch_group(hipStream_t st, const DevTask* d_tasks, int n, const uint8_t* d_seq, const DevParams* d_prm, int32_t* d_out, uint8_t* d_z) {
    if (n <= 0) return;
    static const bool half_rows = getenv("ALFA_DP_GROUP8") != nullptr;
    if (!half_rows) {
        const int per_block = WPB * 4;
        hipLaunchKernelGGL((ksw_group_kernel<MODE, C, 16>), dim3((n + per_block - 1) / per_block), dim3(64 * WPB), 0, st, d_tasks, n, d_seq, d_prm,
                           d_out, d_z);
    } else {
        const int per_block = WPB * 8;
        hipLaunchKernelGGL((ksw_group_kernel<MODE, 2 * C, 8>), dim3((n + per_block - 1) / per_block), dim3(64 * WPB), 0, st, d_tasks, n, d_seq, d_prm,
                           d_out, d_z);
    }
}

template <int MODE>
void launch_all(alfa_dp_batch* b, alfa::Ctx* cx) {
    hipStream_t st = cx->stream;
    const DevParams* d_prm = b->d_tasks.as<DevParams>();
    const DevTask* d_t = (const DevTask*)(d_prm + 1);
    const uint8_t* d_seq = b->d_seq.as<uint8_t>();
    int32_t* d_out = b->d_out.as<int32_t>();
    uint8_t* d_z = b->d_z.as<uint8_t>();
    uint32_t* d_cig = b->d_cig.as<uint32_t>();
    int32_t* d_ncig = MODE == MODE_GLOBAL ? (int32_t*)(d_cig + b->cig_words) : nullptr;
    const int* bc = b->bin_cnt;
    int o = 0;
    // short queries: lane-per-task with LDS-resident state
    static const int QCAP[3] = {30, 62, 126};
    for (int k = 0; k < 3; ++k) {
        if (bc[k]) {
            const size_t lds = (size_t)(QCAP[k] + 2) * 64 * sizeof(int2) + (size_t)(QCAP[k] + 1) * 64;
            if (lds > 48 * 1024) {
                static std::atomic<bool> raised[64];
                if (!raised[b->device & 63].exchange(true))
                    (void)hipFuncSetAttribute((const void*)ksw_lane_kernel<MODE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            }
            hipLaunchKernelGGL((ksw_lane_kernel<MODE, true>), dim3((bc[k] + 63) / 64), dim3(64), lds, st, d_t + o, bc[k], d_seq, d_prm, d_out,
                               (int2*)nullptr, d_z, d_cig, d_ncig, QCAP[k]);
        }
        o += bc[k];
    }
    // bands of up to 16 / 32 / 64 diagonals: 16 lanes per task
    // Every bin is an independent chain (fill kernel -> backtrack): the chains fork onto side streams and join again.
    // Tasks of a bin are ordered by tlen, longest first: the first nbig direction matrices exceed ZC bytes (their backtrack
    // walks HBM), the rest fit ZC (LDS slices); the two backtrack classes of a bin run side by side.
    // streams: 0-2 group bins, 3 the 128-diagonal wave bin (10 for ksw_extend), 5-7 the LDS backtracks of the group bins, 8-9 ksw_lane2_kernel; the
    // 256-diagonal wave bin (long reads: few, long-running wavefronts whose launch is mostly a latency tail) gets a stream per
    // ksw kind -- 4 nw, 11 extend, 13 global -- so that, with lazy joins, the tails of successive batches overlap instead of
    // queueing up (C4 step 20.2 -> 13 ms).  Which streams share one of the 4 hardware queues matters: moving the 128-diagonal
    // bin as well put the C3 stream's three irregular tasks on ksw_lane2_kernel's queue (0.33 -> 0.39 ms of DP per step)
    // (the fork event is recorded at the first use of a side stream: nothing has been enqueued on `st` before that, and a
    // batch whose tasks all run in ksw_lane2_kernel on `st` pays for no cross-stream dependency at all)
    bool forked = false;
    bool used[alfa::Ctx::NSIDE] = {};
    static const int side_mod = [] { const char* e = getenv("ALFA_SIDE_STREAMS"); int v = e ? atoi(e) : alfa::Ctx::NSIDE; return v < 1 ? 1 : (v > alfa::Ctx::NSIDE ? alfa::Ctx::NSIDE : v); }();
    auto side = [&](int k) -> hipStream_t {
        k %= side_mod;
        // a synchronous call orders its side chains behind the caller's stream; with lazy joins the chains are only ordered
        // among themselves (each side stream is in order) and joined at the next fetch / sync: a batch's inputs were uploaded
        // and waited for when it was created, so nothing on `st` -- e.g. the seed kernels of other reads -- has to finish first
        hipStream_t sk = cx->side_stream(k);
        if (sk == st) return st;            // no stream could be made: the main stream itself, nothing to fork or join
        if (!cx->lazy_join) {
            if (!forked) { (void)hipEventRecord(cx->fork_ev, st); forked = true; }
            if (!used[k]) (void)hipStreamWaitEvent(sk, cx->fork_ev, 0);
        }
        used[k] = true;
        return sk;
    };
    for (int g = 0; g < 3; ++g) {
        const int n = bc[3 + g], nbig = b->nbig[g], rowb = GL16 << g;
        if (n > 0) {
            hipStream_t s2 = side(g);
            if (g == 0) launch_group<MODE, 1>(s2, d_t + o, n, d_seq, d_prm, d_out, d_z);
            else if (g == 1) launch_group<MODE, 2>(s2, d_t + o, n, d_seq, d_prm, d_out, d_z);
            else launch_group<MODE, 4>(s2, d_t + o, n, d_seq, d_prm, d_out, d_z);
            if (MODE == MODE_GLOBAL) {
                if (n > nbig) {
                    hipStream_t s3 = nbig > 0 ? side(5 + g) : s2;
                    if (s3 != s2) { (void)hipEventRecord(cx->fill_ev[g], s2); (void)hipStreamWaitEvent(s3, cx->fill_ev[g], 0); }
                    hipLaunchKernelGGL((ksw_backtrack_lds_kernel<ZC, 64>), dim3((n - nbig + 63) / 64), dim3(64), 0, s3, d_t + o + nbig, n - nbig,
                                       d_z, d_cig, d_ncig, rowb);
                }
                const int nlong = std::min(b->nlong[g], nbig);
                if (nlong > 0)
                    hipLaunchKernelGGL(ksw_backtrack_tile_kernel, dim3(nlong), dim3(64), 0, s2, d_t + o, nlong, d_z, d_cig, d_ncig, rowb);
                if (nbig > nlong)
                    hipLaunchKernelGGL(ksw_backtrack_kernel, dim3((nbig - nlong + 63) / 64), dim3(64), 0, s2, d_t + o + nlong, nbig - nlong, d_z, d_cig,
                                       d_ncig, rowb);
            }
        }
        o += n;
    }
    // bands of up to 128 / 256 diagonals: a whole wavefront per task
    if (bc[6]) {
        hipStream_t s2 = side(MODE == MODE_EXTEND ? 10 : 3);
        launch_wave<MODE, 2>(s2, d_t + o, bc[6], d_seq, d_prm, d_out, d_z);
        if (MODE == MODE_GLOBAL)
            hipLaunchKernelGGL(ksw_backtrack_tile_kernel, dim3(bc[6]), dim3(64), 0, s2, d_t + o, bc[6], d_z, d_cig, d_ncig, 64 * 2);
    }
    o += bc[6];
    if (bc[7]) {
        hipStream_t s2 = side(MODE == MODE_NW ? 4 : MODE == MODE_EXTEND ? 11 : 13);
        launch_wave<MODE, 4>(s2, d_t + o, bc[7], d_seq, d_prm, d_out, d_z);
        if (MODE == MODE_GLOBAL)
            hipLaunchKernelGGL(ksw_backtrack_tile_kernel, dim3(bc[7]), dim3(64), 0, s2, d_t + o, bc[7], d_z, d_cig, d_ncig, 64 * 4);
    }
    o += bc[7];
    const int o8 = o;
    o += bc[8];
    // short tasks: band in registers (ksw_lane2_kernel); a ksw_extend task whose band starts to narrow is finished by its own
    // wavefront with the moving-band code (wave_task) and counted in d_bail[0]
    {
        int* d_cnt = b->d_bail.as<int>();
        int n_l2 = 0;
        for (int g = 0; g < NL2; ++g) n_l2 += bc[L2BIN + g];
        if (n_l2 + bc[L2BIN + NL2] > 0) {
            // a synchronous call runs them on `st` (no cross-stream dependency: 25-45 us less per call); with lazy joins
            // (alfa_set_lazy_join: the caller overlaps several batches and the seed kernels, which own `st`) on a side stream
            hipStream_t s8 = cx->lazy_join ? side(8) : st;
            // the long targets first: their few wavefronts are the longest-running ones
            if (b->l2big_blocks && b->l2big_lds > 48 * 1024) {
                static std::atomic<bool> raised[64];
                if (!raised[b->device & 63].exchange(true))
                    (void)hipFuncSetAttribute((const void*)ksw_lane2_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            }
            if (b->l2big_blocks)
                hipLaunchKernelGGL((ksw_lane2_kernel<MODE>), dim3(b->l2big_blocks), dim3(64), b->l2big_lds, side(9), b->l2big, d_t, d_seq, d_prm, d_out,
                                   d_cig, d_ncig, d_cnt);
            if (n_l2)
                hipLaunchKernelGGL((ksw_lane2_kernel<MODE>), dim3(b->l2blocks), dim3(64), b->l2lds, s8, b->l2grid, d_t, d_seq, d_prm, d_out, d_cig,
                                   d_ncig, d_cnt);
            b->launches++;
        }
    }
    bool any = false;
    for (int k = 0; k < alfa::Ctx::NSIDE; ++k)
        if (used[k]) { (void)hipEventRecord(cx->side_ev[k], cx->side[k]); any = true; }
    if (any) cx->sides_pending = true;
    if (!cx->lazy_join) (void)cx->join_sides();
    if (bc[8])
        hipLaunchKernelGGL((ksw_lane_kernel<MODE, false>), dim3((bc[8] + 63) / 64), dim3(64), 0, st, d_t + o8, bc[8], d_seq, d_prm, d_out,
                           b->d_eh.as<int2>(), d_z, d_cig, d_ncig, 0);
}

// (re)fills `b` for tasks ids[] of t[]; buffers of `b` are reused when large enough
int fill_batch(alfa_dp_batch* b, Ctx* cx, int mode, const alfa_dp_params* p, const alfa_dp_task* t, const int* ids, int n) {
    int rc = pack_params(p, &b->hp);
    if (rc) return rc;
    b->mode = mode; b->n = n; b->device = cx->device; b->launched = false; b->kernel_ms = 0;
    b->cig_words = b->cells = b->zbytes = b->eh_items = 0;
    for (int k = 0; k < NBIN; ++k) b->bin_cnt[k] = 0;
    b->l2blocks = b->l2big_blocks = 0; b->l2lds = b->l2big_lds = 0;
    for (int k = 0; k < 3; ++k) b->nbig[k] = b->nlong[k] = 0;
    b->ht.resize((size_t)n);
    b->dense_of.resize((size_t)n);
    if (!b->ev0) { ALFA_HIP(hipEventCreate(&b->ev0)); ALFA_HIP(hipEventCreate(&b->ev1)); }
    if (n == 0) return ALFA_OK;
    std::vector<int> w_eff((size_t)n), bins((size_t)n), perm((size_t)n), rows_eff((size_t)n);
    size_t seq_bytes = 0;
    // tasks of at most this many inner-loop iterations go to the lane-per-task kernel (0 = none); a lane runs its task
    // serially, so the bound caps the kernel's critical path
    static const int64_t lane_work = getenv("ALFA_DP_LANE_WORK") ? atoll(getenv("ALFA_DP_LANE_WORK")) : 0;
    for (int i = 0; i < n; ++i) {
        const alfa_dp_task& x = t[ids ? ids[i] : i];
        if (x.qlen < 0 || x.tlen < 0 || x.w < 0) return alfa::fail(ALFA_EINVAL, "negative length or band in a DP task");
        if ((x.qlen > 0 && !x.query) || (x.tlen > 0 && !x.target)) return alfa::fail(ALFA_EINVAL, "null sequence pointer in a DP task");
        w_eff[i] = effective_w(mode, p, x, b->hp.maxs);
        // ksw_extend never gets past row qlen + w: there beg = i - w reaches qlen and the reference leaves its row loop
        // (ksw.c:393-423), so the rows behind it are dropped here, for every kernel
        rows_eff[i] = mode == MODE_EXTEND ? (int)std::min<int64_t>(x.tlen, (int64_t)x.qlen + w_eff[i] + 1) : x.tlen;
        int bin;
        if (x.qlen == 0 || x.tlen == 0) bin = 8;
        else if (lane_work > 0 && x.qlen <= 126 && (int64_t)x.tlen * std::min<int64_t>(x.qlen, 2 * (int64_t)w_eff[i] + 1) <= lane_work)
            bin = x.qlen <= 30 ? 0 : x.qlen <= 62 ? 1 : 2;
        else {
            const int W = 2 * w_eff[i] + 1;
            bin = W <= 16 ? 3 : W <= 32 ? 4 : W <= 64 ? 5 : W <= 128 ? 6 : W <= 256 ? 7 : 8;
            const int l2 = lane2_class(mode, p, x, w_eff[i], rows_eff[i], b->hp.maxs);
            if (l2 >= 0) bin = L2BIN + l2;
        }
        bins[i] = bin;
        b->bin_cnt[bin]++;
        if (bin >= 3 && bin <= 5 && (int64_t)rows_eff[i] * (GL16 << (bin - 3)) > ZC) b->nbig[bin - 3]++;
        if (bin >= 3 && bin <= 5 && rows_eff[i] >= LONG_TLEN) b->nlong[bin - 3]++;
        seq_bytes += (size_t)x.qlen + (size_t)x.tlen;
    }
    // order: by bin; inside a bin the most work first, so that the 64 tasks of a lane-kernel wave (and co-resident waves of the
    // wave kernel) are alike and retire together
    std::iota(perm.begin(), perm.end(), 0);
    std::vector<int64_t> work((size_t)n);
    for (int i = 0; i < n; ++i) {
        const alfa_dp_task& x = t[ids ? ids[i] : i];
        work[i] = bins[i] < 3 ? (int64_t)rows_eff[i] * std::min<int64_t>(x.qlen, 2 * (int64_t)w_eff[i] + 1) : (int64_t)rows_eff[i];
    }
    if (n < (1 << 26)) {
        // one 64-bit key per task (bin, work descending, index): sorting plain integers is several times faster than a
        // comparator that chases three arrays, and this sort was a third of the host time of a call
        std::vector<uint64_t> key((size_t)n);
        for (int i = 0; i < n; ++i) {
            const uint64_t wk = 0xffffffffull - (uint64_t)std::min<int64_t>(work[i], 0xffffffffll);
            key[i] = (uint64_t)bins[i] << 58 | wk << 26 | (uint64_t)i;
        }
        std::sort(key.begin(), key.end());
        for (int i = 0; i < n; ++i) perm[i] = (int)(key[i] & ((1u << 26) - 1));
    } else
        std::sort(perm.begin(), perm.end(), [&](int a, int c) {
            if (bins[a] != bins[c]) return bins[a] < bins[c];
            if (work[a] != work[c]) return work[a] > work[c];
            return a < c;
        });
    b->seq_bytes = seq_bytes;
    if ((rc = cx->h_in.reserve(seq_bytes + 96)) || (rc = cx->h_tasks.reserve(sizeof(DevTask) * (size_t)n + sizeof(DevParams))) ||
        (rc = b->d_seq.reserve(seq_bytes + 96)) || (rc = b->d_tasks.reserve(sizeof(DevTask) * (size_t)n + sizeof(DevParams))))
        return rc;
    uint8_t* hs = cx->h_in.as<uint8_t>();
    memset(hs, 0, 16);
    DevParams* hprm = cx->h_tasks.as<DevParams>();
    DevTask* ht = (DevTask*)(hprm + 1);
    *hprm = b->hp;
    size_t so = 16;      // ksw_lane2_kernel's staging reads up to 3 bytes in front of a sequence
    const int rowb[NBIN] = {0, 0, 0, GL16 * 1, GL16 * 2, GL16 * 4, 64 * 2, 64 * 4, 0};     // direction bytes per row (HBM); 0 for the lane2 bins
    const int m = b->hp.m;
    int bin_start[NBIN + 1] = {0};
    for (int k = 0; k < NBIN; ++k) bin_start[k + 1] = bin_start[k] + b->bin_cnt[k];
    int64_t wave_zbase = 0;
    int wave_nc = 0;
    for (int k = 0; k < n; ++k) {
        const int li = perm[k];
        const int id = ids ? ids[li] : li;
        const alfa_dp_task& x = t[id];
        DevTask& d = ht[k];
        memset(&d, 0, sizeof(d));
        d.qoff = (int64_t)so;
        // copy, then one branch-free pass for the range check (the compiler vectorises both)
        auto pack = [&](const uint8_t* src, int len) -> bool {
            memcpy(hs + so, src, (size_t)len);
            unsigned bad = 0;
            for (int j = 0; j < len; ++j) bad |= (unsigned)(src[j] >= m);
            so += (size_t)len;
            return bad == 0;
        };
        if (!pack(x.query, x.qlen)) return alfa::fail(ALFA_EINVAL, "query residue code >= m");
        d.toff = (int64_t)so;
        if (!pack(x.target, x.tlen)) return alfa::fail(ALFA_EINVAL, "target residue code >= m");
        d.qlen = x.qlen; d.tlen = rows_eff[li]; d.w = w_eff[li]; d.h0 = x.h0; d.orig = k;
        b->dense_of[k] = li;
        if (bins[li] == 8) { d.ehoff = b->eh_items; b->eh_items += (int64_t)x.qlen + 2; }
        if (mode == MODE_GLOBAL) {
            if (bins[li] < 3) {
                // lane kernel: the 64 tasks of a wave share one interleaved direction-byte region
                const int in_bin = k - bin_start[bins[li]];
                if ((in_bin & 63) == 0) {
                    int64_t nc = 1, tl = 1;
                    for (int kk = k; kk < std::min(k + 64, bin_start[bins[li] + 1]); ++kk) {
                        const int l2 = perm[kk];
                        const alfa_dp_task& y = t[ids ? ids[l2] : l2];
                        nc = std::max<int64_t>(nc, std::min<int64_t>(y.qlen, 2 * (int64_t)w_eff[l2] + 1));
                        tl = std::max<int64_t>(tl, y.tlen);
                    }
                    wave_zbase = b->zbytes;
                    wave_nc = (int)nc;
                    b->zbytes += nc * tl * 64;
                }
                d.zoff = wave_zbase + (in_bin & 63);
                d.cigcap2 = wave_nc;
            } else {
                d.zoff = b->zbytes;
                if (bins[li] == 8) b->zbytes += std::min<int64_t>(x.qlen, 2 * (int64_t)d.w + 1) * x.tlen;
                else b->zbytes += (int64_t)x.tlen * rowb[bins[li]];      // lane2 bins keep their direction bits in LDS
                b->zbytes = (b->zbytes + 15) & ~(int64_t)15;
            }
            d.cigoff = b->cig_words;
            d.cigcap = x.qlen + x.tlen + 2;
            b->cig_words += d.cigcap;
        }
        b->cells += band_cells(x.qlen, x.tlen, d.w);
        b->ht[k] = d;
    }
    // ksw_lane2_kernel's grid: the classes' blocks one after the other; LDS for the neediest class
    // (the classes in the order of their longest wavefront's critical path, so that the launch does not end on one)
    int n_l2 = 0;
    int order[NL2];
    int64_t path[NL2];
    for (int g = 0; g < NL2; ++g) {
        order[g] = g;
        const int cnt = b->bin_cnt[L2BIN + g];
        path[g] = cnt ? (int64_t)(b->ht[bin_start[L2BIN + g]].tlen + L2_CLASS[g].g - 1) * lane2_ipi(mode, g) : -1;
    }
    std::stable_sort(order, order + NL2, [&](int a, int c) { return path[a] > path[c]; });
    for (int k = 0; k < NL2; ++k) {
        const int g = order[k];
        const int s0 = bin_start[L2BIN + g], cnt = b->bin_cnt[L2BIN + g];
        const int tpw = 64 / L2_CLASS[g].g;
        n_l2 += cnt;
        b->l2blocks += (cnt + tpw - 1) / tpw;
        b->l2grid.blk_end[k] = b->l2blocks;
        b->l2grid.task_start[k] = s0;
        b->l2grid.task_cnt[k] = cnt;
        b->l2grid.cls[k] = g;
        b->l2grid.tcap[k] = cnt ? b->ht[s0].tlen : 1;             // tasks of a bin are ordered longest target first
        if (cnt) b->l2lds = std::max(b->l2lds, lane2_lds(mode, L2_CLASS[g].nd, L2_CLASS[g].g, b->l2grid.tcap[k]));
    }
    if (b->bin_cnt[L2BIN + NL2]) {
        const int s0 = bin_start[L2BIN + NL2], cnt = b->bin_cnt[L2BIN + NL2], tpw = 64 / L2_CLASS[0].g;
        b->l2big_blocks = (cnt + tpw - 1) / tpw;
        for (int g = 0; g < NL2; ++g) { b->l2big.blk_end[g] = b->l2big_blocks; b->l2big.task_start[g] = s0; b->l2big.task_cnt[g] = g ? 0 : cnt; b->l2big.tcap[g] = b->ht[s0].tlen; b->l2big.cls[g] = 0; }
        b->l2big_lds = lane2_lds(mode, L2_CLASS[0].nd, L2_CLASS[0].g, b->ht[s0].tlen);
    }
    if (mode == MODE_EXTEND && n_l2) {
        if ((rc = b->d_bail.reserve(256))) return rc;
        ALFA_HIP(hipMemsetAsync(b->d_bail.p, 0, 256, cx->up));           // synchronised with the copies below
        b->launches = 0;
    }
    b->out_bytes = (mode == MODE_EXTEND ? sizeof(ExtOut) : sizeof(int32_t)) * (size_t)n;
    if ((rc = b->d_out.reserve(b->out_bytes)) || (rc = b->h_out.reserve(b->out_bytes))) return rc;
    if (b->eh_items && (rc = b->d_eh.reserve(sizeof(int2) * (size_t)b->eh_items))) return rc;
    if (mode == MODE_GLOBAL) {
        const size_t cb = 4 * (size_t)b->cig_words + 4 * (size_t)n;
        if ((rc = b->d_z.reserve((size_t)b->zbytes + 64)) || (rc = b->d_cig.reserve(cb)) || (rc = b->h_cig.reserve(cb))) return rc;
    }
    // on the upload stream: the host waits for these copies only, not for kernels of other calls still running on cx->stream
    hipStream_t st = cx->up;
    ALFA_HIP(hipMemcpyAsync(b->d_seq.p, hs, seq_bytes + 16, hipMemcpyHostToDevice, st));
    ALFA_HIP(hipMemcpyAsync(b->d_tasks.p, hprm, sizeof(DevParams) + sizeof(DevTask) * (size_t)n, hipMemcpyHostToDevice, st));
    ALFA_HIP(hipStreamSynchronize(st));   // staging buffers of the context are free again
    return ALFA_OK;
}

int launch_batch(alfa_dp_batch* b, Ctx* cx) {
    if (b->n == 0) { b->launched = true; return ALFA_OK; }
    hipStream_t st = cx->stream;
    ALFA_HIP(hipEventRecord(b->ev0, st));
    if (b->mode == MODE_EXTEND) launch_all<MODE_EXTEND>(b, cx);
    else if (b->mode == MODE_NW) launch_all<MODE_NW>(b, cx);
    else launch_all<MODE_GLOBAL>(b, cx);
    ALFA_HIP(hipGetLastError());
    // with lazy joins the batch's kernels run on side streams that `st` has not waited for yet: an event recorded here would
    // bracket almost none of them.  The closing event is then recorded at the fetch-time join (fetch_batch), so that
    // kernel_ms spans the whole batch (and whatever else the stream ran in between: documented in include/alfa_hot.h)
    b->ev1_pending = cx->lazy_join;
    if (!b->ev1_pending) ALFA_HIP(hipEventRecord(b->ev1, st));
    b->launched = true;
    return ALFA_OK;
}

// per-task CIGAR vectors are appended to `cigs` (indexed by the position in ids[]) when GLOBAL
int fetch_batch(alfa_dp_batch* b, Ctx* cx, void* out_host, const int* ids, std::vector<std::vector<uint32_t>>* cigs, int32_t* ncig_host) {
    if (!b->launched) return alfa::fail(ALFA_EINVAL, "fetch before launch");
    const int n = b->n;
    if (n == 0) return ALFA_OK;
    hipStream_t st = cx->stream;
    int jrc = cx->join_sides();
    if (jrc) return jrc;
    if (b->ev1_pending) { ALFA_HIP(hipEventRecord(b->ev1, st)); b->ev1_pending = false; }
    ALFA_HIP(hipMemcpyAsync(b->h_out.p, b->d_out.p, b->out_bytes, hipMemcpyDeviceToHost, st));
    if (b->mode == MODE_GLOBAL)
        ALFA_HIP(hipMemcpyAsync(b->h_cig.p, b->d_cig.p, 4 * (size_t)b->cig_words + 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    ALFA_HIP(hipStreamSynchronize(st));
    float ms = 0;
    ALFA_HIP(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    b->kernel_ms = ms;
    auto dst_index = [&](int k) { const int li = b->dense_of[k]; return ids ? ids[li] : li; };
    if (b->mode == MODE_EXTEND) {
        const ExtOut* r = b->h_out.as<ExtOut>();
        alfa_extend_result* dst = (alfa_extend_result*)out_host;
        for (int k = 0; k < n; ++k) memcpy(&dst[dst_index(k)], &r[k], sizeof(ExtOut));
    } else {
        const int32_t* r = b->h_out.as<int32_t>();
        int32_t* dst = (int32_t*)out_host;
        for (int k = 0; k < n; ++k) dst[dst_index(k)] = r[k];
    }
    if (b->mode == MODE_GLOBAL) {
        const uint32_t* hc = b->h_cig.as<uint32_t>();
        const int32_t* hn = (const int32_t*)(hc + b->cig_words);
        for (int k = 0; k < n; ++k) {
            const int id = dst_index(k);
            const int nc = hn[k];
            ncig_host[id] = nc;
            (*cigs)[id].clear();
            if (nc > 0) {
                const uint32_t* slot = hc + b->ht[k].cigoff;
                (*cigs)[id].assign(slot + b->ht[k].cigcap - nc, slot + b->ht[k].cigcap);
            }
        }
    }
    return ALFA_OK;
}

int current_ctx(Ctx** out, int device = -1) {
    if (alfa::device_count() <= 0) return alfa::fail(ALFA_ENODEV, "no HIP device is usable; libalfa_hot has no CPU fallback");
    int dev = device;
    if (dev < 0) (void)hipGetDevice(&dev);
    Ctx* cx = alfa::thread_ctx(dev);
    if (!cx) return ALFA_ENODEV;
    *out = cx;
    return ALFA_OK;
}

void release_batch(alfa_dp_batch* b) {
    b->d_seq.release(); b->d_tasks.release(); b->d_out.release(); b->d_z.release(); b->d_cig.release(); b->d_eh.release(); b->d_bail.release();
    b->h_out.release(); b->h_cig.release();
    if (b->ev0) { (void)hipEventDestroy(b->ev0); (void)hipEventDestroy(b->ev1); b->ev0 = b->ev1 = nullptr; }
}

int pack_cigars(const std::vector<std::vector<uint32_t>>& cigs, int n, uint32_t** cigar, int64_t* cig_off) {
    int64_t total = 0;
    for (int i = 0; i < n; ++i) { cig_off[i] = total; total += (int64_t)cigs[i].size(); }
    cig_off[n] = total;
    uint32_t* outc = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(total + 1));
    if (!outc) return alfa::fail(ALFA_ENOMEM, "out of host memory for CIGAR output");
    for (int i = 0; i < n; ++i)
        if (!cigs[i].empty()) memcpy(outc + cig_off[i], cigs[i].data(), sizeof(uint32_t) * cigs[i].size());
    *cigar = outc;
    return ALFA_OK;
}

// synchronous call: chunks of bounded device footprint through one per-thread reusable batch object
// the calling thread's recycled batch of a ksw kind; buried when the thread ends (device_ctx.h: a live thread releases it)
alfa_dp_batch* thread_batch(int mode) {
    struct Reuse {
        alfa_dp_batch* b[3] = {nullptr, nullptr, nullptr};
        ~Reuse() {
            for (auto*& x : b)
                if (x) {
                    alfa::bury([](void* p) { alfa_dp_batch* d = (alfa_dp_batch*)p; (void)hipSetDevice(d->device); release_batch(d); delete d; }, x);
                    x = nullptr;
                }
        }
    };
    thread_local Reuse reuse;
    if (!reuse.b[mode]) reuse.b[mode] = new alfa_dp_batch();
    return reuse.b[mode];
}

constexpr int64_t SYNC_LIMIT = (int64_t)6 << 30;      // device bytes of one batch of the synchronous calls
int64_t sync_need(int mode, const alfa_dp_task& x) {
    return (mode == MODE_GLOBAL ? (int64_t)x.tlen * 256 + 4 * ((int64_t)x.qlen + x.tlen) : 0) + (int64_t)x.qlen + x.tlen;
}

int run_sync(int mode, const alfa_dp_params* p, const alfa_dp_task* t, int n, void* out, uint32_t** cigar, int32_t* ncig, int64_t* cig_off) {
    alfa::g_last_timing = alfa::Timing();
    if (n < 0 || (n > 0 && (!t || !out))) return alfa::fail(ALFA_EINVAL, "null task/result array");
    if (mode == MODE_GLOBAL && n >= 0 && (!cigar || !ncig || !cig_off)) return alfa::fail(ALFA_EINVAL, "null CIGAR output argument");
    DevParams tmp;
    int rc = pack_params(p, &tmp);
    if (rc) return rc;
    std::vector<std::vector<uint32_t>> cigs;
    if (mode == MODE_GLOBAL) cigs.resize((size_t)n);
    if (n > 0) {
        Ctx* cx;
        if ((rc = current_ctx(&cx))) return rc;
        alfa_dp_batch* b = thread_batch(mode);
        const int64_t LIMIT = SYNC_LIMIT;
        std::vector<int> ids;
        int64_t need = 0;
        auto flush = [&]() -> int {
            if (ids.empty()) return ALFA_OK;
            int r;
            static const bool call_log = getenv("ALFA_CALL_LOG") != nullptr;      // diagnostics: where a synchronous call's time goes
            const auto t0 = std::chrono::steady_clock::now();
            if ((r = fill_batch(b, cx, mode, p, t, ids.data(), (int)ids.size()))) return r;
            const auto t1 = std::chrono::steady_clock::now();
            if ((r = launch_batch(b, cx))) return r;
            if ((r = fetch_batch(b, cx, out, ids.data(), mode == MODE_GLOBAL ? &cigs : nullptr, ncig))) return r;
            if (call_log) {
                const auto t2 = std::chrono::steady_clock::now();
                fprintf(stderr, "#CALL mode %d tasks %d seq_bytes %zu z_bytes %lld fill_ms %.3f run_fetch_ms %.3f kernel_ms %.3f\n", mode, (int)ids.size(),
                        b->seq_bytes, (long long)b->zbytes, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                        std::chrono::duration<double, std::milli>(t2 - t1).count(), b->kernel_ms);
            }
            alfa::g_last_timing.kernel_ms += b->kernel_ms;
            alfa::g_last_timing.work_units += b->cells;
            alfa::g_last_timing.launches += 1;
            ids.clear(); need = 0;
            return ALFA_OK;
        };
        for (int i = 0; i < n; ++i) {
            const int64_t z = sync_need(mode, t[i]);
            if (!ids.empty() && need + z > LIMIT) { if ((rc = flush())) return rc; }
            ids.push_back(i);
            need += z;
        }
        if ((rc = flush())) return rc;
    }
    if (mode == MODE_GLOBAL) return pack_cigars(cigs, n, cigar, cig_off);
    return ALFA_OK;
}

}  // namespace

extern "C" {

int alfa_ksw_extend_batch(const alfa_dp_params* p, const alfa_dp_task* t, int n, alfa_extend_result* r) {
    return run_sync(MODE_EXTEND, p, t, n, r, nullptr, nullptr, nullptr);
}
int alfa_ksw_nw_batch(const alfa_dp_params* p, const alfa_dp_task* t, int n, int32_t* score) {
    return run_sync(MODE_NW, p, t, n, score, nullptr, nullptr, nullptr);
}
int alfa_ksw_global_batch(const alfa_dp_params* p, const alfa_dp_task* t, int n, int32_t* score, uint32_t** cigar, int32_t* n_cigar,
                          int64_t* cig_off) {
    return run_sync(MODE_GLOBAL, p, t, n, score, cigar, n_cigar, cig_off);
}

int alfa_dp_batch_create(int mode, const alfa_dp_params* p, const alfa_dp_task* t, int n, int device, alfa_dp_batch** out) {
    if (!out || mode < 0 || mode > 2 || n < 0 || (n > 0 && !t)) return alfa::fail(ALFA_EINVAL, "bad argument");
    Ctx* cx;
    int rc = current_ctx(&cx, device);
    if (rc) return rc;
    alfa_dp_batch* b = new alfa_dp_batch();
    if ((rc = fill_batch(b, cx, mode, p, t, nullptr, n))) { release_batch(b); delete b; return rc; }
    b->owner = cx;
    *out = b;
    return ALFA_OK;
}
int alfa_dp_batch_launch(alfa_dp_batch* b) {
    if (!b) return alfa::fail(ALFA_EINVAL, "null batch");
    Ctx* cx;
    int rc = current_ctx(&cx, b->device);
    if (rc) return rc;
    if (cx != b->owner) return alfa::fail(ALFA_EINVAL, "a resident DP batch must be launched and fetched by the host thread that created it");
    return launch_batch(b, cx);
}
int alfa_dp_batch_fetch(alfa_dp_batch* b, void* out, uint32_t** cigar, int32_t* n_cigar, int64_t* cig_off) {
    if (!b || (b->n > 0 && !out)) return alfa::fail(ALFA_EINVAL, "null argument");
    if (b->mode == MODE_GLOBAL && (!cigar || !n_cigar || !cig_off)) return alfa::fail(ALFA_EINVAL, "null CIGAR output argument");
    Ctx* cx;
    int rc = current_ctx(&cx, b->device);
    if (rc) return rc;
    if (cx != b->owner) return alfa::fail(ALFA_EINVAL, "a resident DP batch must be launched and fetched by the host thread that created it");
    std::vector<std::vector<uint32_t>> cigs;
    if (b->mode == MODE_GLOBAL) cigs.resize((size_t)b->n);
    if ((rc = fetch_batch(b, cx, out, nullptr, b->mode == MODE_GLOBAL ? &cigs : nullptr, n_cigar))) return rc;
    if (b->mode == MODE_GLOBAL) return pack_cigars(cigs, b->n, cigar, cig_off);
    return ALFA_OK;
}
int64_t alfa_dp_batch_cells(const alfa_dp_batch* b) { return b ? b->cells : 0; }
int64_t alfa_dp_batch_giveups(alfa_dp_batch* b, int64_t* n_fast) {
    if (!b) return -1;
    int64_t n_l2 = 0;
    for (int g = 0; g < NL2; ++g) n_l2 += b->bin_cnt[L2BIN + g];
    if (n_fast) *n_fast = n_l2;
    if (b->mode != MODE_EXTEND || n_l2 == 0 || !b->launched) return 0;
    if (alfa_sync(b->device) != ALFA_OK) return -1;
    int cnt = 0;
    if (hipMemcpy(&cnt, b->d_bail.as<int>(), sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return cnt / (b->launches > 0 ? b->launches : 1);      // every launch of a batch gives up on the same tasks
}
double alfa_dp_batch_kernel_ms(const alfa_dp_batch* b) {
    if (!b || !b->launched || b->n == 0) return 0;
    if (b->ev1_pending) return 0;      // launched under lazy joins and not fetched yet: no closing event exists (alfa_hot.h)
    float ms = 0;
    if (hipEventElapsedTime(&ms, b->ev0, b->ev1) != hipSuccess) { (void)hipGetLastError(); return b->kernel_ms; }
    return ms;
}
void alfa_dp_batch_free(alfa_dp_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    release_batch(b);
    delete b;
}

int alfa_sync(int device) {
    Ctx* cx;
    int rc = current_ctx(&cx, device);
    if (rc) return rc;
    if ((rc = cx->join_sides())) return rc;
    ALFA_HIP(hipStreamSynchronize(cx->stream));
    return ALFA_OK;
}
int alfa_set_lazy_join(int device, int on) {
    Ctx* cx;
    int rc = current_ctx(&cx, device);
    if (rc) return rc;
    if (!on && (rc = cx->join_sides())) return rc;
    cx->lazy_join = on != 0;
    return ALFA_OK;
}
int alfa_timer_start(int device) {
    Ctx* cx;
    int rc = current_ctx(&cx, device);
    if (rc) return rc;
    ALFA_HIP(hipEventRecord(cx->ev[0], cx->stream));
    return ALFA_OK;
}
int alfa_timer_stop(int device, double* ms) {
    Ctx* cx;
    int rc = current_ctx(&cx, device);
    if (rc) return rc;
    if ((rc = cx->join_sides())) return rc;
    ALFA_HIP(hipEventRecord(cx->ev[1], cx->stream));
    ALFA_HIP(hipEventSynchronize(cx->ev[1]));
    float f = 0;
    ALFA_HIP(hipEventElapsedTime(&f, cx->ev[0], cx->ev[1]));
    if (ms) *ms = f;
    return ALFA_OK;
}

}  // extern "C"

// ================================================================= ksw_global without any allocation
// alfa_ksw_global_noalloc (include/alfa_hot.h): the device-resident form of alfa_ksw_global_batch for callers that cannot
// call hipMalloc (HIP graph capture, tight serving loops).  Inputs and outputs are the caller's device buffers; what the
// library otherwise allocates per batch -- the scoring parameters, the internal task records and the direction bytes of
// the fill kernel -- lives in one file-scope device array.  Each GPU holds its own copy of that array at its own address:
// the address is looked up once per device with hipGetSymbolAddress and only that pointer is ever handed to kernels.
// The kernels are the ones of the allocating entry point (ksw_wave_kernel<GLOBAL,4> + ksw_backtrack_tile_kernel).
namespace {

constexpr size_t NOALLOC_HDR = 256;                               // DevParams at the front of the array
constexpr int NOALLOC_ROWB = 64 * 4;                              // direction bytes per row of ksw_wave_kernel<GLOBAL,4>
static __device__ __align__(256) unsigned char scratch[ALFA_NOALLOC_SCRATCH_BYTES];

std::mutex g_scratch_mu;
unsigned char* g_scratch_ptr[64] = {};                            // per device: device address of that GPU's `scratch`

// device address of `scratch` on `device` (which must be current); looked up once per device
int scratch_for(int device, unsigned char** out) {
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        ALFA_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return alfa::fail(ALFA_EHIP, "hipGetSymbolAddress returned a null address for the scratch array");
        g_scratch_ptr[device] = (unsigned char*)p;
    }
    *out = g_scratch_ptr[device];
    return ALFA_OK;
}

// caller's task records -> the kernels' DevTask records (in the scratch array); a task outside the entry point's envelope
// becomes a harmless 1 x 1 task on the first bytes of d_seq and is flagged (DevTask::pad) for noalloc_finish_kernel
__global__ void noalloc_prep_kernel(DevParams hp, const alfa_dev_task* __restrict__ t, int n, int max_tlen, int cig_stride,
                                    unsigned char* __restrict__ ws, int64_t zbase) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *(DevParams*)ws = hp;
    if (i >= n) return;
    const alfa_dev_task x = t[i];
    DevTask d;
    d.qoff = x.qoff; d.toff = x.toff;
    d.zoff = zbase + (int64_t)i * max_tlen * NOALLOC_ROWB;
    d.cigoff = (int64_t)i * cig_stride;
    d.ehoff = 0;
    d.qlen = x.qlen; d.tlen = x.tlen; d.h0 = 0;
    const int lim = x.qlen > x.tlen ? x.qlen : x.tlen;
    d.w = x.w < lim ? x.w : lim;                                  // the band never binds beyond max(qlen, tlen): identical loops
    d.orig = i; d.cigcap = cig_stride; d.cigcap2 = 0; d.pad = 0;
    const bool ok = x.qlen >= 1 && x.tlen >= 1 && x.tlen <= max_tlen && x.w >= 0 && 2 * (int64_t)d.w + 1 <= NOALLOC_ROWB &&
                    x.qoff >= 0 && x.toff >= 0 && (int64_t)x.qlen + x.tlen + 2 <= cig_stride;
    if (!ok) { d.qoff = d.toff = 0; d.qlen = d.tlen = 1; d.w = 0; d.pad = 1; }
    ((DevTask*)(ws + NOALLOC_HDR))[i] = d;
}
__global__ void noalloc_finish_kernel(const unsigned char* __restrict__ ws, int n, int32_t* __restrict__ score, int32_t* __restrict__ ncig) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (((const DevTask*)(ws + NOALLOC_HDR))[i].pad) { score[i] = (int32_t)0x80000000; ncig[i] = -2; }
}

}  // namespace

extern "C" int alfa_ksw_global_noalloc(const alfa_dp_params* p, const alfa_dev_task* d_tasks, int n, int max_tlen, const uint8_t* d_seq,
                                       int32_t* d_score, uint32_t* d_cigar, int32_t* d_n_cigar, int cig_stride, int device, void* stream) {
    DevParams hp;
    int rc = pack_params(p, &hp);
    if (rc) return rc;
    if (n < 0 || max_tlen < 1 || cig_stride < 3) return alfa::fail(ALFA_EINVAL, "alfa_ksw_global_noalloc: n < 0, max_tlen < 1 or cig_stride < 3");
    if (n == 0) return ALFA_OK;
    if (!d_tasks || !d_seq || !d_score || !d_cigar || !d_n_cigar) return alfa::fail(ALFA_EINVAL, "alfa_ksw_global_noalloc: null device pointer");
    if (device < 0 || device >= 64 || device >= alfa::device_count())
        return alfa::fail(alfa::device_count() <= 0 ? ALFA_ENODEV : ALFA_EINVAL, "alfa_ksw_global_noalloc: no such HIP device");
    const size_t tasks_bytes = ((size_t)n * sizeof(DevTask) + 255) & ~(size_t)255;
    const size_t need = NOALLOC_HDR + tasks_bytes + (size_t)n * (size_t)max_tlen * NOALLOC_ROWB;
    if (need > (size_t)ALFA_NOALLOC_SCRATCH_BYTES)
        return alfa::fail(ALFA_EINVAL, "alfa_ksw_global_noalloc: n * (64 + 256 * max_tlen) = " + std::to_string(need) +
                                           " bytes exceeds the fixed scratch array of " + std::to_string((size_t)ALFA_NOALLOC_SCRATCH_BYTES) +
                                           " bytes; split the batch");
    ALFA_HIP(hipSetDevice(device));
    unsigned char* ws = nullptr;
    if ((rc = scratch_for(device, &ws))) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int64_t zbase = (int64_t)(NOALLOC_HDR + tasks_bytes);
    const DevParams* d_prm = (const DevParams*)ws;
    const DevTask* d_t = (const DevTask*)(ws + NOALLOC_HDR);
    hipLaunchKernelGGL(noalloc_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, st, hp, d_tasks, n, max_tlen, cig_stride, ws, zbase);
    hipLaunchKernelGGL((ksw_wave_kernel<MODE_GLOBAL, 4>), dim3((n + WPB - 1) / WPB), dim3(64 * WPB), 0, st, d_t, n, d_seq, d_prm, d_score, ws);
    hipLaunchKernelGGL(ksw_backtrack_tile_kernel, dim3(n), dim3(64), 0, st, d_t, n, ws, d_cigar, d_n_cigar, NOALLOC_ROWB);
    hipLaunchKernelGGL(noalloc_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ws, n, d_score, d_n_cigar);
    ALFA_HIP(hipGetLastError());
    return ALFA_OK;
}
