// ksw_kernels.hip -- inter-task batched banded integer DP for gfx950 (wave64).
//
// Batch forms of the reference's three live DP routines (src/ksw.c):
//   ksw_extend  :362-448   semi-global extension from h0, H/E clamped at 0, shrinking band, Z-drop
//   ksw_nw      :544-591   banded global score
//   ksw_global  :468-542   banded global + direction bytes + backtrack -> BAM CIGAR
//
// Design (one task per wavefront, no LDS, no barriers, no MFMA -- integer max-plus):
//   * Band-diagonal register layout.  Lane l owns C consecutive band diagonals
//     d = l*C+c, d = j - i + w.  H(i-1,j-1) -> H(i,j) stays in the same register,
//     E(i,j) arrives from diagonal d+1 (one cross-lane shift per row) and the
//     horizontal F recurrence, the only true intra-row dependency, becomes an exact
//     max-plus prefix scan across the wave:
//        F(d+1) = max(F(d) - gape, B(d)),  B(d) = A(d) - gapoe  (A = max(Hdiag+s, E))
//     which is lane-local over C slots plus one 64-lane exclusive max-scan of
//     G(l) = L(l) + l*C*gape.  Integer max/plus is associative, so the scan is
//     bit-identical to the reference's left-to-right loop.
//   * Row maxima (m, mj), the Z-drop test, the gscore cell and the band-shrink scans
//     of ksw_extend are wave reductions over the same registers.
//   * ksw_global packs one direction byte per cell (reference encoding, ksw.c:506-519)
//     into coalesced per-row stores to HBM; a second tiny kernel walks them back.
//   * Tasks whose band does not fit 64*C diagonals (C<=4), and degenerate 0-length
//     tasks, run in a scalar lane-per-task kernel that follows the reference loop
//     literally (the rare tail; never the throughput path).
#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>

#include "device_ctx.h"

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
};

struct DevParams {
    uint64_t matrow[8];   // matrow[t] byte q = mat[t*m+q]
    int32_t m, gapo, gape, zdrop;
};

struct ExtOut { int32_t score, qle, tle, gtle, gscore, max_off; };

// ---------------------------------------------------------------- wave primitives
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// exclusive prefix max across the 64 lanes; lane 0 receives `identity`
__device__ __forceinline__ int wave_excl_max_scan(int v, int identity) {
    const int lane = lane_id();
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int o = __shfl_up(v, off, 64);
        if (lane >= off) v = max(v, o);
    }
    int ex = __shfl_up(v, 1, 64);
    return lane == 0 ? identity : ex;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return v;
}

__device__ __forceinline__ int score_of(uint64_t mr, uint32_t q) { return (int)(int8_t)(mr >> (8 * q)); }

// ---------------------------------------------------------------- wave-per-task kernel
template <int MODE, int C>
__global__ __launch_bounds__(64 * WPB) void ksw_wave_kernel(const DevTask* __restrict__ tasks, int ntask,
                                                           const uint8_t* __restrict__ seq,
                                                           const DevParams* __restrict__ prm,
                                                           int32_t* __restrict__ out, uint8_t* __restrict__ zbuf) {
    const int wave = blockIdx.x * WPB + (threadIdx.x >> 6);
    if (wave >= ntask) return;
    const int lane = lane_id();
    const DevTask T = tasks[wave];
    const uint8_t* __restrict__ q = seq + T.qoff;
    const uint8_t* __restrict__ tg = seq + T.toff;
    const int qlen = T.qlen, tlen = T.tlen, w = T.w;
    const int gapo = prm->gapo, gape = prm->gape, gapoe = gapo + gape;
    const int zdrop = prm->zdrop;
    constexpr int FLOOR = (MODE == MODE_EXTEND) ? 0 : NEG_INF;
    constexpr int ROWB = 64 * C;  // direction bytes per row

    int Hd[C], E[C];
    uint32_t qc[C];
    int h0 = T.h0;
    if (MODE == MODE_EXTEND && h0 < 0) h0 = 0;
    const int eh1 = h0 > gapoe ? h0 - gapoe : 0;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int d = lane * C + c;
        const int j = d - w;  // column of this diagonal on row 0
        qc[c] = (j >= 0 && j < qlen) ? q[j] : 0u;
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

    int beg = 0, end = qlen;                                    // EXTEND state
    int mx = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
    int tnext = tg[0];

    for (int i = 0; i < tlen; ++i) {
        const int tch = __builtin_amdgcn_readfirstlane(tnext);
        if (i + 1 < tlen) tnext = tg[i + 1];
        const uint64_t mr = prm->matrow[tch];
        // new query character entering the top diagonal on the next row
        const int xq = i + 1 - w + ROWB - 1;
        const uint32_t qfeed = (xq >= 0 && xq < qlen) ? q[xq] : 0u;

        if (MODE == MODE_EXTEND) {
            if (beg < i - w) beg = i - w;
            if (end > i + w + 1) end = i + w + 1;
            if (end > qlen) end = qlen;
        } else {
            beg = i > w ? i - w : 0;
            end = i + w + 1 < qlen ? i + w + 1 : qlen;
        }
        const int base = i - w;          // column of diagonal 0 on this row
        const int db = beg - base, de = end - base;

        int A[C], Bv[C];
        bool act[C];
        uint32_t dirH[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int d = lane * C + c;
            act[c] = d >= db && d < de;
            const int hh = Hd[c] + score_of(mr, qc[c]);
            const int e = E[c];
            if (MODE == MODE_EXTEND) { A[c] = hh > e ? hh : e; dirH[c] = 0; }
            else { dirH[c] = hh >= e ? 0u : 1u; A[c] = hh >= e ? hh : e; }
            int bv = A[c] - gapoe;
            if (MODE == MODE_EXTEND) bv = bv > 0 ? bv : 0;
            Bv[c] = act[c] ? bv : FLOOR;
        }
        // lane-local contribution to the F chain, then the cross-lane max-plus scan
        int L = Bv[0];
#pragma unroll
        for (int c = 1; c < C; ++c) L = max(L - gape, Bv[c]);
        const int Cg = C * gape;
        const int gx = wave_excl_max_scan(L + lane * Cg, 0);
        int f = lane == 0 ? FLOOR : gx - (lane - 1) * Cg;

        int Hrow[C], Eo[C];
        uint32_t zpack = 0;
        int m_l = -1, mj_l = -1;       // EXTEND: lane-local row max and its last column
        int zl_l = -1, zr_l = 0x7fffffff;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int F = f;
            int h = A[c];
            uint32_t dd = dirH[c];
            if (MODE == MODE_EXTEND) { h = h > F ? h : F; }
            else { if (!(h >= F)) { dd = 2u; h = F; } }
            Hrow[c] = h;
            int t = h - gapoe;
            if (MODE == MODE_EXTEND) t = t > 0 ? t : 0;
            const int e2 = E[c] - gape;
            const int f2 = F - gape;
            if (MODE == MODE_GLOBAL) {
                dd |= (e2 > t) ? 4u : 0u;
                dd |= (f2 > t) ? 32u : 0u;
                zpack |= (act[c] ? dd : 0u) << (8 * c);
            }
            Eo[c] = act[c] ? (e2 > t ? e2 : t) : FLOOR;
            const int tt = act[c] ? t : FLOOR;
            f = f2 > tt ? f2 : tt;
            if (act[c]) Hd[c] = h;
            if (MODE == MODE_EXTEND && act[c]) {
                if (h >= m_l) { m_l = h; mj_l = base + lane * C + c; }   // ties -> later column (ksw.c:414)
            }
        }
        if (MODE == MODE_GLOBAL) {
            uint8_t* zr = zbuf + T.zoff + (int64_t)i * ROWB + lane * C;
            if (C == 4) *(uint32_t*)zr = zpack;
            else if (C == 2) *(uint16_t*)zr = (uint16_t)zpack;
            else *zr = (uint8_t)zpack;
        }

        if (MODE == MODE_EXTEND) {
            // ---- row maximum m and its last column mj (ksw.c:414-415)
            const int m = wave_max(m_l);                    // -1 when the row is empty
            const int mj = wave_max(m_l == m ? mj_l : -1);
            const bool empty = !(beg < end);
            const int jfin = empty ? beg : end;             // value of the loop variable after the j loop
            int h1 = h0 - (gapo + gape * (i + 1));          // first-column value (ksw.c:395-396)
            if (h1 < 0) h1 = 0;
            const int h1_init = h1;
            if (!empty) {
                // H(i,end-1): the value the reference leaves in h1
                const int dl = end - 1 - base;
                int hv = Hrow[0];
#pragma unroll
                for (int c = 1; c < C; ++c) hv = (dl % C == c) ? Hrow[c] : hv;
                h1 = __shfl(hv, dl / C, 64);
            }
            if (jfin == qlen) {                             // ksw.c:425-428
                max_ie = gscore > h1 ? max_ie : i;
                gscore = gscore > h1 ? gscore : h1;
            }
            const int mrow = m < 0 ? 0 : m;
            const int mjrow = m < 0 ? -1 : mj;
            {
                int dlt = (i - max_i) - (jfin - max_j);
                dlt = dlt < 0 ? -dlt : dlt;
                if (mrow == 0 || (zdrop > 0 && mx - mrow - dlt * gape > zdrop)) break;   // ksw.c:429
            }
            if (mrow > mx) {
                mx = mrow; max_i = i; max_j = mjrow;
                int off = mjrow - i; off = off < 0 ? -off : off;
                max_off = max_off > off ? max_off : off;
            }
            // ---- band shrink (ksw.c:435-438) in terms of this row's H values:
            //   eh[j].h == H(i,j-1) for beg<j<=end, eh[beg].h == h1_init
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
            // inject H(i, beg-1) := h1_init on the diagonal just left of the band (read by row i+1 if beg stays)
            {
                const int di = db - 1;
                if (di >= 0) {
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        if (lane * C + c == di) Hd[c] = h1_init;
                }
            }
            int nbeg;
            if (zl >= 0) nbeg = zl + 2;
            else if (h1_init == 0) nbeg = beg + 1;
            else nbeg = beg;
            beg = nbeg;
            end = (zr != 0x7fffffff) ? zr + 1 : end + 1;
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
            int e_up = __shfl_down(Eo[0], 1, 64);
            uint32_t q_up = __shfl_down(qc[0], 1, 64);
            if (lane == 63) { e_up = FLOOR; q_up = qfeed; }
#pragma unroll
            for (int c = 0; c < C - 1; ++c) { E[c] = Eo[c + 1]; qc[c] = qc[c + 1]; }
            E[C - 1] = e_up;
            qc[C - 1] = q_up;
        }
    }

    if (MODE == MODE_EXTEND) {
        if (lane == 0) {
            ExtOut* o = (ExtOut*)out + T.orig;
            o->score = mx; o->qle = max_j + 1; o->tle = max_i + 1; o->gtle = max_ie + 1; o->gscore = gscore; o->max_off = max_off;
        }
    } else {
        // score = eh[qlen].h = H(tlen-1, qlen-1) when that cell lies inside the band, else MINUS_INF
        const int dl = qlen - tlen + w;
        int score = NEG_INF;
        if (dl >= 0 && dl <= 2 * w) {
            int hv = Hd[0];
#pragma unroll
            for (int c = 1; c < C; ++c) hv = (dl % C == c) ? Hd[c] : hv;
            score = __shfl(hv, dl / C, 64);
        }
        if (lane == 0) out[T.orig] = score;
    }
}

// ---------------------------------------------------------------- backtrack of the wave kernel's direction bytes
// one lane per task; follows ksw.c:524-538 in band-diagonal coordinates.  CIGAR runs are written from the
// end of the task's slot towards its start, so the slot tail holds the CIGAR already in forward order.
template <int C>
__global__ void ksw_backtrack_kernel(const DevTask* __restrict__ tasks, int ntask, const uint8_t* __restrict__ zbuf,
                                     uint32_t* __restrict__ cig, int32_t* __restrict__ ncig) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntask) return;
    const DevTask T = tasks[t];
    constexpr int ROWB = 64 * C;
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

// ---------------------------------------------------------------- generic scalar kernel (one lane per task)
// Literal restatement of the reference loops for tasks outside the wave kernel's envelope.
template <int MODE>
__global__ void ksw_scalar_kernel(const DevTask* __restrict__ tasks, int ntask, const uint8_t* __restrict__ seq,
                                  const DevParams* __restrict__ prm, int32_t* __restrict__ out, int2* __restrict__ ehbuf,
                                  uint8_t* __restrict__ zbuf, uint32_t* __restrict__ cig, int32_t* __restrict__ ncig) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntask) return;
    const DevTask T = tasks[t];
    const uint8_t* q = seq + T.qoff;
    const uint8_t* tg = seq + T.toff;
    const int qlen = T.qlen, tlen = T.tlen;
    int w = T.w;
    const int gapo = prm->gapo, gape = prm->gape, gapoe = gapo + gape;
    int2* eh = ehbuf + T.ehoff;   // .x = h, .y = e ; qlen+2 entries
    if (MODE == MODE_EXTEND) {
        int h0 = T.h0 < 0 ? 0 : T.h0;
        for (int j = 0; j <= qlen + 1; ++j) eh[j] = make_int2(0, 0);
        eh[0].x = h0;
        eh[1].x = h0 > gapoe ? h0 - gapoe : 0;
        for (int j = 2; j <= qlen && eh[j - 1].x > gape; ++j) eh[j].x = eh[j - 1].x - gape;
        int mx = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
        int beg = 0, end = qlen;
        for (int i = 0; i < tlen; ++i) {
            int f = 0, h1, m = 0, mj = -1, j;
            const uint64_t mr = prm->matrow[tg[i]];
            h1 = h0 - (gapo + gape * (i + 1));
            if (h1 < 0) h1 = 0;
            if (beg < i - w) beg = i - w;
            if (end > i + w + 1) end = i + w + 1;
            if (end > qlen) end = qlen;
            for (j = beg; j < end; ++j) {
                int2 p = eh[j];
                int h = p.x, e = p.y;
                p.x = h1;
                h += score_of(mr, q[j]);
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
                eh[j] = p;
                f -= gape;
                f = f > h ? f : h;
            }
            eh[end] = make_int2(h1, 0);
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
            for (j = mj; j >= beg && eh[j].x; --j) {}
            beg = j + 1;
            for (j = mj + 2; j <= end && eh[j].x; ++j) {}
            end = j;
        }
        ExtOut* o = (ExtOut*)out + T.orig;
        o->score = mx; o->qle = max_j + 1; o->tle = max_i + 1; o->gtle = max_ie + 1; o->gscore = gscore; o->max_off = max_off;
    } else {
        const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
        uint8_t* z = zbuf + T.zoff;
        eh[0] = make_int2(0, NEG_INF);
        int j;
        for (j = 1; j <= qlen && j <= w; ++j) eh[j] = make_int2(-(gapo + gape * j), NEG_INF);
        for (; j <= qlen; ++j) eh[j] = make_int2(NEG_INF, NEG_INF);
        for (int i = 0; i < tlen; ++i) {
            int f = NEG_INF, h1;
            const uint64_t mr = prm->matrow[tg[i]];
            const int beg = i > w ? i - w : 0;
            const int end = i + w + 1 < qlen ? i + w + 1 : qlen;
            h1 = beg == 0 ? -(gapo + gape * (i + 1)) : NEG_INF;
            for (j = beg; j < end; ++j) {
                int2 p = eh[j];
                int h = p.x, e = p.y;
                uint32_t d;
                p.x = h1;
                h += score_of(mr, q[j]);
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
                eh[j] = p;
                f -= gape;
                d |= f > h ? 32u : 0u;
                f = f > h ? f : h;
                if (MODE == MODE_GLOBAL) z[(int64_t)i * n_col + (j - beg)] = (uint8_t)d;
            }
            eh[end] = make_int2(h1, NEG_INF);
        }
        out[T.orig] = eh[qlen].x;
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
                const uint32_t zv = z[(int64_t)i * n_col + (k - (i > w ? i - w : 0))];
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

}  // namespace

// ================================================================= host side
namespace {

using alfa::Ctx;

struct Bins {
    std::vector<int> idx[4];   // 0:C1 1:C2 2:C4 3:scalar
};

int pack_params(const alfa_dp_params* p, DevParams* dp) {
    if (!p) return alfa::fail(ALFA_EINVAL, "null scoring parameters");
    if (p->m < 1 || p->m > 8) return alfa::fail(ALFA_EINVAL, "m (residue types) must be in 1..8");
    if (p->gapo < 0 || p->gape < 0) return alfa::fail(ALFA_EINVAL, "gap penalties must be passed as non-negative numbers");
    memset(dp, 0, sizeof(*dp));
    for (int t = 0; t < p->m; ++t)
        for (int q = 0; q < p->m; ++q) dp->matrow[t] |= (uint64_t)(uint8_t)p->mat[t * p->m + q] << (8 * q);
    dp->m = p->m; dp->gapo = p->gapo; dp->gape = p->gape; dp->zdrop = p->zdrop;
    return ALFA_OK;
}

template <int MODE, int C>
void launch_wave(Ctx* cx, const DevTask* d_tasks, int n, const uint8_t* d_seq, const DevParams* d_prm, int32_t* d_out, uint8_t* d_z) {
    if (n <= 0) return;
    dim3 grid((n + WPB - 1) / WPB), block(64 * WPB);
    hipLaunchKernelGGL((ksw_wave_kernel<MODE, C>), grid, block, 0, cx->stream, d_tasks, n, d_seq, d_prm, d_out, d_z);
}

// Runs tasks[lo,hi) (already validated).  out: MODE_EXTEND -> ExtOut[n] indexed by orig, else int32 score[n].
// GLOBAL additionally fills cigs (per-task vectors).
template <int MODE>
int run_range(Ctx* cx, const DevParams& hp, const alfa_dp_task* t, const int* w_eff, const std::vector<int>& ids, void* out_host,
              std::vector<std::vector<uint32_t>>* cigs, int32_t* ncig_host, int64_t* cells_out) {
    const int n = (int)ids.size();
    if (n == 0) return ALFA_OK;
    // ---- classify + order (longest targets first so that co-resident waves finish together)
    std::vector<int> order(ids);
    auto bin_of = [&](int id) {
        const alfa_dp_task& x = t[id];
        if (x.qlen == 0 || x.tlen == 0) return 3;
        const int W = 2 * w_eff[id] + 1;
        return W <= 64 ? 0 : W <= 128 ? 1 : W <= 256 ? 2 : 3;
    };
    std::vector<int> bins(n);
    for (int i = 0; i < n; ++i) bins[i] = bin_of(order[i]);
    std::vector<int> perm(n);
    std::iota(perm.begin(), perm.end(), 0);
    std::sort(perm.begin(), perm.end(), [&](int a, int b) {
        if (bins[a] != bins[b]) return bins[a] < bins[b];
        return t[order[a]].tlen > t[order[b]].tlen;
    });
    int bin_cnt[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) bin_cnt[bins[i]]++;

    // ---- pack sequences and descriptors
    size_t seq_bytes = 0;
    for (int i = 0; i < n; ++i) seq_bytes += (size_t)t[order[i]].qlen + (size_t)t[order[i]].tlen;
    int rc;
    if ((rc = cx->h_in.reserve(seq_bytes + 64)) || (rc = cx->h_tasks.reserve(sizeof(DevTask) * (size_t)n + sizeof(DevParams))) ||
        (rc = cx->d_in.reserve(seq_bytes + 64)) || (rc = cx->d_tasks.reserve(sizeof(DevTask) * (size_t)n + sizeof(DevParams))))
        return rc;
    uint8_t* hs = cx->h_in.as<uint8_t>();
    DevParams* hprm = cx->h_tasks.as<DevParams>();
    DevTask* ht = (DevTask*)(hprm + 1);
    *hprm = hp;
    size_t so = 0;
    int64_t zbytes = 0, cig_words = 0, eh_items = 0, cells = 0;
    const int Cs[4] = {1, 2, 4, 0};
    for (int k = 0; k < n; ++k) {
        const int id = order[perm[k]];
        const int b = bins[perm[k]];
        const alfa_dp_task& x = t[id];
        DevTask& d = ht[k];
        memset(&d, 0, sizeof(d));
        d.qoff = (int64_t)so;
        for (int j = 0; j < x.qlen; ++j) { uint8_t c = x.query[j]; if (c >= hp.m) return alfa::fail(ALFA_EINVAL, "query residue code >= m"); hs[so + j] = c; }
        so += (size_t)x.qlen;
        d.toff = (int64_t)so;
        for (int j = 0; j < x.tlen; ++j) { uint8_t c = x.target[j]; if (c >= hp.m) return alfa::fail(ALFA_EINVAL, "target residue code >= m"); hs[so + j] = c; }
        so += (size_t)x.tlen;
        d.qlen = x.qlen; d.tlen = x.tlen; d.w = w_eff[id]; d.h0 = x.h0; d.orig = id;
        if (b == 3) { d.ehoff = eh_items; eh_items += (int64_t)x.qlen + 2; }
        if (MODE == MODE_GLOBAL) {
            d.zoff = zbytes;
            if (b == 3) { const int64_t ncol = std::min<int64_t>(x.qlen, 2 * (int64_t)d.w + 1); zbytes += ncol * x.tlen; }
            else zbytes += (int64_t)x.tlen * 64 * Cs[b];
            zbytes = (zbytes + 15) & ~(int64_t)15;
            d.cigoff = cig_words;
            d.cigcap = x.qlen + x.tlen + 2;
            cig_words += d.cigcap;
        }
        // algorithmic cells: inner-loop iterations of the static band; EXTEND counts the same envelope
        {
            const int64_t w = d.w;
            for (int i = 0; i < x.tlen; ++i) {
                int64_t b0 = i > w ? i - w : 0, e0 = std::min<int64_t>(i + w + 1, x.qlen);
                if (e0 > b0) cells += e0 - b0;
            }
        }
    }
    if (cells_out) *cells_out += cells;
    const size_t out_bytes = (MODE == MODE_EXTEND ? sizeof(ExtOut) : sizeof(int32_t)) * (size_t)n;
    if ((rc = cx->d_out.reserve(out_bytes)) || (rc = cx->h_out.reserve(out_bytes))) return rc;
    if (eh_items && (rc = cx->d_scratch.reserve(sizeof(int2) * (size_t)eh_items))) return rc;
    if (MODE == MODE_GLOBAL) {
        if ((rc = cx->d_z.reserve((size_t)zbytes + 64)) || (rc = cx->d_out2.reserve(4 * (size_t)cig_words + 4 * (size_t)n)) ||
            (rc = cx->h_out2.reserve(4 * (size_t)cig_words + 4 * (size_t)n)))
            return rc;
    }
    // out buffers are indexed by `orig` relative to the caller's array: remap to dense 0..n-1 for this range
    // (orig was set to the caller's index; we need a dense index for device output)
    std::vector<int> dense_of(n);
    for (int k = 0; k < n; ++k) { dense_of[k] = ht[k].orig; ht[k].orig = k; }

    hipStream_t st = cx->stream;
    ALFA_HIP(hipEventRecord(cx->ev[0], st));
    ALFA_HIP(hipMemcpyAsync(cx->d_in.p, hs, seq_bytes, hipMemcpyHostToDevice, st));
    ALFA_HIP(hipMemcpyAsync(cx->d_tasks.p, hprm, sizeof(DevParams) + sizeof(DevTask) * (size_t)n, hipMemcpyHostToDevice, st));
    const DevParams* d_prm = cx->d_tasks.as<DevParams>();
    const DevTask* d_t = (const DevTask*)(d_prm + 1);
    const uint8_t* d_seq = cx->d_in.as<uint8_t>();
    int32_t* d_out = cx->d_out.as<int32_t>();
    uint8_t* d_z = cx->d_z.as<uint8_t>();
    uint32_t* d_cig = cx->d_out2.as<uint32_t>();
    int32_t* d_ncig = MODE == MODE_GLOBAL ? (int32_t*)(d_cig + cig_words) : nullptr;
    ALFA_HIP(hipEventRecord(cx->ev[1], st));
    int o = 0;
    launch_wave<MODE, 1>(cx, d_t + o, bin_cnt[0], d_seq, d_prm, d_out, d_z);
    if (MODE == MODE_GLOBAL && bin_cnt[0])
        hipLaunchKernelGGL((ksw_backtrack_kernel<1>), dim3((bin_cnt[0] + 63) / 64), dim3(64), 0, st, d_t + o, bin_cnt[0], d_z, d_cig, d_ncig);
    o += bin_cnt[0];
    launch_wave<MODE, 2>(cx, d_t + o, bin_cnt[1], d_seq, d_prm, d_out, d_z);
    if (MODE == MODE_GLOBAL && bin_cnt[1])
        hipLaunchKernelGGL((ksw_backtrack_kernel<2>), dim3((bin_cnt[1] + 63) / 64), dim3(64), 0, st, d_t + o, bin_cnt[1], d_z, d_cig, d_ncig);
    o += bin_cnt[1];
    launch_wave<MODE, 4>(cx, d_t + o, bin_cnt[2], d_seq, d_prm, d_out, d_z);
    if (MODE == MODE_GLOBAL && bin_cnt[2])
        hipLaunchKernelGGL((ksw_backtrack_kernel<4>), dim3((bin_cnt[2] + 63) / 64), dim3(64), 0, st, d_t + o, bin_cnt[2], d_z, d_cig, d_ncig);
    o += bin_cnt[2];
    if (bin_cnt[3])
        hipLaunchKernelGGL((ksw_scalar_kernel<MODE>), dim3((bin_cnt[3] + 63) / 64), dim3(64), 0, st, d_t + o, bin_cnt[3], d_seq, d_prm, d_out,
                           cx->d_scratch.as<int2>(), d_z, d_cig, d_ncig);
    ALFA_HIP(hipGetLastError());
    ALFA_HIP(hipEventRecord(cx->ev[2], st));
    ALFA_HIP(hipMemcpyAsync(cx->h_out.p, d_out, out_bytes, hipMemcpyDeviceToHost, st));
    if (MODE == MODE_GLOBAL)
        ALFA_HIP(hipMemcpyAsync(cx->h_out2.p, d_cig, 4 * (size_t)cig_words + 4 * (size_t)n, hipMemcpyDeviceToHost, st));
    ALFA_HIP(hipEventRecord(cx->ev[3], st));
    ALFA_HIP(hipStreamSynchronize(st));
    float ms = 0;
    ALFA_HIP(hipEventElapsedTime(&ms, cx->ev[0], cx->ev[1])); alfa::g_last_timing.h2d_ms += ms;
    ALFA_HIP(hipEventElapsedTime(&ms, cx->ev[1], cx->ev[2])); alfa::g_last_timing.kernel_ms += ms;
    ALFA_HIP(hipEventElapsedTime(&ms, cx->ev[2], cx->ev[3])); alfa::g_last_timing.d2h_ms += ms;
    alfa::g_last_timing.launches += 1;

    // ---- scatter results back to the caller's order
    if (MODE == MODE_EXTEND) {
        const ExtOut* r = cx->h_out.as<ExtOut>();
        alfa_extend_result* dst = (alfa_extend_result*)out_host;
        for (int k = 0; k < n; ++k) memcpy(&dst[dense_of[k]], &r[k], sizeof(ExtOut));
    } else {
        const int32_t* r = cx->h_out.as<int32_t>();
        int32_t* dst = (int32_t*)out_host;
        for (int k = 0; k < n; ++k) dst[dense_of[k]] = r[k];
    }
    if (MODE == MODE_GLOBAL) {
        const uint32_t* hc = cx->h_out2.as<uint32_t>();
        const int32_t* hn = (const int32_t*)(hc + cig_words);
        for (int k = 0; k < n; ++k) {
            const int id = dense_of[k];
            int nc = hn[k];
            ncig_host[id] = nc;
            if (nc > 0) {
                const uint32_t* slot = hc + ht[k].cigoff;
                (*cigs)[id].assign(slot + ht[k].cigcap - nc, slot + ht[k].cigcap);
            }
        }
    }
    return ALFA_OK;
}

// effective band: ksw_extend's max_gap clamp (ksw.c:381-386) then the exact no-op clamp to max(qlen,tlen)
int effective_w(int mode, const alfa_dp_params* p, const alfa_dp_task& x) {
    int w = x.w;
    if (mode == MODE_EXTEND) {
        int maxs = 0;
        for (int i = 0; i < p->m * p->m; ++i) maxs = maxs > p->mat[i] ? maxs : p->mat[i];
        int max_gap = p->gape > 0 ? (int)((double)(x.qlen * maxs - p->gapo) / p->gape + 1.) : 0x3fffffff;
        max_gap = max_gap > 1 ? max_gap : 1;
        w = w < max_gap ? w : max_gap;
    }
    const int lim = std::max(x.qlen, x.tlen);
    if (w > lim) w = lim;   // the band never binds beyond this: identical loops
    return w;
}

template <int MODE>
int run_batch(const alfa_dp_params* p, const alfa_dp_task* t, int n, void* out, std::vector<std::vector<uint32_t>>* cigs, int32_t* ncig) {
    alfa::g_last_timing = alfa::Timing();
    if (n < 0 || (n > 0 && (!t || !out))) return alfa::fail(ALFA_EINVAL, "null task/result array");
    DevParams hp;
    int rc = pack_params(p, &hp);
    if (rc) return rc;
    if (n == 0) return ALFA_OK;
    int dev = 0;
    if (alfa::device_count() <= 0) return alfa::fail(ALFA_ENODEV, "no HIP device is usable; libalfa_hot has no CPU fallback");
    (void)hipGetDevice(&dev);
    Ctx* cx = alfa::thread_ctx(dev);
    if (!cx) return ALFA_ENODEV;
    std::vector<int> w_eff(n);
    for (int i = 0; i < n; ++i) {
        if (t[i].qlen < 0 || t[i].tlen < 0 || t[i].w < 0) return alfa::fail(ALFA_EINVAL, "negative length or band in a DP task");
        if ((t[i].qlen > 0 && !t[i].query) || (t[i].tlen > 0 && !t[i].target)) return alfa::fail(ALFA_EINVAL, "null sequence pointer in a DP task");
        w_eff[i] = effective_w(MODE, p, t[i]);
    }
    // chunk so that one launch needs at most ~6 GiB of direction bytes / staging
    const int64_t LIMIT = (int64_t)6 << 30;
    std::vector<int> ids;
    int64_t need = 0, cells = 0;
    for (int i = 0; i < n; ++i) {
        int64_t z = MODE == MODE_GLOBAL ? (int64_t)t[i].tlen * 256 : 0;
        z += t[i].qlen + t[i].tlen;
        if (!ids.empty() && need + z > LIMIT) {
            if ((rc = run_range<MODE>(cx, hp, t, w_eff.data(), ids, out, cigs, ncig, &cells))) return rc;
            ids.clear(); need = 0;
        }
        ids.push_back(i);
        need += z;
    }
    if ((rc = run_range<MODE>(cx, hp, t, w_eff.data(), ids, out, cigs, ncig, &cells))) return rc;
    alfa::g_last_timing.work_units = cells;
    return ALFA_OK;
}

}  // namespace

extern "C" {

int alfa_ksw_extend_batch(const alfa_dp_params* p, const alfa_dp_task* t, int n, alfa_extend_result* r) {
    return run_batch<MODE_EXTEND>(p, t, n, r, nullptr, nullptr);
}

int alfa_ksw_nw_batch(const alfa_dp_params* p, const alfa_dp_task* t, int n, int32_t* score) {
    return run_batch<MODE_NW>(p, t, n, score, nullptr, nullptr);
}

int alfa_ksw_global_batch(const alfa_dp_params* p, const alfa_dp_task* t, int n, int32_t* score, uint32_t** cigar, int32_t* n_cigar,
                          int64_t* cig_off) {
    if (n > 0 && (!cigar || !n_cigar || !cig_off)) return alfa::fail(ALFA_EINVAL, "null CIGAR output argument");
    std::vector<std::vector<uint32_t>> cigs((size_t)std::max(n, 0));
    int rc = run_batch<MODE_GLOBAL>(p, t, n, score, &cigs, n_cigar);
    if (rc) return rc;
    int64_t total = 0;
    for (int i = 0; i < n; ++i) { cig_off[i] = total; total += (int64_t)cigs[i].size(); }
    if (cig_off) cig_off[n > 0 ? n : 0] = total;
    uint32_t* outc = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(total + 1));
    if (!outc) return alfa::fail(ALFA_ENOMEM, "out of host memory for CIGAR output");
    for (int i = 0; i < n; ++i)
        if (!cigs[i].empty()) memcpy(outc + cig_off[i], cigs[i].data(), sizeof(uint32_t) * cigs[i].size());
    *cigar = outc;
    return ALFA_OK;
}

}  // extern "C"
