// region_kernels.hip -- candidate regions on the device: the matches of a read strand stay in HBM after the
// emit kernel and one wavefront per strand turns them into the reference's candidate regions
//   identifyAndFilter  src/mapper.cpp:578-704   sort (query asc, len desc) :586, SMEM triage :618-627, sort (ref, query),
//                                              window building and merging :644-676, reserves by a two-pointer sweep :690-700
//   coveredBases       src/mapper.cpp:706-719   per region: sort (query, ref), union of the query intervals
//   nextChain          src/mapper.cpp:721-782   the chain extendChain builds first in a region: from its longest member (the
//                                              anchor tried first, :885-893) forwards and backwards over the sorted members
// so that the host receives regions (bounds, score, members in their final order) instead of rebuilding them per read.
//
// The sequential parts of the reference (triage, window loop, sweeps) are short scans over a few dozen matches: they run on
// lane 0 over LDS; the four sorts are rank sorts (every lane ranks its elements against all others: n^2/64 comparisons, stable
// by construction).  std::sort is not stable, so the order of elements that compare equal is the one thing a re-implementation
// cannot reproduce: where it could matter -- equal (ref, query) or equal (query, ref) with different lengths -- the strand is
// handed back to the host (status 1), as are strands with more matches or regions than the LDS arrays hold.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "region_kernels.h"

namespace alfa {

namespace {

// Two sizes of the same kernel.  A wavefront's state lives in LDS and its size decides how many strands a CU works on at a time:
// at 512 matches / 128 regions it is 15.9 KB -- 10 wavefronts per CU, for a kernel that waits on LDS and memory most of its time --
// while nearly every strand of a short read has a handful of matches.  So strands of up to SMALL_CAP matches run in a 3 KB
// instance (launched first, for all strands); the full-size instance, launched behind it, takes the strands the small one left:
// those with more matches, and those whose first chain outgrew the small instance's scratch (status 2 in between).
constexpr int SMALL_CAP = 64;

template <int RCAP, int RREG>
struct LdsT {
    int64_t R[RCAP];
    int32_t Q[RCAP], L[RCAP];
    uint16_t o1[RCAP], sm[RCAP], rs[RCAP], tmp[RCAP];
    uint8_t cls[RCAP];
    int64_t rb[RREG], re[RREG];
    uint16_t lo[RREG], hi[RREG], rlo[RREG], rhi[RREG];
    int ns, nr, nreg, fail, score;
};

}  // namespace

template <int RCAP, int RREG, bool SMALL>
__global__ __launch_bounds__(64) void regions_kernel(const alfa_match* __restrict__ m_all, const int64_t* __restrict__ base,
                                                    const int32_t* __restrict__ qlen, const int32_t* __restrict__ edit, int nq,
                                                    RegionParams rp, alfa_region* __restrict__ reg_out, alfa_match* __restrict__ rm_out,
                                                    int32_t* __restrict__ chain_out,
                                                    int32_t* __restrict__ nreg_out, int32_t* __restrict__ nrm_out, int32_t* __restrict__ status) {
    __shared__ LdsT<RCAP, RREG> s;
    const int qi = blockIdx.x, lane = threadIdx.x;
    if (qi >= nq) return;
    const int64_t b0 = base[qi];
    const int64_t nn = base[qi + 1] - b0;
    const int64_t rm_base = 2 * b0 + 8 * (int64_t)qi;          // the strand's slots: n regions at b0, 2n + 8 members here
    if (SMALL && nn > SMALL_CAP) return;                       // the full-size instance's
    if (!SMALL && !(nn > SMALL_CAP || status[qi] == 2)) return;      // done by the small instance
    if (nn == 0 || nn > RCAP) {
        if (lane == 0) { nreg_out[qi] = 0; nrm_out[qi] = 0; status[qi] = nn ? 1 : 0; }
        return;
    }
    const int n = (int)nn;
    const int qLength = qlen[qi], editDist = edit[qi];
    for (int i = lane; i < n; i += 64) {
        const alfa_match m = m_all[b0 + i];
        s.R[i] = m.ref; s.Q[i] = m.query; s.L[i] = m.len;
    }
    if (lane == 0) { s.ns = 0; s.nr = 0; s.nreg = 0; s.fail = 0; }
    __syncthreads();

    // ---- sort by (query asc, len desc)  :586   (ties: same query and length, different ref -- they stay adjacent and are
    // re-sorted by ref below, so their order here is immaterial)
    for (int i = lane; i < n; i += 64) {
        const int q = s.Q[i], l = s.L[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const int qj = s.Q[j], lj = s.L[j];
            rank += (qj < q) | ((qj == q) & ((lj > l) | ((lj == l) & (j < i))));
        }
        s.o1[rank] = (uint16_t)i;
    }
    __syncthreads();

    // ---- SMEM triage  :600-640
    if (lane == 0) {
        int64_t previousREnd = -1, currentSmemRange = 0;
        int index = 0;
        while (index < n) {
            const int rangeStart = index;
            const int64_t ps = s.Q[s.o1[index]], pe = ps + s.L[s.o1[index]];
            index++;
            while (index < n && s.Q[s.o1[index]] == ps && (int64_t)s.Q[s.o1[index]] + s.L[s.o1[index]] == pe) index++;
            if (pe > previousREnd) currentSmemRange = index - rangeStart;
            else currentSmemRange += index - rangeStart;
            const bool keep = (pe > previousREnd && currentSmemRange <= rp.max_smem_occ) ||
                              (currentSmemRange <= rp.max_region_mems && pe - ps >= rp.min_mem_reg_length);
            for (int j = rangeStart; j < index; ++j) s.cls[j] = keep ? 1 : 0;
            previousREnd = pe;
        }
    }
    __syncthreads();

    // ---- smems and reserves, each sorted by (ref, query)  :642, :688
    {
        int fail = 0;
        for (int k = lane; k < n; k += 64) {
            const int i = s.o1[k];
            const int c = s.cls[k];
            const int64_t r = s.R[i];
            const int q = s.Q[i], l = s.L[i];
            int rank = 0;
            for (int k2 = 0; k2 < n; ++k2) {
                const int j = s.o1[k2];
                const int64_t rj = s.R[j];
                const int qj = s.Q[j];
                const bool same = (rj == r) & (qj == q);
                const bool less = (rj < r) | ((rj == r) & (qj < q)) | (same & (j < i));
                const bool mine = s.cls[k2] == c;
                rank += mine & less;
                fail |= mine & same & (s.L[j] != l);
            }
            if (c) s.sm[rank] = (uint16_t)i; else s.rs[rank] = (uint16_t)i;
            atomicAdd(c ? &s.ns : &s.nr, 1);
        }
        if (fail) s.fail = 1;
    }
    __syncthreads();

    // ---- windows  :644-676, reserves  :690-700
    if (lane == 0) {
        const int ns = s.ns, nr = s.nr;
        int nreg = 0;
        int begin = 0;
        bool fail = false;
        auto R = [&](int k) { return s.R[s.sm[k]]; };
        auto Qq = [&](int k) { return (int64_t)s.Q[s.sm[k]]; };
        auto Ll = [&](int k) { return (int64_t)s.L[s.sm[k]]; };
        while (begin < ns && !fail) {
            int64_t refEnd = R(begin) - Qq(begin) + qLength + editDist;
            int longest = begin, end = begin + 1;
            while (end < ns && R(end) + Ll(end) <= refEnd) {
                if (Ll(longest) < Ll(end)) { longest = end; refEnd = R(longest) - Qq(longest) + qLength + editDist; }
                end++;
            }
            const int64_t refBegin = R(longest) - Qq(longest) - editDist;
            begin--;
            while (begin >= 0 && R(begin) >= refBegin) begin--;
            if (nreg == 0 || s.re[nreg - 1] < refBegin) {
                if (nreg >= RREG) { fail = true; break; }
                s.lo[nreg] = (uint16_t)(begin + 1); s.hi[nreg] = (uint16_t)end;
                s.rb[nreg] = refBegin; s.re[nreg] = refEnd;
                s.rlo[nreg] = 0; s.rhi[nreg] = 0;
                nreg++;
            } else {
                // the appended members beginFix+1 .. end-1 start where the region's members ended (begin was the previous `end`)
                s.hi[nreg - 1] = (uint16_t)end;
                s.rb[nreg - 1] = s.rb[nreg - 1] < refBegin ? s.rb[nreg - 1] : refBegin;
                s.re[nreg - 1] = s.re[nreg - 1] < refEnd ? s.re[nreg - 1] : refEnd;     // the reference takes min for both ends (:673-674)
            }
            begin = end;
        }
        if (!fail && nreg > 0) {
            int cr = 0, ci = 0;
            while (cr < nr && ci < nreg) {
                const int64_t rr = s.R[s.rs[cr]];
                if (rr < s.rb[ci]) cr++;
                else if (s.re[ci] < rr) ci++;
                else {
                    if (s.rhi[ci] == s.rlo[ci]) s.rlo[ci] = (uint16_t)cr;
                    s.rhi[ci] = (uint16_t)(cr + 1);
                    cr++;
                }
            }
        }
        s.nreg = nreg;
        if (fail) s.fail = 1;
    }
    __syncthreads();
    if (s.fail) {
        if (lane == 0) { nreg_out[qi] = 0; nrm_out[qi] = 0; status[qi] = 1; }
        return;
    }

    // ---- per region: members sorted by (query, ref), covered bases  :706-719
    const int nreg = s.nreg;
    const int64_t rm_cap = 2 * nn + 8;
    int cursor = 0;
    bool bad = false;
    for (int r = 0; r < nreg; ++r) {
        const int c1 = s.hi[r] - s.lo[r], c2 = s.rhi[r] - s.rlo[r], m = c1 + c2;
        if (m > RCAP || cursor + m > rm_cap) { bad = true; break; }
        auto member = [&](int k) -> int { return k < c1 ? s.sm[s.lo[r] + k] : s.rs[s.rlo[r] + (k - c1)]; };
        int fail = 0;
        for (int k = lane; k < m; k += 64) {
            const int i = member(k);
            const int64_t rr = s.R[i];
            const int q = s.Q[i], l = s.L[i];
            int rank = 0;
            for (int k2 = 0; k2 < m; ++k2) {
                const int j = member(k2);
                const int64_t rj = s.R[j];
                const int qj = s.Q[j];
                const bool same = (qj == q) & (rj == rr);
                rank += (qj < q) | ((qj == q) & (rj < rr)) | (same & (k2 < k));
                fail |= same & (s.L[j] != l);
            }
            s.tmp[rank] = (uint16_t)i;
        }
        if (fail) s.fail = 1;                   // (an ambiguity: final, whatever the instance)
        __syncthreads();
        for (int k = lane; k < m; k += 64) {
            const int i = s.tmp[k];
            alfa_match o;
            o.ref = s.R[i]; o.query = s.Q[i]; o.len = s.L[i];
            rm_out[rm_base + cursor + k] = o;
        }
        if (lane == 0) {
            int i0 = s.tmp[0];
            int trueCover = s.L[i0];
            int prevEnd = s.Q[i0] + s.L[i0] - 1;
            for (int k = 1; k < m; ++k) {
                const int i = s.tmp[k];
                const int start = s.Q[i] > prevEnd + 1 ? s.Q[i] : prevEnd + 1;
                const int stop = prevEnd > s.Q[i] + s.L[i] - 1 ? prevEnd : s.Q[i] + s.L[i] - 1;
                prevEnd = stop;
                trueCover += stop - start + 1 > 0 ? stop - start + 1 : 0;
            }
            // ---- nextChain from the anchor extendChain tries first: the longest member, the later one among equals
            int anchor = 0;
            for (int k = 1; k < m; ++k) if (s.L[s.tmp[k]] >= s.L[s.tmp[anchor]]) anchor = k;
            const int64_t MAX_OVERLAP = -10;
            int curEdit = 0, coverage = s.L[s.tmp[anchor]];
            int nback = 0, nfwd = 0;
            uint16_t* back = s.o1;                      // scratch: the first sort's order is no longer needed
            uint16_t* fwd = s.o1 + RCAP / 2;
            const int64_t aR = s.R[s.tmp[anchor]], aQ = s.Q[s.tmp[anchor]], aL = s.L[s.tmp[anchor]];
            int64_t refLB = aR, qLB = aQ, refRB = aR + aL - 1, qRB = aQ + aL - 1;
            for (int k = anchor + 1; k < m; ++k) {
                const int i = s.tmp[k];
                const int64_t mq = s.Q[i], mr = s.R[i], ml = s.L[i];
                const int64_t qDist = mq - (qRB + 1), refDist = mr - (refRB + 1);
                const int64_t qLong = mq + ml - qLB, rLong = mr + ml - refLB;
                const int64_t dd = qDist - refDist < 0 ? refDist - qDist : qDist - refDist;
                if (qDist >= MAX_OVERLAP && refDist >= MAX_OVERLAP && dd <= (int64_t)(rp.error_percent * (double)(qLong > rLong ? qLong : rLong))) {
                    if (nfwd < RCAP / 2) fwd[nfwd] = (uint16_t)k;
                    nfwd++;
                    curEdit += (int)(dd > 1 ? dd : 1);
                    refLB = mr; refRB = mr + ml - 1; qLB = mq; qRB = mq + ml - 1;
                    const int64_t mn = qDist < refDist ? qDist : refDist;
                    coverage += mn < 0 ? (int)(ml + mn) : (int)ml;
                }
            }
            curEdit += (int)(qRB + 1) < qLength ? 1 : 0;
            refLB = aR; qLB = aQ; refRB = aR + aL - 1; qRB = aQ + aL - 1;
            for (int k = anchor - 1; k >= 0; --k) {
                const int i = s.tmp[k];
                const int64_t mq = s.Q[i], mr = s.R[i], ml = s.L[i];
                const int64_t qDist = qLB - (mq + ml), refDist = refLB - (mr + ml);
                const int64_t qLong = qRB - mq + 1, rLong = refRB - mr + 1;
                const int64_t dd = qDist - refDist < 0 ? refDist - qDist : qDist - refDist;
                if (qDist >= MAX_OVERLAP && refDist >= MAX_OVERLAP && dd <= (int64_t)(rp.error_percent * (double)(qLong > rLong ? qLong : rLong))) {
                    if (nback < RCAP / 2) back[nback] = (uint16_t)k;
                    nback++;
                    curEdit += (int)(dd > 1 ? dd : 1);
                    refLB = mr; refRB = mr + ml - 1; qLB = mq; qRB = mq + ml - 1;
                    const int64_t mn = qDist < refDist ? qDist : refDist;
                    coverage += mn < 0 ? (int)(ml + mn) : (int)ml;
                }
            }
            curEdit += qLB > 0 ? 1 : 0;
            if (nback > RCAP / 2 || nfwd > RCAP / 2) s.fail = SMALL ? 2 : 1;          // (small instance: the full-size one tries again)
            else {
                int32_t* co = chain_out + rm_base + cursor;
                int c = 0;
                for (int k = nback - 1; k >= 0; --k) co[c++] = back[k];
                co[c++] = anchor;
                for (int k = 0; k < nfwd; ++k) co[c++] = fwd[k];
            }
            alfa_region o;
            o.ref_begin = s.rb[r]; o.ref_end = s.re[r]; o.score = trueCover; o.first = cursor; o.count = m;
            o.chain_count = nback + 1 + nfwd; o.chain_anchor = anchor; o.chain_edit = curEdit; o.chain_coverage = coverage; o.reserved = 0;
            reg_out[b0 + r] = o;
        }
        cursor += m;
        __syncthreads();
    }
    if (lane == 0) {
        const bool f = bad || s.fail;
        nreg_out[qi] = f ? 0 : nreg;
        nrm_out[qi] = f ? 0 : cursor;
        status[qi] = f ? ((SMALL && !bad && s.fail == 2) ? 2 : 1) : 0;
    }
}

void launch_regions(hipStream_t st, const alfa_match* d_m, const int64_t* d_base, const int32_t* d_qlen, const int32_t* d_edit, int nq,
                    const RegionParams& rp, alfa_region* d_reg, alfa_match* d_rm, int32_t* d_chain, int32_t* d_nreg, int32_t* d_nrm, int32_t* d_status) {
    if (nq <= 0) return;
    hipLaunchKernelGGL((regions_kernel<SMALL_CAP, SMALL_CAP, true>), dim3(nq), dim3(64), 0, st, d_m, d_base, d_qlen, d_edit, nq, rp, d_reg, d_rm, d_chain,
                       d_nreg, d_nrm, d_status);
    hipLaunchKernelGGL((regions_kernel<REGION_MATCH_CAP, REGION_CAP, false>), dim3(nq), dim3(64), 0, st, d_m, d_base, d_qlen, d_edit, nq, rp, d_reg, d_rm,
                       d_chain, d_nreg, d_nrm, d_status);
}

}  // namespace alfa
