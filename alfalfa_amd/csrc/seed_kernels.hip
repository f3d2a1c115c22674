// seed_kernels.hip -- batched exact-match seed retrieval from the HBM-resident enhanced sparse
// suffix array (SA + LCP + child table + 10-mer table), gfx950.
//
// Batch form of sparseSA::SMAM (src/sparseSA.cpp:1121-1124 -> findSMAM :1063-1114 ->
// traverse_faster :733-791 -> top_down_child :671-697 -> collectAllSMAMs :901-907 ->
// find_LmaximalStrict :858-871), the reference's default seed type (CLI "--seed SMEM").
//
// Work decomposition: the reference walks, for each phase k<K, the prefixes k, k+q*K, k+2q*K ...
// of a read and (without suffix links, the default index) restarts from the root for every prefix,
// so every (read, prefix) pair is an independent query.  Results are produced deterministically in the
// reference's emission order without a global sort:
//   kernel A  count the matches every query will emit and give each emitting query its offset inside
//             the read; emitting queries append a 24-byte record to a (segmented) work list
//   scan      exclusive sum of the per-read totals
//   kernel B  one lane per work-list record re-enumerates its SA interval, left-extends
//             (find_LmaximalStrict) and writes match_t records at base[read] + offset
// Kernel A exists in two forms: smam_count_kernel (the throughput path: persistent workgroups, the 10-mer presence
// bitmap in LDS, three queued stages filter / table / evaluate -- see the comment above it) and
// smam_count_simple_kernel (one workgroup per read, every lane walks its query to the end; indexes without a
// child table or without the 10-mer table).
//
// The other seed types use the same count / scan / emit structure:
//   MEM   (sparseSA::MEM  :1116-1119 -> findMEM :993-1060 -> collectAllMEMs :873-899), CLI "--seed MEM"
//   SMEM  (sparseSA::SMEM :1127-1208 -> collectXmiAndMli :910-966), CLI "--seed PSMEM": the per-read second phase
//         (tuples ordered by end position, beginLong/smemCount tables, the curMinPos sweep) runs on one lane of the
//         read's workgroup between the parallel traversal and the parallel counting stages.
// Indexes without a child table (--no-child) take the binary-search descent (traverse :701-728, top_down_faster :614-666)
// for SMAM and MEM.  On suffix-link indexes (--suflink) the prefixes of a phase form a chain (smam_suflink_kernel: lane =
// (read, phase)).  SMEMP (unreachable from the reference's command line): smemp_count_kernel, same three stages as SMEM with
// the stack of traverse_keep re-derived instead of stored.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/alfa_index.h"
#include "device_ctx.h"
#include "ksw_wave.h"
#include "map_kernels.h"
#include "region_kernels.h"

struct alfa_index {
    int device = 0;
    int64_t N = 0, n = 0, K = 1, logN = 0, NKm1 = 0, nM = 0;
    int has_child = 0, has_suflink = 0, has_kmer = 0;
    uint8_t nul_sub = 0;     // a byte value that occurs nowhere in S (and is not 0): what a NUL inside a read is packed as
    double kbit_fill = 0;    // fraction of the 10-mers that have a sampled suffix: near 1 the presence bitmap filters nothing
    alfa::DevBuf S, SA, LCP, Midx, Mval, CHILD, KMR, KMR2, KBIT, ISA;
    alfa::DevBuf STARTPOS;   // start offsets of the concatenated sequences (chromosome bounds for the mapping kernel)
    int n_seq = 0;
    // alfa_map_reads_batch on long reads: the direction bytes of the ksw_global calls of the reads in flight take gigabytes (a
    // slot per resident workgroup, sized for the longest read of the call).  Calls on this index share a few such buffers
    // instead of every calling thread keeping its own: the GPU runs four kernels side by side at most (its hardware queues),
    // and sixteen threads' worth of such buffers is memory the driver has to clear when the process ends -- long enough for
    // the next process to stall behind it.  A call takes a buffer for the time of its map_kernel.
    static constexpr int ZPOOL = 8;          // at most; zpool of them are used (ALFA_MAP_ZPOOL, default 4: one per hardware queue)
    int zpool = [] { const char* e = getenv("ALFA_MAP_ZPOOL"); const int v = e ? atoi(e) : 4; return v < 1 ? 1 : v > ZPOOL ? ZPOOL : v; }();
    std::mutex zmu;
    std::condition_variable zcv;
    alfa::DevBuf zbuf[ZPOOL];
    bool zbusy[ZPOOL] = {false, false, false, false, false, false, false, false};
};

namespace {

constexpr int KMER = 10;
constexpr uint32_t TABLE = 1u << (2 * KMER);
constexpr int SEED_BLOCK = 128;

struct DIdx {
    const uint8_t* __restrict__ S;
    const uint32_t* __restrict__ SA;
    const uint8_t* __restrict__ LCP;
    const int64_t* __restrict__ Midx;
    const int32_t* __restrict__ Mval;
    const int32_t* __restrict__ CHILD;
    const uint2* __restrict__ KMR;      // the reference's 10-mer table {left, right}
    const uint4* __restrict__ KMR2;     // derived at upload: {left, right, SA[left], right == left ? S[SA[left]+10..+13] : SA[left+1]}
    const uint32_t* __restrict__ KBIT;  // derived at upload: bit per 10-mer, set when right > 0 (128 KiB: stays in L2)
    const int32_t* __restrict__ ISA;    // suffix-link indexes only
    int64_t N;
    int32_t n, nM, K, has_kmer, has_child;
    int32_t logN, NKm1;
};

struct SeedQ {          // one sparseSA::SMAM(P, ...) call
    int64_t poff;       // offset of P in the packed read buffer
    int32_t plen, min_len, max_count, sparse_q;
    int32_t npref;      // number of prefix queries of this call
    int32_t max_smem;   // max_smem_cont (SMEM only)
    int64_t soff;       // SMEM: offset (in ints) of this read's tables in the global scratch, when they do not fit in LDS
    // findSMAM/findMEM prefix enumeration, precomputed on the host (wave-uniform divisions are not worth VALU time):
    int32_t pm_nh, pm_kh;       // prefixes of the phases k < kh: nh each, of the others nh-1
    uint32_t pm_m1, pm_m2;      // ceil(2^32/nh), ceil(2^32/(nh-1)) when the multiply-high quotient is exact, else 0
};

struct WorkRec2 {       // MEM / SMEM: one emitting prefix query with both of its intervals
    int32_t read, woff, prefix;
    int32_t ms, me, md;        // mli
    int32_t xs, xe, xd;        // xmi
    int32_t calc, nes, nesp;   // SMEM: calcMli, nextEndStop, nextEndStopPrefix
};

struct IvalD { int start, end, depth; };

struct WorkRec {        // one emitting prefix query
    int32_t read, woff, start, end, depth, prefix;
};

__device__ __forceinline__ uint32_t bitadd(uint32_t c) {
    // sparseSA.h:38-63: a/A c/C g/G t/T -> 0..3, anything else all ones
    c |= 0x20u;
    return c == 'a' ? 0u : c == 'c' ? 1u : c == 'g' ? 2u : c == 't' ? 3u : 0xFFFFFFFFu;
}

// bitadd of four residues at once: byte k of the result is the 2-bit code of byte k of `w`, with 0x80 set when that byte is
// not one of acgtACGT or when k >= nvalid (positions behind the read)
__device__ __forceinline__ uint32_t codes_of4(uint32_t w, int nvalid) {
    const uint32_t x = w | 0x20202020u;
    auto eq = [&](uint32_t letter) {            // 0x80 in every byte of x that equals `letter`, exact per byte
        const uint32_t t = x ^ (letter * 0x01010101u);
        return ~(((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t) & 0x80808080u;
    };
    const uint32_t za = eq('a'), zc = eq('c'), zg = eq('g'), zt = eq('t');
    const uint32_t code = (zc >> 7) | (zg >> 6) | (zt >> 7) | (zt >> 6);
    const uint32_t other = ~(za | zc | zg | zt) & 0x80808080u;
    const uint32_t behind = nvalid >= 4 ? 0u : (nvalid <= 0 ? 0x80808080u : 0x80808080u << (8 * nvalid));
    return code | other | behind;
}

__device__ __forceinline__ int d_child(const DIdx& X, int i) { return (i >= 0 && i < X.n) ? X.CHILD[i] : -1; }

// vec_uchar::operator[] (sparseSA.h:79-84)
__device__ __forceinline__ int d_lcp(const DIdx& X, int i) {
    if (i < 0 || i >= X.n) return 0;
    const uint32_t b = X.LCP[i];
    if (b != 255u) return (int)b;
    int lo = 0, hi = X.nM;   // lower_bound on the overflow table
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (X.Midx[mid] < (int64_t)i) lo = mid + 1; else hi = mid;
    }
    return X.Mval[lo];
}

__device__ __forceinline__ uint32_t d_s(const DIdx& X, int64_t pos) { return (pos >= 0 && pos < X.N + 16) ? X.S[pos] : 0u; }

// 8 bytes from any byte address (gfx950 runs in unaligned access mode: one global_load_dwordx2 / ds_read_b64)
__device__ __forceinline__ uint64_t ld8(const uint8_t* p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// Every byte string the kernels compare 8 bytes at a time has 16 readable bytes in front of it and 16 behind it
// (S on the device, the packed reads, the LDS copy of a read), so that chunked loads never leave their buffer.
constexpr int PADB = 16;

struct PatD {           // read staged in LDS; out-of-range reads give '\0' like std::string::operator[](size())
    const uint8_t* p;
    int len;
    __device__ __forceinline__ uint32_t at(int i) const { return (i >= 0 && i < len) ? p[i] : 0u; }
    __device__ __forceinline__ uint64_t load8(int i) const { return ld8(p + i); }   // caller keeps i in [-8, len+8]
};
struct PatG {   // read bytes straight from global memory (kernel B touches only a few of them)
    const uint8_t* p;
    int len;
    __device__ __forceinline__ uint32_t at(int i) const { return (i >= 0 && i < len) ? p[i] : 0u; }
    __device__ __forceinline__ uint64_t load8(int i) const { return ld8(p + i); }
};


// number of leading positions t < L with S[spos+t] == P[c+t]; needs spos+L <= N+16 and c+L <= |P|
template <class PAT>
__device__ __forceinline__ int match_fwd(const DIdx& X, const PAT& P, int64_t spos, int c, int L) {
    int m = 0;
    while (m < L) {
        const uint64_t a0 = ld8(X.S + spos + m), a1 = ld8(X.S + spos + m + 8);
        const uint64_t b0 = P.load8(c + m), b1 = P.load8(c + m + 8);
        const uint64_t x0 = a0 ^ b0, x1 = a1 ^ b1;
        if (x0) { m += __builtin_ctzll(x0) >> 3; break; }
        if (x1) { m += 8 + (__builtin_ctzll(x1) >> 3); break; }
        m += 16;
    }
    return m < L ? m : L;
}

// lcp-interval value of [start,end]  (sparseSA.cpp:751-754, 768-771)
__device__ __forceinline__ int ival_lcp(const DIdx& X, int start, int end) {
    const int ce = d_child(X, end), cs = d_child(X, start);
    return d_lcp(X, (start < ce && ce <= end) ? ce : cs);
}

// top_down_child  sparseSA.cpp:671-697
// (loads whose addresses do not depend on each other are issued together; the decisions are the reference's)
__device__ bool td_child(const DIdx& X, uint32_t c, int& start, int& end, int depth) {
    const int ce = d_child(X, end), cs = d_child(X, start);
    const int64_t sa0 = (int64_t)X.SA[start];
    int right = ce;
    if (start >= right || right > end) right = cs;
    if (d_s(X, sa0 + depth) == c) { end = right - 1; return true; }
    int left = right;
    for (;;) {
        // here left == right: the next sibling boundary, this boundary's lcp and this child's first suffix
        const int nxt = d_child(X, right);
        const int lcp_r = d_lcp(X, right);
        const int64_t sal = (int64_t)X.SA[left];
        const uint32_t ch = d_s(X, sal + depth);
        if (!(nxt > right && lcp_r == d_lcp(X, nxt))) {
            if (ch == c) { start = left; return true; }
            return false;
        }
        right = nxt;
        if (ch == c) { start = left; end = right - 1; return true; }
        left = right;
    }
}

// traverse_faster  sparseSA.cpp:733-791 with maxBranching = -1 (the only value the reference passes), after its 10-mer
// table jump: continue the descent of [start,end] at `depth` until a mismatch, the end of the read or depth == min_len.
// The function is resumable: stopping at some depth and calling it again with a larger min_len visits the same states
// as one uninterrupted call (the reference itself resumes this way in findMEM/SMEM: mli -> xmi).
// sa_hint: SA[start] when the caller already knows it (start == end), else -1.
__device__ void traverse_from(const DIdx& X, const PatD& P, int prefix, int& start, int& end, int& depth, int min_len, int64_t sa_hint) {
    if (depth >= min_len) return;
    int c = prefix + depth;
    if (c >= P.len) return;
    if (start == end) {
        // one suffix left: first character (:758) and the loop of :781-786 are one run of comparisons; the text ends at N.
        // (Identical to the reference for reads without NUL bytes: S is NUL-padded behind N.)
        const int64_t pos0 = (sa_hint >= 0 ? sa_hint : (int64_t)X.SA[start]) + depth;
        const int L = min(P.len - c, min_len - depth);
        const int Ls = (int)max((int64_t)0, min((int64_t)L, X.N - pos0));
        depth += match_fwd(X, P, pos0, c, Ls);
        return;
    }
    bool found;
    if (ival_lcp(X, start, end) == depth) found = td_child(X, P.at(c), start, end, depth);
    else found = P.at(c) == d_s(X, (int64_t)X.SA[start] + depth);
    bool mism = false;
    while (found && !mism && c < P.len && depth < min_len) {
        ++c; ++depth;
        if (start != end) {
            const int childLCP = ival_lcp(X, start, end);
            const int minimum = childLCP < min_len ? childLCP : min_len;
            // edge walk (the byte loop of :773-776), 16 bytes per round trip; S reads as '\0' from N+16 on
            const int64_t pos0 = (int64_t)X.SA[start] + depth;
            const int L = min(P.len - c, minimum - depth);
            if (L > 0) {
                const int Ls = (int)max((int64_t)0, min((int64_t)L, X.N + 16 - pos0));
                const int m = match_fwd(X, P, pos0, c, Ls);
                if (m < L) { mism = true; depth += m; c += m + 1; }
                else { depth += L; c += L; }
            }
            found = c < P.len && !mism && depth < min_len && td_child(X, P.at(c), start, end, depth);
        } else {
            // single suffix left (:781-786): compare on until a mismatch, the end of the text or of the read
            const int64_t pos0 = (int64_t)X.SA[start] + depth;
            const int L = min(P.len - c, min_len - depth);
            if (L > 0) {
                const int Ls = (int)max((int64_t)0, min((int64_t)L, X.N - pos0));
                const int m = match_fwd(X, P, pos0, c, Ls);
                if (m < L) { mism = true; depth += m; c += m + 1; }
                else { depth += L; c += L; }
            }
        }
    }
}

// 10-mer code from the per-read array of 2-bit codes (one byte per residue, 0x80 = not ACGT / beyond the read), 8+2 bytes
__device__ __forceinline__ uint32_t kmer_code_packed(const uint8_t* codes, int prefix) {
    const uint64_t v = ld8(codes + prefix);
    const uint32_t w = (uint32_t)ld8(codes + prefix + 8) & 0xffffu;
    if ((v & 0x8080808080808080ull) | (w & 0x8080u)) return 0xFFFFFFFFu;
    uint64_t x = __builtin_bswap64(v);                              // first residue -> most significant
    x = (x | (x >> 6)) & 0x000F000F000F000Full;
    x = (x | (x >> 12)) & 0x000000FF000000FFull;
    x = (x | (x >> 24)) & 0xFFFFull;
    return ((uint32_t)x << 4) | ((w & 0xffu) << 2) | (w >> 8);
}

__device__ __forceinline__ uint32_t kmer_code(const PatD& P, int prefix) {
    uint32_t index = 0;
#pragma unroll
    for (int i = 0; i < KMER; ++i) index = (index << 2) | bitadd(P.at(prefix + i));
    return index;
}

__device__ void traverse_child(const DIdx& X, const PatD& P, int prefix, int& start, int& end, int& depth, int min_len) {
    int64_t sa_hint = -1;
    if (X.has_kmer && depth == 0 && min_len >= KMER) {          // :734-746
        const uint32_t index = kmer_code(P, prefix);
        if (index >= TABLE) return;
        if (!((X.KBIT[index >> 5] >> (index & 31)) & 1u)) return;   // right == 0: no sampled suffix starts with this 10-mer
        const uint4 e = X.KMR2[index];
        depth = KMER; start = (int)e.x; end = (int)e.y;
        if (start == end) sa_hint = (int64_t)e.z;
    }
    traverse_from(X, P, prefix, start, end, depth, min_len, sa_hint);
}

// find_LmaximalStrict  sparseSA.cpp:858-871.  Returns true (and the match) when one is emitted.
// The reference steps left one character at a time, at most `border` times, and emits only when it stops on a left
// mismatch, the text start or the read start.  Same decision here from the length m of the common left extension
// (capped at min(border, prefix, i)), compared 16 bytes per round trip: emitted iff m < border.
template <class PAT>
__device__ __forceinline__ bool lmax_strict(const DIdx& X, const PAT& P, int prefix, int64_t i, int len, int min_len, int border,
                                            int64_t& mref, int& mq, int& mlen) {
    if (border <= 0) return false;
    const int lim = (int)min((int64_t)min(border, prefix), i);
    int m = 0;
    while (m < lim) {
        // S[i-m-8 .. i-m-1] against P[prefix-m-8 .. prefix-m-1]: the byte next to the match is the most significant one
        const uint64_t a0 = ld8(X.S + i - m - 8), a1 = ld8(X.S + i - m - 16);
        const uint64_t b0 = P.load8(prefix - m - 8), b1 = P.load8(prefix - m - 16);
        const uint64_t x0 = a0 ^ b0, x1 = a1 ^ b1;
        if (x0) { m += __builtin_clzll(x0) >> 3; break; }
        if (x1) { m += 8 + (__builtin_clzll(x1) >> 3); break; }
        m += 16;
    }
    if (m > lim) m = lim;
    if (m >= border) return false;
    if (len + m < min_len) return false;
    mref = i - m; mq = prefix - m; mlen = len + m;
    return true;
}

// top_down_faster  sparseSA.cpp:614-666 (binary search on the i-th character, no child table)
__device__ bool td_binary(const DIdx& X, uint32_t c, int i, int& start, int& end) {
    int l, r, m, r2 = end, l2 = start;
    bool found = false;
    const int cmp_first = (int)c - (int)d_s(X, (int64_t)X.SA[start] + i);
    const int cmp_last = (int)c - (int)d_s(X, (int64_t)X.SA[end] + i);
    if (cmp_first < 0) { l = start + 1; l2 = start; }
    else if (cmp_last > 0) { l = end + 1; l2 = end; }
    else {
        l = start; r = end;
        if (cmp_first == 0) { found = true; r2 = r; }
        else {
            while (r - l > 1) {
                m = (l + r) / 2;
                const int vgl = (int)c - (int)d_s(X, (int64_t)X.SA[m] + i);
                if (vgl <= 0) {
                    if (!found && vgl == 0) { found = true; l2 = m; r2 = r; }
                    r = m;
                } else l = m;
            }
            l = r;
        }
        if (!found) l2 = l - 1;
        if (cmp_last == 0) l2 = end;
        else {
            while (r2 - l2 > 1) {
                m = (l2 + r2) / 2;
                const int vgl = (int)c - (int)d_s(X, (int64_t)X.SA[m] + i);
                if (vgl < 0) r2 = m; else l2 = m;
            }
        }
    }
    start = l; end = l2;
    return l <= l2;
}

// traverse  sparseSA.cpp:701-728 with maxBranching = -1
__device__ void traverse_plain(const DIdx& X, const PatD& P, int prefix, int& start, int& end, int& depth, int min_len) {
    if (X.has_kmer && depth == 0 && min_len >= KMER) {
        uint32_t index = 0;
#pragma unroll
        for (int i = 0; i < KMER; ++i) index = (index << 2) | bitadd(P.at(prefix + i));
        if (index >= TABLE) return;
        const uint2 lr = X.KMR[index];
        if (lr.y == 0) return;
        depth = KMER; start = (int)lr.x; end = (int)lr.y;
    }
    if (depth >= min_len) return;
    while (prefix + depth < P.len) {
        int s2 = start, e2 = end;
        if (!td_binary(X, P.at(prefix + depth), depth, s2, e2)) return;
        ++depth; start = s2; end = e2;
        if (depth == min_len) return;
    }
}

__device__ __forceinline__ void traverse_any(const DIdx& X, const PatD& P, int prefix, IvalD& v, int min_len) {
    if (X.has_child) traverse_child(X, P, prefix, v.start, v.end, v.depth, min_len);
    else traverse_plain(X, P, prefix, v.start, v.end, v.depth, min_len);
}

// collectAllMEMs  sparseSA.cpp:873-899.  Counts (and with EMIT writes) the matches of one findMEM prefix.
template <bool EMIT, class PAT>
__device__ int collect_mem(const DIdx& X, const PAT& P, int prefix, IvalD mli, IvalD xmi, int min_len, int maxLeft, alfa_match* o) {
    int cnt = 0;
    auto one = [&](int i, int depth) {
        int64_t mr; int mq, ml;
        if (lmax_strict(X, P, prefix, (int64_t)X.SA[i], depth, min_len, maxLeft, mr, mq, ml)) {
            if (EMIT) { alfa_match m; m.ref = mr; m.query = mq; m.len = ml; o[cnt] = m; }
            ++cnt;
        }
    };
    const int n = X.n;
    const int xe = xmi.end;
    for (int i = xmi.start; i <= xe; ++i) one(i, xmi.depth);
    if (mli.start == xmi.start && mli.end == xmi.end) return cnt;
    while (xmi.depth >= mli.depth) {
        if (xmi.end + 1 < n) xmi.depth = max(d_lcp(X, xmi.start), d_lcp(X, xmi.end + 1));
        else xmi.depth = d_lcp(X, xmi.start);
        if (xmi.depth >= mli.depth) {
            while (xmi.start > 0 && d_lcp(X, xmi.start) >= xmi.depth) { --xmi.start; one(xmi.start, xmi.depth); }
            while (xmi.end + 1 < n && d_lcp(X, xmi.end + 1) >= xmi.depth) { ++xmi.end; one(xmi.end, xmi.depth); }
        }
    }
    return cnt;
}

// collectXmiAndMli  sparseSA.cpp:910-966
template <bool EMIT, class PAT>
__device__ int collect_xmi_mli(const DIdx& X, const PAT& P, int prefix, IvalD mli, IvalD xmi, bool calcMli, int nextEndStop,
                               int nextEndStopPrefix, int min_len, int maxCount, int maxLeft, alfa_match* o) {
    int cnt = 0;
    auto one = [&](int i, int depth) {
        int64_t mr; int mq, ml;
        if (lmax_strict(X, P, prefix, (int64_t)X.SA[i], depth, min_len, maxLeft, mr, mq, ml)) {
            if (EMIT) { alfa_match m; m.ref = mr; m.query = mq; m.len = ml; o[cnt] = m; }
            ++cnt;
        }
    };
    const int64_t n = X.n;
    const int K = X.K;
    if (xmi.depth + prefix < min_len) return 0;
    for (int i = xmi.start; i <= xmi.end; ++i) one(i, xmi.depth);
    if (!calcMli || (mli.start == xmi.start && mli.end == xmi.end)) return cnt;
    int64_t xs = xmi.start, xe = xmi.end;
    int xd;
    auto redepth = [&]() {
        if (xe + 1 < n) xd = max(d_lcp(X, (int)xs), d_lcp(X, (int)xe + 1));
        else xd = d_lcp(X, (int)xs);
    };
    redepth();
    while ((xe - xs + 1) < maxCount && prefix + xd > nextEndStop && xd >= mli.depth && xd + prefix >= min_len) {
        const int64_t prevStart = xs, prevStop = xe;
        const int64_t room = (int64_t)maxCount - (xe - xs + 1);
        const int64_t minStart = max((int64_t)0, xs - room);
        while (xs >= minStart && xs > 0 && d_lcp(X, (int)xs) >= xd) --xs;
        const int64_t maxEnd = min(n, xe + 1 + room);
        while (xe + 1 < maxEnd && d_lcp(X, (int)xe + 1) >= xd) ++xe;
        if ((xe - xs + 1) < maxCount) {
            for (int64_t i = xs; i < prevStart; ++i) one((int)i, xd);
            for (int64_t i = prevStop + 1; i <= xe; ++i) one((int)i, xd);
            redepth();
        }
    }
    if ((xe - xs + 1) < maxCount && prefix + xd == nextEndStop && xd >= mli.depth && xd + prefix >= min_len &&
        prefix <= nextEndStopPrefix + K) {
        const int64_t prevStart = xs, prevStop = xe;
        const int64_t room = (int64_t)maxCount - (xe - xs + 1);
        const int64_t minStart = max((int64_t)0, xs - room);
        while (xs >= minStart && xs > 0 && d_lcp(X, (int)xs) >= xd) --xs;
        const int64_t maxEnd = min(n, xe + room);
        while (xe + 1 < maxEnd && d_lcp(X, (int)xe + 1) >= xd) ++xe;
        if ((xe - xs + 1) < maxCount) {
            for (int64_t i = xs; i < prevStart; ++i) one((int)i, xd);
            for (int64_t i = prevStop + 1; i <= xe; ++i) one((int)i, xd);
        }
    }
    return cnt;
}

// flat reference-order index t -> (phase k, step jj) of findSMAM's double loop
struct PrefixMap {
    int step, nh, kh, Lmax;
    uint32_t m1 = 0, m2 = 0;      // ceil(2^32 / nh), ceil(2^32 / (nh-1)): exact quotients for t, divisor < 2^16 (set_magic)
    __device__ __forceinline__ int prefix_of(int t) const {
        int k, jj;
        if (t < kh * nh) { k = t / nh; jj = t - k * nh; }
        else { const int t2 = t - kh * nh; const int q = t2 / (nh - 1); k = kh + q; jj = t2 - q * (nh - 1); }
        return k + jj * step;
    }
    // same with the host's multiply-high constants: m1 == 0 means nh <= 1 (quotient = t), m2 == 0 means nh-1 <= 1
    __device__ __forceinline__ int prefix_of_fast(int t) const {
        int k, jj;
        if (t < kh * nh) { k = m1 ? (int)__umulhi((uint32_t)t, m1) : t; jj = t - k * nh; }
        else {
            const int t2 = t - kh * nh;
            const int q = m2 ? (int)__umulhi((uint32_t)t2, m2) : t2;
            k = kh + q; jj = t2 - q * (nh - 1);
        }
        return k + jj * step;
    }
    // the same without a divergent branch (selects): the filter stage runs it for every query
    __device__ __forceinline__ int prefix_of_select(int t) const {
        const int split = kh * nh;
        const bool first = t < split;
        const int tt = first ? t : t - split;
        const uint32_t mm = first ? m1 : m2;
        const int nn = first ? nh : nh - 1;
        const int q = mm ? (int)__umulhi((uint32_t)tt, mm) : tt;
        return (first ? q : kh + q) + (tt - q * nn) * step;
    }
};

// block-wide exclusive scan of one int per thread (SEED_BLOCK threads); total returned to every thread
__device__ __forceinline__ int block_excl_scan(int v, int* sh, int& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) sh[wid] = inc;
    __syncthreads();
    int base = 0;
    constexpr int NW = SEED_BLOCK / 64;
    int tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) { const int s = sh[i]; if (i < wid) base += s; tot += s; }
    __syncthreads();
    total = tot;
    return base + inc - v;
}

// Work-list slots for the lanes of a converged wave that `want` one: a single atomic per wave.  (One atomic per lane on one
// address was the bottleneck of the first version of these kernels: ~10^6 same-address device atomics per launch, each
// wave's life dominated by the wait for its turn.)
__device__ __forceinline__ int wave_alloc_slot(int* counter, bool want) {
    const uint64_t m = __ballot(want);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, 0, 64);
    return base + __popcll(m & ((1ull << lane) - 1ull));
}

// ---------------------------------------------------------------- kernel A: traverse + count
// Counters (rocprofv3, profiles/) show the one-read-per-workgroup form bound by the vector L1's miss path: ~250 L1->L2
// requests per read strand, the L1 stalled on pending L2 data for 3/4 of the kernel, waves waiting 80 % of their life.
// Nearly half of those requests are probes of the 10-mer table that find it empty, and most of the rest are the first
// characters of random 10-mer hits.  This form removes both:
//   * persistent workgroups, one per CU (1024 threads), keep the 128 KiB 10-mer presence bitmap in LDS;
//   * the companion table KMR2 carries, for single-suffix entries, SA[left] and the 4 text bytes behind the 10-mer, so a
//     random hit is resolved from the table entry alone;
//   * a wave owns a read strand (its bytes in the wave's LDS slice) and takes the next one from a global counter; offsets
//     come from a wave-level scan, so the main loop has no workgroup barrier.
constexpr int PBLOCK = 1024;
constexpr int PWAVES = PBLOCK / 64;
constexpr int KBIT_BYTES = (int)(TABLE / 8);
constexpr int WQ_BYTES = 128 * 8;                // per-wave queue of bitmap survivors {prefix, 10-mer code}

__device__ __forceinline__ int wave_excl_scan(int v, int& total) {
    const int lane = threadIdx.x & 63;
    int inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    total = __shfl(inc, 63, 64);
    return inc - v;
}

// A survivor of the table stage that may still emit matches: everything the evaluation stage needs.
struct Q3 { int32_t read, prefix, start, end; uint32_t z, w; };     // prefix bit 31: query without 10-mer jump (|P| < 10)
constexpr int Q3_CAP = 128;

// Evaluation of one survivor of the table stage (stage 3 of smam_count_kernel; the whole per-query work of the sorted path):
// the deepest match of P[prefix..] inside [start,end] and how many matches it emits.  ent.prefix bit 31: no 10-mer jump.
__device__ __forceinline__ void eval_entry(const DIdx& X, const SeedQ& Q, const uint8_t* __restrict__ reads, const int K, const Q3& ent, int& cnt,
                                           int& start, int& end, int& depth) {
    cnt = 0; start = ent.start; end = ent.end; depth = 0;
    const int prefix = ent.prefix & 0x7fffffff;
    PatD P;
    P.len = Q.plen;
    P.p = reads + Q.poff;                               // the read's LDS copy may be gone: global bytes
    const int step = Q.sparse_q * K;
    const int min_lenK = Q.min_len - (step - 1);
    const int minMin_len = Q.sparse_q > 1 ? Q.min_len - K : Q.min_len;
    const int maxCount = Q.sparse_q * Q.max_count;
    // Hits with 1..4 sampled suffixes behind the 10-mer (all but a fraction of a percent on a non-repetitive
    // reference) are evaluated directly.  The child-table descent reaches, by construction, the deepest match M of
    // P[prefix..] among the suffixes of the interval and ends on the sub-interval of the suffixes that attain it (at a
    // branching node when no child continues P, inside an edge on a mismatch, or at the end of the read or text):
    // with so few suffixes, M follows from comparing each of them with the read, all text chunks being independent
    // loads (16 bytes per suffix and round trip) instead of ~10 dependent CHILD/LCP/SA/S probes.  A single suffix
    // is the run of comparisons of :758,:781-786.
    int64_t sas[4] = {0, 0, 0, 0};
    int lim[4] = {-1, -1, -1, -1}, mi[4] = {-1, -1, -1, -1};     // -1: unused slot
    bool go[4] = {false, false, false, false};
    bool direct = false;
    const int c0 = prefix + KMER;
    if (ent.prefix >= 0) {
        depth = KMER;
        const int nsuf = end - start + 1;
        if (nsuf >= 1 && nsuf <= 4) {
            direct = true;
            sas[0] = (int64_t)ent.z;
            if (nsuf >= 2) sas[1] = (int64_t)ent.w;
            if (nsuf >= 3) sas[2] = (int64_t)X.SA[start + 2];
            if (nsuf >= 4) sas[3] = (int64_t)X.SA[start + 3];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nsuf) {
                    lim[k] = (int)max((int64_t)0, min((int64_t)(Q.plen - c0), X.N - (sas[k] + KMER)));
                    mi[k] = 0;
                    go[k] = lim[k] > 0;
                }
        }
    }
    if (direct) {
        for (int off = 0; go[0] || go[1] || go[2] || go[3]; off += 16) {
            uint64_t a0[4], a1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (go[k]) { a0[k] = ld8(X.S + sas[k] + KMER + off); a1[k] = ld8(X.S + sas[k] + KMER + off + 8); }
            const uint64_t b0 = P.load8(c0 + off), b1 = P.load8(c0 + off + 8);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (go[k]) {
                    const uint64_t x0 = a0[k] ^ b0, x1 = a1[k] ^ b1;
                    int m = 16;
                    if (x0) m = __builtin_ctzll(x0) >> 3;
                    else if (x1) m = 8 + (__builtin_ctzll(x1) >> 3);
                    if (!(m >= 16 && lim[k] > off + 16)) { mi[k] = min(off + m, lim[k]); go[k] = false; }
                }
        }
        const int M = max(max(mi[0], mi[1]), max(mi[2], mi[3]));
        int ns = 3, ne = 0;
#pragma unroll
        for (int k = 3; k >= 0; --k) if (mi[k] == M) ns = k;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (mi[k] == M) ne = k;
        depth = KMER + M;
        end = start + ne; start = start + ns;
        // keep the SA values of the surviving suffixes addressable by (rank - start)
#pragma unroll
        for (int k = 0; k < 3; ++k) if (ns > k) { sas[0] = sas[1]; sas[1] = sas[2]; sas[2] = sas[3]; }
    } else {
        traverse_from(X, P, prefix, start, end, depth, Q.plen, -1);
    }
    if (depth > 1 && depth >= min_lenK && (end - start + 1) <= maxCount) {
        int64_t mr; int mq, ml;
        if (direct) {
            for (int k = 0; k <= end - start; ++k) {
                const int64_t sk = k == 0 ? sas[0] : k == 1 ? sas[1] : k == 2 ? sas[2] : sas[3];
                cnt += lmax_strict(X, P, prefix, sk, depth, minMin_len, step, mr, mq, ml) ? 1 : 0;
            }
        } else
            for (int i = start; i <= end; ++i)
                cnt += lmax_strict(X, P, prefix, (int64_t)X.SA[i], depth, minMin_len, step, mr, mq, ml) ? 1 : 0;
    }
}

__global__ __launch_bounds__(PBLOCK) void smam_count_kernel(DIdx X, const SeedQ* __restrict__ qs, const uint8_t* __restrict__ reads,
                                                           unsigned long long* __restrict__ read_total, WorkRec* __restrict__ work,
                                                           int* __restrict__ seg_count, int seg_cap, int nq, int lds_stride,
                                                           Q3* __restrict__ q3_all) {
    extern __shared__ uint8_t smem[];
    __shared__ int sh_wcount;                                   // records in this workgroup's segment of the work list
    uint32_t* bits = (uint32_t*)smem;
    for (int i = threadIdx.x; i < (int)(TABLE / 32); i += PBLOCK) bits[i] = X.KBIT[i];
    if (threadIdx.x == 0) sh_wcount = 0;
    __syncthreads();
    work += (size_t)blockIdx.x * seg_cap;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint8_t* mybuf = smem + KBIT_BYTES + (size_t)wid * (2 * lds_stride + WQ_BYTES);
    const int K = X.K;
    // Three stages with a queue between each pair; a stage only runs when its input queue holds a full wave of work:
    //   1 filter    prefix, 10-mer code, presence bit (LDS only)                       queries  -> wq (LDS, this read)
    //   2 table     KMR2 entry; single-suffix hits that mismatch within the entry's 4 text bytes and cannot reach min_lenK
    //               are finished here (9 in 10 survivors)                             wq       -> q3 (HBM, across reads)
    //   3 evaluate  text comparisons / child-table descent, left extension, offsets   q3       -> work list
    // Queue order is (read, query) order, and only stage 3 emits, so offsets inside a read stay in the reference's order.
    Q3* q3 = q3_all + (size_t)(blockIdx.x * PWAVES + wid) * Q3_CAP;
    int q3n = 0;
    int carry_r = -1, carry_run = 0;                             // offset reached inside the last read stage 3 has seen

    auto evaluate = [&](int take) {
        // this wave's q3 stores are visible to its own loads: one CU, one vector L1 -- a workgroup-scope fence (a wait for
        // the stores) is enough; an agent-scope fence would write back / invalidate the XCD's L2 and costs 5x the kernel
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        const bool have = lane < take;
        Q3 ent;
        ent.read = -1; ent.prefix = 0; ent.start = 0; ent.end = 0; ent.z = 0; ent.w = 0;
        if (have) ent = q3[lane];
        Q3 mv = ent;
        const bool moves = lane + take < q3n;
        if (moves) mv = q3[lane + take];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (moves) q3[lane] = mv;                               // the rest of the queue moves to the front
        q3n -= take;
        // (the body of eval_entry, kept in line: this kernel sits on the 128-register budget of a 16-wave workgroup and the
        // call, although inlined, made the allocator spill 19 registers: 0.31 -> 0.44 ms)
        int cnt = 0, start = ent.start, end = ent.end, depth = 0;
        const int prefix = ent.prefix & 0x7fffffff;
        if (have) {
            const SeedQ Q = qs[ent.read];
            PatD P;
            P.len = Q.plen;
            P.p = reads + Q.poff;                               // the read's LDS copy may be gone: global bytes
            const int step = Q.sparse_q * K;
            const int min_lenK = Q.min_len - (step - 1);
            const int minMin_len = Q.sparse_q > 1 ? Q.min_len - K : Q.min_len;
            const int maxCount = Q.sparse_q * Q.max_count;
            // Hits with 1..4 sampled suffixes behind the 10-mer (all but a fraction of a percent on a non-repetitive
            // reference) are evaluated directly.  The child-table descent reaches, by construction, the deepest match M of
            // P[prefix..] among the suffixes of the interval and ends on the sub-interval of the suffixes that attain it (at a
            // branching node when no child continues P, inside an edge on a mismatch, or at the end of the read or text):
            // with so few suffixes, M follows from comparing each of them with the read, all text chunks being independent
            // loads (16 bytes per suffix and round trip) instead of ~10 dependent CHILD/LCP/SA/S probes.  A single suffix
            // is the run of comparisons of :758,:781-786.
            int64_t sas[4] = {0, 0, 0, 0};
            int lim[4] = {-1, -1, -1, -1}, mi[4] = {-1, -1, -1, -1};     // -1: unused slot
            bool go[4] = {false, false, false, false};
            bool direct = false;
            const int c0 = prefix + KMER;
            if (ent.prefix >= 0) {
                depth = KMER;
                const int nsuf = end - start + 1;
                if (nsuf >= 1 && nsuf <= 4) {
                    direct = true;
                    sas[0] = (int64_t)ent.z;
                    if (nsuf >= 2) sas[1] = (int64_t)ent.w;
                    if (nsuf >= 3) sas[2] = (int64_t)X.SA[start + 2];
                    if (nsuf >= 4) sas[3] = (int64_t)X.SA[start + 3];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < nsuf) {
                            lim[k] = (int)max((int64_t)0, min((int64_t)(Q.plen - c0), X.N - (sas[k] + KMER)));
                            mi[k] = 0;
                            go[k] = lim[k] > 0;
                        }
                }
            }
            if (direct) {
                for (int off = 0; go[0] || go[1] || go[2] || go[3]; off += 16) {
                    uint64_t a0[4], a1[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (go[k]) { a0[k] = ld8(X.S + sas[k] + KMER + off); a1[k] = ld8(X.S + sas[k] + KMER + off + 8); }
                    const uint64_t b0 = P.load8(c0 + off), b1 = P.load8(c0 + off + 8);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (go[k]) {
                            const uint64_t x0 = a0[k] ^ b0, x1 = a1[k] ^ b1;
                            int m = 16;
                            if (x0) m = __builtin_ctzll(x0) >> 3;
                            else if (x1) m = 8 + (__builtin_ctzll(x1) >> 3);
                            if (!(m >= 16 && lim[k] > off + 16)) { mi[k] = min(off + m, lim[k]); go[k] = false; }
                        }
                }
                const int M = max(max(mi[0], mi[1]), max(mi[2], mi[3]));
                int ns = 3, ne = 0;
#pragma unroll
                for (int k = 3; k >= 0; --k) if (mi[k] == M) ns = k;
#pragma unroll
                for (int k = 0; k < 4; ++k) if (mi[k] == M) ne = k;
                depth = KMER + M;
                end = start + ne; start = start + ns;
                // keep the SA values of the surviving suffixes addressable by (rank - start)
#pragma unroll
                for (int k = 0; k < 3; ++k) if (ns > k) { sas[0] = sas[1]; sas[1] = sas[2]; sas[2] = sas[3]; }
            } else {
                traverse_from(X, P, prefix, start, end, depth, Q.plen, -1);
            }
            if (depth > 1 && depth >= min_lenK && (end - start + 1) <= maxCount) {
                int64_t mr; int mq, ml;
                if (direct) {
                    for (int k = 0; k <= end - start; ++k) {
                        const int64_t sk = k == 0 ? sas[0] : k == 1 ? sas[1] : k == 2 ? sas[2] : sas[3];
                        cnt += lmax_strict(X, P, prefix, sk, depth, minMin_len, step, mr, mq, ml) ? 1 : 0;
                    }
                } else
                    for (int i = start; i <= end; ++i)
                        cnt += lmax_strict(X, P, prefix, (int64_t)X.SA[i], depth, minMin_len, step, mr, mq, ml) ? 1 : 0;
            }
        }
        // offsets: an exclusive scan over the wave, made relative to the first entry of the same read, plus what earlier
        // evaluations already handed out inside that read
        int total;
        const int excl = wave_excl_scan(cnt, total);
        const int r_prev = __shfl_up(ent.read, 1, 64);
        const bool head = lane == 0 || r_prev != ent.read;
        int hl = head ? lane : -1;                              // lane of the head of my run: inclusive max-scan
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(hl, off, 64); if (lane >= off) hl = max(hl, o); }
        const int seg_base = __shfl(excl, hl, 64);
        const int carried = ent.read == carry_r ? carry_run : 0;
        const int woff = excl - seg_base + carried;
        const int r_next = __shfl_down(ent.read, 1, 64);
        const bool tail = have && (lane == take - 1 || r_next != ent.read);
        const int run_total = excl + cnt - seg_base;
        if (tail && run_total > 0) atomicAdd(&read_total[ent.read], (unsigned long long)run_total);
        const int last_r = __shfl(ent.read, take - 1, 64);
        const int last_total = __shfl(run_total + carried, take - 1, 64);
        carry_r = last_r; carry_run = last_total;
        const int slot = wave_alloc_slot(&sh_wcount, cnt > 0);
        if (cnt > 0 && slot < seg_cap) {
            WorkRec w;
            w.read = ent.read; w.woff = woff; w.start = start; w.end = end; w.depth = depth; w.prefix = prefix;
            work[slot] = w;
        }
    };

    // The descriptor of a read and the first 512 bytes of its text are fetched one read ahead (the descriptor two ahead, so
    // that the text address is known when its load is issued): both latencies hide behind the work on the current read.
    const int rstride = gridDim.x * PWAVES;
    const int r0 = blockIdx.x * PWAVES + wid;
    SeedQ Qcur = qs[r0 < nq ? r0 : 0];
    SeedQ Qnext = qs[r0 + rstride < nq ? r0 + rstride : 0];
    uint64_t vcur = (r0 < nq && 8 * lane < Qcur.plen + 16) ? ld8(reads + Qcur.poff + 8 * lane) : 0ull;
    for (int r = r0; r < nq; r += rstride) {      // wave-uniform
        const SeedQ Q = Qcur;
        const uint64_t vfirst = vcur;
        Qcur = Qnext;
        if (r + rstride < nq && 8 * lane < Qcur.plen + 16) vcur = ld8(reads + Qcur.poff + 8 * lane);
        Qnext = qs[r + 2 * rstride < nq ? r + 2 * rstride : 0];
        PatD P;
        P.len = Q.plen;
        const bool in_lds = Q.plen + 2 * PADB + 8 <= lds_stride;        // + 8: the staging loop writes whole 8-byte words
        uint8_t* codes = mybuf + lds_stride;                    // second half of the wave's slice: one 2-bit code per residue
        if (in_lds) {
            // eight residues per lane and trip (one 250-bp read = one trip): the bytes, and their 2-bit codes computed on four
            // bytes at a time (0x80 set for anything but acgt/ACGT and for positions behind the read)
            for (int off = 8 * lane; off < Q.plen + 16; off += 512) {
                const uint64_t v = off < 512 ? vfirst : ld8(reads + Q.poff + off);
                *(uint64_t*)(mybuf + PADB + off) = v;
                const uint32_t c0 = codes_of4((uint32_t)v, Q.plen - off), c1 = codes_of4((uint32_t)(v >> 32), Q.plen - off - 4);
                *(uint64_t*)(codes + off) = (uint64_t)c0 | ((uint64_t)c1 << 32);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            P.p = mybuf + PADB;
        } else {
            P.p = reads + Q.poff;
        }
        const int step = Q.sparse_q * K;
        const int min_lenK = Q.min_len - (step - 1);
        PrefixMap pm;
        pm.step = step;
        pm.Lmax = Q.plen - min_lenK;
        pm.nh = Q.pm_nh; pm.kh = Q.pm_kh; pm.m1 = Q.pm_m1; pm.m2 = Q.pm_m2;
        const bool use_kmer = Q.plen >= KMER;                   // traverse_faster is called with min_len = |P| (:1073)
        uint2* wq = (uint2*)(mybuf + 2 * lds_stride);           // {prefix, 10-mer code}, capacity 128
        int qn = 0, base = 0;
        for (;;) {
            // ---- stage 1
            // common case (read in LDS, 10-mer table): straight-line code, no divergent branch
            while (in_lds && use_kmer && qn < 64 && base < Q.npref) {
                const int t = base + lane;
                const bool valid = t < Q.npref;
                const int pf = pm.prefix_of_select(valid ? t : 0);
                const uint32_t packed = kmer_code_packed(codes, min(pf, Q.plen));
                const uint32_t index = pf + KMER <= Q.plen ? packed : 0xFFFFFFFFu;
                const uint32_t ix = index < TABLE ? index : 0u;
                const bool keep = valid && index < TABLE && ((bits[ix >> 5] >> (ix & 31)) & 1u);
                const uint64_t m = __ballot(keep);
                if (keep) wq[qn + __popcll(m & ((1ull << lane) - 1ull))] = make_uint2((uint32_t)pf, index);
                qn += __popcll(m);
                base += 64;
            }
            while (qn < 64 && base < Q.npref) {
                const int t = base + lane;
                bool keep = t < Q.npref;
                int pf = 0;
                uint32_t index = 0xFFFFFFFFu;
                if (keep) {
                    pf = pm.prefix_of_fast(t);
                    if (use_kmer) {
                        index = !in_lds ? kmer_code(P, pf) : (pf + KMER <= Q.plen ? kmer_code_packed(codes, pf) : 0xFFFFFFFFu);
                        keep = index < TABLE && ((bits[index >> 5] >> (index & 31)) & 1u);
                    }
                }
                const uint64_t m = __ballot(keep);
                if (keep) wq[qn + __popcll(m & ((1ull << lane) - 1ull))] = make_uint2((uint32_t)pf, index);
                qn += __popcll(m);
                base += 64;
            }
            if (qn == 0) break;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int take = qn < 64 ? qn : 64;
            const bool have = lane < take;
            const uint2 ent = have ? wq[lane] : make_uint2(0u, 0xFFFFFFFFu);
            const uint2 mv = (lane + 64 < qn) ? wq[lane + 64] : make_uint2(0u, 0u);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane + 64 < qn) wq[lane] = mv;                  // the rest of the queue moves to the front
            qn -= take;
            // ---- stage 2
            bool defer = false;
            Q3 out;
            out.read = r; out.prefix = (int)ent.x; out.start = 0; out.end = X.n - 1; out.z = 0; out.w = 0;
            if (have) {
                if (!use_kmer) { defer = true; out.prefix |= (int)0x80000000; }
                else {
                    const uint4 e = X.KMR2[ent.y];
                    out.start = (int)e.x; out.end = (int)e.y; out.z = e.z; out.w = e.w;
                    defer = true;
                    if (e.x == e.y) {
                        // single suffix: its first 4 characters come with the entry; most random hits end here
                        const int c0 = (int)ent.x + KMER;
                        const int lim0 = (int)max((int64_t)0, min((int64_t)(Q.plen - c0), X.N - ((int64_t)e.z + KMER)));
                        int d = KMER;
                        bool resolved = lim0 <= 0;
                        if (lim0 > 0) {
                            const uint32_t x = e.w ^ (uint32_t)P.load8(c0);
                            const int m4 = x ? (__builtin_ctz(x) >> 3) : 4;
                            if (m4 < 4 || lim0 <= 4) { d += min(m4, lim0); resolved = true; }
                        }
                        if (resolved && !(d > 1 && d >= min_lenK)) defer = false;      // finished, and too short to emit
                    }
                }
            }
            const uint64_t dm = __ballot(defer);
            if (defer) q3[q3n + __popcll(dm & ((1ull << lane) - 1ull))] = out;
            q3n += __popcll(dm);
            // ---- stage 3 whenever a full wave of work is waiting
            if (q3n >= 64) evaluate(64);
        }
        __builtin_amdgcn_wave_barrier();                        // the LDS slice is rewritten for the next read
    }
    while (q3n > 0) evaluate(q3n < 64 ? q3n : 64);
    __syncthreads();
    if (threadIdx.x == 0) seg_count[blockIdx.x] = sh_wcount;
}

// ---------------------------------------------------------------- kernel A, sorted form: indexes beyond the caches
// On an index that does not fit L2 / Infinity Cache (C4: 175 MB, every 10-mer holds ~8 suffixes) every probe of the table, of
// SA / LCP / CHILD and of the text costs a whole 64-byte sector: the persistent kernel moves 13x its algorithmic bytes and is
// bound by memory transactions (profiles/r02_c4_pmc_*).  The queries of a batch are independent, so they are evaluated in
// 10-MER ORDER instead: the ~12 queries of a batch that share a 10-mer (and their neighbours in the table) then touch the
// same sectors at the same time.  Emission order is restored for free: every query writes its count at its FLAT index
// (read-major, then the reference's prefix order), one exclusive scan over that array yields the offsets of all matches.
//   sorted_keys_kernel   (read, t) -> key = 10-mer code (TABLE: no table jump; TABLE+1: emits nothing), value = read << 32 | t
//   hipCUB radix sort    21 key bits
//   sorted_eval_kernel   one lane per sorted query: table entry, eval_entry (direct evaluation / child-table descent), count
//   hipCUB exclusive sum over the flat counts; sorted_base_kernel picks the per-read bases out of it
//   sorted_emit_kernel   one lane per emitting query: left extension and the match records
struct SortedRec { int32_t read, start, end, depth, prefix, pad; };

__global__ __launch_bounds__(256) void sorted_keys_kernel(DIdx X, const SeedQ* __restrict__ qs, const uint8_t* __restrict__ reads,
                                                         const int64_t* __restrict__ qbase, uint32_t* __restrict__ keys,
                                                         uint64_t* __restrict__ vals) {
    const int r = blockIdx.x;
    const SeedQ Q = qs[r];
    const int64_t f0 = qbase[r];
    PatD P;
    P.len = Q.plen;
    P.p = reads + Q.poff;
    PrefixMap pm;
    pm.step = Q.sparse_q * X.K;
    pm.Lmax = Q.plen - (Q.min_len - (pm.step - 1));
    pm.nh = Q.pm_nh; pm.kh = Q.pm_kh; pm.m1 = Q.pm_m1; pm.m2 = Q.pm_m2;
    const bool use_kmer = Q.plen >= KMER;
    for (int t = threadIdx.x; t < Q.npref; t += 256) {
        const int pf = pm.prefix_of_fast(t);
        uint32_t key = TABLE;                                   // no 10-mer jump: the descent starts at the root
        if (use_kmer) {
            const uint32_t index = pf + KMER <= Q.plen ? kmer_code(P, pf) : 0xFFFFFFFFu;
            key = (index < TABLE && ((X.KBIT[index >> 5] >> (index & 31)) & 1u)) ? index : TABLE + 1;
        }
        keys[f0 + t] = key;
        vals[f0 + t] = ((uint64_t)(uint32_t)r << 32) | (uint32_t)t;
    }
}

__global__ __launch_bounds__(256) void sorted_eval_kernel(DIdx X, const SeedQ* __restrict__ qs, const uint8_t* __restrict__ reads,
                                                         const int64_t* __restrict__ qbase, const uint32_t* __restrict__ keys,
                                                         const uint64_t* __restrict__ vals, int64_t total, int64_t* __restrict__ cnt_out,
                                                         SortedRec* __restrict__ rec) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= total) return;
    const uint32_t key = keys[s];
    const uint64_t v = vals[s];
    const int r = (int)(v >> 32), t = (int)(uint32_t)v;
    const int64_t f = qbase[r] + t;
    if (key > TABLE) { cnt_out[f] = 0; return; }
    const SeedQ Q = qs[r];
    PrefixMap pm;
    pm.step = Q.sparse_q * X.K;
    pm.Lmax = Q.plen - (Q.min_len - (pm.step - 1));
    pm.nh = Q.pm_nh; pm.kh = Q.pm_kh; pm.m1 = Q.pm_m1; pm.m2 = Q.pm_m2;
    Q3 ent;
    ent.read = r; ent.prefix = pm.prefix_of_fast(t); ent.start = 0; ent.end = X.n - 1; ent.z = 0; ent.w = 0;
    if (key == TABLE) ent.prefix |= (int)0x80000000;
    else {
        const uint4 e = X.KMR2[key];
        ent.start = (int)e.x; ent.end = (int)e.y; ent.z = e.z; ent.w = e.w;
    }
    int cnt, start, end, depth;
    eval_entry(X, Q, reads, X.K, ent, cnt, start, end, depth);
    cnt_out[f] = cnt;
    if (cnt > 0) {
        SortedRec o;
        o.read = r; o.start = start; o.end = end; o.depth = depth; o.prefix = ent.prefix & 0x7fffffff; o.pad = 0;
        rec[f] = o;
    }
}

// read_base[r] = offset of read r's first match = the scanned count at the read's first flat index (read_base[nq] = total)
__global__ void sorted_base_kernel(const int64_t* __restrict__ scan, const int64_t* __restrict__ cnt, const int64_t* __restrict__ qbase, int nq,
                                   int64_t total_q, int64_t* __restrict__ read_base) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nq) return;
    const int64_t f = qbase[r];
    read_base[r] = f < total_q ? scan[f] : (total_q > 0 ? scan[total_q - 1] + cnt[total_q - 1] : 0);
}

__global__ __launch_bounds__(256) void sorted_emit_kernel(DIdx X, const SeedQ* __restrict__ qs, const uint8_t* __restrict__ reads,
                                                         const int64_t* __restrict__ scan, const int64_t* __restrict__ cnt,
                                                         const SortedRec* __restrict__ rec, int64_t total_q, const int64_t* __restrict__ read_base,
                                                         int nq, alfa_match* __restrict__ out, int64_t out_cap) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= total_q || read_base[nq] > out_cap || cnt[f] == 0) return;
    const SortedRec w = rec[f];
    const SeedQ Q = qs[w.read];
    PatG P; P.p = reads + Q.poff; P.len = Q.plen;
    const int K = X.K;
    const int minMin_len = Q.sparse_q > 1 ? Q.min_len - K : Q.min_len;
    const int border = Q.sparse_q * K;
    alfa_match* o = out + scan[f];
    for (int i = w.start; i <= w.end; ++i) {
        int64_t mr; int mq, ml;
        if (lmax_strict(X, P, w.prefix, (int64_t)X.SA[i], w.depth, minMin_len, border, mr, mq, ml)) {
            alfa_match m; m.ref = mr; m.query = mq; m.len = ml;
            *o++ = m;
        }
    }
}

// ---------------------------------------------------------------- kernel A, plain form: one read per workgroup, every lane walks
// its query to the end (used for indexes without a child table)
__global__ __launch_bounds__(SEED_BLOCK) void smam_count_simple_kernel(DIdx X, const SeedQ* __restrict__ qs, const uint8_t* __restrict__ reads,
                                                               int64_t* __restrict__ read_total, WorkRec* __restrict__ work,
                                                               int* __restrict__ work_count, int work_cap, int lds_p) {
    extern __shared__ uint8_t smem[];
    __shared__ int sh_scan[SEED_BLOCK / 64];
    const int r = blockIdx.x;
    const SeedQ Q = qs[r];
    const int tid = threadIdx.x;
    PatD P;
    P.len = Q.plen;
    if (lds_p) {
        for (int i = tid; i < Q.plen; i += SEED_BLOCK) smem[PADB + i] = reads[Q.poff + i];
        __syncthreads();
        P.p = smem + PADB;
    } else {
        P.p = reads + Q.poff;
    }
    const int K = X.K;
    const int step = Q.sparse_q * K;
    const int min_lenK = Q.min_len - (step - 1);
    const int minMin_len = Q.sparse_q > 1 ? Q.min_len - K : Q.min_len;
    PrefixMap pm;
    pm.step = step;
    pm.Lmax = Q.plen - min_lenK;
    pm.nh = pm.Lmax >= 0 ? pm.Lmax / step + 1 : 0;
    {
        const int kk = K < pm.Lmax + 1 ? K : pm.Lmax + 1;
        const int r0 = pm.Lmax >= 0 ? pm.Lmax % step : 0;
        pm.kh = kk < r0 + 1 ? kk : r0 + 1;
    }
    const int maxCount = Q.sparse_q * Q.max_count;
    const int border = step;
    int running = 0;
    for (int base = 0; base < Q.npref; base += SEED_BLOCK) {
        const int t = base + tid;
        int cnt = 0, start = 0, end = X.n - 1, depth = 0, prefix = 0;
        if (t < Q.npref) {
            prefix = pm.prefix_of(t);
            if (X.has_child) traverse_child(X, P, prefix, start, end, depth, Q.plen);
            else traverse_plain(X, P, prefix, start, end, depth, Q.plen);
            if (depth > 1 && depth >= min_lenK && (end - start + 1) <= maxCount) {
                for (int i = start; i <= end; ++i) {
                    int64_t mr; int mq, ml;
                    cnt += lmax_strict(X, P, prefix, (int64_t)X.SA[i], depth, minMin_len, border, mr, mq, ml) ? 1 : 0;
                }
            }
        }
        int total;
        const int excl = block_excl_scan(cnt, sh_scan, total);
        const int slot = wave_alloc_slot(work_count, cnt > 0);
        if (cnt > 0) {
            if (slot < work_cap) {
                WorkRec w;
                w.read = r; w.woff = running + excl; w.start = start; w.end = end; w.depth = depth; w.prefix = prefix;
                work[slot] = w;
            }
        }
        running += total;
    }
    if (tid == 0) read_total[r] = running;
}

// ---------------------------------------------------------------- kernel B: emit
__global__ void smam_emit_kernel(DIdx X, const SeedQ* __restrict__ qs, const uint8_t* __restrict__ reads,
                                 const int64_t* __restrict__ read_base, int nq, const WorkRec* __restrict__ work,
                                 const int* __restrict__ seg_count, int seg_cap, alfa_match* __restrict__ out, int64_t out_cap) {
    // the work list is segmented (one segment per producing workgroup of the persistent kernel, else a single one):
    // blockIdx.y = segment.  An overflowing segment or match array is detected by the host at fetch time, which
    // re-launches with larger buffers; records beyond a segment's capacity were never written.
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int nwork = min(seg_count[blockIdx.y], seg_cap);
    if (read_base[nq] > out_cap || g >= nwork) return;
    const WorkRec w = work[(size_t)blockIdx.y * seg_cap + g];
    const SeedQ Q = qs[w.read];
    PatG P; P.p = reads + Q.poff; P.len = Q.plen;
    const int K = X.K;
    const int minMin_len = Q.sparse_q > 1 ? Q.min_len - K : Q.min_len;
    const int border = Q.sparse_q * K;
    alfa_match* o = out + read_base[w.read] + w.woff;
    for (int i = w.start; i <= w.end; ++i) {
        int64_t mr; int mq, ml;
        if (lmax_strict(X, P, w.prefix, (int64_t)X.SA[i], w.depth, minMin_len, border, mr, mq, ml)) {
            alfa_match m; m.ref = mr; m.query = mq; m.len = ml;
            *o++ = m;
        }
    }
}

// ---------------------------------------------------------------- MEM: traverse + count
__global__ __launch_bounds__(SEED_BLOCK) void mem_count_kernel(DIdx X, const SeedQ* __restrict__ qs, const uint8_t* __restrict__ reads,
                                                              int64_t* __restrict__ read_total, WorkRec2* __restrict__ work,
                                                              int* __restrict__ work_count, int work_cap, int lds_p) {
    extern __shared__ uint8_t smem[];
    __shared__ int sh_scan[SEED_BLOCK / 64];
    const int r = blockIdx.x;
    const SeedQ Q = qs[r];
    const int tid = threadIdx.x;
    PatD P;
    P.len = Q.plen;
    if (lds_p) {
        for (int i = tid; i < Q.plen; i += SEED_BLOCK) smem[PADB + i] = reads[Q.poff + i];
        __syncthreads();
        P.p = smem + PADB;
    } else {
        P.p = reads + Q.poff;
    }
    const int K = X.K;
    const int step = Q.sparse_q * K;
    const int min_lenK = Q.min_len - (step - 1);
    PrefixMap pm;
    pm.step = step;
    pm.Lmax = Q.plen - min_lenK;
    pm.nh = pm.Lmax >= 0 ? pm.Lmax / step + 1 : 0;
    {
        const int kk = K < pm.Lmax + 1 ? K : pm.Lmax + 1;
        const int r0 = pm.Lmax >= 0 ? pm.Lmax % step : 0;
        pm.kh = kk < r0 + 1 ? kk : r0 + 1;
    }
    int running = 0;
    for (int base = 0; base < Q.npref; base += SEED_BLOCK) {
        const int t = base + tid;
        int cnt = 0, prefix = 0;
        IvalD mli = {0, X.n - 1, 0}, xmi = {0, X.n - 1, 0};
        if (t < Q.npref) {
            prefix = pm.prefix_of(t);
            traverse_any(X, P, prefix, mli, min_lenK);                    // findMEM :1006
            if (mli.depth > 1 && mli.depth >= min_lenK) {
                xmi = mli;
                traverse_any(X, P, prefix, xmi, Q.plen);                  // :1017
                cnt = collect_mem<false>(X, P, prefix, mli, xmi, Q.min_len, step, nullptr);
            }
        }
        int total;
        const int excl = block_excl_scan(cnt, sh_scan, total);
        const int slot = wave_alloc_slot(work_count, cnt > 0);
        if (cnt > 0) {
            if (slot < work_cap) {
                WorkRec2 w;
                w.read = r; w.woff = running + excl; w.prefix = prefix;
                w.ms = mli.start; w.me = mli.end; w.md = mli.depth; w.xs = xmi.start; w.xe = xmi.end; w.xd = xmi.depth;
                w.calc = 0; w.nes = 0; w.nesp = 0;
                work[slot] = w;
            }
        }
        running += total;
    }
    if (tid == 0) read_total[r] = running;
}

// ---------------------------------------------------------------- SMEM: traverse, per-read selection, count
// Per-read tables (ints, each `W = plen + 2` wide), in LDS when they fit, else in the global scratch at Q.soff:
constexpr int SMEM_TABLES = 14;
enum { T_MS, T_ME, T_MD, T_XS, T_XE, T_XD, T_ORD, T_SEL, T_NES, T_NESP, T_BL, T_SC, T_PB, T_CS };

__global__ __launch_bounds__(SEED_BLOCK) void smem_count_kernel(DIdx X, const SeedQ* __restrict__ qs, const uint8_t* __restrict__ reads,
                                                               int64_t* __restrict__ read_total, WorkRec2* __restrict__ work,
                                                               int* __restrict__ work_count, int work_cap, int lds_p, int lds_tab,
                                                               int* g_scratch) {
    extern __shared__ uint8_t smem[];
    __shared__ int sh_scan[SEED_BLOCK / 64];
    __shared__ int sh_nt;
    const int r = blockIdx.x;
    const SeedQ Q = qs[r];
    const int tid = threadIdx.x;
    PatD P;
    P.len = Q.plen;
    const int pbytes = ((Q.plen + 15) & ~15) + 2 * PADB;
    if (lds_p) {
        for (int i = tid; i < Q.plen; i += SEED_BLOCK) smem[PADB + i] = reads[Q.poff + i];
        P.p = smem + PADB;
    } else {
        P.p = reads + Q.poff;
    }
    const int W = (Q.plen > Q.npref ? Q.plen : Q.npref) + 2;
    int* tab = lds_tab ? (int*)(smem + (lds_p ? pbytes : 0)) : g_scratch + Q.soff;
    const int K = X.K;
    const int sparseQ = Q.sparse_q;
    const int step = sparseQ * K;
    const int min_lenK = Q.min_len - (step - 1);
    const int minMin_len = sparseQ > 1 ? Q.min_len - K : Q.min_len;
    // candidate prefixes are 0 .. plen - min_lenK (sparseSA.cpp:1140); Q.npref counts them
    const int np = Q.npref;                                  // == Lmax + 1 (0 when the call is a no-op)
    __syncthreads();
    // stage 1: mli of every candidate prefix
    for (int p = tid; p < np; p += SEED_BLOCK) {
        IvalD mli = {0, X.n - 1, 0};
        traverse_child(X, P, p, mli.start, mli.end, mli.depth, min_lenK);
        tab[T_MS * W + p] = mli.start; tab[T_ME * W + p] = mli.end; tab[T_MD * W + p] = mli.depth;
        tab[T_XD * W + p] = sparseQ > 1 ? -1 : -2;
    }
    __syncthreads();
    // the reference's prefix loop skips (sparseQ-1)*K positions after a prefix == K-1 (mod K) -- but only when that prefix
    // matched deeper than one character (the `continue` at :1146 bypasses the skip), so the visited set depends on the data
    if (sparseQ > 1 && tid == 0) {
        for (int p = 0; p < np; ++p) {
            tab[T_XD * W + p] = -2;
            if (tab[T_MD * W + p] <= 1) continue;
            if (p % K == K - 1) p += (sparseQ - 1) * K;
        }
    }
    __syncthreads();
    // stage 2: xmi of every visited prefix whose mli reached min_lenK
    for (int p = tid; p < np; p += SEED_BLOCK) {
        const int md = tab[T_MD * W + p];
        int xd = -1;
        if (tab[T_XD * W + p] == -2 && md > 1 && md >= min_lenK) {
            IvalD xmi = {tab[T_MS * W + p], tab[T_ME * W + p], md};
            traverse_child(X, P, p, xmi.start, xmi.end, xmi.depth, Q.plen);
            if (xmi.depth + p >= Q.min_len) { tab[T_XS * W + p] = xmi.start; tab[T_XE * W + p] = xmi.end; xd = xmi.depth; }
        }
        tab[T_XD * W + p] = xd;
    }
    for (int e = tid; e < W; e += SEED_BLOCK) { tab[T_BL * W + e] = -1; tab[T_SC * W + e] = -1; tab[T_CS * W + e] = 0; }
    __syncthreads();
    // per-read selection (sparseSA.cpp:1160-1207) on one lane: counting sort by end position (descending; prefixes stay
    // ascending inside a group = compIntervTupleRev :52-54), then the two sweeps over the groups
    if (tid == 0) {
        int* ord = tab + T_ORD * W; int* sel = tab + T_SEL * W; int* nes = tab + T_NES * W; int* nesp = tab + T_NESP * W;
        int* bl = tab + T_BL * W; int* sc = tab + T_SC * W; int* pb = tab + T_PB * W; int* cs = tab + T_CS * W;
        const int* xd = tab + T_XD * W; const int* xs = tab + T_XS * W; const int* xe = tab + T_XE * W;
        const int plen = Q.plen;
        int nt = 0;
        for (int p = 0; p < np; ++p) if (xd[p] >= 0) { cs[p + xd[p]]++; ++nt; }      // end position <= plen
        int pos = 0;
        for (int e = plen; e >= 0; --e) { const int c = cs[e]; cs[e] = pos; pos += c; }
        for (int p = 0; p < np; ++p) if (xd[p] >= 0) ord[cs[p + xd[p]]++] = p;
        // first sweep: beginLong / smemCount
        int i = 0;
        while (i < nt) {
            int64_t curSize = 0;
            const int curEnd = ord[i] + xd[ord[i]];
            bool foundMax = false;
            while (i < nt && ord[i] + xd[ord[i]] == curEnd) {
                const int p = ord[i];
                if (!foundMax || p <= bl[curEnd] + K) curSize += (int64_t)xe[p] - xs[p] + 1;
                if (!foundMax && curSize >= Q.max_smem) { bl[curEnd] = p; foundMax = true; }
                ++i;
            }
            sc[curEnd] = (int)curSize;
        }
        // pb[e] = largest e' <= e with smemCount[e'] >= max_smem_cont (the nextEndStop search :1190-1192)
        int last = -1;
        for (int e = 0; e <= plen; ++e) { if (sc[e] >= Q.max_smem) last = e; pb[e] = last; }
        // second sweep: which tuples are expanded, and with which parameters
        i = 0;
        int curMinPos = plen;
        while (i < nt) {
            const int curEnd = ord[i] + xd[ord[i]];
            const bool skipEnd = sc[curEnd] >= Q.max_count || ord[i] > curMinPos + K;
            if (!skipEnd && sc[curEnd] >= Q.max_smem) curMinPos = min(curMinPos, bl[curEnd]);
            while (i < nt && ord[i] + xd[ord[i]] == curEnd) {
                const int p = ord[i];
                int s = 0;
                if (!skipEnd && p <= curMinPos + K && (sc[curEnd] < Q.max_smem || p <= bl[curEnd] + K)) {
                    if (sc[curEnd] >= Q.max_smem) { s = 1; nes[p] = 0; nesp[p] = 0; }
                    else {
                        const int stop = curEnd >= 1 ? pb[curEnd - 1] : -1;
                        s = 2; nes[p] = stop; nesp[p] = stop >= 0 ? bl[stop] : curEnd;
                    }
                }
                sel[p] = s;
                ++i;
            }
        }
        sh_nt = nt;
    }
    __syncthreads();
    // stage 3: count what every selected tuple emits, in the sorted order
    const int nt = sh_nt;
    int running = 0;
    for (int base = 0; base < nt; base += SEED_BLOCK) {
        const int j = base + tid;
        int cnt = 0, p = 0, s = 0;
        IvalD mli = {0, 0, 0}, xmi = {0, 0, 0};
        if (j < nt) {
            p = tab[T_ORD * W + j];
            s = tab[T_SEL * W + p];
            if (s) {
                mli = {tab[T_MS * W + p], tab[T_ME * W + p], tab[T_MD * W + p]};
                xmi = {tab[T_XS * W + p], tab[T_XE * W + p], tab[T_XD * W + p]};
                cnt = collect_xmi_mli<false>(X, P, p, mli, xmi, s == 2, tab[T_NES * W + p], tab[T_NESP * W + p], minMin_len, Q.max_count,
                                             step, nullptr);
            }
        }
        int total;
        const int excl = block_excl_scan(cnt, sh_scan, total);
        const int slot = wave_alloc_slot(work_count, cnt > 0);
        if (cnt > 0) {
            if (slot < work_cap) {
                WorkRec2 w;
                w.read = r; w.woff = running + excl; w.prefix = p;
                w.ms = mli.start; w.me = mli.end; w.md = mli.depth; w.xs = xmi.start; w.xe = xmi.end; w.xd = xmi.depth;
                w.calc = s == 2; w.nes = tab[T_NES * W + p]; w.nesp = tab[T_NESP * W + p];
                work[slot] = w;
            }
        }
        running += total;
    }
    if (tid == 0) read_total[r] = running;
}

// ---------------------------------------------------------------- SMEMP (sparseSA.cpp:1211-1254; unreachable from the reference's CLI)
// traverse_keep  sparseSA.cpp:793-845: the descent of traverse_faster that also keeps the stack of the lcp-intervals it
// passes (those of at most maxBr suffixes).  collectIntervals (:968-990) walks that stack from the deepest entry upwards
// while two conditions hold that are both monotone along the stack (interval size grows, depth shrinks), so the entries it
// uses are exactly those satisfying them: nothing has to be stored.  `ring(lower, upper)` is called for every pair of
// adjacent stack entries in push order; `have`/`back` return whether the stack ended non-empty and its last entry.
template <class PAT, class F>
__device__ void traverse_stack(const DIdx& X, const PAT& P, int prefix, IvalD mli, int min_len, int maxBr, F&& ring, bool& have, IvalD& back) {
    IvalD cur = mli;
    int cnt = 1;
    back = mli;
    auto size = [](const IvalD& v) { return v.end - v.start + 1; };
    auto push = [&](const IvalD& nw) { if (cnt > 0) ring(back, nw); back = nw; ++cnt; };
    if (size(cur) > maxBr) cnt = 0;
    have = cnt > 0;
    if (cur.depth >= min_len) return;
    int c = prefix + cur.depth;
    bool found = c < P.len;
    const int curLCP = cur.start != cur.end ? ival_lcp(X, cur.start, cur.end) : -1;
    if (found && cur.start != cur.end && curLCP == cur.depth) {
        found = td_child(X, P.at(c), cur.start, cur.end, cur.depth);
        if (found && size(cur) <= maxBr) push(cur);
    } else if (found) {
        found = P.at(c) == d_s(X, (int64_t)X.SA[cur.start] + cur.depth);
    }
    bool mism = false;
    while (found && !mism && c < P.len && cur.depth < min_len) {
        ++c; ++cur.depth;
        if (cur.start != cur.end) {
            const int childLCP = ival_lcp(X, cur.start, cur.end);
            const int minimum = childLCP < min_len ? childLCP : min_len;
            const int64_t sa0 = (int64_t)X.SA[cur.start];
            while (!mism && c < P.len && cur.depth < minimum) {
                mism = d_s(X, sa0 + cur.depth) != P.at(c);
                ++c; cur.depth += !mism;
            }
            found = c < P.len && !mism && cur.depth < min_len && td_child(X, P.at(c), cur.start, cur.end, cur.depth);
            if (found && size(cur) <= maxBr) push(cur);
        } else {
            const int64_t sa0 = (int64_t)X.SA[cur.start];
            while (!mism && c < P.len && cur.depth < min_len) {
                const int64_t pos = sa0 + cur.depth;
                mism = pos >= X.N || d_s(X, pos) != P.at(c);
                ++c; cur.depth += !mism;
            }
        }
    }
    have = cnt > 0;
    if (have && back.start == cur.start && back.end == cur.end && back.depth < cur.depth) back.depth = cur.depth;      // :843-844
}

// collectIntervals  sparseSA.cpp:968-990 for one prefix whose stack ends in `xmi`: the suffixes of xmi, then (calcMli) for
// every qualifying stack entry, deepest first, the suffixes LEFT of the entry above it (the reference's second loop runs to
// xmi.end and is always empty).  The stack is re-derived by a second descent; with EMIT the groups are written back to
// front so that the deepest comes first, `total` being the count of a previous call.
template <bool EMIT, class PAT>
__device__ int collect_intervals(const DIdx& X, const PAT& P, int prefix, IvalD mli, IvalD xmi, bool calcMli, int nextEndStop, int min_len,
                                 int maxCount, int maxLeft, int total, alfa_match* o) {
    if (xmi.depth + prefix < min_len) return 0;
    int cnt = 0;
    for (int i = xmi.start; i <= xmi.end; ++i) {
        int64_t mr; int mq, ml;
        if (lmax_strict(X, P, prefix, (int64_t)X.SA[i], xmi.depth, min_len, maxLeft, mr, mq, ml)) {
            if (EMIT) { alfa_match m; m.ref = mr; m.query = mq; m.len = ml; o[cnt] = m; }
            ++cnt;
        }
    }
    if (!calcMli) return cnt;
    const int K = X.K;
    const int thr = max(nextEndStop - K, min_len);
    int tail = total;
    auto ring = [&](const IvalD& lo, const IvalD& up) {
        if (!(lo.end - lo.start + 1 <= maxCount / K && prefix + lo.depth > thr)) return;
        int c = 0;
        for (int i = lo.start; i < up.start; ++i) {
            int64_t mr; int mq, ml;
            if (lmax_strict(X, P, prefix, (int64_t)X.SA[i], lo.depth, min_len, maxLeft, mr, mq, ml)) ++c;
        }
        if (EMIT) {
            tail -= c;
            int w = 0;
            for (int i = lo.start; i < up.start; ++i) {
                int64_t mr; int mq, ml;
                if (lmax_strict(X, P, prefix, (int64_t)X.SA[i], lo.depth, min_len, maxLeft, mr, mq, ml)) {
                    alfa_match m; m.ref = mr; m.query = mq; m.len = ml;
                    o[tail + w++] = m;
                }
            }
        }
        cnt += c;
    };
    bool have; IvalD back;
    traverse_stack(X, P, prefix, mli, P.len, maxCount, ring, have, back);
    return cnt;
}

__global__ __launch_bounds__(SEED_BLOCK) void smemp_count_kernel(DIdx X, const SeedQ* __restrict__ qs, const uint8_t* __restrict__ reads,
                                                                int64_t* __restrict__ read_total, WorkRec2* __restrict__ work,
                                                                int* __restrict__ work_count, int work_cap, int lds_p, int lds_tab,
                                                                int* g_scratch) {
    extern __shared__ uint8_t smem[];
    __shared__ int sh_scan[SEED_BLOCK / 64];
    const int r = blockIdx.x;
    const SeedQ Q = qs[r];
    const int tid = threadIdx.x;
    PatD P;
    P.len = Q.plen;
    const int pbytes = ((Q.plen + 15) & ~15) + 2 * PADB;
    if (lds_p) {
        for (int i = tid; i < Q.plen; i += SEED_BLOCK) smem[PADB + i] = reads[Q.poff + i];
        P.p = smem + PADB;
    } else {
        P.p = reads + Q.poff;
    }
    const int W = (Q.plen > Q.npref ? Q.plen : Q.npref) + 2;
    int* tab = lds_tab ? (int*)(smem + (lds_p ? pbytes : 0)) : g_scratch + Q.soff;
    const int K = X.K;
    const int sparseQ = Q.sparse_q;
    const int step = sparseQ * K;
    const int min_lenK = Q.min_len - (step - 1);
    const int minMin_len = sparseQ > 1 ? Q.min_len - K : Q.min_len;
    const int np = Q.npref;
    __syncthreads();
    // stage 1: mli of every candidate prefix (:1224)
    for (int p = tid; p < np; p += SEED_BLOCK) {
        IvalD mli = {0, X.n - 1, 0};
        traverse_child(X, P, p, mli.start, mli.end, mli.depth, min_lenK);
        tab[T_MS * W + p] = mli.start; tab[T_ME * W + p] = mli.end; tab[T_MD * W + p] = mli.depth;
        tab[T_XD * W + p] = sparseQ > 1 ? -1 : -2;
    }
    __syncthreads();
    if (sparseQ > 1 && tid == 0) {          // the data-dependent skip of :1225,:1252 (as in SMEM)
        for (int p = 0; p < np; ++p) {
            tab[T_XD * W + p] = -2;
            if (tab[T_MD * W + p] <= 1) continue;
            if (p % K == K - 1) p += (sparseQ - 1) * K;
        }
    }
    __syncthreads();
    // stage 2: the last stack entry of every visited prefix whose mli reached min_lenK (:1226-1228)
    for (int p = tid; p < np; p += SEED_BLOCK) {
        const int md = tab[T_MD * W + p];
        int xd = -1;
        if (tab[T_XD * W + p] == -2 && md > 1 && md >= min_lenK) {
            const IvalD mli = {tab[T_MS * W + p], tab[T_ME * W + p], md};
            bool have; IvalD back;
            traverse_stack(X, P, p, mli, Q.plen, Q.max_count, [](const IvalD&, const IvalD&) {}, have, back);
            if (have) { tab[T_XS * W + p] = back.start; tab[T_XE * W + p] = back.end; xd = back.depth; }
        }
        tab[T_XD * W + p] = xd;
    }
    __syncthreads();
    // the running decisions of :1230-1248, in prefix order, on one lane
    if (tid == 0) {
        int prevEnd = 0, prevEndPrefix = 0;
        int64_t prevSize = -1;
        for (int p = 0; p < np; ++p) {
            const int xd = tab[T_XD * W + p];
            if (xd < 0) { tab[T_SEL * W + p] = 0; continue; }
            const int64_t sz = (int64_t)tab[T_XE * W + p] - tab[T_XS * W + p] + 1;
            const int end = p + xd;
            const bool extendXmi = end > prevEnd || prevSize <= Q.max_smem || p <= prevEndPrefix + K || sz <= Q.max_smem;
            const bool extendMli = extendXmi && sz <= Q.max_smem && (end > prevEnd || prevSize <= Q.max_smem);
            tab[T_SEL * W + p] = extendXmi ? (extendMli ? 2 : 1) : 0;
            tab[T_NES * W + p] = prevEnd;
            if (prevEnd < end) { prevEnd = end; prevSize = sz; prevEndPrefix = prevSize > Q.max_smem ? p : -1; }
            else { prevSize += sz; if (prevEndPrefix < 0 && prevSize > Q.max_smem) prevEndPrefix = p; }
        }
    }
    __syncthreads();
    // stage 3: what every selected prefix emits, in prefix order
    int running = 0;
    for (int base = 0; base < np; base += SEED_BLOCK) {
        const int p = base + tid;
        int cnt = 0, s = 0;
        IvalD mli = {0, 0, 0}, xmi = {0, 0, 0};
        if (p < np && (s = tab[T_SEL * W + p]) != 0) {
            mli = {tab[T_MS * W + p], tab[T_ME * W + p], tab[T_MD * W + p]};
            xmi = {tab[T_XS * W + p], tab[T_XE * W + p], tab[T_XD * W + p]};
            cnt = collect_intervals<false>(X, P, p, mli, xmi, s == 2, tab[T_NES * W + p], minMin_len, Q.max_count, step, 0, nullptr);
        }
        int total;
        const int excl = block_excl_scan(cnt, sh_scan, total);
        const int slot = wave_alloc_slot(work_count, cnt > 0);
        if (cnt > 0 && slot < work_cap) {
            WorkRec2 w;
            w.read = r; w.woff = running + excl; w.prefix = p;
            w.ms = mli.start; w.me = mli.end; w.md = mli.depth; w.xs = xmi.start; w.xe = xmi.end; w.xd = xmi.depth;
            w.calc = s == 2; w.nes = tab[T_NES * W + p]; w.nesp = cnt;
            work[slot] = w;
        }
        running += total;
    }
    if (tid == 0) read_total[r] = running;
}

// ---------------------------------------------------------------- MEM / SMEM: emit
template <int TYPE>
__global__ void emit2_kernel(DIdx X, const SeedQ* __restrict__ qs, const uint8_t* __restrict__ reads, const int64_t* __restrict__ read_base,
                             int nq, const WorkRec2* __restrict__ work, const int* __restrict__ seg_count, int seg_cap,
                             alfa_match* __restrict__ out, int64_t out_cap) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int nwork = min(seg_count[blockIdx.y], seg_cap);
    if (read_base[nq] > out_cap || g >= nwork) return;
    const WorkRec2 w = work[(size_t)blockIdx.y * seg_cap + g];
    const SeedQ Q = qs[w.read];
    PatG P; P.p = reads + Q.poff; P.len = Q.plen;
    const int K = X.K;
    const int step = Q.sparse_q * K;
    alfa_match* o = out + read_base[w.read] + w.woff;
    const IvalD mli = {w.ms, w.me, w.md}, xmi = {w.xs, w.xe, w.xd};
    if (TYPE == ALFA_SEED_MEM) collect_mem<true>(X, P, w.prefix, mli, xmi, Q.min_len, step, o);
    else if (TYPE == ALFA_SEED_SMEMP) collect_intervals<true>(X, P, w.prefix, mli, xmi, w.calc != 0, w.nes, Q.sparse_q > 1 ? Q.min_len - K : Q.min_len,
                                                              Q.max_count, step, w.nesp /* the count pass's total */, o);
    else collect_xmi_mli<true>(X, P, w.prefix, mli, xmi, w.calc != 0, w.nes, w.nesp, Q.sparse_q > 1 ? Q.min_len - K : Q.min_len,
                               Q.max_count, step, o);
}

// ---------------------------------------------------------------- SMAM on a suffix-link index (--suflink)
// expand_link src/sparseSA.h:223-239, suffixlink src/sparseSA.cpp:849-855
__device__ bool d_suffixlink(const DIdx& X, IvalD& m) {
    m.depth -= X.K;
    if (m.depth <= 0) return false;
    const int64_t is = (int64_t)X.SA[m.start] / X.K + 1, ie = (int64_t)X.SA[m.end] / X.K + 1;
    if (is >= X.n || ie >= X.n) return false;          // (the reference reads one past its ISA here)
    int start = X.ISA[is], end = X.ISA[ie];
    m.start = start; m.end = end;                      // kept (unexpanded) when the expansion below gives up, as in the reference
    const int64_t thresh = 2 * (int64_t)m.depth * X.logN;
    int64_t exp = 0;
    while (start > 0 && d_lcp(X, start) >= m.depth) { if (++exp >= thresh) return false; --start; }
    while (end < X.NKm1 && d_lcp(X, end + 1) >= m.depth) { if (++exp >= thresh) return false; ++end; }
    m.start = start; m.end = end;
    return true;
}

// With suffix links the prefixes of one phase k form a chain (findSMAM :1063-1114: the interval reached for one prefix is
// carried, through sparseQ suffix links, into the next), so the unit of parallel work is (read, phase): lane k of the read's
// workgroup walks phase k.  Pass 1 (EMIT = false) counts per phase; pass 2 re-walks and writes at the scanned offsets.
template <int TYPE, bool EMIT>
__global__ __launch_bounds__(SEED_BLOCK) void smam_suflink_kernel(DIdx X, const SeedQ* __restrict__ qs, const uint8_t* __restrict__ reads,
                                                                 int64_t* __restrict__ read_total, int32_t* __restrict__ phase_cnt,
                                                                 const int64_t* __restrict__ read_base, alfa_match* __restrict__ out,
                                                                 int64_t out_cap, int nq, int lds_p) {
    extern __shared__ uint8_t smem[];
    __shared__ int sh_scan[SEED_BLOCK / 64];
    const int r = blockIdx.x;
    const SeedQ Q = qs[r];
    const int tid = threadIdx.x;
    PatD P;
    P.len = Q.plen;
    if (lds_p) {
        for (int i = tid; i < Q.plen; i += SEED_BLOCK) smem[PADB + i] = reads[Q.poff + i];
        __syncthreads();
        P.p = smem + PADB;
    } else {
        P.p = reads + Q.poff;
    }
    if (EMIT && read_base[nq] > out_cap) return;
    const int K = X.K;
    const int step = Q.sparse_q * K;
    const int min_lenK = Q.min_len - (step - 1);
    const int minMin_len = Q.sparse_q > 1 ? Q.min_len - K : Q.min_len;
    const int maxCount = Q.sparse_q * Q.max_count;
    const bool active = Q.min_len >= K;                       // sparseSA::SMAM :1122
    int running = 0;
    for (int kb = 0; kb < K; kb += SEED_BLOCK) {
        const int k = kb + tid;
        int cnt = 0;
        int my_off = 0;
        if (EMIT) {
            const int c = (active && k < K) ? phase_cnt[(int64_t)r * K + k] : 0;
            int total;
            my_off = running + block_excl_scan(c, sh_scan, total);
            running += total;
        }
        if (active && k < K) {
            alfa_match* o = EMIT ? out + read_base[r] + my_off : nullptr;
            IvalD xmi = {0, X.n - 1, 0};
            int prefix = k;
            if (TYPE == ALFA_SEED_SMAM) {
                while (prefix <= Q.plen - min_lenK) {
                    traverse_any(X, P, prefix, xmi, Q.plen);
                    if (xmi.depth <= 1) { xmi = {0, X.n - 1, 0}; prefix += step; continue; }
                    if (xmi.depth >= min_lenK && (xmi.end - xmi.start + 1) <= maxCount) {
                        for (int i = xmi.start; i <= xmi.end; ++i) {
                            int64_t mr; int mq, ml;
                            if (lmax_strict(X, P, prefix, (int64_t)X.SA[i], xmi.depth, minMin_len, step, mr, mq, ml)) {
                                if (EMIT) { alfa_match m; m.ref = mr; m.query = mq; m.len = ml; o[cnt] = m; }
                                ++cnt;
                            }
                        }
                    }
                    prefix += step;
                    int i = 0;
                    bool ok = true;
                    while (i < Q.sparse_q && (ok = d_suffixlink(X, xmi))) ++i;
                    if (!ok) xmi = {0, X.n - 1, 0};
                }
            } else {
                // findMEM :993-1060 with suffix links: mli (matched to min_lenK) and xmi (matched to the end) are both carried
                IvalD mli = {0, X.n - 1, 0};
                while (prefix <= Q.plen - min_lenK) {
                    traverse_any(X, P, prefix, mli, min_lenK);
                    if (mli.depth > xmi.depth) xmi = mli;
                    if (mli.depth <= 1) { mli = {0, X.n - 1, 0}; xmi = mli; prefix += step; continue; }
                    bool ok = true;
                    int i = 0;
                    if (mli.depth >= min_lenK) {
                        traverse_any(X, P, prefix, xmi, Q.plen);
                        cnt += collect_mem<EMIT>(X, P, prefix, mli, xmi, Q.min_len, step, EMIT ? o + cnt : nullptr);
                        prefix += step;
                        while (i < Q.sparse_q && (ok = d_suffixlink(X, mli))) { (void)d_suffixlink(X, xmi); ++i; }
                        if (!ok) { mli = {0, X.n - 1, 0}; xmi = mli; }
                    } else {
                        prefix += step;
                        while (i < Q.sparse_q && (ok = d_suffixlink(X, mli))) ++i;
                        if (!ok) { mli = {0, X.n - 1, 0}; xmi = mli; }
                        else xmi = mli;
                    }
                }
            }
        }
        if (!EMIT) {
            if (active && k < K) phase_cnt[(int64_t)r * K + k] = cnt;
            int total;
            (void)block_excl_scan(cnt, sh_scan, total);
            running += total;
        }
    }
    if (!EMIT && tid == 0) read_total[r] = running;
}

// number of prefix queries of one SMAM call (mirrors findSMAM's loop bounds, sparseSA.cpp:1066-1072)
int64_t count_prefixes(int64_t K, int plen, int min_len, int sparse_q) {
    if (min_len < K) return 0;                       // sparseSA.cpp:1122
    const int64_t step = (int64_t)sparse_q * K;
    if (step <= 0) return -1;
    const int64_t min_lenK = min_len - (step - 1);
    const int64_t L = (int64_t)plen - min_lenK;
    int64_t tot = 0;
    for (int64_t k = 0; k < K && k <= L; ++k) tot += (L - k) / step + 1;
    return tot;
}

// candidate prefixes of one SMEM call: 0 .. |P| - min_lenK (sparseSA.cpp:1128,1140)
int64_t count_prefixes_smem(int64_t K, int plen, int min_len, int sparse_q) {
    if (min_len < K) return 0;
    const int64_t min_lenK = min_len - ((int64_t)sparse_q * K - 1);
    return std::max<int64_t>(0, (int64_t)plen - min_lenK + 1);
}

// S with PADB zero bytes in front and 32 behind (the kernels compare 16 bytes per step)
int upload_padded(alfa::DevBuf& b, const void* src, size_t bytes) {
    int rc = b.reserve(bytes + PADB + 32 + 16);
    if (rc) return rc;
    ALFA_HIP(hipMemset(b.p, 0, bytes + PADB + 32 + 16));
    if (bytes) ALFA_HIP(hipMemcpy((char*)b.p + PADB, src, bytes, hipMemcpyHostToDevice));
    return ALFA_OK;
}

int upload(alfa::DevBuf& b, const void* src, size_t bytes, size_t pad = 0) {
    int rc = b.reserve(bytes + pad + 16);
    if (rc) return rc;
    if (bytes) ALFA_HIP(hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
    if (pad) ALFA_HIP(hipMemset((char*)b.p + bytes, 0, pad));
    return ALFA_OK;
}

}  // namespace

extern "C" {

int alfa_index_upload(const alfa_index_view* v, int device, alfa_index** out) {
    if (!v || !out) return alfa::fail(ALFA_EINVAL, "null argument");
    if (!v->S || !v->SA || !v->LCP || v->n <= 0) return alfa::fail(ALFA_EINVAL, "incomplete index view");
    if (v->n >= 0x7fffffffLL) return alfa::fail(ALFA_EINVAL, "index too large for int32 SA ranks");
    alfa::Ctx* cx = alfa::thread_ctx(device);
    if (!cx) return ALFA_ENODEV;
    alfa_index* ix = new alfa_index();
    ix->device = device;
    ix->N = v->N; ix->n = v->n; ix->K = v->K; ix->logN = v->logN; ix->NKm1 = v->NKm1; ix->nM = v->nM;
    ix->has_child = v->has_child && v->CHILD; ix->has_suflink = v->has_suflink && v->ISA; ix->has_kmer = v->has_kmer && v->KMR;
    int rc = ALFA_OK;
    if ((rc = upload_padded(ix->S, v->S, (size_t)v->N)) || (rc = upload(ix->SA, v->SA, 4 * (size_t)v->n)) ||
        (rc = upload(ix->LCP, v->LCP, (size_t)v->n)) || (rc = upload(ix->Midx, v->LCP_M_idx, 8 * (size_t)v->nM)) ||
        (rc = upload(ix->Mval, v->LCP_M_val, 4 * (size_t)v->nM)) ||
        (ix->has_child && (rc = upload(ix->CHILD, v->CHILD, 4 * (size_t)v->n))) ||
        (ix->has_kmer && (rc = upload(ix->KMR, v->KMR, 8 * (size_t)TABLE))) ||
        (ix->has_suflink && v->ISA && (rc = upload(ix->ISA, v->ISA, 4 * (size_t)v->n))) ||
        (v->n_seq > 0 && v->startpos && (rc = upload(ix->STARTPOS, v->startpos, 8 * (size_t)v->n_seq)))) {
        alfa_index_free(ix);
        return rc;
    }
    ix->n_seq = (v->n_seq > 0 && v->startpos) ? v->n_seq : 0;
    {
        // A NUL byte inside a read matches no text character in the reference (it only ever compares P[x] with S[y]), but the
        // kernels' chunked compares would take it for the NUL padding behind the text: reads are packed with their NULs replaced
        // by a byte value that occurs nowhere in S, which is equivalent for every comparison the algorithms make.
        bool seen[256] = {};
        for (int64_t i = 0; i < v->N; ++i) seen[v->S[i]] = true;
        for (int c = 1; c < 256 && !ix->nul_sub; ++c)
            if (!seen[c] && c != 'a' && c != 'c' && c != 'g' && c != 't') ix->nul_sub = (uint8_t)c;
    }
    if (ix->has_kmer) {
        // device-side companions of the 10-mer table (layout is ours; content is a function of KMR and SA only)
        std::vector<uint32_t> k2(4 * (size_t)TABLE), kb(TABLE / 32, 0u);
        size_t nset = 0;
        for (size_t i = 0; i < TABLE; ++i) {
            const uint32_t l = v->KMR[2 * i], r = v->KMR[2 * i + 1];
            if (r > 0 && (l > r || r >= (uint64_t)v->n)) {
                alfa_index_free(ix);
                return alfa::fail(ALFA_EINVAL, "10-mer table entry outside the suffix array (left <= right < n violated)");
            }
            const uint32_t sa = (l < (uint64_t)v->n) ? v->SA[l] : 0u;
            uint32_t nxt = 0;        // one suffix: S[sa+10 .. sa+13] (the host S carries 16 NUL bytes behind N); more: SA[left+1]
            if (r > l) nxt = (l + 1 < (uint64_t)v->n) ? v->SA[l + 1] : 0u;
            else if ((int64_t)sa + KMER + 4 <= v->N + 16) memcpy(&nxt, v->S + sa + KMER, 4);
            k2[4 * i] = l; k2[4 * i + 1] = r; k2[4 * i + 2] = sa; k2[4 * i + 3] = nxt;
            if (r > 0) { kb[i >> 5] |= 1u << (i & 31); ++nset; }
        }
        ix->kbit_fill = (double)nset / (double)TABLE;
        if ((rc = upload(ix->KMR2, k2.data(), 16 * (size_t)TABLE)) || (rc = upload(ix->KBIT, kb.data(), TABLE / 8))) {
            alfa_index_free(ix);
            return rc;
        }
    }
    *out = ix;
    return ALFA_OK;
}

// ---- index files straight to device memory (second half; sparseSA::load src/sparseSA.cpp:390-476)
namespace {
// the 10-mer table's device companions from arrays that are already on the device (alfa_index_upload derives them on the host)
__global__ void kmr2_kernel(const uint2* __restrict__ KMR, const uint32_t* __restrict__ SA, const uint8_t* __restrict__ S0, int64_t n, int64_t N,
                            uint4* __restrict__ K2) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= TABLE) return;
    const uint2 e = KMR[i];
    const uint32_t l = e.x, r = e.y;
    const uint32_t sa = (l < (uint64_t)n) ? SA[l] : 0u;
    uint32_t nxt = 0;
    if (r > l) nxt = (l + 1 < (uint64_t)n) ? SA[l + 1] : 0u;
    else if ((int64_t)sa + KMER + 4 <= N + 16) {            // S0 = S[0]: zero bytes behind the text, as the host copy has
        const uint8_t* p = S0 + sa + KMER;
        nxt = (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24;
    }
    K2[i] = make_uint4(l, r, sa, nxt);
}

// `bytes` of file f (from its current position) into b, through two pinned buffers: the read of one chunk overlaps the copy of the last
int stream_file(FILE* f, alfa::DevBuf& b, size_t bytes, size_t pad, hipStream_t st, alfa::PinBuf pin[2], hipEvent_t ev[2]) {
    int rc = b.reserve(bytes + pad + 16);
    if (rc) return rc;
    constexpr size_t CH = (size_t)32 << 20;
    size_t done = 0;
    for (int k = 0; done < bytes; ++k) {
        const int s = k & 1;
        const size_t m = std::min(CH, bytes - done);
        if ((rc = pin[s].reserve(CH))) return rc;
        if (k >= 2) ALFA_HIP(hipEventSynchronize(ev[s]));           // the copy that last used this buffer
        if (fread(pin[s].p, 1, m, f) != m) return alfa::fail(ALFA_EINVAL, "index file truncated");
        ALFA_HIP(hipMemcpyAsync((char*)b.p + done, pin[s].p, m, hipMemcpyHostToDevice, st));
        ALFA_HIP(hipEventRecord(ev[s], st));
        done += m;
    }
    if (pad) ALFA_HIP(hipMemsetAsync((char*)b.p + bytes, 0, pad, st));
    ALFA_HIP(hipStreamSynchronize(st));
    return ALFA_OK;
}
}  // namespace

int alfa_index_load_to_device(const char* prefix, int device, alfa_host_index** host, alfa_index** out) {
    if (!prefix || !host || !out) return alfa::fail(ALFA_EINVAL, "null argument");
    int rc;
    const bool own_host = *host == nullptr;
    if (own_host && (rc = alfa_host_index_load_text(prefix, host))) return rc;
    const alfa_index_view* v = alfa_host_index_view(*host);
    alfa::Ctx* cx = alfa::thread_ctx(device);
    if (!cx) { if (own_host) { alfa_host_index_free(*host); *host = nullptr; } return ALFA_ENODEV; }
    alfa_index* ix = new alfa_index();
    ix->device = device;
    ix->N = v->N; ix->K = v->K; ix->logN = v->logN; ix->NKm1 = v->NKm1;
    alfa::PinBuf pin[2];
    hipEvent_t ev[2] = {nullptr, nullptr};
    std::vector<FILE*> open_files;
    auto fail_out = [&](int code) {
        for (FILE* f : open_files) fclose(f);
        for (auto& e : ev) if (e) (void)hipEventDestroy(e);
        pin[0].release(); pin[1].release();
        alfa_index_free(ix);
        if (own_host) { alfa_host_index_free(*host); *host = nullptr; }
        return code;
    };
    for (auto& e : ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return fail_out(alfa::fail(ALFA_EHIP, "hipEventCreate"));
    const std::string b(prefix);
    auto open = [&](const char* ext) { FILE* f = fopen((b + ext).c_str(), "rb"); if (f) open_files.push_back(f); return f; };
    auto count = [&](FILE* f, uint32_t& n) { return fread(&n, 4, 1, f) == 1; };
    hipStream_t st = cx->stream;
    if ((rc = upload_padded(ix->S, v->S, (size_t)v->N))) return fail_out(rc);
    uint32_t n = 0;
    {
        FILE* f = open(".sa");
        if (!f || !count(f, n) || n == 0) return fail_out(alfa::fail(ALFA_EINVAL, "unable to read " + b + ".sa"));
        if (n >= 0x7fffffffu) return fail_out(alfa::fail(ALFA_EINVAL, "index too large for int32 SA ranks"));
        if ((rc = stream_file(f, ix->SA, 4 * (size_t)n, 0, st, pin, ev))) return fail_out(rc);
    }
    ix->n = n;
    {
        FILE* f = open(".lcp");
        uint32_t nl = 0, nm = 0;
        if (!f || !count(f, nl) || !count(f, nm) || nl != n) return fail_out(alfa::fail(ALFA_EINVAL, "unable to read " + b + ".lcp"));
        if ((rc = stream_file(f, ix->LCP, (size_t)nl, 0, st, pin, ev))) return fail_out(rc);
        std::vector<int64_t> midx(nm);
        std::vector<int32_t> mval(nm);
        for (uint32_t i = 0; i < nm; ++i) {
            struct { uint64_t idx; int32_t val; int32_t pad; } it;      // vec_uchar::item_t on disk (src/sparseSA.h:60-70)
            if (fread(&it, sizeof(it), 1, f) != 1) return fail_out(alfa::fail(ALFA_EINVAL, b + ".lcp is truncated"));
            midx[i] = (int64_t)it.idx; mval[i] = it.val;
        }
        ix->nM = nm;
        if ((rc = upload(ix->Midx, midx.data(), 8 * (size_t)nm)) || (rc = upload(ix->Mval, mval.data(), 4 * (size_t)nm))) return fail_out(rc);
    }
    ix->has_child = v->has_child; ix->has_suflink = v->has_suflink; ix->has_kmer = v->has_kmer;
    if (ix->has_child) {
        FILE* f = open(".child");
        uint32_t nc = 0;
        if (!f || !count(f, nc) || nc != n) return fail_out(alfa::fail(ALFA_EINVAL, "unable to read " + b + ".child"));
        if ((rc = stream_file(f, ix->CHILD, 4 * (size_t)nc, 0, st, pin, ev))) return fail_out(rc);
    }
    if (ix->has_suflink) {
        FILE* f = open(".isa");
        uint32_t ni = 0;
        if (!f || !count(f, ni) || ni != n) return fail_out(alfa::fail(ALFA_EINVAL, "unable to read " + b + ".isa"));
        if ((rc = stream_file(f, ix->ISA, 4 * (size_t)ni, 0, st, pin, ev))) return fail_out(rc);
    }
    if (v->n_seq > 0 && v->startpos && (rc = upload(ix->STARTPOS, v->startpos, 8 * (size_t)v->n_seq))) return fail_out(rc);
    ix->n_seq = (v->n_seq > 0 && v->startpos) ? v->n_seq : 0;
    {
        bool seen[256] = {};                                 // (see alfa_index_upload)
        for (int64_t i = 0; i < v->N; ++i) seen[v->S[i]] = true;
        for (int c = 1; c < 256 && !ix->nul_sub; ++c)
            if (!seen[c] && c != 'a' && c != 'c' && c != 'g' && c != 't') ix->nul_sub = (uint8_t)c;
    }
    if (ix->has_kmer) {
        FILE* f = open(".kmer");
        uint32_t nk = 0;
        if (!f || !count(f, nk) || nk != TABLE) return fail_out(alfa::fail(ALFA_EINVAL, "unable to read " + b + ".kmer"));
        std::vector<uint32_t> kmr(2 * (size_t)TABLE), kb(TABLE / 32, 0u);
        if (fread(kmr.data(), 8, TABLE, f) != TABLE) return fail_out(alfa::fail(ALFA_EINVAL, b + ".kmer is truncated"));
        size_t nset = 0;
        for (size_t i = 0; i < TABLE; ++i) {
            const uint32_t l = kmr[2 * i], r = kmr[2 * i + 1];
            if (r > 0 && (l > r || r >= n)) return fail_out(alfa::fail(ALFA_EINVAL, "10-mer table entry outside the suffix array (left <= right < n violated)"));
            if (r > 0) { kb[i >> 5] |= 1u << (i & 31); ++nset; }
        }
        ix->kbit_fill = (double)nset / (double)TABLE;
        if ((rc = upload(ix->KMR, kmr.data(), 8 * (size_t)TABLE)) || (rc = upload(ix->KBIT, kb.data(), TABLE / 8)) ||
            (rc = ix->KMR2.reserve(16 * (size_t)TABLE + 16))) return fail_out(rc);
        hipLaunchKernelGGL(kmr2_kernel, dim3(TABLE / 256), dim3(256), 0, st, ix->KMR.as<uint2>(), ix->SA.as<uint32_t>(), ix->S.as<uint8_t>() + PADB,
                           (int64_t)n, ix->N, ix->KMR2.as<uint4>());
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return fail_out(alfa::fail(ALFA_EHIP, "kmr2_kernel"));
    }
    for (FILE* f : open_files) fclose(f);
    for (auto& e : ev) (void)hipEventDestroy(e);
    pin[0].release(); pin[1].release();
    *out = ix;
    return ALFA_OK;
}

void alfa_index_free(alfa_index* ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    ix->S.release(); ix->SA.release(); ix->LCP.release(); ix->Midx.release(); ix->Mval.release(); ix->CHILD.release(); ix->KMR.release(); ix->KMR2.release(); ix->KBIT.release(); ix->ISA.release(); ix->STARTPOS.release();
    for (alfa::DevBuf& b : ix->zbuf) b.release();
    delete ix;
}

int alfa_index_device(const alfa_index* ix) { return ix ? ix->device : -1; }

}  // extern "C"

struct alfa_seed_plan {
    alfa_index* ix = nullptr;
    int dev = 0;                 // ix->device, kept for the release of a recycled plan
    int nq = 0, max_plen = 0;
    int64_t total_pref = 0;
    size_t read_bytes = 0;
    int nseg = 1, seg_cap = 0;   // work list: nseg segments of seg_cap records (persistent SMAM kernel: one per workgroup)
    bool persistent = false;
    bool suflink = false;        // SMAM on a suffix-link index: (read, phase) chains, no work list
    bool sorted = false;         // SMAM on an index beyond the caches: queries evaluated in 10-mer order (sorted_*_kernel)
    size_t sort_tmp = 0;
    alfa::DevBuf d_qbase, d_keys, d_keys2, d_vals, d_vals2, d_fcnt, d_fscan, d_frec, d_sorttmp;
    int stride = 0;              // persistent kernel: LDS bytes per wave for its read
    int64_t out_cap = 0;
    alfa::DevBuf d_reads, d_q, d_tot, d_base, d_work, d_cnt, d_out, d_tmp, d_scr, d_q3;
    alfa::DevBuf d_reg, d_rm, d_chain, d_aux;     // alfa_seed_regions_batch: regions, their members, first chains, per-strand ints
    // alfa_map_reads_batch: per-read parameters and strand offsets, DP scratch, slotted and dense results, scoring parameters
    alfa::DevBuf d_marena;
    alfa::PinBuf h_mint, h_mstage;
    hipEvent_t mev0 = nullptr, mev1 = nullptr;
    size_t tmp_bytes = 0;
    int mem_type = ALFA_SEED_SMAM;
    size_t work_rec = sizeof(WorkRec);
    int max_w = 0;          // SMEM: widest per-read table
    bool lds_tab = true;    // SMEM: tables fit in LDS
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool launched = false;
    const alfa::Ctx* owner = nullptr;   // launch and fetch are ordered on the creating thread's stream: other threads are refused
};

namespace {

DIdx make_didx(const alfa_index* ix) {
    DIdx X;
    X.S = ix->S.as<uint8_t>() + PADB; X.SA = ix->SA.as<uint32_t>(); X.LCP = ix->LCP.as<uint8_t>();
    X.Midx = ix->Midx.as<int64_t>(); X.Mval = ix->Mval.as<int32_t>(); X.CHILD = ix->CHILD.as<int32_t>();
    X.KMR = ix->KMR.as<uint2>(); X.KMR2 = ix->KMR2.as<uint4>(); X.KBIT = ix->KBIT.as<uint32_t>(); X.ISA = ix->ISA.as<int32_t>();
    X.logN = (int32_t)ix->logN; X.NKm1 = (int32_t)ix->NKm1;
    X.N = ix->N; X.n = (int32_t)ix->n; X.nM = (int32_t)ix->nM; X.K = (int32_t)ix->K; X.has_kmer = ix->has_kmer; X.has_child = ix->has_child;
    return X;
}

// the plan's work-list counters: behind the per-read totals in d_tot (plan_fill reserves room for 4096 of them)
static inline int* plan_cnt(const alfa_seed_plan* pl) { return (int*)((char*)pl->d_tot.p + ((8 * ((size_t)pl->nq + 1) + 255) & ~(size_t)255)); }

int plan_fill(alfa_seed_plan* pl, alfa::Ctx* cx, alfa_index* ix, int mem_type, const alfa_seed_query* q, int nq) {
    if (mem_type != ALFA_SEED_SMAM && mem_type != ALFA_SEED_MEM && mem_type != ALFA_SEED_SMEM && mem_type != ALFA_SEED_SMEMP)
        return alfa::fail(ALFA_EINVAL, "unknown seed type");
    const bool smem_like = mem_type == ALFA_SEED_SMEM || mem_type == ALFA_SEED_SMEMP;
    if (!ix->has_child && smem_like)
        return alfa::fail(ALFA_ENOTSUP, "SMEM / SMEMP need an index with a child table (the reference calls traverse_faster unconditionally)");

    pl->ix = ix; pl->dev = ix->device; pl->nq = nq; pl->launched = false; pl->mem_type = mem_type;
    pl->work_rec = mem_type == ALFA_SEED_SMAM ? sizeof(WorkRec) : sizeof(WorkRec2);
    if (!pl->ev0) { ALFA_HIP(hipEventCreate(&pl->ev0)); ALFA_HIP(hipEventCreate(&pl->ev1)); }
    size_t bytes = 0;
    int max_plen = 0;
    for (int i = 0; i < nq; ++i) {
        if (q[i].plen < 0 || (q[i].plen > 0 && !q[i].P)) return alfa::fail(ALFA_EINVAL, "bad read in seed query");
        if (q[i].sparse_q < 1) return alfa::fail(ALFA_EINVAL, "sparse_q must be >= 1");
        bytes += (size_t)q[i].plen;
        max_plen = std::max(max_plen, q[i].plen);
    }
    pl->read_bytes = bytes; pl->max_plen = max_plen;
    if (nq == 0) return ALFA_OK;
    int rc;
    bytes += PADB;    // front pad; the reserve below adds the pad behind the last read
    if ((rc = cx->h_in.reserve(bytes + 64)) || (rc = cx->h_tasks.reserve(sizeof(SeedQ) * (size_t)nq)) ||
        (rc = pl->d_reads.reserve(bytes + 64)) || (rc = pl->d_q.reserve(sizeof(SeedQ) * (size_t)nq)) ||
        (rc = pl->d_tot.reserve(((8 * ((size_t)nq + 1) + 255) & ~(size_t)255) + 4 * 4096 + 64)) || (rc = pl->d_base.reserve(8 * ((size_t)nq + 1) + 64)))
        return rc;
    uint8_t* hp = cx->h_in.as<uint8_t>();
    SeedQ* hq = cx->h_tasks.as<SeedQ>();
    memset(hp, 0, PADB);
    bool magic_ok = true;       // every read short enough for the exact multiply-high prefix mapping (< 65536 prefixes)
    int64_t total_pref = 0, scr_ints = 0;
    int max_w = 0;
    // offsets first (a running sum), then the descriptors and the read bytes -- the bulk of a call's host time (about 50 ns per
    // read strand) -- in slices on a few threads when the call is large: one caller packing 200 000 strands is the case the
    // PCIe-inclusive figure measures; the mapper's workers, which already pack their own slices side by side, stay below it
    {
        size_t po = PADB;
        for (int i = 0; i < nq; ++i) { hq[i].poff = (int64_t)po; po += (size_t)q[i].plen; }
    }
    struct Part { int64_t total_pref = 0; int max_w = 0; bool magic_ok = true; int err = 0; };
    auto pack_range = [&](int lo, int hi, Part& P) {
        int c_plen = -1, c_min = -1, c_sq = -1;
        int64_t c_np = 0;
        int32_t c_nh = 0, c_kh = 0;
        uint32_t c_m1 = 0, c_m2 = 0;
        bool c_magic = true;
        for (int i = lo; i < hi; ++i) {
            SeedQ& s = hq[i];
            const size_t po = (size_t)s.poff;
            s.plen = q[i].plen; s.min_len = q[i].min_len; s.max_count = q[i].max_count; s.sparse_q = q[i].sparse_q;
            // the prefix enumeration depends on (plen, min_len, sparse_q) only: reads of one length share it (a dozen 64-bit
            // divisions per strand were the larger half of the packing time)
            if (q[i].plen != c_plen || q[i].min_len != c_min || q[i].sparse_q != c_sq) {
                c_plen = q[i].plen; c_min = q[i].min_len; c_sq = q[i].sparse_q;
                c_np = smem_like ? count_prefixes_smem(ix->K, c_plen, c_min, c_sq) : count_prefixes(ix->K, c_plen, c_min, c_sq);
                // mirrors the device PrefixMap
                const int64_t stp = (int64_t)c_sq * ix->K;
                const int64_t Lm = (int64_t)c_plen - (c_min - (stp - 1));
                const int64_t nh = Lm >= 0 ? Lm / stp + 1 : 0;
                const int64_t kk = std::min<int64_t>(ix->K, Lm + 1), rr = Lm >= 0 ? Lm % stp : 0;
                c_nh = (int32_t)nh;
                c_kh = (int32_t)std::min<int64_t>(kk, rr + 1);
                c_m1 = c_m2 = 0;
                c_magic = !(c_np >= 65536 || nh >= 65536);
                if (c_np < 65536 && nh >= 2 && nh < 65536) {
                    c_m1 = (uint32_t)((0x100000000ull + (uint64_t)nh - 1) / (uint64_t)nh);
                    if (nh >= 3) c_m2 = (uint32_t)((0x100000000ull + (uint64_t)(nh - 1) - 1) / (uint64_t)(nh - 1));
                }
            }
            const int64_t np = c_np;
            if (np < 0 || np > 0x3fffffff) { P.err = 1; return; }
            s.npref = (int32_t)np;
            s.pm_nh = c_nh; s.pm_kh = c_kh; s.pm_m1 = c_m1; s.pm_m2 = c_m2;
            if (!c_magic) P.magic_ok = false;
            s.max_smem = q[i].max_smem_cont;
            s.soff = 0;
            P.max_w = std::max(P.max_w, (int)std::max<int64_t>(q[i].plen, np) + 2);
            P.total_pref += np;
            if (q[i].plen) {
                memcpy(hp + po, q[i].P, (size_t)q[i].plen);
                if (memchr(q[i].P, 0, (size_t)q[i].plen)) {
                    if (!ix->nul_sub) { P.err = 2; return; }
                    for (int j = 0; j < q[i].plen; ++j)
                        if (!hp[po + j]) hp[po + j] = ix->nul_sub;
                }
            }
        }
    };
    {
        static const int pack_threads = [] {
            const char* e = getenv("ALFA_PACK_THREADS");
            int t = e ? atoi(e) : (int)std::min<unsigned>(8, std::max<unsigned>(1, std::thread::hardware_concurrency() / 2));
            return t < 1 ? 1 : (t > 32 ? 32 : t);
        }();
        // (the mapper's slices of short reads: 32 768 strands = 8 MB per call, sixteen callers at once: one thread each; its
        // whole-batch calls of long reads: 50 000 strands = 300 MB from one caller)
        const int nt = nq >= 65536 ? std::min(pack_threads, nq / 8192)
                                   : (bytes >= ((size_t)64 << 20) && nq >= 64) ? std::min<int>(pack_threads, (int)(bytes >> 24)) : 1;
        std::vector<Part> parts((size_t)std::max(nt, 1));
        if (nt <= 1) pack_range(0, nq, parts[0]);
        else {
            std::vector<std::thread> th;
            for (int t = 1; t < nt; ++t) th.emplace_back([&, t] { pack_range((int)((int64_t)nq * t / nt), (int)((int64_t)nq * (t + 1) / nt), parts[(size_t)t]); });
            pack_range(0, (int)((int64_t)nq / nt), parts[0]);
            for (auto& x : th) x.join();
        }
        for (const Part& P : parts) {
            if (P.err == 1) return alfa::fail(ALFA_EINVAL, "seed query has too many prefixes");
            if (P.err == 2) return alfa::fail(ALFA_EINVAL, "a read contains a NUL byte and the text uses every other byte value");
            total_pref += P.total_pref; max_w = std::max(max_w, P.max_w); magic_ok = magic_ok && P.magic_ok;
        }
        if (smem_like)            // per-read tables in the global scratch: a running sum over the reads
            for (int i = 0; i < nq; ++i) {
                hq[i].soff = scr_ints;
                scr_ints += (int64_t)SMEM_TABLES * ((int)std::max<int64_t>(q[i].plen, hq[i].npref) + 2);
            }
    }
    pl->total_pref = total_pref;
    pl->max_w = max_w;
    if (smem_like) {
        const size_t pbytes = max_plen <= 48 * 1024 ? (((size_t)max_plen + 15) & ~(size_t)15) + 2 * PADB : 0;
        pl->lds_tab = pbytes + (size_t)SMEM_TABLES * 4 * (size_t)max_w + 64 <= 60 * 1024;
        if (!pl->lds_tab && (rc = pl->d_scr.reserve(4 * (size_t)scr_ints + 64))) return rc;
    }
    static const bool force_simple = getenv("ALFA_SEED_SIMPLE") != nullptr;   // A/B switch for profiling
    pl->suflink = (mem_type == ALFA_SEED_SMAM || mem_type == ALFA_SEED_MEM) && ix->has_suflink;
    pl->persistent = mem_type == ALFA_SEED_SMAM && ix->has_child && ix->has_kmer && !force_simple && magic_ok && !pl->suflink;
    if (pl->suflink && (rc = pl->d_scr.reserve(4 * (size_t)nq * (size_t)ix->K + 64))) return rc;
    // indexes whose 10-mer table is (nearly) full: the bitmap filters nothing and the index is far beyond the caches
    static const char* sorted_env = getenv("ALFA_SEED_SORTED");
    pl->sorted = pl->persistent && total_pref > 0 && total_pref < 0x7fffffffLL &&
                 (sorted_env ? atoi(sorted_env) != 0 : (ix->kbit_fill >= 0.9 && ix->n >= (1 << 21)));
    if (pl->sorted) {
        pl->persistent = false;
        const size_t tp = (size_t)total_pref;
        if ((rc = pl->d_qbase.reserve(8 * ((size_t)nq + 1))) || (rc = pl->d_keys.reserve(4 * tp)) || (rc = pl->d_keys2.reserve(4 * tp)) ||
            (rc = pl->d_vals.reserve(8 * tp)) || (rc = pl->d_vals2.reserve(8 * tp)) || (rc = pl->d_fcnt.reserve(8 * tp)) ||
            (rc = pl->d_fscan.reserve(8 * tp)) || (rc = pl->d_frec.reserve(sizeof(SortedRec) * tp)))
            return rc;
        size_t t1 = 0, t2 = 0;
        ALFA_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, t1, pl->d_keys.as<uint32_t>(), pl->d_keys2.as<uint32_t>(), pl->d_vals.as<uint64_t>(),
                                                    pl->d_vals2.as<uint64_t>(), (int)tp, 0, 21, cx->stream));
        ALFA_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, t2, pl->d_fcnt.as<int64_t>(), pl->d_fscan.as<int64_t>(), (int)tp, cx->stream));
        pl->sort_tmp = std::max(t1, t2) + 256;
        if ((rc = pl->d_sorttmp.reserve(pl->sort_tmp))) return rc;
        // per-read first flat index: staged behind the descriptors
        if ((rc = cx->h_in2.reserve(8 * ((size_t)nq + 1)))) return rc;
        int64_t* hb = cx->h_in2.as<int64_t>();
        hb[0] = 0;
        for (int i = 0; i < nq; ++i) hb[i + 1] = hb[i] + hq[i].npref;
        ALFA_HIP(hipMemcpyAsync(pl->d_qbase.p, hb, 8 * ((size_t)nq + 1), hipMemcpyHostToDevice, cx->up));
    }
    pl->nseg = 1;
    if (pl->persistent) {
        static std::atomic<int> n_cu[64];          // several worker threads may get here at once: the set-up is idempotent
        int ncu = n_cu[ix->device & 63].load();
        if (!ncu) {
            hipDeviceProp_t prop;
            ALFA_HIP(hipGetDeviceProperties(&prop, ix->device));
            ALFA_HIP(hipFuncSetAttribute((const void*)smam_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
            ncu = std::max(1, prop.multiProcessorCount);
            n_cu[ix->device & 63].store(ncu);
        }
        pl->nseg = std::max(1, std::min(ncu, (nq + PWAVES - 1) / PWAVES));
        const int room = ((160 * 1024 - KBIT_BYTES - 1024) / PWAVES - WQ_BYTES) / 2;      // LDS bytes a wave can have for its read (x2: codes)
        pl->stride = std::min(room & ~15, (max_plen + 2 * PADB + 8 + 15) & ~15);              // longer reads are read from global memory
        if ((rc = pl->d_q3.reserve(sizeof(Q3) * (size_t)pl->nseg * PWAVES * Q3_CAP + 64))) return rc;
    }
    const int64_t total_cap = std::min<int64_t>(std::max<int64_t>(4096, 4 * (int64_t)nq), std::max<int64_t>(total_pref, 1));
    pl->seg_cap = pl->nseg == 1 ? (int)total_cap : (int)std::max<int64_t>(256, 2 * total_cap / pl->nseg);
    pl->out_cap = std::max<int64_t>(4096, 16 * (int64_t)nq);
    pl->tmp_bytes = 0;
    ALFA_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, pl->tmp_bytes, pl->d_tot.as<int64_t>(), pl->d_base.as<int64_t>(), nq + 1, cx->stream));
    if ((rc = pl->d_work.reserve(pl->work_rec * (size_t)pl->nseg * (size_t)pl->seg_cap + 64)) ||
        (rc = pl->d_out.reserve(sizeof(alfa_match) * (size_t)pl->out_cap)) || (rc = pl->d_tmp.reserve(pl->tmp_bytes + 64)))
        return rc;
    // on the upload stream: the host waits for these copies only, not for kernels of other calls still running on cx->stream
    ALFA_HIP(hipMemcpyAsync(pl->d_reads.p, hp, bytes, hipMemcpyHostToDevice, cx->up));
    ALFA_HIP(hipMemcpyAsync(pl->d_q.p, hq, sizeof(SeedQ) * (size_t)nq, hipMemcpyHostToDevice, cx->up));
    ALFA_HIP(hipStreamSynchronize(cx->up));
    return ALFA_OK;
}

int plan_launch(alfa_seed_plan* pl, alfa::Ctx* cx) {
    pl->launched = true;
    if (pl->nq == 0) return ALFA_OK;
    hipStream_t st = cx->stream;
    const DIdx X = make_didx(pl->ix);
    const int nq = pl->nq;
    const int lds_p = pl->max_plen <= 48 * 1024;
    const size_t lds_bytes = lds_p ? (size_t)pl->max_plen + 2 * PADB + 16 : 0;
    // the work-list counters (one per segment: at most one per CU) sit behind the per-read totals: one memset clears both
    if (pl->nseg > 4096) return alfa::fail(ALFA_EINVAL, "more work-list segments than the plan has counters for");
    int* d_cnt = plan_cnt(pl);
    int64_t* d_tot = pl->d_tot.as<int64_t>();
    int64_t* d_base = pl->d_base.as<int64_t>();
    const SeedQ* d_q = pl->d_q.as<SeedQ>();
    const uint8_t* d_reads = pl->d_reads.as<uint8_t>();
    ALFA_HIP(hipMemsetAsync(d_tot, 0, (size_t)((char*)d_cnt - (char*)d_tot) + 4 * (size_t)pl->nseg, st));       // the persistent kernel accumulates per-read totals
    ALFA_HIP(hipEventRecord(pl->ev0, st));
    if (pl->sorted) {
        const int64_t tp = pl->total_pref;
        const int64_t* d_qbase = pl->d_qbase.as<int64_t>();
        hipLaunchKernelGGL(sorted_keys_kernel, dim3(nq), dim3(256), 0, st, X, d_q, d_reads, d_qbase, pl->d_keys.as<uint32_t>(), pl->d_vals.as<uint64_t>());
        size_t tb = pl->sort_tmp;
        ALFA_HIP(hipcub::DeviceRadixSort::SortPairs(pl->d_sorttmp.p, tb, pl->d_keys.as<uint32_t>(), pl->d_keys2.as<uint32_t>(), pl->d_vals.as<uint64_t>(),
                                                    pl->d_vals2.as<uint64_t>(), (int)tp, 0, 21, st));
        hipLaunchKernelGGL(sorted_eval_kernel, dim3((unsigned)((tp + 255) / 256)), dim3(256), 0, st, X, d_q, d_reads, d_qbase, pl->d_keys2.as<uint32_t>(),
                           pl->d_vals2.as<uint64_t>(), tp, pl->d_fcnt.as<int64_t>(), pl->d_frec.as<SortedRec>());
        ALFA_HIP(hipGetLastError());
        ALFA_HIP(hipEventRecord(pl->ev1, st));
        tb = pl->sort_tmp;
        ALFA_HIP(hipcub::DeviceScan::ExclusiveSum(pl->d_sorttmp.p, tb, pl->d_fcnt.as<int64_t>(), pl->d_fscan.as<int64_t>(), (int)tp, st));
        hipLaunchKernelGGL(sorted_base_kernel, dim3((nq + 1 + 255) / 256), dim3(256), 0, st, pl->d_fscan.as<int64_t>(), pl->d_fcnt.as<int64_t>(), d_qbase, nq, tp,
                           d_base);
        hipLaunchKernelGGL(sorted_emit_kernel, dim3((unsigned)((tp + 255) / 256)), dim3(256), 0, st, X, d_q, d_reads, pl->d_fscan.as<int64_t>(),
                           pl->d_fcnt.as<int64_t>(), pl->d_frec.as<SortedRec>(), tp, (const int64_t*)d_base, nq, pl->d_out.as<alfa_match>(), pl->out_cap);
        ALFA_HIP(hipGetLastError());
        return ALFA_OK;
    }
    if (pl->suflink) {
        if (pl->mem_type == ALFA_SEED_SMAM)
            hipLaunchKernelGGL((smam_suflink_kernel<ALFA_SEED_SMAM, false>), dim3(nq), dim3(SEED_BLOCK), lds_bytes, st, X, d_q, d_reads, d_tot,
                               pl->d_scr.as<int32_t>(), (const int64_t*)nullptr, (alfa_match*)nullptr, (int64_t)0, nq, lds_p);
        else
            hipLaunchKernelGGL((smam_suflink_kernel<ALFA_SEED_MEM, false>), dim3(nq), dim3(SEED_BLOCK), lds_bytes, st, X, d_q, d_reads, d_tot,
                               pl->d_scr.as<int32_t>(), (const int64_t*)nullptr, (alfa_match*)nullptr, (int64_t)0, nq, lds_p);
    } else if (pl->persistent) {
        // one workgroup per CU, the 10-mer bitmap + one read slice per wave in LDS
        const size_t lds = (size_t)KBIT_BYTES + (size_t)PWAVES * (2 * pl->stride + WQ_BYTES);      // read bytes + 2-bit codes + survivor queue per wave
        hipLaunchKernelGGL(smam_count_kernel, dim3(pl->nseg), dim3(PBLOCK), lds, st, X, d_q, d_reads, (unsigned long long*)d_tot,
                           pl->d_work.as<WorkRec>(), d_cnt, pl->seg_cap, nq, pl->stride, pl->d_q3.as<Q3>());
    } else if (pl->mem_type == ALFA_SEED_SMAM) {
        hipLaunchKernelGGL(smam_count_simple_kernel, dim3(nq), dim3(SEED_BLOCK), lds_bytes, st, X, d_q, d_reads, d_tot, pl->d_work.as<WorkRec>(),
                           d_cnt, pl->seg_cap, lds_p);
    } else if (pl->mem_type == ALFA_SEED_MEM) {
        hipLaunchKernelGGL(mem_count_kernel, dim3(nq), dim3(SEED_BLOCK), lds_bytes, st, X, d_q, d_reads, d_tot, pl->d_work.as<WorkRec2>(), d_cnt,
                           pl->seg_cap, lds_p);
    } else {
        const size_t pbytes = lds_p ? (((size_t)pl->max_plen + 15) & ~(size_t)15) + 2 * PADB : 0;
        const size_t lds2 = pbytes + (pl->lds_tab ? (size_t)SMEM_TABLES * 4 * (size_t)pl->max_w : 0) + 16;
        if (pl->mem_type == ALFA_SEED_SMEMP) hipLaunchKernelGGL(smemp_count_kernel, dim3(nq), dim3(SEED_BLOCK), lds2, st, X, d_q, d_reads, d_tot, pl->d_work.as<WorkRec2>(), d_cnt,
                           pl->seg_cap, lds_p, pl->lds_tab ? 1 : 0, pl->d_scr.as<int>());
        else hipLaunchKernelGGL(smem_count_kernel, dim3(nq), dim3(SEED_BLOCK), lds2, st, X, d_q, d_reads, d_tot, pl->d_work.as<WorkRec2>(), d_cnt,
                           pl->seg_cap, lds_p, pl->lds_tab ? 1 : 0, pl->d_scr.as<int>());
    }
    ALFA_HIP(hipGetLastError());
    ALFA_HIP(hipEventRecord(pl->ev1, st));
    size_t tb = pl->tmp_bytes;
    ALFA_HIP(hipcub::DeviceScan::ExclusiveSum(pl->d_tmp.p, tb, d_tot, d_base, nq + 1, st));
    const dim3 eg((pl->seg_cap + 127) / 128, pl->nseg), eb(128);
    if (pl->suflink && pl->mem_type == ALFA_SEED_SMAM)
        hipLaunchKernelGGL((smam_suflink_kernel<ALFA_SEED_SMAM, true>), dim3(nq), dim3(SEED_BLOCK), lds_bytes, st, X, d_q, d_reads, d_tot,
                           pl->d_scr.as<int32_t>(), (const int64_t*)d_base, pl->d_out.as<alfa_match>(), pl->out_cap, nq, lds_p);
    else if (pl->suflink)
        hipLaunchKernelGGL((smam_suflink_kernel<ALFA_SEED_MEM, true>), dim3(nq), dim3(SEED_BLOCK), lds_bytes, st, X, d_q, d_reads, d_tot,
                           pl->d_scr.as<int32_t>(), (const int64_t*)d_base, pl->d_out.as<alfa_match>(), pl->out_cap, nq, lds_p);
    else if (pl->mem_type == ALFA_SEED_SMAM)
        hipLaunchKernelGGL(smam_emit_kernel, eg, eb, 0, st, X, d_q, d_reads, d_base, nq, pl->d_work.as<WorkRec>(), d_cnt, pl->seg_cap,
                           pl->d_out.as<alfa_match>(), pl->out_cap);
    else if (pl->mem_type == ALFA_SEED_MEM)
        hipLaunchKernelGGL(emit2_kernel<ALFA_SEED_MEM>, eg, eb, 0, st, X, d_q, d_reads, d_base, nq, pl->d_work.as<WorkRec2>(), d_cnt, pl->seg_cap,
                           pl->d_out.as<alfa_match>(), pl->out_cap);
    else if (pl->mem_type == ALFA_SEED_SMEMP)
        hipLaunchKernelGGL(emit2_kernel<ALFA_SEED_SMEMP>, eg, eb, 0, st, X, d_q, d_reads, d_base, nq, pl->d_work.as<WorkRec2>(), d_cnt, pl->seg_cap,
                           pl->d_out.as<alfa_match>(), pl->out_cap);
    else
        hipLaunchKernelGGL(emit2_kernel<ALFA_SEED_SMEM>, eg, eb, 0, st, X, d_q, d_reads, d_base, nq, pl->d_work.as<WorkRec2>(), d_cnt, pl->seg_cap,
                           pl->d_out.as<alfa_match>(), pl->out_cap);
    ALFA_HIP(hipGetLastError());
    return ALFA_OK;
}

// waits for the launch, reads the per-query offsets back and, if the work list or the match array overflowed, grows them and
// launches again: afterwards the matches of query i are d_out[offsets[i] .. offsets[i+1]) on the device
int plan_settle(alfa_seed_plan* pl, alfa::Ctx* cx, int64_t* offsets) {
    const int nq = pl->nq;
    hipStream_t st = cx->stream;
    int rc;
    for (int attempt = 0;; ++attempt) {
        std::vector<int> segc((size_t)pl->nseg);
        ALFA_HIP(hipMemcpyAsync(segc.data(), plan_cnt(pl), 4 * (size_t)pl->nseg, hipMemcpyDeviceToHost, st));
        ALFA_HIP(hipMemcpyAsync(offsets, pl->d_base.p, 8 * ((size_t)nq + 1), hipMemcpyDeviceToHost, st));
        ALFA_HIP(hipStreamSynchronize(st));
        const int64_t total = offsets[nq];
        const int nwork = *std::max_element(segc.begin(), segc.end());
        if (nwork <= pl->seg_cap && total <= pl->out_cap) break;
        if (attempt >= 2) return alfa::fail(ALFA_EHIP, "seed work list kept overflowing");
        // grow to the need reported by the device and run again
        if (nwork > pl->seg_cap) {
            pl->seg_cap = pl->nseg == 1 ? nwork : nwork + nwork / 8 + 64;
            if ((rc = pl->d_work.reserve(pl->work_rec * (size_t)pl->nseg * (size_t)pl->seg_cap + 64))) return rc;
        }
        if (total > pl->out_cap) {
            pl->out_cap = total;
            if ((rc = pl->d_out.reserve(sizeof(alfa_match) * (size_t)pl->out_cap))) return rc;
        }
        if ((rc = plan_launch(pl, cx))) return rc;
    }
    return ALFA_OK;
}

int plan_fetch(alfa_seed_plan* pl, alfa::Ctx* cx, alfa_match** matches, int64_t* offsets) {
    if (!pl->launched) return alfa::fail(ALFA_EINVAL, "fetch before launch");
    const int nq = pl->nq;
    offsets[0] = 0;
    if (nq == 0) { *matches = (alfa_match*)malloc(sizeof(alfa_match)); return ALFA_OK; }
    hipStream_t st = cx->stream;
    int rc;
    if ((rc = plan_settle(pl, cx, offsets))) return rc;
    const int64_t total = offsets[nq];
    alfa_match* hm = (alfa_match*)malloc(sizeof(alfa_match) * (size_t)(total + 1));
    if (!hm) return alfa::fail(ALFA_ENOMEM, "out of host memory for the match array");
    if (total > 0) {
        hipError_t e = hipMemcpyAsync(hm, pl->d_out.p, sizeof(alfa_match) * (size_t)total, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { free(hm); return alfa::fail(ALFA_EHIP, hipGetErrorString(e)); }
    }
    *matches = hm;
    return ALFA_OK;
}

void plan_release(alfa_seed_plan* pl) {
    pl->d_reads.release(); pl->d_q.release(); pl->d_tot.release(); pl->d_base.release(); pl->d_work.release(); pl->d_out.release();
    pl->d_tmp.release(); pl->d_scr.release(); pl->d_cnt.release(); pl->d_q3.release();
    pl->d_reg.release(); pl->d_rm.release(); pl->d_chain.release(); pl->d_aux.release();
    pl->d_marena.release();
    pl->h_mint.release(); pl->h_mstage.release();
    if (pl->mev0) { (void)hipEventDestroy(pl->mev0); (void)hipEventDestroy(pl->mev1); pl->mev0 = pl->mev1 = nullptr; }
    for (alfa::DevBuf* b : {&pl->d_qbase, &pl->d_keys, &pl->d_keys2, &pl->d_vals, &pl->d_vals2, &pl->d_fcnt, &pl->d_fscan, &pl->d_frec, &pl->d_sorttmp}) b->release();
    if (pl->ev0) { (void)hipEventDestroy(pl->ev0); (void)hipEventDestroy(pl->ev1); pl->ev0 = pl->ev1 = nullptr; }
}

// the calling thread's recycled plans (slot 0: alfa_seed_batch, 1: alfa_seed_regions_batch, 2: alfa_map_reads_batch); buried when the thread ends
// (device_ctx.h: a live thread releases them)
alfa_seed_plan* thread_plan(int slot) {
    struct Reuse {
        alfa_seed_plan* p[3] = {nullptr, nullptr, nullptr};
        ~Reuse() {
            for (auto*& x : p)
                if (x) {
                    x->ix = nullptr;          // the index may be freed before the plan is released
                    alfa::bury([](void* q) { alfa_seed_plan* d = (alfa_seed_plan*)q; (void)hipSetDevice(d->dev); plan_release(d); delete d; }, x);
                    x = nullptr;
                }
        }
    };
    thread_local Reuse reuse;
    if (!reuse.p[slot]) reuse.p[slot] = new alfa_seed_plan();
    return reuse.p[slot];
}

}  // namespace

extern "C" {

int alfa_seed_plan_create(alfa_index* ix, int mem_type, const alfa_seed_query* q, int nq, alfa_seed_plan** out) {
    if (!ix || !out || nq < 0 || (nq > 0 && !q)) return alfa::fail(ALFA_EINVAL, "null argument");
    alfa::Ctx* cx = alfa::thread_ctx(ix->device);
    if (!cx) return ALFA_ENODEV;
    alfa_seed_plan* pl = new alfa_seed_plan();
    int rc = plan_fill(pl, cx, ix, mem_type, q, nq);
    if (rc) { plan_release(pl); delete pl; return rc; }
    pl->owner = cx;
    *out = pl;
    return ALFA_OK;
}
int alfa_seed_plan_launch(alfa_seed_plan* pl) {
    if (!pl) return alfa::fail(ALFA_EINVAL, "null plan");
    alfa::Ctx* cx = alfa::thread_ctx(pl->ix->device);
    if (!cx) return ALFA_ENODEV;
    if (cx != pl->owner) return alfa::fail(ALFA_EINVAL, "a resident seed plan must be launched and fetched by the host thread that created it");
    return plan_launch(pl, cx);
}
int alfa_seed_plan_fetch(alfa_seed_plan* pl, alfa_match** matches, int64_t* offsets) {
    if (!pl || !matches || !offsets) return alfa::fail(ALFA_EINVAL, "null argument");
    alfa::Ctx* cx = alfa::thread_ctx(pl->ix->device);
    if (!cx) return ALFA_ENODEV;
    if (cx != pl->owner) return alfa::fail(ALFA_EINVAL, "a resident seed plan must be launched and fetched by the host thread that created it");
    return plan_fetch(pl, cx, matches, offsets);
}
int64_t alfa_seed_plan_queries(const alfa_seed_plan* pl) { return pl ? pl->total_pref : 0; }
int alfa_seed_plan_sorted(const alfa_seed_plan* pl) { return pl && pl->sorted ? 1 : 0; }
double alfa_seed_plan_kernel_ms(const alfa_seed_plan* pl) {
    if (!pl || !pl->launched || pl->nq == 0) return 0;
    float ms = 0;
    if (hipEventElapsedTime(&ms, pl->ev0, pl->ev1) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return ms;
}
void alfa_seed_plan_free(alfa_seed_plan* pl) {
    if (!pl) return;
    (void)hipSetDevice(pl->ix->device);
    plan_release(pl);
    delete pl;
}

int alfa_seed_batch(alfa_index* ix, int mem_type, const alfa_seed_query* q, int nq, alfa_match** matches, int64_t* offsets) {
    alfa::g_last_timing = alfa::Timing();
    if (!ix || !matches || !offsets || nq < 0 || (nq > 0 && !q)) return alfa::fail(ALFA_EINVAL, "null argument");
    alfa::Ctx* cx = alfa::thread_ctx(ix->device);
    if (!cx) return ALFA_ENODEV;
    alfa_seed_plan* reuse = thread_plan(0);         // buffers are recycled from call to call
    int rc;
    if ((rc = plan_fill(reuse, cx, ix, mem_type, q, nq)) || (rc = plan_launch(reuse, cx)) || (rc = plan_fetch(reuse, cx, matches, offsets))) return rc;
    alfa::g_last_timing.kernel_ms = alfa_seed_plan_kernel_ms(reuse);
    alfa::g_last_timing.work_units = reuse->total_pref;
    alfa::g_last_timing.launches = 1;
    return ALFA_OK;
}

// seeds, then identifyAndFilter + coveredBases on the matches where they lie (region_kernels.hip); see include/alfa_hot.h
int alfa_seed_regions_batch(alfa_index* ix, int mem_type, const alfa_seed_query* q, const int32_t* edit_dist, int nq, const alfa_region_params* p,
                            alfa_match** matches, int64_t* match_off, alfa_region** regions, int64_t* region_off, alfa_match** members,
                            int64_t* member_off, int32_t** chains, int32_t* status) {
    alfa::g_last_timing = alfa::Timing();
    if (!ix || !p || !matches || !match_off || !regions || !region_off || !members || !member_off || !chains || nq < 0 ||
        (nq > 0 && (!q || !edit_dist || !status)))
        return alfa::fail(ALFA_EINVAL, "null argument");
    alfa::Ctx* cx = alfa::thread_ctx(ix->device);
    if (!cx) return ALFA_ENODEV;
    alfa_seed_plan* pl = thread_plan(1);
    int rc;
    *matches = nullptr; *regions = nullptr; *members = nullptr; *chains = nullptr;
    if ((rc = plan_fill(pl, cx, ix, mem_type, q, nq)) || (rc = plan_launch(pl, cx)) || (rc = plan_fetch(pl, cx, matches, match_off))) return rc;
    region_off[0] = member_off[0] = 0;
    *regions = (alfa_region*)malloc(sizeof(alfa_region));
    *members = (alfa_match*)malloc(sizeof(alfa_match));
    *chains = (int32_t*)malloc(sizeof(int32_t));
    if (!*regions || !*members || !*chains) return alfa::fail(ALFA_ENOMEM, "out of host memory");
    if (nq == 0) return ALFA_OK;
    const int64_t total = match_off[nq];
    const size_t rm_slots = 2 * (size_t)total + 8 * (size_t)nq;
    hipStream_t st = cx->stream;
    if ((rc = pl->d_reg.reserve(sizeof(alfa_region) * (size_t)(total + 1))) || (rc = pl->d_rm.reserve(sizeof(alfa_match) * (rm_slots + 1))) || (rc = pl->d_chain.reserve(4 * (rm_slots + 1))) ||
        (rc = pl->d_aux.reserve(4 * 5 * (size_t)nq + 64)) || (rc = cx->h_tasks.reserve(4 * 5 * (size_t)nq + 64)))
        return rc;
    int32_t* h_aux = cx->h_tasks.as<int32_t>();                 // [qlen | edit | nreg | nrm | status]
    for (int i = 0; i < nq; ++i) { h_aux[i] = q[i].plen; h_aux[nq + i] = edit_dist[i]; }
    int32_t* d_aux = pl->d_aux.as<int32_t>();
    ALFA_HIP(hipMemcpyAsync(d_aux, h_aux, 4 * 2 * (size_t)nq, hipMemcpyHostToDevice, st));
    alfa::RegionParams rp;
    rp.max_smem_occ = p->max_seed_candidates < 0 ? INT64_MAX : p->max_seed_candidates;      // the reference holds them in size_t
    rp.max_region_mems = p->max_region_mems < 0 ? INT64_MAX : p->max_region_mems;
    rp.min_mem_reg_length = p->min_mem_reg_length;
    rp.error_percent = p->error_percent;
    alfa::launch_regions(st, pl->d_out.as<alfa_match>(), pl->d_base.as<int64_t>(), d_aux, d_aux + nq, nq, rp, pl->d_reg.as<alfa_region>(),
                         pl->d_rm.as<alfa_match>(), pl->d_chain.as<int32_t>(), d_aux + 2 * (size_t)nq, d_aux + 3 * (size_t)nq, d_aux + 4 * (size_t)nq);
    ALFA_HIP(hipGetLastError());
    ALFA_HIP(hipMemcpyAsync(h_aux + 2 * (size_t)nq, d_aux + 2 * (size_t)nq, 4 * 3 * (size_t)nq, hipMemcpyDeviceToHost, st));
    ALFA_HIP(hipStreamSynchronize(st));
    const int32_t *h_nreg = h_aux + 2 * (size_t)nq, *h_nrm = h_aux + 3 * (size_t)nq, *h_st = h_aux + 4 * (size_t)nq;
    for (int i = 0; i < nq; ++i) {
        region_off[i + 1] = region_off[i] + h_nreg[i];
        member_off[i + 1] = member_off[i] + h_nrm[i];
        status[i] = h_st[i];
    }
    const int64_t nreg_total = region_off[nq], nrm_total = member_off[nq];
    if (nreg_total == 0) return ALFA_OK;
    // the device arrays are slotted per strand (worst-case sizes); they come back whole and are compacted here
    alfa_region* hr_all = (alfa_region*)malloc(sizeof(alfa_region) * (size_t)(total + 1));
    alfa_match* hm_all = (alfa_match*)malloc(sizeof(alfa_match) * (rm_slots + 1));
    int32_t* hc_all = (int32_t*)malloc(4 * (rm_slots + 1));
    int32_t* hc = (int32_t*)realloc(*chains, 4 * (size_t)(nrm_total + 1));
    if (hc) *chains = hc;
    alfa_region* hr = (alfa_region*)realloc(*regions, sizeof(alfa_region) * (size_t)(nreg_total + 1));
    alfa_match* hm = (alfa_match*)realloc(*members, sizeof(alfa_match) * (size_t)(nrm_total + 1));
    if (hr) *regions = hr;
    if (hm) *members = hm;
    if (!hr_all || !hm_all || !hc_all || !hr || !hm || !hc) { free(hr_all); free(hm_all); free(hc_all); return alfa::fail(ALFA_ENOMEM, "out of host memory for the region arrays"); }
    hipError_t e = hipMemcpyAsync(hr_all, pl->d_reg.p, sizeof(alfa_region) * (size_t)total, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(hm_all, pl->d_rm.p, sizeof(alfa_match) * rm_slots, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(hc_all, pl->d_chain.p, 4 * rm_slots, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { free(hr_all); free(hm_all); free(hc_all); return alfa::fail(ALFA_EHIP, hipGetErrorString(e)); }
    for (int i = 0; i < nq; ++i) {
        if (h_nreg[i] > 0) memcpy(hr + region_off[i], hr_all + match_off[i], sizeof(alfa_region) * (size_t)h_nreg[i]);
        if (h_nrm[i] > 0) memcpy(hm + member_off[i], hm_all + 2 * match_off[i] + 8 * (int64_t)i, sizeof(alfa_match) * (size_t)h_nrm[i]);
        if (h_nrm[i] > 0) memcpy(hc + member_off[i], hc_all + 2 * match_off[i] + 8 * (int64_t)i, 4 * (size_t)h_nrm[i]);
    }
    free(hr_all); free(hm_all); free(hc_all);
    alfa::g_last_timing.kernel_ms = alfa_seed_plan_kernel_ms(pl);
    alfa::g_last_timing.work_units = pl->total_pref;
    alfa::g_last_timing.launches = 1;
    return ALFA_OK;
}

// whole reads: seeds and regions of both strands, then the trial loop with its DP on the device (map_kernels.hip); see
// include/alfa_hot.h
int alfa_map_reads_batch(alfa_index* ix, const alfa_map_options* o, const alfa_dp_params* dp, const alfa_map_read* reads, int n,
                         int32_t* status, int64_t* aln_off, alfa_map_alignment** alns, int64_t* seed_off, alfa_match** seeds,
                         int64_t* op_off, alfa_map_cigar_op** ops) {
    alfa::g_last_timing = alfa::Timing();
    if (!ix || !o || !dp || !aln_off || !alns || !seed_off || !seeds || !op_off || !ops || n < 0 || (n > 0 && (!reads || !status)))
        return alfa::fail(ALFA_EINVAL, "null argument");
    if (dp->m < 1 || dp->m > 8 || dp->gapo < 0 || dp->gape < 0) return alfa::fail(ALFA_EINVAL, "bad scoring parameters");
    if (n > (1 << 24)) return alfa::fail(ALFA_EINVAL, "more than 2^24 reads in one call: split the batch");
    alfa::Ctx* cx = alfa::thread_ctx(ix->device);
    if (!cx) return ALFA_ENODEV;
    static_assert(sizeof(alfa_map_alignment) == sizeof(alfa::mapcore::AlnOut), "alfa_map_alignment == mapcore::AlnOut");
    static_assert(sizeof(alfa_map_cigar_op) == sizeof(alfa::mapcore::CigOp), "alfa_map_cigar_op == mapcore::CigOp");
    *alns = (alfa_map_alignment*)malloc(sizeof(alfa_map_alignment));
    *seeds = (alfa_match*)malloc(sizeof(alfa_match));
    *ops = (alfa_map_cigar_op*)malloc(sizeof(alfa_map_cigar_op));
    if (!*alns || !*seeds || !*ops) return alfa::fail(ALFA_ENOMEM, "out of host memory");
    aln_off[0] = seed_off[0] = op_off[0] = 0;
    if (n == 0) return ALFA_OK;
    if (ix->n_seq <= 0) return alfa::fail(ALFA_EINVAL, "the index was uploaded without its sequence start offsets");
    alfa_seed_plan* pl = thread_plan(2);
    const int nq = 2 * n;
    int rc;
    static const bool call_log = getenv("ALFA_CALL_LOG") != nullptr;      // diagnostics: where the call's time goes
    const auto tc0 = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    int max_plen = 0, max_edit = 0;
    std::vector<alfa_seed_query> q((size_t)nq);
    for (int i = 0; i < n; ++i) {
        const alfa_map_read& r = reads[i];
        if (r.plen < 0 || (r.plen > 0 && (!r.P || !r.Prc))) return alfa::fail(ALFA_EINVAL, "bad read");
        q[2 * i] = alfa_seed_query{r.P, r.plen, r.min_len, o->max_seed_candidates, r.sparse_q, o->max_smem_start};
        q[2 * i + 1] = alfa_seed_query{r.Prc, r.plen, r.min_len, o->max_seed_candidates, r.sparse_q, o->max_smem_start};
        max_plen = std::max(max_plen, r.plen);
        max_edit = std::max(max_edit, r.edit_dist);
    }
    if ((rc = plan_fill(pl, cx, ix, o->mem_type, q.data(), nq))) return rc;
    std::vector<int64_t> h_off((size_t)nq + 1);      // strand offsets in the packed read buffer (plan_fill's descriptors are still in h_tasks)
    {
        const SeedQ* hq = cx->h_tasks.as<SeedQ>();
        for (int i = 0; i < nq; ++i) h_off[i] = hq[i].poff;
    }
    const double ms_fill = since(tc0);
    if ((rc = plan_launch(pl, cx))) return rc;
    std::vector<int64_t> moff((size_t)nq + 1);
    moff[0] = 0;
    if ((rc = plan_settle(pl, cx, moff.data()))) return rc;
    const double ms_seeds = since(tc0) - ms_fill;
    const int64_t total = moff[nq];
    const size_t rm_slots = 2 * (size_t)total + 8 * (size_t)nq;
    hipStream_t st = cx->stream;
    // ---- regions of every strand (regions_kernel)
    if ((rc = pl->d_reg.reserve(sizeof(alfa_region) * (size_t)(total + 1))) || (rc = pl->d_rm.reserve(sizeof(alfa_match) * (rm_slots + 1))) ||
        (rc = pl->d_chain.reserve(4 * (rm_slots + 1))) || (rc = pl->d_aux.reserve(4 * 5 * (size_t)nq + 64)) || (rc = cx->h_tasks.reserve(4 * 5 * (size_t)nq + 64)))
        return rc;
    int32_t* h_aux = cx->h_tasks.as<int32_t>();                 // [qlen | edit | nreg | nrm | status]
    for (int i = 0; i < nq; ++i) { const alfa_map_read& r = reads[i >> 1]; h_aux[i] = r.plen; h_aux[nq + i] = r.region_edit_dist > 0 ? r.region_edit_dist : r.edit_dist; }
    int32_t* d_aux = pl->d_aux.as<int32_t>();
    ALFA_HIP(hipMemcpyAsync(d_aux, h_aux, 4 * 2 * (size_t)nq, hipMemcpyHostToDevice, st));
    alfa::RegionParams rp;
    rp.max_smem_occ = o->max_seed_candidates < 0 ? INT64_MAX : o->max_seed_candidates;
    rp.max_region_mems = o->max_region_mems < 0 ? INT64_MAX : o->max_region_mems;
    rp.min_mem_reg_length = o->min_mem_reg_length;
    rp.error_percent = o->error_percent;
    alfa::launch_regions(st, pl->d_out.as<alfa_match>(), pl->d_base.as<int64_t>(), d_aux, d_aux + nq, nq, rp, pl->d_reg.as<alfa_region>(),
                         pl->d_rm.as<alfa_match>(), pl->d_chain.as<int32_t>(), d_aux + 2 * (size_t)nq, d_aux + 3 * (size_t)nq, d_aux + 4 * (size_t)nq);
    ALFA_HIP(hipGetLastError());
    // ---- the trial loop, one wavefront per read (map_kernel)
    using alfa::mapcore::ALN_CAP;
    using alfa::mapcore::CHAINOUT_CAP;
    using alfa::mapcore::OPS_CAP;
    const int qcap = (max_plen + 15) & ~15;
    const int64_t stride64 = (16 + (int64_t)qcap + 16 + (int64_t)max_plen + max_edit + 64 + 255) & ~(int64_t)255;
    if (stride64 > (1 << 30)) return alfa::fail(ALFA_EINVAL, "read too long for the mapping kernel's scratch");
    const size_t nints = (size_t)n + 6 * ((size_t)n + 1) + 16;        // (+ 16 words of diagnostics: A.reasons)              // status[n] | naln | nseed | nops | aln_off | seed_off | op_off, n+1 each (int32)
    // scratch of a read in flight, per workgroup of the launch: its staged strings, the direction bytes of its ksw_global call
    // (the whole read at 256 bytes per row: every piece fits) and the BAM operations of that call's walk.  The launch has as
    // many workgroups as the GPU keeps resident -- fewer when the slots would exceed 12 GiB per call (reads beyond
    // ~10 kbp; ALFA_MAP_ZSCRATCH_GIB changes the bound)
    const int64_t zworst = 256 * ((int64_t)max_plen + max_edit + 8);
    const int bam_cap = max_plen + max_edit + 8;
    const int64_t zstride = o->cigar_alignments > 0 ? zworst : 0;
    const int64_t zslot = o->cigar_alignments > 0 ? ((zstride + 4 * (int64_t)bam_cap + 255) & ~(int64_t)255) : 0;
    int slots = std::min(n, alfa::map_resident_workgroups());
    static const int64_t zscratch_gib = [] { const char* e = getenv("ALFA_MAP_ZSCRATCH_GIB"); const int v = e ? atoi(e) : 12; return (int64_t)(v < 1 ? 1 : v > 64 ? 64 : v); }();
    if (zslot > 0) slots = (int)std::max<int64_t>(1, std::min<int64_t>(slots, std::max<int64_t>(256, (zscratch_gib << 30) / zslot)));
    size_t tmp_bytes = 0;
    ALFA_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, (int32_t*)nullptr, (int32_t*)nullptr, n + 1, st));
    // ONE device arena and ONE pinned arena per calling thread for everything below (each hipMalloc / hipHostMalloc is a trip
    // through the runtime's allocator lock, and sixteen worker threads make their first calls at the same moment)
    size_t dsz = 0;
    auto carve = [&](size_t bytes) { const size_t at = dsz; dsz += (bytes + 255) & ~(size_t)255; return at; };
    // reads of very different lengths (long-read data): the longest first, so that what the workgroups take last is short --
    // a call ends when its last read does
    const bool by_length = max_plen >= 1000;
    // the call's parameters -- per-read thresholds, scoring, strand offsets, the order -- have one layout on both sides and go up in ONE copy
    auto al256 = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t r_prm = al256(sizeof(alfa::mapcore::ReadParams) * (size_t)n), r_off = r_prm + al256(sizeof(DevParams));
    const size_t r_ord = r_off + al256(8 * ((size_t)nq + 1)), par_bytes = r_ord + al256(by_length ? 4 * (size_t)n : 0);
    const size_t o_par = carve(par_bytes), o_rp = o_par, o_prm = o_par + r_prm, o_off = o_par + r_off, o_ord = o_par + r_ord;
    const size_t o_int = carve(4 * (nints + 8)), o_tmp = carve(tmp_bytes + 64);
    const size_t o_aln = carve(sizeof(alfa::mapcore::AlnOut) * (size_t)n * ALN_CAP), o_aln2 = carve(sizeof(alfa::mapcore::AlnOut) * (size_t)n * ALN_CAP);
    const size_t o_seed = carve(sizeof(alfa_match) * (size_t)n * CHAINOUT_CAP), o_seed2 = carve(sizeof(alfa_match) * (size_t)n * CHAINOUT_CAP);
    // a read's extended CIGAR: at 10 % error about one operation per five residues per reported alignment; room for one per two
    // residues (more: op_count = -1 for that alignment, the caller recovers it), within 2 GiB per call (x 2: slotted and dense)
    const int ops_cap = o->cigar_alignments > 0
                            ? (int)std::max<int64_t>(OPS_CAP, std::min<int64_t>((int64_t)max_plen / 2 + 64, ((int64_t)2 << 30) / (8 * (int64_t)n)))
                            : OPS_CAP;
    const size_t o_ops = carve(sizeof(alfa::mapcore::CigOp) * (size_t)n * (size_t)ops_cap), o_ops2 = carve(sizeof(alfa::mapcore::CigOp) * (size_t)n * (size_t)ops_cap);
    const size_t zbytes = (size_t)zslot * (size_t)slots + 64;
    const bool zshared = zbytes > ((size_t)1 << 30);           // long reads: one of the index's shared buffers (alfa_index::zbuf)
    const size_t o_scr = carve((size_t)stride64 * (size_t)slots + 64), o_z = carve(zshared ? 64 : zbytes);
    // pinned: [ReadParams n | DevParams | strand offsets 2n+1 | the int block] ; the dense results are staged in a second pinned buffer
    size_t hsz = 0;
    auto hcarve = [&](size_t bytes) { const size_t at = hsz; hsz += (bytes + 63) & ~(size_t)63; return at; };
    const size_t p_par = hcarve(par_bytes), p_rp = p_par, p_prm = p_par + r_prm, p_off = p_par + r_off, p_ord = p_par + r_ord;
    const size_t p_int = hcarve(4 * (nints + 8));
    if ((rc = pl->d_marena.reserve(dsz)) || (rc = pl->h_mint.reserve(hsz))) return rc;
    char* const dA = pl->d_marena.as<char>();
    char* const hA = pl->h_mint.as<char>();
    if (!pl->mev0) { ALFA_HIP(hipEventCreate(&pl->mev0)); ALFA_HIP(hipEventCreate(&pl->mev1)); }
    {
        alfa::mapcore::ReadParams* hr = (alfa::mapcore::ReadParams*)(hA + p_rp);
        for (int i = 0; i < n; ++i) {
            const alfa_map_read& r = reads[i];
            hr[i].plen = r.plen; hr[i].editDist = r.edit_dist; hr[i].minScore = r.min_score; hr[i].minCoverage = r.min_coverage;
            hr[i].minScore2 = r.min_score2; hr[i].minCoverage2 = r.min_coverage2;
        }
        dev_params_from(*dp, *(DevParams*)(hA + p_prm));
        memcpy(hA + p_off, h_off.data(), 8 * (size_t)nq);
        if (by_length) {
            int32_t* ho = (int32_t*)(hA + p_ord);
            for (int i = 0; i < n; ++i) ho[i] = i;
            std::stable_sort(ho, ho + n, [reads](int32_t a, int32_t b) { return reads[a].plen > reads[b].plen; });
        }
        ALFA_HIP(hipMemcpyAsync(dA + o_par, hA + p_par, par_bytes, hipMemcpyHostToDevice, st));
    }
    int32_t* d_int = (int32_t*)(dA + o_int);
    int32_t* d_status = d_int;
    int32_t* d_naln = d_int + n;                     // n + 1 entries, the last stays 0
    int32_t* d_nseed = d_naln + (n + 1);
    int32_t* d_nops = d_nseed + (n + 1);
    int32_t* d_aoff = d_nops + (n + 1);
    int32_t* d_soff = d_aoff + (n + 1);
    int32_t* d_ooff = d_soff + (n + 1);
    ALFA_HIP(hipMemsetAsync(d_int, 0, 4 * (nints + 8), st));
    alfa::MapKernelArgs A;
    A.S = ix->S.as<uint8_t>() + PADB; A.N = ix->N; A.startpos = ix->STARTPOS.as<int64_t>(); A.n_seq = ix->n_seq;
    A.reads = pl->d_reads.as<uint8_t>(); A.read_off = (const int64_t*)(dA + o_off); A.rp = (const alfa::mapcore::ReadParams*)(dA + o_rp); A.n = n;
    A.base = pl->d_base.as<int64_t>(); A.reg = pl->d_reg.as<alfa_region>(); A.members = pl->d_rm.as<alfa_match>();
    A.nreg = d_aux + 2 * (size_t)nq; A.rstatus = d_aux + 4 * (size_t)nq;
    A.dp_params = dA + o_prm; A.scratch = (uint8_t*)(dA + o_scr); A.scratch_stride = (int32_t)stride64; A.qcap = qcap;
    A.zscratch = (uint8_t*)(dA + o_z); A.zstride = zstride; A.zslot = zslot; A.bam_cap = bam_cap;
    A.order = by_length ? (const int32_t*)(dA + o_ord) : nullptr;
    A.slots = slots; A.next = d_int + nints; A.reasons = d_int + nints - 16;          // (the int block has 8 spare words behind its arrays; all zeroed below)
    A.opt.cigarCount = o->cigar_alignments > 0 ? o->cigar_alignments : 0; A.opt.rescueOnly = o->rescue_only ? 1 : 0;
    A.opt.match = o->match; A.opt.mismatch = o->mismatch; A.opt.openGap = o->open_gap; A.opt.extendGap = o->extend_gap;
    A.opt.global = o->global; A.opt.fixedBandSize = o->fixed_band; A.opt.maxTrial = o->max_trial; A.opt.maxTrialBest = o->max_trial_best;
    A.opt.maxAlnCount = o->max_aln_count; A.opt.rescue = o->rescue; A.opt.errorPercent = o->error_percent;
    A.status = d_status; A.naln = d_naln; A.nseed = d_nseed; A.nops = d_nops;
    A.alns = (alfa::mapcore::AlnOut*)(dA + o_aln); A.seeds = (alfa_match*)(dA + o_seed); A.ops = (alfa::mapcore::CigOp*)(dA + o_ops); A.ops_cap = ops_cap;
    struct ZHold {           // the shared buffer, from the kernel's launch until the host has seen the call's results
        alfa_index* ix; int k = -1;
        void release() {
            if (k < 0) return;
            { std::lock_guard<std::mutex> g(ix->zmu); ix->zbusy[k] = false; }
            ix->zcv.notify_one();
            k = -1;
        }
        ~ZHold() { release(); }
    } zhold{ix};
    if (zshared) {
        {
            std::unique_lock<std::mutex> g(ix->zmu);
            int k = -1;
            ix->zcv.wait(g, [&] {
                for (int j = 0; j < ix->zpool && k < 0; ++j) if (!ix->zbusy[j] && ix->zbuf[j].cap >= zbytes) k = j;     // one that is large enough already
                for (int j = 0; j < ix->zpool && k < 0; ++j) if (!ix->zbusy[j]) k = j;
                return k >= 0;
            });
            ix->zbusy[k] = true;
            zhold.k = k;
        }
        if ((rc = ix->zbuf[zhold.k].reserve(zbytes))) return rc;
        A.zscratch = ix->zbuf[zhold.k].as<uint8_t>();
    }
    ALFA_HIP(hipEventRecord(pl->mev0, st));
    alfa::launch_map(st, A);
    ALFA_HIP(hipGetLastError());
    ALFA_HIP(hipEventRecord(pl->mev1, st));
    if (n + 1 <= (1 << 20) && !getenv("ALFA_MAP_SCAN_LIB")) alfa::launch_map_offsets(      // (the variable: tests compare the two paths)
        st, n + 1, d_naln, d_nseed, d_nops, d_aoff, d_soff, d_ooff);
    else {
        size_t tb = tmp_bytes;
        ALFA_HIP(hipcub::DeviceScan::ExclusiveSum(dA + o_tmp, tb, d_naln, d_aoff, n + 1, st));
        tb = tmp_bytes;
        ALFA_HIP(hipcub::DeviceScan::ExclusiveSum(dA + o_tmp, tb, d_nseed, d_soff, n + 1, st));
        tb = tmp_bytes;
        ALFA_HIP(hipcub::DeviceScan::ExclusiveSum(dA + o_tmp, tb, d_nops, d_ooff, n + 1, st));
    }
    alfa::launch_map_compact(st, n, d_naln, d_nseed, d_nops, d_aoff, d_soff, d_ooff, A.alns, A.seeds, A.ops, ops_cap, (alfa::mapcore::AlnOut*)(dA + o_aln2),
                             (alfa_match*)(dA + o_seed2), (alfa::mapcore::CigOp*)(dA + o_ops2));
    ALFA_HIP(hipGetLastError());
    int32_t* h_int = (int32_t*)(hA + p_int);
    ALFA_HIP(hipMemcpyAsync(h_int, d_int, 4 * nints, hipMemcpyDeviceToHost, st));
    const double ms_enq = since(tc0) - ms_fill - ms_seeds;
    ALFA_HIP(hipStreamSynchronize(st));
    zhold.release();
    const double ms_map = since(tc0) - ms_fill - ms_seeds - ms_enq;
    const int32_t* h_status = h_int;
    const int32_t* h_aoff = h_int + n + 3 * ((size_t)n + 1);
    const int32_t* h_soff = h_aoff + (n + 1);
    const int32_t* h_ooff = h_soff + (n + 1);
    for (int i = 0; i < n; ++i) status[i] = h_status[i];
    for (int i = 0; i <= n; ++i) { aln_off[i] = h_aoff[i]; seed_off[i] = h_soff[i]; op_off[i] = h_ooff[i]; }
    const int64_t ta = aln_off[n], ts = seed_off[n], to = op_off[n];
    alfa_map_cigar_op* ho = (alfa_map_cigar_op*)realloc(*ops, sizeof(alfa_map_cigar_op) * (size_t)(to + 1));
    if (ho) *ops = ho;
    alfa_map_alignment* ha = (alfa_map_alignment*)realloc(*alns, sizeof(alfa_map_alignment) * (size_t)(ta + 1));
    if (ha) *alns = ha;
    alfa_match* hs = (alfa_match*)realloc(*seeds, sizeof(alfa_match) * (size_t)(ts + 1));
    if (hs) *seeds = hs;
    if (!ha || !hs || !ho) return alfa::fail(ALFA_ENOMEM, "out of host memory for the alignments");
    // the dense results come back through pinned staging (a copy into pageable memory is staged by the runtime anyway, under
    // its own locks) and are handed out in malloc'd arrays, as the header promises
    const size_t ba = sizeof(alfa_map_alignment) * (size_t)ta, bs = sizeof(alfa_match) * (size_t)ts, bo = sizeof(alfa_map_cigar_op) * (size_t)to;
    if ((rc = pl->h_mstage.reserve(ba + bs + bo + 64))) return rc;
    char* hS = pl->h_mstage.as<char>();
    if (ta > 0) ALFA_HIP(hipMemcpyAsync(hS, dA + o_aln2, ba, hipMemcpyDeviceToHost, st));
    if (ts > 0) ALFA_HIP(hipMemcpyAsync(hS + ba, dA + o_seed2, bs, hipMemcpyDeviceToHost, st));
    if (to > 0) ALFA_HIP(hipMemcpyAsync(hS + ba + bs, dA + o_ops2, bo, hipMemcpyDeviceToHost, st));
    ALFA_HIP(hipStreamSynchronize(st));
    if (ta > 0) memcpy(ha, hS, ba);
    if (ts > 0) memcpy(hs, hS + ba, bs);
    if (to > 0) memcpy(ho, hS + ba + bs, bo);
    float ms = 0;
    if (hipEventElapsedTime(&ms, pl->mev0, pl->mev1) != hipSuccess) { (void)hipGetLastError(); ms = 0; }
    alfa::g_last_timing.kernel_ms = ms;
    alfa::g_last_timing.work_units = n;
    alfa::g_last_timing.launches = 1;
    if (call_log)
    {
        const int32_t* why = h_int + nints - 16;
        fprintf(stderr, "#MAPWHY handed back by cause 1..11:");
        for (int k = 1; k <= 11; ++k) fprintf(stderr, " %d", why[k]);
        fprintf(stderr, "\n");
    }
    if (call_log) {
        size_t mfree = 0, mtotal = 0;
        (void)hipMemGetInfo(&mfree, &mtotal);
        fprintf(stderr, "#MAPCALL reads %d matches %lld alns %lld fill_ms %.3f seeds_ms %.3f enqueue_ms %.3f regions_map_wait_ms %.3f total_ms %.3f map_kernel_ms %.3f seed_kernel_ms %.3f arena_mb %.0f zscratch_mb %.0f prefixes %lld device_used_mb %.0f\n",
                n, (long long)total, (long long)ta, ms_fill, ms_seeds, ms_enq, ms_map, since(tc0), ms, alfa_seed_plan_kernel_ms(pl), dsz / 1048576.0, zbytes / 1048576.0,
                (long long)pl->total_pref, (mtotal - mfree) / 1048576.0);
    }
    return ALFA_OK;
}

}  // extern "C"
