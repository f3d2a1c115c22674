// map_kernels.hip -- whole reads on the device (the part round 2 left on the host): after the seed kernels
// and regions_kernel, one wavefront per read runs the reference's trial loop
//   unpairedMatch        src/mapper.cpp:1269-1302
//   processLisIntervals  src/mapper.cpp:1118-1204
//   extendAlignmentChain src/mapper.cpp:869-1081  (nextChain :721-782, left/rightExtension :784-867)
// over the read's candidate regions, with every ksw_nw / ksw_extend call it makes executed in place by the same wavefront
// (wave_task of ksw_wave.h: band-diagonal registers, all 64 lanes on the one task, the reference's moving band followed
// exactly), so that round k's DP results feed round k + 1's chain selection without leaving the GPU.  The control logic is
// csrc/map_core.h, compiled here for the device: every lane executes the same statements on the read's state in LDS, which
// keeps the control flow wave-uniform around the collective DP calls; the loops that are worth it (loading a region's seeds,
// ranking the anchors, staging the nt4 strings of a DP task) are strided over the lanes.
// A read the kernel cannot finish exactly as the reference would -- a capacity of map_core.h exceeded, a DP task outside the
// wave kernel's envelope, regions_kernel's hand-backs, the re-seeding rescue -- comes back with status NEED_CALLER and
// the caller maps it through the batch entry points as before.
#include <hip/hip_runtime.h>

#include <cstdint>

#define MC_FN __device__ __forceinline__
#include "map_kernels.h"
#include "ksw_wave.h"

namespace alfa {

namespace {

using mapcore::AlnOut;
using mapcore::ExtRes;
using mapcore::Work;

// Utils::ORDVALUE src/utils.h:55-80 for a byte of the read or of the text
__device__ __forceinline__ uint8_t nt4_of(uint8_t c) {
    const uint8_t l = c | 0x20;
    return l == 'a' ? 0 : l == 'c' ? 1 : l == 'g' ? 2 : l == 't' ? 3 : 4;
}

// -DALFA_MAP_PROFILE (tools/probes/map_profile.sh; never in the product build): cycles per phase, summed over all reads
#ifdef ALFA_MAP_PROFILE
__device__ unsigned long long g_map_prof[16];     // 0 stage, 1 ksw_nw, 2 ksw_extend, 3 ksw_global fill, 4 walk, 5 whole read, 6 reads, 8.. map_core.h's timers
#define PROF_T0 const long long prof_t0 = clock64()
#define PROF_ADD(k) prof[k] += clock64() - prof_t0
#else
#define PROF_T0
#define PROF_ADD(k)
#endif

constexpr int SEQ_OFF = 16;        // wave_task's 8-byte windows may start a few bytes in front of a sequence
constexpr int ZTILE = 2048;        // direction bytes of a ksw_global call that fit here stay in LDS, rows packed to the band's width

// Every member function is inlined by force: one that is left as a call takes the DevEnv by address, which puts it -- and the
// copy of the kernel's arguments it refers to -- into scratch memory, a round trip to HBM for every field the trial loop
// touches (2.71 -> 1.94 ms per 100 000 reads of C3 for the attribute alone).
struct DevEnv {
    const MapKernelArgs& A;
    int ln, r, slot;
    const uint8_t* Pf;
    const uint8_t* Pr;
    uint8_t* scr;
    uint8_t* ztile;
    const DevParams* prm;
    const alfa_region* regF; const alfa_region* regR;
    const alfa_match* memF; const alfa_match* memR;
    int nF, nR;
#ifdef ALFA_MAP_PROFILE
    long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    bool xlds = true;            // xop(): where the last call's operations lie, and the window of 64 of them held in the lanes
    int xbase = -64, xcap = 0;
    uint32_t xwin = 0;

    __device__ int lane() const { return ln; }
    __device__ int lanes() const { return 64; }
    __device__ void sync() const { __syncthreads(); }

    __device__ __forceinline__ bool load_region(Work& W, int k, int& n, bool& fw, int& score) {
        fw = k < nF;
        const alfa_region g = fw ? regF[k] : regR[k - nF];
        const alfa_match* m = (fw ? memF : memR) + g.first;
        n = g.count; score = g.score;
        if (n > mapcore::MEMBER_CAP || n < 1) return false;
        __syncthreads();
        for (int i = ln; i < n; i += 64) { const alfa_match x = m[i]; W.mref[i] = x.ref; W.mq[i] = x.query; W.ml[i] = x.len; }
        __syncthreads();
        return true;
    }
    // alfa_host_index_chrom_bounds (sparseSA::getChromBounds src/sparseSA.h:180-190)
    __device__ __forceinline__ void chrom_bounds(int64_t pos, int64_t& cs, int64_t& ce) const {
        int lo = 0, hi = A.n_seq;                   // first start > pos
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (A.startpos[mid] <= pos) lo = mid + 1; else hi = mid; }
        const int seq = lo - 1 < 0 ? 0 : lo - 1;
        cs = A.startpos[seq];
        ce = (A.n_seq > 1 && seq < A.n_seq - 1) ? A.startpos[seq + 1] - 1 : A.N - 1;
    }
    // the task's two strings as nt4 codes into this read's scratch slot; false when they do not fit
    __device__ __forceinline__ bool stage(bool fw, bool left, int qFrom, int qLen, int64_t tFrom, int tLen) {
        if (qLen < 1 || tLen < 1 || qLen > A.qcap || SEQ_OFF + A.qcap + SEQ_OFF + tLen + 16 > A.scratch_stride) return false;
        if (left ? (tFrom >= A.N || tFrom - tLen + 1 < 0) : (tFrom < 0 || tFrom + tLen > A.N)) return false;
        const uint8_t* P = fw ? Pf : Pr;
        uint8_t* q = scr + SEQ_OFF;
        uint8_t* t = scr + SEQ_OFF + A.qcap + SEQ_OFF;
        PROF_T0;
        __syncthreads();
        if (left) {          // both strings reversed: position j is j + 1 residues to the left of the anchor
            for (int j = ln; j < qLen; j += 64) q[j] = nt4_of(P[qFrom - j]);
            for (int j = ln; j < tLen; j += 64) t[j] = nt4_of(A.S[tFrom - j]);
        } else {
            for (int j = ln; j < qLen; j += 64) q[j] = nt4_of(P[qFrom + j]);
            for (int j = ln; j < tLen; j += 64) t[j] = nt4_of(A.S[tFrom + j]);
        }
        __threadfence_block();
        __syncthreads();
        PROF_ADD(0);
        return true;
    }
    __device__ __forceinline__ DevTask task(int qLen, int tLen, int w, int h0) const {
        DevTask T;
        T.qoff = SEQ_OFF; T.toff = SEQ_OFF + A.qcap + SEQ_OFF; T.zoff = 0; T.cigoff = 0; T.ehoff = 0;
        T.qlen = qLen; T.tlen = tLen; T.w = w; T.h0 = h0; T.orig = 0; T.cigcap = 0; T.cigcap2 = 0; T.pad = 0;
        return T;
    }
    // ksw_nw(qLen, P[qFrom..], tLen, S[tFrom..], ..., w)   src/ksw.c:544, call site src/mapper.cpp:982
    __device__ __forceinline__ bool nw(bool fw, int qFrom, int qLen, int64_t tFrom, int tLen, int w, int& score) {
        const int lim = qLen > tLen ? qLen : tLen;
        if (w > lim) w = lim;                       // the band never binds beyond this: identical loops (ksw_kernels.hip effective_w)
        if (w < 0 || 2 * w + 1 > 256) return false;
        if (!stage(fw, false, qFrom, qLen, tFrom, tLen)) return false;
        const DevTask T = task(qLen, tLen, w, 0);
        ExtOut eo;
        int s = 0;
        PROF_T0;
        if (2 * w + 1 <= 64) wave_task_run<MODE_NW, 1>(T, scr, prm, nullptr, eo, s);          // (a row costs by cells per lane, not by band width)
        else if (2 * w + 1 <= 128) wave_task_run<MODE_NW, 2>(T, scr, prm, nullptr, eo, s);
        else wave_task_run<MODE_NW, 4>(T, scr, prm, nullptr, eo, s);
        PROF_ADD(1);
        score = s;
        return true;
    }
    // ksw_extend(qLen, query, tLen, target, 5, mat, gapo, gape, w, 100, h0, ...)   src/ksw.c:362, call sites src/mapper.cpp:802,845
    // left: query[j] = P[qAnchor - 1 - j], target[j] = S[tAnchor - 1 - j]; right: query[j] = P[qAnchor + 1 + j], target[j] = S[tAnchor + 1 + j]
    __device__ __forceinline__ bool extend(bool fw, bool left, int qAnchor, int qLen, int64_t tAnchor, int tLen, int w, int h0, ExtRes& r) {
#ifdef ALFA_MAP_PROFILE
        struct Incl { long long& acc; long long t0; __device__ ~Incl() { acc += clock64() - t0; } } incl{prof[15], clock64()};
#endif
        // effective band and rows exactly as the batch entry point derives them (ksw_kernels.hip: effective_w, rows_eff)
        const int gapo = prm->gapo, gape = prm->gape, maxs = prm->maxs;
        int max_gap = gape > 0 ? (int)((double)(qLen * maxs - gapo) / gape + 1.) : 0x3fffffff;
        max_gap = max_gap > 1 ? max_gap : 1;
        w = w < max_gap ? w : max_gap;
        const int lim = qLen > tLen ? qLen : tLen;
        if (w > lim) w = lim;
        if (w < 0 || 2 * w + 1 > 256) return false;
        const int64_t rows64 = (int64_t)qLen + w + 1;
        const int rows = tLen < rows64 ? tLen : (int)rows64;        // ksw_extend leaves its row loop at row qlen + w (ksw.c:393-423)
        if ((int64_t)(h0 < 0 ? 0 : h0) + (int64_t)qLen * maxs >= (1 << 22)) return false;      // packed row-maximum key (wave_task PACK)
        if (!stage(fw, left, left ? qAnchor - 1 : qAnchor + 1, qLen, left ? tAnchor - 1 : tAnchor + 1, rows)) return false;
        const DevTask T = task(qLen, rows, w, h0);
        ExtOut eo;
        int s = 0;
        PROF_T0;
        if (2 * w + 1 <= 64) wave_task_run<MODE_EXTEND, 1, true>(T, scr, prm, nullptr, eo, s);
        else if (2 * w + 1 <= 128) wave_task_run<MODE_EXTEND, 2, true>(T, scr, prm, nullptr, eo, s);
        else wave_task_run<MODE_EXTEND, 4, true>(T, scr, prm, nullptr, eo, s);
        PROF_ADD(2);
        r.score = eo.score; r.qle = eo.qle; r.tle = eo.tle; r.gtle = eo.gtle; r.gscore = eo.gscore; r.max_off = eo.max_off;
        return true;
    }
    // ksw_global(qLen, P[qFrom..], tLen, S[tFrom..], ..., band)   src/ksw.c:468, call sites src/mapper.cpp:200,239,256: the wave-wide
    // fill writes one direction byte per cell to this read's slot of A.zscratch, then the walk of ksw.c:524-538 (every lane
    // the same steps) leaves the BAM operations in forward order at W.bam[first .. first + n).  -1: not done here.
    __device__ __forceinline__ int global_cigar(Work& W, bool fw, int qFrom, int qLen, int64_t tFrom, int tLen, int w, int& first) {
#ifdef ALFA_MAP_PROFILE
        struct Incl { long long& acc; long long t0; __device__ ~Incl() { acc += clock64() - t0; } } incl{prof[7], clock64()};
#endif
        if (qLen < 1 || tLen < 1) return -1;                 // degenerate calls (the reference's loops do not run): the caller's path
        const int lim = qLen > tLen ? qLen : tLen;
        if (w > lim) w = lim;
        if (w < 0 || 2 * w + 1 > 256 || tLen > qLen + w) return -1;      // (tlen > qlen + w: the reference indexes outside its matrix)
        const int C4 = 2 * w + 1 > 128;
        const bool C1 = 2 * w + 1 <= 64;                     // one cell per lane: the cheapest row
        // short pieces (the common case on short reads: a gap or an overhang of a few dozen residues) keep their direction
        // bytes in LDS, rows packed to the band's width -- the walk is a chain of dependent reads, 60 of them through HBM were
        // half of the kernel's time.  Longer ones (long reads: pieces of hundreds to thousands of rows) go to the workgroup's
        // slot of A.zscratch at the batch kernels' row stride and are walked through LDS a tile of rows at a time; their BAM
        // operations go to the slot as well.
        const int wpad = (2 * w + 1 + 4) & ~3;               // > 2w + 1, a multiple of 4
        const bool in_lds = !C4 && tLen * wpad <= ZTILE;
        const int rowb = in_lds ? wpad : (C4 ? 256 : C1 ? 64 : 128);
        if (!in_lds && (int64_t)tLen * rowb > A.zstride) return -1;
        if (!stage(fw, false, qFrom, qLen, tFrom, tLen)) return -1;
        uint8_t* zg = A.zscratch + (int64_t)slot * A.zslot;
        uint8_t* z = in_lds ? ztile : zg;
        const DevTask T = task(qLen, tLen, w, 0);
        ExtOut eo;
        int s = 0;
        {
            PROF_T0;
            if (in_lds && C1) wave_task_run<MODE_GLOBAL, 1, false, true, true>(T, scr, prm, z, eo, s, wpad);
            else if (in_lds) wave_task_run<MODE_GLOBAL, 2, false, true, true>(T, scr, prm, z, eo, s, wpad);
            else if (C1) wave_task_run<MODE_GLOBAL, 1, false, false, true>(T, scr, prm, z, eo, s);
            else if (!C4) wave_task_run<MODE_GLOBAL, 2, false, false, true>(T, scr, prm, z, eo, s);
            else wave_task_run<MODE_GLOBAL, 4, false, false, true>(T, scr, prm, z, eo, s);
            __threadfence_block();
            __syncthreads();
            PROF_ADD(3);
        }
        PROF_T0;
        // the walk's operations: at most one per step, qLen + tLen steps -- LDS when that is sure to fit, else the slot
        const bool ops_lds = qLen + tLen <= mapcore::BAM_CAP;
        uint32_t* bam = ops_lds ? W.bam : (uint32_t*)(zg + A.zstride);
        const int cap = ops_lds ? mapcore::BAM_CAP : A.bam_cap;
        int pos = cap, n = 0;
        uint32_t cur_op = 0xffffffffu, cur_len = 0;
        bool over = false;
        auto push = [&](uint32_t op, uint32_t len) {
            if (op == cur_op) { cur_len += len; return; }
            if (cur_op != 0xffffffffu) { if (pos > 0) bam[--pos] = cur_len << 4 | cur_op; else over = true; ++n; }
            cur_op = op; cur_len = len;
        };
        int i = tLen - 1, k = (i + w + 1 < qLen ? i + w + 1 : qLen) - 1;
        int which = 0;
        // The walk of ksw.c:524-538 is a chain of dependent one-byte reads, a few hundred cycles per step for a lone wavefront.
        // Most steps are diagonal, though, and a diagonal run needs no chain: in state 0 the lanes read the cells (i - t, k - t),
        // t = lane, at once (same column of the band, consecutive rows), a ballot gives the length of the run of cells whose
        // H field says "diagonal", and their equality bits become '=' / 'X' runs by bit scans.  Gap steps stay serial.
        const int TR = in_lds ? tLen : ZTILE / rowb;         // rows in LDS at a time: all of them, or a tile of 32 / 16
        const uint8_t* zt = in_lds ? z : ztile;
        int lo = in_lds ? 0 : i + 1;                         // first row of the tile in LDS (long pieces: none yet)
        while (i >= 0 && k >= 0) {
            if (i < lo) {                                    // every lane takes the same path: i and k are wave-uniform
                lo = i - TR + 1 > 0 ? i - TR + 1 : 0;
                const int bytes = (i - lo + 1) * rowb;
                __syncthreads();
                for (int o = ln * 16; o < bytes; o += 64 * 16) *(uint4*)(ztile + o) = *(const uint4*)(z + (int64_t)lo * rowb + o);
                __syncthreads();
            }
            if (which == 0) {
                int lim = i - lo < k ? i - lo : k;           // cells of this diagonal inside the tile (and the matrix), less one
                lim = (lim < 63 ? lim : 63) + 1;
                uint32_t zc = 3u;
                if (ln < lim) zc = zt[(i - ln - lo) * rowb + (k - i + w)];
                const uint64_t stop = __ballot(ln >= lim || (zc & 3u) != 0u);
                const int L = stop ? __builtin_ctzll(stop) : 64;
                if (L > 0) {
                    const uint64_t eqm = __ballot((zc & 64u) != 0u);
                    for (int t = 0; t < L;) {
                        const bool e = (eqm >> t) & 1;
                        const uint64_t diff = (e ? ~eqm : eqm) >> t;
                        int run = diff ? __builtin_ctzll(diff) : 64;
                        if (run > L - t) run = L - t;
                        push(e ? mapcore::XOP_EQ : mapcore::XOP_X, (uint32_t)run);
                        t += run;
                    }
                    i -= L; k -= L;
                    continue;
                }
            }
            const uint32_t zv = zt[(i - lo) * rowb + (k - i + w)];
            which = (zv >> (which << 1)) & 3;
            if (which == 0) { push((zv & 64u) ? mapcore::XOP_EQ : mapcore::XOP_X, 1); --i; --k; }
            else if (which == 1) { push(mapcore::XOP_DEL, 1); --i; }
            else { push(mapcore::XOP_INS, 1); --k; }
        }
        if (i >= 0) push(mapcore::XOP_DEL, (uint32_t)(i + 1));
        if (k >= 0) push(mapcore::XOP_INS, (uint32_t)(k + 1));
        if (cur_op != 0xffffffffu) { if (pos > 0) bam[--pos] = cur_len << 4 | cur_op; else over = true; ++n; }
        PROF_ADD(4);
        if (over) return -1;
        first = pos;
        W.bamp = bam;
        xlds = ops_lds;
        xbase = -64;
        xcap = cap;
        return n;
    }
    // operation i of the last global_cigar call.  Short calls: LDS.  Long calls: the workgroup's slot in HBM, where one load per
    // operation would be a chain of ~1 us round trips: the lanes fetch 64 operations at a time and hand them round
    __device__ __forceinline__ uint32_t xop(const Work& W, int i) {
        if (xlds) return W.bam[i];
        if ((unsigned)(i - xbase) >= 64u) {
            xbase = i;
            xwin = i + ln < xcap ? W.bamp[i + ln] : 0u;
        }
        return (uint32_t)__builtin_amdgcn_readlane((int)xwin, __builtin_amdgcn_readfirstlane(i - xbase));
    }
    // bit i: nt4(P[qpos + i]) == nt4(S[tpos + i]), i < n <= 64
    __device__ __forceinline__ uint64_t equal_mask(bool fw, int qpos, int64_t tpos, int n) const {
        const uint8_t* P = fw ? Pf : Pr;
        bool e = false;
        if (ln < n) e = nt4_of(P[qpos + ln]) == nt4_of(A.S[tpos + ln]);
        return __ballot(e);
    }
};

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) void map_kernel(const MapKernelArgs A) {
    __shared__ Work W;
    __shared__ __align__(16) uint8_t ztile[ZTILE];
    __shared__ int next_read;
    const int ln = threadIdx.x, slot = blockIdx.x;
    // every workgroup takes reads until none is left (the grid is what the GPU holds at a time: `slots` workgroups)
    for (;;) {
        __syncthreads();
        if (ln == 0) next_read = atomicAdd(A.next, 1);
        __syncthreads();
        if (next_read >= A.n) break;
        const int r = A.order ? A.order[next_read] : next_read;
        const int sf = 2 * r, sr = 2 * r + 1;
        const int64_t bf = A.base[sf], br = A.base[sr], be = A.base[sr + 1];
        int status = mapcore::DONE, naln = 0, nseed = 0, nops = 0;
        // why a read is handed back (A.reasons): 1 regions_kernel handed a strand back, 2 no seed (the caller's seed rescue),
        // 3 regions > REGION_CAP2, 4 sort depth, 5 TREE_CAP, 6 ksw_nw / 7 ksw_extend call beyond the kernel's band or scratch,
        // 8 BOUND_CAP / ALN_CAP / CHAINOUT_CAP, 9 BOUND_CAP, 10 region members > MEMBER_CAP, 11 no alignment (re-seeding rescue)
        int why = 0;
        if (A.rstatus[sf] | A.rstatus[sr]) { status = mapcore::NEED_CALLER; why = 1; }          // regions_kernel handed a strand back
        else if (be == bf) { status = (A.opt.rescue || A.opt.rescueOnly) ? mapcore::NO_SEED : mapcore::DONE; why = 2; }      // no seed on either strand: the seed rescue rounds (src/mapper.cpp:542-555) stay with the caller
        else {
            DevEnv env{A, ln, r, slot, A.reads + A.read_off[sf], A.reads + A.read_off[sr], A.scratch + (int64_t)slot * A.scratch_stride, ztile,
                       (const DevParams*)A.dp_params, A.reg + bf, A.reg + br, A.members + 2 * bf + 8 * (int64_t)sf,
                       A.members + 2 * br + 8 * (int64_t)sr, A.nreg[sf], A.nreg[sr]};
            const int nreg = env.nF + env.nR;
            if (nreg > mapcore::REGION_CAP2) { status = mapcore::NEED_CALLER; why = 3; }
            else {
                for (int k = ln; k < nreg; k += 64) W.rscore[k] = k < env.nF ? env.regF[k].score : env.regR[k - env.nF].score;
                static_assert(sizeof(mapcore::ChainSeed) == sizeof(alfa_match), "ChainSeed == alfa_match");
                W.chain = (mapcore::ChainSeed*)(A.seeds + (int64_t)r * mapcore::CHAINOUT_CAP);      // straight into the read's slot of the output
                W.ops = A.ops + (int64_t)r * A.ops_cap;          // (every lane stores the same values)
                W.opsCap = A.ops_cap;
                W.bamp = W.bam;
                __syncthreads();
#ifdef ALFA_MAP_PROFILE
                const long long rt0 = clock64();
#endif
                status = mapcore::map_read(env, W, A.opt, A.rp[r], nreg);
#ifdef ALFA_MAP_PROFILE
                if (ln == 0) {
                    env.prof[5] = clock64() - rt0; env.prof[6] = 1;
                    for (int k = 0; k < 16; ++k) atomicAdd(&g_map_prof[k], (unsigned long long)env.prof[k]);
                }
#endif
                why = W.fail;
                if (status == mapcore::DONE) {
                    naln = W.nAln; nseed = W.nChainOut; nops = W.nOps;
                    __syncthreads();
                    for (int k = ln; k < naln; k += 64) A.alns[(int64_t)r * mapcore::ALN_CAP + k] = W.alns[k];
                }
            }
        }
        if (ln == 0) {
            if (status != mapcore::DONE) atomicAdd(A.reasons + (why & 15), 1);
            A.status[r] = status;
            A.naln[r] = naln;
            A.nseed[r] = nseed;
            A.nops[r] = nops;
        }
    }
}

__global__ void map_compact_kernel(int n, const int32_t* __restrict__ naln, const int32_t* __restrict__ nseed, const int32_t* __restrict__ nops,
                                   const int32_t* __restrict__ aln_off, const int32_t* __restrict__ seed_off, const int32_t* __restrict__ op_off,
                                   const AlnOut* __restrict__ alns, const alfa_match* __restrict__ seeds, const mapcore::CigOp* __restrict__ ops, int ops_cap,
                                   AlnOut* __restrict__ out_alns, alfa_match* __restrict__ out_seeds, mapcore::CigOp* __restrict__ out_ops) {
    const int r = blockIdx.x;
    if (r >= n) return;
    const int na = naln[r], ns = nseed[r], no = nops[r];
    for (int k = threadIdx.x; k < na; k += blockDim.x) out_alns[aln_off[r] + k] = alns[(int64_t)r * mapcore::ALN_CAP + k];
    for (int k = threadIdx.x; k < ns; k += blockDim.x) out_seeds[seed_off[r] + k] = seeds[(int64_t)r * mapcore::CHAINOUT_CAP + k];
    for (int k = threadIdx.x; k < no; k += blockDim.x) out_ops[op_off[r] + k] = ops[(int64_t)r * ops_cap + k];
}

// exclusive prefix sums of the three per-read count arrays (n1 = n + 1 entries each), one workgroup per array: a call's
// arrays are a few ten thousand entries, for which a library scan's two launches each cost more than the scan
constexpr int OFFS_ITEMS = 4;
__global__ __launch_bounds__(1024) void map_offsets_kernel(int n1, const int32_t* __restrict__ a, const int32_t* __restrict__ b, const int32_t* __restrict__ c,
                                                          int32_t* __restrict__ oa, int32_t* __restrict__ ob, int32_t* __restrict__ oc) {
    const int32_t* __restrict__ in = blockIdx.x == 0 ? a : blockIdx.x == 1 ? b : c;
    int32_t* __restrict__ out = blockIdx.x == 0 ? oa : blockIdx.x == 1 ? ob : oc;
    __shared__ int32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int32_t carry = 0;
    for (int start = 0; start < n1; start += 1024 * OFFS_ITEMS) {
        const int at = start + tid * OFFS_ITEMS;
        int32_t v[OFFS_ITEMS];
        int32_t mine = 0;
#pragma unroll
        for (int k = 0; k < OFFS_ITEMS; ++k) { v[k] = at + k < n1 ? in[at + k] : 0; mine += v[k]; }
        int32_t inc = mine;                                  // inclusive scan over the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int32_t o = __shfl_up(inc, off, 64); if (lane >= off) inc += o; }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        int32_t before = carry, total = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const int32_t w = wsum[k]; if (k < wv) before += w; total += w; }
        int32_t run = before + inc - mine;
#pragma unroll
        for (int k = 0; k < OFFS_ITEMS; ++k) { if (at + k < n1) out[at + k] = run; run += v[k]; }
        carry += total;
        __syncthreads();
    }
}

}  // namespace

void launch_map_offsets(hipStream_t st, int n1, const int32_t* a, const int32_t* b, const int32_t* c, int32_t* oa, int32_t* ob, int32_t* oc) {
    if (n1 <= 0) return;
    hipLaunchKernelGGL(map_offsets_kernel, dim3(3), dim3(1024), 0, st, n1, a, b, c, oa, ob, oc);
}

// 4 wavefronts per SIMD (128 registers) and the LDS of a CU (10 KB per workgroup) both allow 16 one-wave workgroups per CU
int map_resident_workgroups() {
    static const int v = [] {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t p;
            if (hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) cus = p.multiProcessorCount;
        }
        return cus * 16;
    }();
    return v;
}

#ifdef ALFA_MAP_PROFILE
extern "C" int alfa_map_profile(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_map_prof), sizeof(g_map_prof)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_map_prof), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif

void launch_map(hipStream_t st, const MapKernelArgs& a) {
    if (a.n <= 0 || a.slots <= 0) return;
    hipLaunchKernelGGL(map_kernel, dim3(a.slots), dim3(64), 0, st, a);
}

void launch_map_compact(hipStream_t st, int n, const int32_t* naln, const int32_t* nseed, const int32_t* nops, const int32_t* aln_off,
                        const int32_t* seed_off, const int32_t* op_off, const mapcore::AlnOut* alns, const alfa_match* seeds,
                        const mapcore::CigOp* ops, int ops_cap, mapcore::AlnOut* out_alns, alfa_match* out_seeds, mapcore::CigOp* out_ops) {
    if (n <= 0) return;
    hipLaunchKernelGGL(map_compact_kernel, dim3(n), dim3(64), 0, st, n, naln, nseed, nops, aln_off, seed_off, op_off, alns, seeds, ops, ops_cap,
                       out_alns, out_seeds, out_ops);
}

}  // namespace alfa
