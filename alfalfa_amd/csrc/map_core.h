// map_core.h -- the per-read trial loop of the mapper (second half), written once for two builds:
//   * map_kernels.hip compiles it for the GPU: one wavefront per read runs it (every lane executes the same statements on the
//     read's state in LDS -- identical values, identical control flow -- so that the banded-DP calls in the middle of it are
//     wave-wide collectives: wave_task of ksw_wave.h with all 64 lanes on the one task);
//   * tests/ compiles it with g++ against the CPU oracle's DP to check the logic in a GPU-less container
//     (tests/cpu_mapper_main.cpp: `_cpu_mapper align --core-trials`, SAM goldens).
// What it restates (reference file:line, semantics exactly, structure ours):
//   processLisIntervals   src/mapper.cpp:1118-1204   regions in score order, hasUnique :1090-1112, trial counters :1134-1169
//   extendAlignmentChain  src/mapper.cpp:869-1081    anchors longest first :885-893, containment skip :895-914, nextChain
//                                                    :721-782, closed-form gap cases :951-966, inner band :978-983, order of
//                                                    the two extensions :994-1001, duplicate suppression :1004-1029
//   leftExtension / rightExtension  :784-867         windows, h0 = current score, global-vs-local decision :806,849, the
//                                                    chromosome-edge gap charge :820-823,863-866
//   unpairedMatch         src/mapper.cpp:1269-1302   first pass, then the looser second pass over the same regions (:1283-1289);
//                                                    the re-seeding rescue (:1290-1300) is left to the caller (NEED_CALLER)
// and the one library routine whose behaviour the result depends on: std::sort's order of regions with EQUAL scores
// (compIntervals :1114 is not a total order) -- libstdc++'s introsort is restated (sort_like_libstdcxx) so that ties fall as
// they do in the reference binary.
// Everything that involves floating point and depends only on the read length (editDist, minScore, minCoverage, ...) is
// computed by the caller on the host and arrives in ReadParams: the device never rounds differently from the reference.
#pragma once
#include <stdint.h>

#ifndef MC_FN
#define MC_FN inline
#endif
// phase timers of the profiling build of the kernel (map_kernels.hip under -DALFA_MAP_PROFILE); nothing anywhere else
#if defined(ALFA_MAP_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define MC_T0(v) const long long v = clock64()
#define MC_ADD(env, k, v) (env).prof[k] += clock64() - v
#else
#define MC_T0(v)
#define MC_ADD(env, k, v)
#endif

namespace alfa {
namespace mapcore {

constexpr int MEMBER_CAP = 160;     // seeds of one candidate region
constexpr int REGION_CAP2 = 128;    // candidate regions of a read, both strands
constexpr int TREE_CAP = 96;        // distinct read positions in hasUnique's interval tree, per strand
constexpr int BOUND_CAP = 24;       // chains tried in one region (alnBounds)
constexpr int ALN_CAP = 8;          // alignments reported per read
constexpr int CHAINOUT_CAP = 96;    // seeds of all reported chains of a read
constexpr int OPS_CAP = 160;        // extended CIGAR operations of a read's reported alignments: the least a caller provides room for
constexpr int BAM_CAP = 96;         // extended operations of one short ksw_global call (longer calls: the environment's memory)

enum Status { DONE = 0, NEED_CALLER = 1, NO_SEED = 2, NO_ALIGNMENT = 3 };      // NEED_CALLER: the caller maps this read itself (a capacity was exceeded, the
                                                // re-seeding rescue is due, a DP task is outside the wave kernel's envelope)
// An alignment whose extended CIGAR was not made here (it is not among the first `cigarCount` by score, or a capacity was
// exceeded, or a ksw_global call was degenerate) has opCount = -1: the caller recovers it through ksw_global as before.

struct Options {                    // align_opt (src/options.h), the fields the trial loop reads
    int32_t match, mismatch, openGap, extendGap;     // openGap / extendGap as alnOptions holds them: negative
    int32_t global, fixedBandSize, maxTrial, maxTrialBest, maxAlnCount, rescue;
    int32_t cigarCount, rescueOnly; // rescueOnly: only the looser pass, and no alignment = done (the RESCUE2 round src/mapper.cpp:1290-1300); extended CIGARs for the first cigarCount alignments by score (alignmentCount; 0 = none)
    double errorPercent;
};

struct ReadParams {                 // derived on the host from the read length (src/mapper.cpp:1224,1280-1287)
    int32_t plen, editDist;
    int32_t minScore, minCoverage;          // first pass
    int32_t minScore2, minCoverage2;        // second pass (rescue 1), maxAlnCount 20
};

struct Trial { int64_t refB, refE; int32_t qB, qE; };
struct ExtRes { int32_t score, qle, tle, gtle, gscore, max_off; };      // = alfa_extend_result
struct AlnOut { int64_t refB, refE; int32_t qB, qE, score, fw, chainFirst, chainCount, opFirst, opCount, editDist, pad_; };
struct ChainSeed { int64_t ref; int32_t q, len; };       // = alfa_match
struct CigOp { int32_t len, code; };            // code: the operation's character ('=', 'X', 'I', 'D', 'S'); len as the reference prints it

struct Work {                       // one read's state (LDS on the device)
    int64_t mref[MEMBER_CAP];                       // the current region's seeds, in the order coveredBases leaves them
    int32_t mq[MEMBER_CAP], ml[MEMBER_CAP];
    uint16_t byLen[MEMBER_CAP];                     // member indices by (len, index) ascending
    uint8_t used[MEMBER_CAP];
    uint16_t fwd[MEMBER_CAP], back[MEMBER_CAP];     // the chain: members after / before the anchor, in the order they were taken
    Trial bounds[BOUND_CAP];
    int32_t accScore[BOUND_CAP];
    int64_t accBegin[BOUND_CAP];
    int32_t treeQ[2][TREE_CAP], treeL[2][TREE_CAP];
    int32_t treeN[2];
    uint16_t rorder[REGION_CAP2];
    int32_t rscore[REGION_CAP2];
    AlnOut alns[ALN_CAP];                           // the read's alignments, in the order the reference appends them
    ChainSeed* chain;                               // their chains' seeds, CHAINOUT_CAP of them: memory of the environment's (the device
                                                    // build: the read's slot of the call's output -- written when an alignment is
                                                    // accepted, read once per piece when its CIGAR is made; 1.5 KB that LDS does not hold)
    CigOp* ops;                                     // the read's extended CIGAR operations (caller's memory: written, never read back)
    int32_t opsCap;
    uint32_t bam[BAM_CAP];
    const uint32_t* bamp;      // the operations of the last global_cigar call: `bam`, or memory of the environment's (long pieces)
    uint16_t aorder[ALN_CAP];
    int32_t nAln, nChainOut, nOps, fail;
};

// ---- std::sort as libstdc++ implements it (bits/stl_algo.h: __sort -> __introsort_loop + __final_insertion_sort, threshold
// 16, median of three to the first position, unguarded Hoare partition), on an index array with `before(a, b)` as the
// comparator.  Returns false when the depth limit is hit (the library would switch to heap sort: not restated; never seen).
template <class Before>
MC_FN void lin_insert(uint16_t* a, int last, Before before) {        // __unguarded_linear_insert
    const uint16_t val = a[last];
    int next = last - 1;
    while (before(val, a[next])) { a[last] = a[next]; last = next; --next; }
    a[last] = val;
}
template <class Before>
MC_FN void ins_sort(uint16_t* a, int first, int last, Before before) {   // __insertion_sort on [first, last)
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (before(a[i], a[first])) {
            const uint16_t val = a[i];
            for (int k = i; k > first; --k) a[k] = a[k - 1];
            a[first] = val;
        } else
            lin_insert(a, i, before);
    }
}
template <class Before>
MC_FN bool sort_like_libstdcxx(uint16_t* a, int n, uint32_t* stk, int stkCap, Before before) {
    if (n <= 1) return true;
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) ++lg;
    // __introsort_loop, its recursion on the right part as an explicit stack of (first, last, depth) in the caller's memory
    // (stkCap entries of three words; the device build hands over LDS -- an array on the function's own stack is scratch
    // memory there, the only one the kernel would have).  The depth limit bounds the stack by 2 lg n + 1 entries.
    int sp = 0;
    stk[0] = 0; stk[1] = (uint32_t)n; stk[2] = (uint32_t)(2 * lg); sp = 1;
    while (sp > 0) {
        --sp;
        int first = (int)stk[3 * sp], last = (int)stk[3 * sp + 1], depth = (int)stk[3 * sp + 2];
        // the library recurses into [cut, last) FIRST and then loops on [first, cut); partitions of disjoint ranges commute,
        // so the order in which the two parts are processed does not change the result
        while (last - first > 16) {
            if (depth == 0) return false;
            --depth;
            const int mid = first + (last - first) / 2;
            {   // __move_median_to_first(first, first + 1, mid, last - 1)
                const int A = first + 1, B = mid, C = last - 1;
                int pick;
                if (before(a[A], a[B])) pick = before(a[B], a[C]) ? B : (before(a[A], a[C]) ? C : A);
                else pick = before(a[A], a[C]) ? A : (before(a[B], a[C]) ? C : B);
                const uint16_t t = a[first]; a[first] = a[pick]; a[pick] = t;
            }
            int lo = first + 1, hi = last;                          // __unguarded_partition(first + 1, last, first)
            for (;;) {
                while (before(a[lo], a[first])) ++lo;
                --hi;
                while (before(a[first], a[hi])) --hi;
                if (!(lo < hi)) break;
                const uint16_t t = a[lo]; a[lo] = a[hi]; a[hi] = t;
                ++lo;
            }
            const int cut = lo;
            if (sp >= stkCap) return false;
            stk[3 * sp] = (uint32_t)cut; stk[3 * sp + 1] = (uint32_t)last; stk[3 * sp + 2] = (uint32_t)depth; ++sp;
            last = cut;
        }
    }
    // __final_insertion_sort
    if (n > 16) {
        ins_sort(a, 0, 16, before);
        for (int i = 16; i < n; ++i) lin_insert(a, i, before);
    } else
        ins_sort(a, 0, n, before);
    return true;
}

// hasUnique src/mapper.cpp:1090-1112: the read positions covered so far on this strand as (position -> seed length), kept
// sorted by position; a region is "unique" when one of its seeds is not inside the interval of the entry at or before it
MC_FN bool covers_new_bases(Work& W, int strand, int n) {
    bool unique = false;
    int32_t* tq = W.treeQ[strand];
    int32_t* tl = W.treeL[strand];
    int tn = W.treeN[strand];
    for (int k = 0; k < n; ++k) {
        const int32_t q = W.mq[k], len = W.ml[k];
        int ub = 0;                                   // first entry with position > q
        while (ub < tn && tq[ub] <= q) ++ub;
        const bool hasLower = ub > 0;
        if (!hasLower || tq[ub - 1] + tl[ub - 1] < q + len) {
            if (hasLower && tq[ub - 1] == q) tl[ub - 1] = len;
            else {
                if (tn >= TREE_CAP) { W.fail = 5; break; }
                for (int j = tn; j > ub; --j) { tq[j] = tq[j - 1]; tl[j] = tl[j - 1]; }
                tq[ub] = q; tl[ub] = len;
                ++tn;
            }
            unique = true;
        }
    }
    W.treeN[strand] = tn;
    return unique;
}

// nextChain src/mapper.cpp:721-782: greedy collinear chain around the anchor over the region's seeds (sorted by read position):
// forwards, then backwards; marks what it takes; returns the edit estimate, `coverage` = read bases the chain covers
MC_FN int grow_chain(Work& W, int n, int anchor, double errorPercent, int plen, int& coverage, int& nFwd, int& nBack) {
    const int64_t MAX_OVERLAP = -10;
    int edits = 0;
    coverage = W.ml[anchor];
    nFwd = nBack = 0;
    for (int dir = 0; dir < 2; ++dir) {
        // the chain's current end on this side: [lb, rb] in the read and in the text
        int64_t rLB = W.mref[anchor], qLB = W.mq[anchor];
        int64_t rRB = rLB + W.ml[anchor] - 1, qRB = qLB + W.ml[anchor] - 1;
        for (int k = dir == 0 ? anchor + 1 : anchor - 1; dir == 0 ? k < n : k >= 0; k += dir == 0 ? 1 : -1) {
            const int64_t mq = W.mq[k], mr = W.mref[k], ml = W.ml[k];
            const int64_t qGap = dir == 0 ? mq - (qRB + 1) : qLB - (mq + ml);
            const int64_t rGap = dir == 0 ? mr - (rRB + 1) : rLB - (mr + ml);
            const int64_t qSpan = dir == 0 ? mq + ml - qLB : qRB - mq + 1;
            const int64_t rSpan = dir == 0 ? mr + ml - rLB : rRB - mr + 1;
            const int64_t skew = qGap - rGap < 0 ? rGap - qGap : qGap - rGap;
            if (qGap >= MAX_OVERLAP && rGap >= MAX_OVERLAP && skew <= (int64_t)(errorPercent * (double)(qSpan > rSpan ? qSpan : rSpan))) {
                if (dir == 0) W.fwd[nFwd++] = (uint16_t)k; else W.back[nBack++] = (uint16_t)k;
                edits += (int)(skew > 1 ? skew : 1);
                rLB = mr; rRB = mr + ml - 1; qLB = mq; qRB = mq + ml - 1;
                W.used[k] = 1;
                const int64_t mn = qGap < rGap ? qGap : rGap;
                coverage += mn < 0 ? (int)(ml + mn) : (int)ml;
            }
        }
        if (dir == 0) edits += (int)(qRB + 1) < plen ? 1 : 0;
        else edits += qLB > 0 ? 1 : 0;
    }
    return edits;
}

// member index of the c-th seed of the current chain (back members reversed, the anchor, forward members)
MC_FN int chain_at(const Work& W, int anchor, int nBack, int c) {
    return c < nBack ? W.back[nBack - 1 - c] : (c == nBack ? anchor : W.fwd[c - nBack - 1]);
}

// extendAlignmentChain src/mapper.cpp:869-1081 on the region held in W (n seeds); appends accepted alignments through env
template <class Env>
MC_FN int try_region(Env& env, Work& W, const Options& O, const ReadParams& R, int n, bool fw, int minScore, int minCoverage,
                     int64_t chrStart, int64_t chrEnd) {
    int best = minScore - 10 < -1 ? minScore - 10 : -1;
    MC_T0(t_rank);
    // anchors: by (length, index) ascending, tried from the back (:885-893)
    for (int i = env.lane(); i < n; i += env.lanes()) {
        const int32_t l = W.ml[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (W.ml[j] < l) | ((W.ml[j] == l) & (j < i));
        W.byLen[rank] = (uint16_t)i;
        W.used[i] = 0;
    }
    env.sync();
    MC_ADD(env, 11, t_rank);
    int nBounds = 0, nAcc = 0;
    const int plen = R.plen, editDist = R.editDist;
    for (int oi = n - 1; oi >= 0; --oi) {
        const int a = W.byLen[oi];
        if (W.used[a]) continue;
        const int64_t aR = W.mref[a];
        const int32_t aQ = W.mq[a], aL = W.ml[a];
        bool go = true;
        for (int j = 0; j < nBounds && go; ++j) {            // inside an alignment already made here, near its diagonal (:895-914)
            const Trial b = W.bounds[j];
            const bool outside = aQ < b.qB || aQ + aL >= b.qE || aR < b.refB || aR + aL >= b.refE;
            if (!outside) {
                int qd = aQ - b.qB, rd = (int)(aR - b.refB);
                int lim = (int)(((qd < rd ? qd : rd) * O.match + O.openGap) / O.extendGap + 1.);
                lim = lim < O.fixedBandSize ? lim : O.fixedBandSize;
                go = (qd > rd ? qd : rd) >= lim;
                if (go) {
                    qd = b.qE - (aQ + aL); rd = (int)(b.refE - (aR + aL));
                    lim = (int)(((qd < rd ? qd : rd) * O.match + O.openGap) / O.extendGap + 1.);
                    lim = lim < O.fixedBandSize ? lim : O.fixedBandSize;
                    go = (qd > rd ? qd : rd) >= lim;
                }
            }
        }
        if (!go) continue;
        int score, nFwd, nBack;
        MC_T0(t_gc);
        const int minEdit = grow_chain(W, n, a, O.errorPercent, plen, score, nFwd, nBack);      // the score starts as the chain's coverage
        MC_ADD(env, 12, t_gc);
        const int chainLen = nBack + 1 + nFwd;
        if (!(minEdit <= editDist && (chainLen > 1 || aL >= minCoverage))) continue;
        const int band = O.fixedBandSize < editDist - minEdit + 1 ? O.fixedBandSize : editDist - minEdit + 1;
        const int first = chain_at(W, a, nBack, 0);
        int qB = W.mq[first], qE = qB + W.ml[first] - 1;
        int64_t rB = W.mref[first], rE = rB + W.ml[first] - 1;
        for (int c = 1; c < chainLen; ++c) {                 // gaps between consecutive seeds (:943-986)
            const int k = chain_at(W, a, nBack, c);
            const int64_t sR = W.mref[k];
            const int32_t sQ = W.mq[k], sL = W.ml[k];
            int64_t qd = sQ - qE, rd = sR - rE;
            if (qd < 0 || rd < 0) {
                if (qd >= rd) qd -= rd; else rd -= qd;
            }
            if (qd <= 0) score += (int)(O.openGap + O.extendGap * rd);
            else if (rd <= 0) score += (int)(O.openGap + O.extendGap * qd);
            else if (qd == 1) score += (int)(O.openGap + O.extendGap * (rd - 1));
            else if (rd == 1) score += (int)(O.openGap + O.extendGap * (qd - 1));
            else if (rd == 2 && qd == 2) score += O.mismatch;
            else {
                const int tLen = (int)(sR - rE - 1), qLen = sQ - qE - 1;
                const int d = qLen - tLen < 0 ? tLen - qLen : qLen - tLen;
                int inner = editDist - minEdit + d + 3;
                inner = inner < O.fixedBandSize ? inner : O.fixedBandSize;
                const int wide = d + (int)(d * O.errorPercent);
                inner = inner > wide ? inner : wide;
                int s = 0;
                if (!env.nw(fw, qE + 1, qLen, rE + 1, tLen, inner, s)) { W.fail = 6; return best; }
                score += s;
            }
            rE = sR + sL - 1;
            qE = sQ + sL - 1;
        }
        // the two extensions, shorter overhang first (:994-1001); each starts from the score so far
        const bool leftFirst = qB <= (plen - 1) - qE;
        for (int step = 0; step < 2; ++step) {
            const bool left = (step == 0) == leftFirst;
            if (left) {                                       // leftExtension :784-824
                const int ed = editDist - minEdit + 1;
                if (qB > 0 && rB > chrStart) {
                    int64_t lo = rB - qB - ed;
                    lo = lo > chrStart ? lo : chrStart;
                    const int tLen = (int)(rB - lo);
                    ExtRes r;
                    if (!env.extend(fw, true, qB, qB, rB, tLen, band < qB ? band : qB, score, r)) { W.fail = 7; return best; }
                    score = r.score;
                    if (O.global || r.gscore >= score + (O.mismatch - O.match) * 3) { score = r.gscore; rB -= r.gtle; qB = 0; }
                    else { qB -= r.qle; rB -= r.tle; }
                } else if (qB > 0 && rB == chrStart && O.global)
                    score += O.openGap + O.extendGap * qB;
            } else {                                          // rightExtension :826-867
                if (qE < plen - 1 && rE < chrEnd) {
                    int64_t hi = rE + (plen - 1 - qE) + editDist;
                    hi = hi < chrEnd ? hi : chrEnd;
                    const int tLen = (int)(hi - rE), qLen = plen - 1 - qE;
                    ExtRes r;
                    if (!env.extend(fw, false, qE, qLen, rE, tLen, band < qLen ? band : qLen, score, r)) { W.fail = 7; return best; }
                    score = r.score;
                    if (O.global || r.gscore >= score + (O.mismatch - O.match) * 3) { score = r.gscore; rE += r.gtle; qE = plen - 1; }
                    else { qE += r.qle; rE += r.tle; }
                } else if (qE < plen - 1 && rE == chrEnd && O.global)
                    score += O.openGap + O.extendGap * (plen - 1 - qE);
            }
        }
        best = best > score ? best : score;
        if (score >= minScore) {
            bool dup = false;                                 // :1004-1029
            for (int k = 0; k < nAcc; ++k) {
                const int64_t dd = W.accBegin[k] - rB;
                if (W.accScore[k] >= score && (dd < 0 ? -dd : dd) < editDist) dup = true;
            }
            if (!dup) {
                if (nAcc >= BOUND_CAP || W.nAln >= ALN_CAP || W.nChainOut + chainLen > CHAINOUT_CAP) { W.fail = 8; return best; }
                AlnOut o;
                o.refB = rB; o.refE = rE; o.qB = qB; o.qE = qE; o.score = score; o.fw = fw ? 1 : 0;
                o.chainFirst = W.nChainOut; o.chainCount = chainLen; o.opFirst = 0; o.opCount = -1; o.editDist = 0; o.pad_ = 0;
                W.alns[W.nAln] = o;
                for (int c = 0; c < chainLen; ++c) {
                    const int k = chain_at(W, a, nBack, c);
                    ChainSeed cs; cs.ref = W.mref[k]; cs.q = W.mq[k]; cs.len = W.ml[k];
                    W.chain[W.nChainOut + c] = cs;
                }
                W.nChainOut += chainLen;
                W.nAln += 1;
                W.accScore[nAcc] = score; W.accBegin[nAcc] = rB;
                ++nAcc;
            }
        }
        if (nBounds >= BOUND_CAP) { W.fail = 9; return best; }
        Trial t; t.refB = rB; t.refE = rE; t.qB = qB; t.qE = qE;
        W.bounds[nBounds++] = t;
    }
    return best >= minScore ? best : -1;
}

// processLisIntervals src/mapper.cpp:1118-1204 over the read's regions in W.rorder
template <class Env>
MC_FN void run_regions(Env& env, Work& W, const Options& O, const ReadParams& R, int nRegions, int maxAln, int minCoverage, int minScore) {
    int trial[2] = {0, 0};            // [0] reverse strand, [1] forward strand
    W.treeN[0] = W.treeN[1] = 0;
    int bestSoFar = 0;
    for (int li = 0; li < nRegions && W.nAln < maxAln && !W.fail; ++li) {
        int n, score;
        bool fw;
        MC_T0(t_lr);
        if (!env.load_region(W, W.rorder[li], n, fw, score)) { W.fail = 10; return; }
        MC_ADD(env, 8, t_lr);
        const int s = fw ? 1 : 0;
        MC_T0(t_cn);
        const bool unique = covers_new_bases(W, s, n);
        MC_ADD(env, 9, t_cn);
        if (W.fail) return;
        if ((trial[s] < O.maxTrial || unique) && (n > 1 || score >= minCoverage)) {
            int64_t cs, ce;
            env.chrom_bounds(W.mref[0], cs, ce);
            MC_T0(t_tr);
            const int ret = try_region(env, W, O, R, n, fw, minScore, minCoverage, cs, ce);
            MC_ADD(env, 10, t_tr);
            if (W.fail) return;
            trial[s]++;
            if (ret >= 0 && (!O.maxTrialBest || bestSoFar < ret)) trial[s] = 0;
            bestSoFar = bestSoFar > ret ? bestSoFar : ret;
        }
    }
}

// bandSizeEstimate src/mapper.cpp:44-52
MC_FN int band_estimate(int fixedBand, int qLen, int refLen, int matchScore, int gapOpen, int gapExtend, int alnScore) {
    const int d = qLen - refLen > refLen - qLen ? qLen - refLen : refLen - qLen;
    const int minBand = d + 3;
    const int scoreBand = ((qLen < refLen ? qLen : refLen) * matchScore - alnScore + gapOpen) / (-1 * gapExtend) + 1;
    const int maxGapBand = minBand - 3 + ((qLen >> 1) * matchScore + gapOpen) / (-1 * gapExtend) + 1;
    int band = maxGapBand < scoreBand ? maxGapBand : scoreBand;
    band = fixedBand < band ? fixedBand : band;
    band = minBand > band ? minBand : band;
    return band;
}

MC_FN bool push_op(Work& W, int32_t len, char code) {
    if (W.nOps >= W.opsCap) return false;
    CigOp o; o.len = len; o.code = (int32_t)code;
    W.ops[W.nOps++] = o;
    return true;
}

// one ksw_global call of constructCIGAR (src/mapper.cpp:200,239,256) and recoverCIGAR4 (:80-117) on its result.  The
// environment returns the call's alignment as EXTENDED operations already -- len << 4 | code, code 0 '=', 1 'I', 2 'D', 3 'X',
// adjacent operations of one kind merged, in order: the BAM operations with every M run split into maximal runs of equal /
// unequal residues (nt4 codes are compared: n equals n) -- env.xop(W, first + k), k < n.  The device knows a cell's equality
// while it fills the matrix and its walk writes these runs directly; the CPU build expands the oracle's BAM operations.
// Adds mismatches + gap bases to `edit`.
constexpr uint32_t XOP_EQ = 0, XOP_INS = 1, XOP_DEL = 2, XOP_X = 3;
template <class Env>
MC_FN bool dp_piece(Env& env, Work& W, bool fw, int qFrom, int qLen, int64_t tFrom, int tLen, int band, int& edit) {
    int first = 0;
    MC_T0(t_dp);
    const int n = env.global_cigar(W, fw, qFrom, qLen, tFrom, tLen, band, first);
    if (n < 0) return false;
    for (int k = 0; k < n; ++k) {
        const uint32_t c = env.xop(W, first + k);
        const uint32_t code = c & 0xf;
        const int len = (int)(c >> 4);
        if (!push_op(W, len, "=IDX"[code])) return false;
        if (code != XOP_EQ) edit += len;
    }
    MC_ADD(env, 13, t_dp);
    return true;
}

// constructCIGAR src/mapper.cpp:172-271 for one alignment: soft clips, the chain's seeds as '=' runs, the closed-form gap
// cases (:207-243, mirroring the scoring cases), one ksw_global per remaining gap and for the head and the tail
template <class Env>
MC_FN bool cigar_one(Env& env, Work& W, const Options& O, const ReadParams& R, AlnOut& a) {
    const bool fw = a.fw != 0;
    int edit = 0;
    a.opFirst = W.nOps;
    if (a.qB && !push_op(W, a.qB, 'S')) return false;
    const int c0 = a.chainFirst, cn = a.chainCount;
    // (the chain lies in the environment's memory: every seed is read once, the next one a piece ahead of its use)
    ChainSeed cur = W.chain[c0];
    if (cur.q > a.qB) {
        const int qLen = cur.q - a.qB, refLen = (int)(cur.ref - a.refB);
        const int band = band_estimate(O.fixedBandSize, qLen, refLen, O.match, O.openGap, O.extendGap, a.score);
        if (!dp_piece(env, W, fw, a.qB, qLen, a.refB, refLen, band, edit)) return false;
    }
    if (!push_op(W, cur.len, '=')) return false;
    ChainSeed nxt = cn > 1 ? W.chain[c0 + 1] : cur;
    for (int i = 1; i < cn; ++i) {
        const ChainSeed prv = cur;
        cur = nxt;
        if (i + 1 < cn) nxt = W.chain[c0 + i + 1];
        int qDist = cur.q - (prv.q + prv.len - 1);
        int refDist = (int)(cur.ref - (prv.ref + prv.len - 1));
        if (qDist < 0 || refDist < 0) {
            if (qDist >= refDist) qDist -= refDist; else refDist -= qDist;
        }
        bool ok = true;
        if (qDist <= 0) { ok = push_op(W, refDist, 'D') && push_op(W, cur.len - 1 + qDist, '='); edit += refDist; }
        else if (refDist <= 0) { ok = push_op(W, qDist, 'I') && push_op(W, cur.len - 1 + refDist, '='); edit += qDist; }
        else if (qDist == 1) { ok = push_op(W, refDist - 1, 'D') && push_op(W, cur.len, '='); edit += refDist - 1; }
        else if (refDist == 1) { ok = push_op(W, qDist - 1, 'I') && push_op(W, cur.len, '='); edit += qDist - 1; }
        else if (refDist == 2 && qDist == 2) { ok = push_op(W, 1, 'X') && push_op(W, cur.len, '='); edit += 1; }
        else {
            const int qLen = cur.q - 1 - (prv.q + prv.len) + 1;
            const int refLen = (int)(cur.ref - 1 - (prv.ref + prv.len) + 1);
            const int band = band_estimate(O.fixedBandSize, qLen, refLen, O.match, O.openGap, O.extendGap, a.score);
            ok = dp_piece(env, W, fw, prv.q + prv.len, qLen, prv.ref + prv.len, refLen, band, edit) && push_op(W, cur.len, '=');
        }
        if (!ok) return false;
    }
    if (cur.q + cur.len - 1 < a.qE) {
        const int qLen = a.qE - (cur.q + cur.len) + 1, refLen = (int)(a.refE - (cur.ref + cur.len) + 1);
        const int band = band_estimate(O.fixedBandSize, qLen, refLen, O.match, O.openGap, O.extendGap, a.score);
        if (!dp_piece(env, W, fw, cur.q + cur.len, qLen, cur.ref + cur.len, refLen, band, edit)) return false;
    }
    if (R.plen - 1 - a.qE && !push_op(W, R.plen - 1 - a.qE, 'S')) return false;
    a.opCount = W.nOps - a.opFirst;
    a.editDist = edit;
    return true;
}

// read_t::postprocess src/mapper.cpp:322-379, the part in front of the MAPQ: the alignments ordered by score (std::sort with
// compAlignments :295 -- at most ALN_CAP = 8 elements, where libstdc++ runs a plain insertion sort: stable), extended CIGARs
// for the first O.cigarCount of them.  An alignment whose CIGAR cannot be made here keeps opCount = -1.
template <class Env>
MC_FN void build_cigars(Env& env, Work& W, const Options& O, const ReadParams& R) {
    W.nOps = 0;
    const int n = W.nAln;
    for (int k = 0; k < n; ++k) W.aorder[k] = (uint16_t)k;
    for (int i = 1; i < n; ++i) {                    // insertion sort by score, descending, stable
        const uint16_t v = W.aorder[i];
        int j = i;
        while (j > 0 && W.alns[W.aorder[j - 1]].score < W.alns[v].score) { W.aorder[j] = W.aorder[j - 1]; --j; }
        W.aorder[j] = v;
    }
    const int want = n < O.cigarCount ? n : O.cigarCount;
    for (int j = 0; j < want; ++j) {
        AlnOut& a = W.alns[W.aorder[j]];
        const int mark = W.nOps;
        if (!cigar_one(env, W, O, R, a)) { W.nOps = mark; a.opFirst = 0; a.opCount = -1; a.editDist = 0; }
    }
}

// unpairedMatch src/mapper.cpp:1269-1302 from the candidate regions on (W.rscore[k] = score of region k, k in the order
// calculateCandRegions creates them: forward strand first); returns DONE or NEED_CALLER
template <class Env>
MC_FN int map_read(Env& env, Work& W, const Options& O, const ReadParams& R, int nRegions) {
    W.nAln = 0; W.nChainOut = 0; W.nOps = 0; W.fail = 0;
    if (nRegions > REGION_CAP2) { W.fail = 3; return NEED_CALLER; }
    for (int k = 0; k < nRegions; ++k) W.rorder[k] = (uint16_t)k;
    const int32_t* sc = W.rscore;
    if (!sort_like_libstdcxx(W.rorder, nRegions, W.bam, BAM_CAP / 3, [sc](uint16_t x, uint16_t y) { return sc[x] > sc[y]; })) { W.fail = 4; return NEED_CALLER; }   // compIntervals :1114
    if (nRegions == 0) return DONE;                          // :1272: no regions, no rescue
    if (!O.rescueOnly) {
        run_regions(env, W, O, R, nRegions, O.maxAlnCount, R.minCoverage, R.minScore);
        if (W.fail) return NEED_CALLER;
    }
    if (W.nAln == 0 && (O.rescue || O.rescueOnly)) {
        run_regions(env, W, O, R, nRegions, 20, R.minCoverage2, R.minScore2);
        if (W.fail) return NEED_CALLER;
        if (W.nAln == 0 && !O.rescueOnly) { W.fail = 11; return NO_ALIGNMENT; }      // RESCUE2 (:1290-1300) is a call of its own
    }
    MC_T0(t_bc);
    if (O.cigarCount > 0) build_cigars(env, W, O, R);
    MC_ADD(env, 14, t_bc);
    return DONE;
}

}  // namespace mapcore
}  // namespace alfa
