// dp_oracle.cpp -- CPU restatement of the reference's dynProg class (src/dp.cpp), SURVEY rows a13/a14.
//
// TEST INFRASTRUCTURE ONLY (see alfa_oracle.h): the product never links or loads this file.
// Pinned against the compiled reference (oracle/_ref/libalfalfa_ref.so, ref_dynprog) by tools/make_goldens.py and
// tests/golden/dpband.npz.
//
// Faithful to the reference including its boundary rules (generic UP/LEFT values on the band edges, src/dp.cpp:226,247),
// its tie order in the traceback (= then X then D then I, :369-387) and its start-cell search (:133-184).  Two behaviours of
// the reference cannot be reproduced as values and are reported through `ret` instead:
//   ret 2  the reference reads a matrix cell this call never wrote (its matrices are uninitialised `new int[]`; happens
//          e.g. in findTraceBackPosStatic with an asymmetric band, :164-181) -- result undefined there
//   ret 3  the reference prints "Error in dynamic programming" and calls exit(1) (:402,:453,:505,:557,:866,:1006)
// ret 0/1 are the reference's own return values (1 = dpBandStatic gave up, :679-684).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "alfa_oracle.h"

namespace {

struct Cnt { int64_t cells = 0; };

struct Mat {                       // one of M / UP / LEFT with "written in this call" tracking
    int W = 0;
    std::vector<int> v;
    std::vector<uint8_t> w;
    bool* undef = nullptr;
    void init(int L2, int L1, bool* u) { W = L1 + 2; v.assign((size_t)(L2 + 1) * W, 0); w.assign(v.size(), 0); undef = u; }
    void set(long i, long j, int x) { v[(size_t)i * W + j] = x; w[(size_t)i * W + j] = 1; }
    int get(long i, long j) const {
        if (!w[(size_t)i * W + j]) *undef = true;
        return v[(size_t)i * W + j];
    }
};

struct DP {
    // dynProg's fields (src/dp.h:79-90)
    Mat M, UP, LEFT;
    int L1 = 0, L2 = 0, bandSize = 0, bandLeft = 0, bandRight = 0;
    bool banded = false, undef = false, failed = false;
    int match, mismatch, openGap, extendGap;
    const char* ref; const char* query;
    int64_t cells = 0;

    // dp_scores::updateScoreMatrixDna src/dp.cpp:83-94
    int sc(unsigned char a, unsigned char b) const {
        const bool up = a >= 'A' && a <= 'Z' && b >= 'A' && b <= 'Z';
        const bool lo = a >= 'a' && a <= 'z' && b >= 'a' && b <= 'z';
        if (up || lo) return a == b ? match : mismatch;
        return -200000;
    }
    bool affine() const { return openGap != 0; }

    // updateMatrix src/dp.cpp:96-129
    void update_matrix(bool freeRefB, bool freeQueryB, bool backward) {
        M.init(L2, L1, &undef);
        if (affine()) { UP.init(L2, L1, &undef); LEFT.init(L2, L1, &undef); }
        M.set(0, 0, 0);
        for (int i = 1; i <= L2; i++) M.set(i, 0, (freeQueryB && !backward) ? 0 : openGap + i * extendGap);
        for (int i = 1; i <= L1; i++) M.set(0, i, (freeRefB && !backward) ? 0 : openGap + i * extendGap);
        if (affine()) {
            UP.set(0, 0, 0); LEFT.set(0, 0, 0);
            for (int i = 1; i <= L1; i++) { UP.set(0, i, M.get(0, i) + openGap); LEFT.set(0, i, M.get(0, i) + openGap); }
            for (int i = 1; i <= L2; i++) { UP.set(i, 0, M.get(i, 0) + openGap); LEFT.set(i, 0, M.get(i, 0) + openGap); }
        }
    }

    // dpFillStatic / dpFillStaticBackward / dpFillOptStatic / dpFillOptStaticBackward src/dp.cpp:209-355
    // fwd: characters ref[refB+j-1], query[queryB+i-1]; backward: ref[refE-j+1], query[queryE-i+1]; `clamp` = the
    // forward fills' "local" rule (:214,:301): scores below 0 are raised to 0
    void fill(bool forward, long refB, long refE, int queryB, int queryE, bool clamp) {
        const int bandL = std::max(bandLeft, bandSize), bandR = std::max(bandRight, bandSize);
        const int e = extendGap, o = openGap;
        auto d_at = [&](long i, long j) {
            return forward ? sc((unsigned char)ref[refB + j - 1], (unsigned char)query[queryB + i - 1])
                           : sc((unsigned char)ref[refE - j + 1], (unsigned char)query[queryE - i + 1]);
        };
        auto cl = [&](int x) { return clamp ? std::max(x, 0) : x; };
        for (long i = 1; i <= L2; i++) {
            long idx2 = i - bandL;
            const long colRB = i + bandR;
            if (idx2 > 0) {
                // (the reference also computes this cell when idx2 > L1, i.e. right of the matrix; nothing reads it)
                if (idx2 <= L1) {
                    cells++;
                    const int d = d_at(i, idx2);
                    if (affine()) {
                        int m = M.get(i - 1, idx2 - 1) + d;
                        const int up = std::max(M.get(i - 1, idx2) + e + o, UP.get(i - 1, idx2) + e);
                        UP.set(i, idx2, up);
                        m = std::max(m, up);
                        LEFT.set(i, idx2, m + e + 2 * o);
                        M.set(i, idx2, cl(m));
                    } else {
                        const int m = std::max(M.get(i - 1, idx2 - 1) + d, M.get(i - 1, idx2) + e);
                        M.set(i, idx2, cl(m));
                    }
                }
                idx2++;
            }
            for (long j = std::max(1L, idx2); j <= std::min((long)L1, colRB - 1); j++) {
                cells++;
                const int d = d_at(i, j);
                if (affine()) {
                    const int up = std::max(M.get(i - 1, j) + e + o, UP.get(i - 1, j) + e);
                    UP.set(i, j, up);
                    int m = M.get(i - 1, j - 1) + d;
                    const int left = std::max(M.get(i, j - 1) + e + o, LEFT.get(i, j - 1) + e);
                    LEFT.set(i, j, left);
                    m = std::max(m, left);
                    m = std::max(m, up);
                    M.set(i, j, cl(m));
                } else {
                    int m = std::max(M.get(i - 1, j - 1) + d, M.get(i, j - 1) + e);
                    m = std::max(m, M.get(i - 1, j) + e);
                    M.set(i, j, cl(m));
                }
            }
            if (colRB <= L1) {
                cells++;
                const long c = colRB;
                const int d = d_at(i, c);
                if (affine()) {
                    int m = M.get(i - 1, c - 1) + d;
                    const int left = std::max(M.get(i, c - 1) + e + o, LEFT.get(i, c - 1) + e);
                    LEFT.set(i, c, left);
                    m = std::max(m, left);
                    UP.set(i, c, m + e + 2 * o);
                    M.set(i, c, cl(m));
                } else {
                    const int m = std::max(M.get(i - 1, c - 1) + d, M.get(i, c - 1) + e);
                    M.set(i, c, cl(m));
                }
            }
        }
    }

    // findTraceBackPosStatic src/dp.cpp:133-184
    void find_start(int& i, int& j, bool fRB, bool fRE, bool fQB, bool fQE, bool local) {
        int maximum = M.get(i, j);
        const int bandL = std::max(bandLeft, bandSize), bandR = std::max(bandRight, bandSize);
        if ((fQE && fRE) || (fQB && fRB) || local) {
            for (int idx = L2; idx >= 0; idx--)
                for (int idx2 = std::min(idx + bandR, L1); idx2 >= std::max(0, idx - bandL); idx2--)
                    if (M.get(idx, idx2) > maximum) { maximum = M.get(idx, idx2); i = idx; j = idx2; }
        } else if (fQE || fQB) {
            for (int idx = std::min(L2, L1 + bandR); idx >= std::min(L2, std::max(0, L1 - bandL)); idx--)
                if (M.get(idx, j) > maximum) { maximum = M.get(idx, j); i = idx; }
        } else if (fRE || fRB) {
            for (int idx = std::min(L1, L2 + bandR); idx >= std::min(L1, std::max(0, L2 - bandL)); idx--)
                if (M.get(i, idx) > maximum) { maximum = M.get(i, idx); j = idx; }
        }
    }

    // the four band-aware tracebacks src/dp.cpp:357-561.  `both` = loop while (i>0 || j>0) (Static) vs (i>0 && j>0) (Full);
    // `loc` keeps the linear-gap loops running "while local" (:468,:520) -- they can only end through the error exit.
    void traceback(int& i, int& j, bool both, bool loc, bool forward, long refB, long refE, int queryB, int queryE, int& editDist,
                   std::string& ss) {
        const int bandL = std::max(bandLeft, bandSize), bandR = std::max(bandRight, bandSize);
        const bool aff = affine();
        while ((both ? (i > 0 || j > 0) : (i > 0 && j > 0)) || (!aff && loc)) {
            bool equal = false;
            if (j > 0 && i > 0)
                equal = forward ? ref[refB + j - 1] == query[queryB + i - 1] : ref[refE - j + 1] == query[queryE - i + 1];
            if (j > 0 && i > 0 && M.get(i, j) == M.get(i - 1, j - 1) + match && equal) { ss += '='; i--; j--; }
            else if (j > 0 && i > 0 && M.get(i, j) == M.get(i - 1, j - 1) + mismatch) { ss += 'X'; editDist++; i--; j--; }
            else if (j > 0 && j > i - bandL && ((aff ? M.get(i, j) == LEFT.get(i, j) : M.get(i, j) == M.get(i, j - 1) + extendGap) || i == 0)) {
                ss += 'D'; editDist++; j--;
            } else if (i > 0 && j < i + bandR && ((aff ? M.get(i, j) == UP.get(i, j) : M.get(i, j) == M.get(i - 1, j) + extendGap) || j == 0)) {
                ss += 'I'; editDist++; i--;
            } else { failed = true; return; }
        }
    }

    // dpTraceBackScore src/dp.cpp:839-870 (no band guards, affine matrices, edit distance only)
    void traceback_score(int& i, int& j, long refB, int queryB, int& editDist) {
        while (i > 0 && j > 0) {
            const bool equal = ref[refB + j - 1] == query[queryB + i - 1];
            if (M.get(i, j) == M.get(i - 1, j - 1) + match && equal) { i--; j--; }
            else if (M.get(i, j) == M.get(i - 1, j - 1) + mismatch) { editDist++; i--; j--; }
            else if (M.get(i, j) == LEFT.get(i, j)) { editDist++; j--; }
            else if (M.get(i, j) == UP.get(i, j)) { editDist++; i--; }
            else { failed = true; return; }
        }
    }
};

// run-length encoding of the error string src/dp.cpp:741-751
void rle(const std::string& es, std::vector<char>& chars, std::vector<int>& lens) {
    size_t it = 0;
    while (it < es.size()) {
        size_t t = it;
        while (it + 1 < es.size() && es[it + 1] == es[t]) it++;
        it++;
        chars.push_back(es[t]);
        lens.push_back((int)(it - t));
    }
}

int one(int variant, const alfa_dpband_scores* s, const alfa_dpband_task& t, alfa_dpband_result& r, std::vector<char>& chars,
        std::vector<int>& lens, int64_t& cells) {
    DP dp;
    dp.match = s->match; dp.mismatch = s->mismatch; dp.openGap = s->openGap; dp.extendGap = s->extendGap;
    dp.ref = t.ref; dp.query = t.query;
    long refB = t.refB, refE = t.refE;
    int queryB = t.queryB, queryE = t.queryE;
    const bool fRB = t.freeRefB, fRE = t.freeRefE, fQB = t.freeQueryB, fQE = t.freeQueryE, local = t.local;
    dp.L1 = (int)(refE - refB + 1);
    dp.L2 = queryE - queryB + 1;
    int editDist = 0, dpScore = 0, ret = 0;
    std::string es;
    bool want_cigar = t.out_type == 3 || t.out_type == 4;
    const bool aff = dp.affine();
    if (variant == ALFA_DPBAND_STATIC) {                                   // dpBandStatic src/dp.cpp:648-759
        dp.bandSize = t.band1;
        dp.banded = true;
        const int L1 = dp.L1, L2 = dp.L2;
        if ((fQB && !fRB && L2 + dp.bandSize < L1) || (fRB && !fQB && L2 - dp.bandSize > L1) ||
            (!(fQB || fRB) && !fRE && L2 + dp.bandSize < L1) || (!(fQB || fRB) && !fQE && L2 - dp.bandSize > L1)) {
            if (std::max(L2, L1) - std::min(L2, L1) < t.band2) dp.bandSize = std::max(L2, L1) - std::min(L2, L1);
            else {
                r.ret = 1; r.dpScore = s->mismatch * (int)t.query_len; r.editDist = (int)t.query_len;
                r.refB = refB; r.refE = refE; r.queryB = queryB; r.queryE = queryE; r.n_cigar = 0;
                return 1;
            }
        }
        const bool backward = (fQB || fRB) && !local;
        dp.update_matrix(fRB, fQB, backward);
        dp.fill(!backward, refB, refE, queryB, queryE, !backward && (local || (fQB && fRB)));
        int i = std::min(L2, L1 + dp.bandSize), j = std::min(L1, L2 + dp.bandSize);
        dp.find_start(i, j, fRB, fRE, fQB, fQE, local);
        dpScore = dp.M.get(i, j);
        if (t.out_type > 0) {
            if (fQE && i < L2) queryE = queryB + i - 1;
            if (fRE && j < L1) refE = refB + j - 1;
            if (fQB && i < L2) queryB = queryE - i + 1;
            if (fRB && j < L1) refB = refE - j + 1;
            dp.traceback(i, j, true, local || (fQB && fRB), !backward, refB, refE, queryB, queryE, editDist, es);
            if (!dp.failed && want_cigar) {
                if (!backward) std::reverse(es.begin(), es.end());
                rle(es, chars, lens);
            }
        }
    } else if (variant == ALFA_DPBAND_FULL || variant == ALFA_DPBAND_SCORE) {   // dpBandFull :761-837, dpBandScore :872-924
        dp.bandSize = 0; dp.banded = true; dp.bandLeft = t.band1; dp.bandRight = t.band2;
        const int L1 = dp.L1, L2 = dp.L2;
        dp.update_matrix(fRB, fQB, false);
        dp.fill(true, refB, refE, queryB, queryE, local || (fQB && fRB));
        int i = std::min(L2, L1 + dp.bandLeft), j = std::min(L1, L2 + dp.bandRight);
        dp.find_start(i, j, fRB, fRE, fQB, fQE, local);
        dpScore = dp.M.get(i, j);
        if (t.out_type > 0) {
            if (fQE && i < L2) queryE = queryB + i - 1;
            if (fRE && j < L1) refE = refB + j - 1;
            if (variant == ALFA_DPBAND_FULL) dp.traceback(i, j, false, local || (fQB && fRB), true, refB, refE, queryB, queryE, editDist, es);
            else dp.traceback_score(i, j, refB, queryB, editDist);
            if (!dp.failed) {
                if (i > 0 && !fQB) editDist += i;
                if (j > 0 && !fRB) editDist += j;
                if (variant == ALFA_DPBAND_FULL && want_cigar) {
                    if (i > 0 && !fQB) es.append((size_t)i, 'I');
                    else if (j > 0 && !fRB) es.append((size_t)j, 'D');
                    std::reverse(es.begin(), es.end());
                    rle(es, chars, lens);
                }
            }
        }
        if (!dp.failed) {
            if (fQB && i > 0) queryB = queryB + i;
            if (fRB && j > 0) refB = refB + j;
        }
    } else {                                                               // dpBasic src/dp.cpp:926-1040
        const int L1 = dp.L1, L2 = dp.L2;
        dp.update_matrix(fRB, fQB, false);
        // :937-941 rewrites the borders with the same rule; the fill is the unbanded linear-gap recurrence
        for (long i = 1; i <= L2; i++)
            for (long j = 1; j <= L1; j++) {
                dp.cells++;
                const int d = dp.sc((unsigned char)t.ref[refB + j - 1], (unsigned char)t.query[queryB + i - 1]);
                int m = std::max(dp.M.get(i - 1, j - 1) + d, dp.M.get(i, j - 1) + s->extendGap);
                m = std::max(m, dp.M.get(i - 1, j) + s->extendGap);
                dp.M.set(i, j, m);
            }
        int i = L2, j = L1;
        int maximum = dp.M.get(i, j);
        for (int idx = L1; idx >= 0; idx--)
            if (dp.M.get(i, idx) > maximum) { maximum = dp.M.get(i, idx); j = idx; }
        dpScore = dp.M.get(i, j);
        if (t.out_type > 0) {
            if (fRE && j < L1) refE = refB + j - 1;
            while (i > 0) {
                bool equal = false;
                if (j > 0 && i > 0) equal = t.ref[refB + j - 1] == t.query[queryB + i - 1];
                if (j > 0 && i > 0 && dp.M.get(i, j) == dp.M.get(i - 1, j - 1) + s->match && equal) { es += '='; i--; j--; }
                else if (j > 0 && i > 0 && dp.M.get(i, j) == dp.M.get(i - 1, j - 1) + s->mismatch) { es += 'X'; editDist++; i--; j--; }
                else if (j > 0 && (dp.M.get(i, j) == dp.M.get(i, j - 1) + s->extendGap || i == 0)) { es += 'D'; editDist++; j--; }
                else if (i > 0 && (dp.M.get(i, j) == dp.M.get(i - 1, j) + s->extendGap || j == 0)) { es += 'I'; editDist++; i--; }
                else { dp.failed = true; break; }
            }
            if (!dp.failed) {
                if (i > 0 && !fQB) editDist += i;
                if (j > 0 && !fRB) editDist += j;
                if (want_cigar) {
                    if (i > 0 && !fQB) es.append((size_t)i, 'I');
                    else if (j > 0 && !fRB) es.append((size_t)j, 'D');
                    std::reverse(es.begin(), es.end());
                    rle(es, chars, lens);
                }
            }
        }
        if (!dp.failed && fRB && j > 0) refB = refB + j;
    }
    cells += dp.cells;
    (void)aff;
    if (dp.undef || dp.failed) {
        chars.clear(); lens.clear();
        memset(&r, 0, sizeof(r));
        r.ret = dp.undef ? 2 : 3;
        return r.ret;
    }
    r.ret = ret; r.dpScore = dpScore; r.editDist = editDist;
    r.refB = refB; r.refE = refE; r.queryB = queryB; r.queryE = queryE;
    return ret;
}

}  // namespace

// the argument checks every implementation of the boundary shares (the reference asserts or reads out of bounds there)
extern "C" int orc_dp_band_check(int variant, const alfa_dpband_scores* s, const alfa_dpband_task* t) {
    if (variant < 0 || variant > 3 || !s || !t) return -22;
    if ((s->affine != 0) != (s->openGap != 0)) return -22;                     // matrices are allocated from `affine`, used from openGap
    if (variant == ALFA_DPBAND_SCORE && s->openGap == 0) return -22;           // dpTraceBackScore reads UP/LEFT (:854,:858)
    const int64_t L1 = t->refE - t->refB + 1, L2 = (int64_t)t->queryE - t->queryB + 1;
    if (L1 < 0 || L2 < 0 || t->refB < 0 || t->queryB < 0 || t->refE >= t->ref_len || t->queryE >= t->query_len) return -22;
    if (variant == ALFA_DPBAND_STATIC) {
        if (t->band1 <= 0 || t->local) return -22;                              // asserts :662-663
        if ((t->freeQueryB || t->freeRefB) && (t->freeQueryE || t->freeRefE)) return -22;   // assert :664
    } else if (variant != ALFA_DPBAND_BASIC) {
        if (t->band1 <= 0 || t->band2 <= 0) return -22;                         // assert :776,:887
    }
    for (int64_t k = t->refB; k <= t->refE; k++) if ((unsigned char)t->ref[k] >= 127) return -22;          // scoreMatrix[127][127]
    for (int64_t k = t->queryB; k <= t->queryE; k++) if ((unsigned char)t->query[k] >= 127) return -22;
    return 0;
}

extern "C" int orc_dp_band_batch(int variant, const alfa_dpband_scores* s, const alfa_dpband_task* t, int n, alfa_dpband_result* r,
                                 char** cig_chars, int32_t** cig_lens) {
    std::vector<char> allc;
    std::vector<int> alll;
    int64_t cells = 0;
    for (int k = 0; k < n; k++) {
        int rc = orc_dp_band_check(variant, s, &t[k]);
        if (rc) return rc;
        std::vector<char> c;
        std::vector<int> l;
        memset(&r[k], 0, sizeof(r[k]));
        one(variant, s, t[k], r[k], c, l, cells);
        r[k].n_cigar = (int)c.size();
        r[k].cig_off = (int64_t)allc.size();
        allc.insert(allc.end(), c.begin(), c.end());
        alll.insert(alll.end(), l.begin(), l.end());
    }
    orc_counters_add_dpband(cells);
    *cig_chars = (char*)malloc(allc.size() + 1);
    *cig_lens = (int32_t*)malloc(sizeof(int32_t) * (alll.size() + 1));
    if (!allc.empty()) { memcpy(*cig_chars, allc.data(), allc.size()); memcpy(*cig_lens, alll.data(), sizeof(int32_t) * alll.size()); }
    return 0;
}
