// oracle/alfa_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see alfa_oracle.h).
//
// Scalar CPU restatement of ALFALFA's seed retrieval (enhanced sparse suffix
// array search) and ksw banded DP, written against SURVEY.md / the reference
// sources as a specification.  Every function cites the reference file:line it
// follows.  Semantics (including quirks) are reproduced, code is not: the index
// is a plain C view, intervals are three scalars, results go to a growable POD
// array, and every index-array access goes through a counting accessor that
// implements the "algorithmic bytes" census of SURVEY.md 8d.
#include "alfa_oracle.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

thread_local orc_counters g_cnt;

const int KMER = 10;                  // sparseSA.h:36
const uint32_t TABLE = 1u << (2 * KMER);  // sparseSA.h:37

// sparseSA.h:38-63 (BITADD): a/A=0 c/C=1 g/G=2 t/T=3, everything else all-ones
inline uint32_t bitadd(unsigned char c) {
    switch (c) {
        case 'a': case 'A': return 0u;
        case 'c': case 'C': return 1u;
        case 'g': case 'G': return 2u;
        case 't': case 'T': return 3u;
        default: return 0xFFFFFFFFu;
    }
}

struct Ival {  // interval_t, sparseSA.h:120-128
    int64_t start, end;
    int depth;
    int64_t size() const { return end - start + 1; }
};

struct Pat {  // the query string; reads past the end return '\0' like std::string::operator[] at size()
    const char* p;
    int len;
    char at(int64_t i) const {
        g_cnt.p_bytes++;
        return (i >= 0 && i < len) ? p[i] : '\0';
    }
};

struct Idx {
    const alfa_index_view* v;
    int64_t n;
    // counting accessors
    int64_t sa(int64_t i) const { g_cnt.sa_reads++; return (int64_t)v->SA[i]; }
    int child(int64_t i) const { g_cnt.child_reads++; return (i >= 0 && i < n) ? v->CHILD[i] : -1; }
    // vec_uchar::operator[] sparseSA.h:79-84: 255 -> lower_bound in the overflow table
    int lcp(int64_t i) const {
        g_cnt.lcp_reads++;
        if (i < 0 || i >= n) return 0;  // reference reads out of bounds here only when the value is unused
        uint8_t b = v->LCP[i];
        if (b != 255) return b;
        const int64_t* lo = std::lower_bound(v->LCP_M_idx, v->LCP_M_idx + v->nM, i);
        return v->LCP_M_val[lo - v->LCP_M_idx];
    }
    char s(int64_t i) const {
        g_cnt.s_bytes++;
        return (i >= 0 && i < v->N + 16) ? (char)v->S[i] : '\0';
    }
};

typedef std::vector<alfa_match> MatchVec;

// top_down_faster  sparseSA.cpp:614-666 (binary search without a child table)
bool td_binary(const Idx& X, char c, int64_t i, int64_t& start, int64_t& end) {
    int64_t l, r, m, r2 = end, l2 = start, vgl;
    bool found = false;
    int64_t cmp_first = (int64_t)c - (int64_t)X.s(X.sa(start) + i);
    int64_t cmp_last = (int64_t)c - (int64_t)X.s(X.sa(end) + i);
    if (cmp_first < 0) { l = start + 1; l2 = start; }
    else if (cmp_last > 0) { l = end + 1; l2 = end; }
    else {
        l = start; r = end;
        if (cmp_first == 0) { found = true; r2 = r; }
        else {
            while (r - l > 1) {
                m = (l + r) / 2;
                vgl = (int64_t)c - (int64_t)X.s(X.sa(m) + i);
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
                vgl = (int64_t)c - (int64_t)X.s(X.sa(m) + i);
                if (vgl < 0) r2 = m; else l2 = m;
            }
        }
    }
    start = l; end = l2;
    return l <= l2;
}

// top_down_child  sparseSA.cpp:671-697: walk the child list of `cur`, one S probe per child
bool td_child(const Idx& X, char c, Ival& cur) {
    int64_t left = cur.start;
    int64_t right = X.child(cur.end);
    if (cur.start >= right || right > cur.end) right = X.child(cur.start);
    if (X.s(X.sa(cur.start) + cur.depth) == c) { cur.end = right - 1; return true; }
    left = right;
    for (;;) {
        int nxt = X.child(right);
        if (!(nxt > right && X.lcp(right) == X.lcp(nxt))) break;
        right = nxt;
        if (X.s(X.sa(left) + cur.depth) == c) { cur.start = left; cur.end = right - 1; return true; }
        left = right;
    }
    if (X.s(X.sa(left) + cur.depth) == c) { cur.start = left; return true; }
    return false;
}

// the 10-mer table jump shared by traverse / traverse_faster  sparseSA.cpp:702-714, 734-746
// returns false when the caller must return immediately
bool kmer_jump(const Idx& X, const Pat& P, int prefix, Ival& cur, int min_len) {
    if (X.v->has_kmer && cur.depth == 0 && min_len >= KMER) {
        uint32_t index = 0;
        for (int i = 0; i < KMER; i++) index = (index << 2) | bitadd((unsigned char)P.at(prefix + i));
        if (index < TABLE) {
            g_cnt.kmr_reads++;
            uint32_t left = X.v->KMR[2 * (size_t)index], right = X.v->KMR[2 * (size_t)index + 1];
            if (right > 0) { cur.depth = KMER; cur.start = left; cur.end = right; return true; }
        }
        return false;
    }
    return true;
}

// LCP value of the lcp-interval [start,end]  (sparseSA.cpp:751-754, 768-771)
inline int ival_lcp(const Idx& X, const Ival& cur) {
    int ce = X.child(cur.end);
    if (cur.start < ce && ce <= cur.end) return X.lcp(ce);
    return X.lcp(X.child(cur.start));
}

// traverse  sparseSA.cpp:701-728
void traverse_plain(const Idx& X, const Pat& P, int prefix, Ival& cur, int min_len, int maxBr) {
    g_cnt.queries++;
    if (!kmer_jump(X, P, prefix, cur, min_len)) return;
    if (cur.depth >= min_len || cur.size() <= maxBr) return;
    while (prefix + cur.depth < P.len) {
        int64_t start = cur.start, end = cur.end;
        if (!td_binary(X, P.at(prefix + cur.depth), cur.depth, start, end)) return;
        cur.depth += 1; cur.start = start; cur.end = end;
        if (cur.depth == min_len || cur.size() <= maxBr) return;
    }
}

// traverse_faster  sparseSA.cpp:733-791
void traverse_child(const Idx& X, const Pat& P, int prefix, Ival& cur, int min_len, int maxBr) {
    g_cnt.queries++;
    if (!kmer_jump(X, P, prefix, cur, min_len)) return;
    if (cur.depth >= min_len || cur.size() <= maxBr) return;
    int64_t c = prefix + cur.depth;
    bool found = c < P.len;
    int curLCP = cur.size() > 1 ? ival_lcp(X, cur) : -1;  // only used when size()>1 (sparseSA.cpp:755)
    if (found && cur.size() > 1 && curLCP == cur.depth) found = td_child(X, P.at(c), cur);
    else if (found) found = P.at(c) == X.s(X.sa(cur.start) + cur.depth);
    bool mism = false;
    while (found && !mism && c < P.len && (cur.depth < min_len && cur.size() > maxBr)) {
        c++; cur.depth++;
        if (cur.start != cur.end) {
            int childLCP = ival_lcp(X, cur);
            int minimum = std::min(childLCP, min_len);
            while (!mism && c < P.len && cur.depth < minimum) {
                mism = X.s(X.sa(cur.start) + cur.depth) != P.at(c);
                c++; cur.depth += !mism;
            }
            found = c < P.len && !mism && cur.depth < min_len && td_child(X, P.at(c), cur);
        } else {
            while (!mism && c < P.len && cur.depth < min_len) {
                int64_t pos = X.sa(cur.start) + cur.depth;
                mism = pos >= X.v->N || X.s(pos) != P.at(c);
                c++; cur.depth += !mism;
            }
        }
    }
}

// traverse_keep  sparseSA.cpp:793-845 (SMEMP only): keeps the stack of visited intervals
void traverse_stack(const Idx& X, const Pat& P, int prefix, std::vector<Ival>& st, int min_len, int maxBr) {
    g_cnt.queries++;
    Ival cur = st.back();
    if (cur.size() > maxBr) st.pop_back();
    if (cur.depth >= min_len || cur.size() <= -1) return;
    int64_t c = prefix + cur.depth;
    bool found = c < P.len;
    int curLCP = cur.size() > 1 ? ival_lcp(X, cur) : -1;
    if (found && cur.size() > 1 && curLCP == cur.depth) {
        found = td_child(X, P.at(c), cur);
        if (found && cur.size() <= maxBr) st.push_back(cur);
    } else if (found) found = P.at(c) == X.s(X.sa(cur.start) + cur.depth);
    bool mism = false;
    while (found && !mism && c < P.len && (cur.depth < min_len && cur.size() > -1)) {
        c++; cur.depth++;
        if (cur.start != cur.end) {
            int childLCP = ival_lcp(X, cur);
            int minimum = std::min(childLCP, min_len);
            while (!mism && c < P.len && cur.depth < minimum) {
                mism = X.s(X.sa(cur.start) + cur.depth) != P.at(c);
                c++; cur.depth += !mism;
            }
            found = c < P.len && !mism && cur.depth < min_len && td_child(X, P.at(c), cur);
            if (found && cur.size() <= maxBr) st.push_back(cur);
        } else {
            while (!mism && c < P.len && cur.depth < min_len) {
                int64_t pos = X.sa(cur.start) + cur.depth;
                mism = pos >= X.v->N || X.s(pos) != P.at(c);
                c++; cur.depth += !mism;
            }
        }
    }
    if (!st.empty() && st.back().start == cur.start && st.back().end == cur.end && st.back().depth < cur.depth)
        st.back().depth = cur.depth;
}

inline void traverse_any(const Idx& X, const Pat& P, int prefix, Ival& cur, int min_len, int maxBr) {
    if (X.v->has_child) traverse_child(X, P, prefix, cur, min_len, maxBr);
    else traverse_plain(X, P, prefix, cur, min_len, maxBr);
}

// expand_link sparseSA.h:223-239 / suffixlink sparseSA.cpp:849-855
bool expand_link(const Idx& X, Ival& link) {
    int64_t thresh = 2 * (int64_t)link.depth * X.v->logN, exp = 0;
    int64_t start = link.start, end = link.end;
    while (X.lcp(start) >= link.depth) { exp++; if (exp >= thresh) return false; start--; }
    while (end < X.v->NKm1 && X.lcp(end + 1) >= link.depth) { exp++; if (exp >= thresh) return false; end++; }
    link.start = start; link.end = end;
    return true;
}
bool suffixlink(const Idx& X, Ival& m) {
    m.depth -= (int)X.v->K;
    if (m.depth <= 0) return false;
    m.start = X.v->ISA[X.sa(m.start) / X.v->K + 1];
    m.end = X.v->ISA[X.sa(m.end) / X.v->K + 1];
    return expand_link(X, m);
}

// find_LmaximalStrict  sparseSA.cpp:858-871: step left at most `border` chars; emit only on a left
// mismatch / text start / read start found INSIDE the loop.
void lmax_strict(const Idx& X, const Pat& P, int prefix, int64_t i, int len, MatchVec& out, int min_len, int border) {
    for (int k = 0; k < border; k++) {
        if (prefix == 0 || i == 0 || P.at(prefix - 1) != X.s(i - 1)) {
            if (len >= min_len) { alfa_match m; m.ref = i; m.query = prefix; m.len = len; out.push_back(m); g_cnt.matches++; }
            return;
        }
        prefix--; i--; len++;
    }
}

// collectAllSMAMs  sparseSA.cpp:901-907
void collect_smam(const Idx& X, const Pat& P, int prefix, const Ival& xmi, MatchVec& out, int min_len, int maxCount, int maxLeft) {
    if (xmi.size() <= maxCount)
        for (int64_t i = xmi.start; i <= xmi.end; i++) lmax_strict(X, P, prefix, X.sa(i), xmi.depth, out, min_len, maxLeft);
}

// collectAllMEMs  sparseSA.cpp:873-899
void collect_mem(const Idx& X, const Pat& P, int prefix, const Ival& mli, Ival xmi, MatchVec& out, int min_len, int maxLeft) {
    const int64_t n = X.n;
    int64_t xe = xmi.end;
    for (int64_t i = xmi.start; i <= xe; i++) lmax_strict(X, P, prefix, X.sa(i), xmi.depth, out, min_len, maxLeft);
    if (mli.start == xmi.start && mli.end == xmi.end) return;
    while (xmi.depth >= mli.depth) {
        if (xmi.end + 1 < n) xmi.depth = std::max(X.lcp(xmi.start), X.lcp(xmi.end + 1));
        else xmi.depth = X.lcp(xmi.start);
        if (xmi.depth >= mli.depth) {
            while (X.lcp(xmi.start) >= xmi.depth) {
                xmi.start--;
                lmax_strict(X, P, prefix, X.sa(xmi.start), xmi.depth, out, min_len, maxLeft);
            }
            while (xmi.end + 1 < n && X.lcp(xmi.end + 1) >= xmi.depth) {
                xmi.end++;
                lmax_strict(X, P, prefix, X.sa(xmi.end), xmi.depth, out, min_len, maxLeft);
            }
        }
    }
}

// findSMAM  sparseSA.cpp:1063-1114
void find_smam(const Idx& X, int k, const Pat& P, MatchVec& out, int min_len, int maxCount, int sparseQ) {
    const int K = (int)X.v->K;
    const int64_t n = X.n;
    if (k < 0 || k >= K) return;
    int prefix = k;
    Ival xmi = {0, n - 1, 0};
    int min_lenK = min_len - (sparseQ * K - 1);
    int minMin_len = sparseQ > 1 ? min_len - K : min_len;
    while (prefix <= P.len - min_lenK) {
        traverse_any(X, P, prefix, xmi, P.len, -1);
        if (xmi.depth <= 1) { xmi = {0, n - 1, 0}; prefix += sparseQ * K; continue; }
        if (xmi.depth >= min_lenK)
            collect_smam(X, P, prefix, xmi, out, minMin_len, sparseQ * maxCount, sparseQ * K);
        prefix += sparseQ * K;
        if (!X.v->has_suflink) { xmi = {0, n - 1, 0}; continue; }
        int i = 0; bool ok = true;
        while (i < sparseQ && (ok = suffixlink(X, xmi))) i++;
        if (!ok) { xmi = {0, n - 1, 0}; continue; }
    }
}

// findMEM  sparseSA.cpp:993-1060
void find_mem(const Idx& X, int k, const Pat& P, MatchVec& out, int min_len, int /*maxCount*/, int sparseQ) {
    const int K = (int)X.v->K;
    const int64_t n = X.n;
    if (k < 0 || k >= K) return;
    int prefix = k;
    Ival mli = {0, n - 1, 0}, xmi = {0, n - 1, 0};
    int min_lenK = min_len - (sparseQ * K - 1);
    while (prefix <= P.len - min_lenK) {
        traverse_any(X, P, prefix, mli, min_lenK, -1);
        if (mli.depth > xmi.depth) xmi = mli;
        if (mli.depth <= 1) { mli = {0, n - 1, 0}; xmi = {0, n - 1, 0}; prefix += sparseQ * K; continue; }
        if (mli.depth >= min_lenK) {
            traverse_any(X, P, prefix, xmi, P.len, -1);
            collect_mem(X, P, prefix, mli, xmi, out, min_len, sparseQ * K);
            prefix += sparseQ * K;
            if (!X.v->has_suflink) { mli = {0, n - 1, 0}; xmi = {0, n - 1, 0}; continue; }
            int i = 0; bool ok = true;
            while (i < sparseQ && (ok = suffixlink(X, mli))) { suffixlink(X, xmi); i++; }
            if (!ok) { mli = {0, n - 1, 0}; xmi = {0, n - 1, 0}; continue; }
        } else {
            prefix += sparseQ * K;
            if (!X.v->has_suflink) { mli = {0, n - 1, 0}; xmi = {0, n - 1, 0}; continue; }
            int i = 0; bool ok = true;
            while (i < sparseQ && (ok = suffixlink(X, mli))) i++;
            if (!ok) { mli = {0, n - 1, 0}; xmi = {0, n - 1, 0}; continue; }
            xmi = mli;
        }
    }
}

struct Tuple {  // intervTuple_t sparseSA.h:131-140
    Ival mli, xmi;
    int prefix;
    int xmiEnd() const { return prefix + xmi.depth; }
};

// collectXmiAndMli  sparseSA.cpp:910-966
void collect_xmi_mli(const Idx& X, const Pat& P, const Tuple& t, MatchVec& out, bool calcMli, int nextEndStop,
                     int nextEndStopPrefix, int min_len, int maxCount, int maxLeft) {
    const int64_t n = X.n;
    const int K = (int)X.v->K;
    int prefix = t.prefix;
    Ival xmi = t.xmi, mli = t.mli;
    if (xmi.depth + prefix < min_len) return;
    for (int64_t i = xmi.start; i <= xmi.end; i++) lmax_strict(X, P, prefix, X.sa(i), xmi.depth, out, min_len, maxLeft);
    if (!calcMli || (mli.start == xmi.start && mli.end == xmi.end)) return;
    if (xmi.end + 1 < n) xmi.depth = std::max(X.lcp(xmi.start), X.lcp(xmi.end + 1));
    else xmi.depth = X.lcp(xmi.start);
    while (xmi.size() < maxCount && prefix + xmi.depth > nextEndStop && xmi.depth >= mli.depth && xmi.depth + prefix >= min_len) {
        int64_t prevStart = xmi.start, prevStop = xmi.end;
        int64_t minStart = std::max((int64_t)0, xmi.start - (maxCount - xmi.size()));
        while (xmi.start >= minStart && X.lcp(xmi.start) >= xmi.depth) xmi.start--;
        int64_t maxEnd = std::min(n, xmi.end + 1 + (maxCount - xmi.size()));
        while (xmi.end + 1 < maxEnd && X.lcp(xmi.end + 1) >= xmi.depth) xmi.end++;
        if (xmi.size() < maxCount) {
            for (int64_t i = xmi.start; i < prevStart; i++) lmax_strict(X, P, prefix, X.sa(i), xmi.depth, out, min_len, maxLeft);
            for (int64_t i = prevStop + 1; i <= xmi.end; i++) lmax_strict(X, P, prefix, X.sa(i), xmi.depth, out, min_len, maxLeft);
            if (xmi.end + 1 < n) xmi.depth = std::max(X.lcp(xmi.start), X.lcp(xmi.end + 1));
            else xmi.depth = X.lcp(xmi.start);
        }
    }
    if (xmi.size() < maxCount && prefix + xmi.depth == nextEndStop && xmi.depth >= mli.depth && xmi.depth + prefix >= min_len &&
        prefix <= nextEndStopPrefix + K) {
        int64_t prevStart = xmi.start, prevStop = xmi.end;
        int64_t minStart = std::max((int64_t)0, xmi.start - (maxCount - xmi.size()));
        while (xmi.start >= minStart && X.lcp(xmi.start) >= xmi.depth) xmi.start--;
        int64_t maxEnd = std::min(n, xmi.end + (maxCount - xmi.size()));
        while (xmi.end + 1 < maxEnd && X.lcp(xmi.end + 1) >= xmi.depth) xmi.end++;
        if (xmi.size() < maxCount) {
            for (int64_t i = xmi.start; i < prevStart; i++) lmax_strict(X, P, prefix, X.sa(i), xmi.depth, out, min_len, maxLeft);
            for (int64_t i = prevStop + 1; i <= xmi.end; i++) lmax_strict(X, P, prefix, X.sa(i), xmi.depth, out, min_len, maxLeft);
        }
    }
}

// sparseSA::SMEM  sparseSA.cpp:1127-1208   (CLI "--seed PSMEM")
void seed_smem(const Idx& X, const Pat& P, MatchVec& out, int min_len, int maxCount, int sparseQ, int max_smem_cont) {
    const int K = (int)X.v->K;
    const int64_t n = X.n;
    if (min_len < K) return;
    int min_lenK = min_len - (sparseQ * K - 1);
    Ival mli = {0, n - 1, 0}, xmi = {0, n - 1, 0};
    std::vector<Tuple> iv;
    for (int prefix = 0; prefix <= P.len - min_lenK; prefix++) {
        traverse_child(X, P, prefix, mli, min_lenK, -1);
        if (mli.depth > xmi.depth) xmi = mli;
        if (mli.depth <= 1) { mli = {0, n - 1, 0}; xmi = {0, n - 1, 0}; continue; }
        if (mli.depth >= min_lenK) {
            traverse_child(X, P, prefix, xmi, P.len, -1);
            if (xmi.depth + prefix >= min_len) { Tuple t; t.mli = mli; t.xmi = xmi; t.prefix = prefix; iv.push_back(t); }
        }
        mli = {0, n - 1, 0}; xmi = {0, n - 1, 0};
        if (prefix % K == K - 1) prefix += (sparseQ - 1) * K;
    }
    // compIntervTupleRev sparseSA.cpp:52-54: end position descending, then prefix ascending (a total order:
    // at most one tuple per prefix)
    std::sort(iv.begin(), iv.end(), [](const Tuple& a, const Tuple& b) {
        int ea = a.prefix + a.xmi.depth, eb = b.prefix + b.xmi.depth;
        return ea > eb || (ea == eb && a.prefix < b.prefix);
    });
    std::vector<int> beginLong(P.len + 1, -1), smemCount(P.len + 1, -1);
    size_t i = 0;
    while (i < iv.size()) {
        int64_t curSize = 0;
        int curEnd = iv[i].xmiEnd();
        bool foundMax = false;
        while (i < iv.size() && iv[i].xmiEnd() == curEnd) {
            if (!foundMax || (iv[i].prefix <= beginLong[curEnd] + K)) curSize += iv[i].xmi.size();
            if (!foundMax && curSize >= max_smem_cont) { beginLong[iv[i].xmiEnd()] = iv[i].prefix; foundMax = true; }
            i++;
        }
        smemCount[iv[i - 1].xmiEnd()] = (int)curSize;
    }
    i = 0;
    int curMinPos = P.len;
    int minMin_len = sparseQ > 1 ? min_len - K : min_len;
    while (i < iv.size()) {
        int curEnd = iv[i].xmiEnd();
        bool skipEnd = smemCount[curEnd] >= maxCount || iv[i].prefix > curMinPos + K;
        if (!skipEnd && smemCount[curEnd] >= max_smem_cont) curMinPos = std::min(curMinPos, beginLong[curEnd]);
        while (i < iv.size() && iv[i].xmiEnd() == curEnd) {
            if (!skipEnd && iv[i].prefix <= curMinPos + K &&
                (smemCount[curEnd] < max_smem_cont || iv[i].prefix <= beginLong[curEnd] + K)) {
                if (smemCount[curEnd] >= max_smem_cont) {
                    collect_xmi_mli(X, P, iv[i], out, false, 0, 0, minMin_len, maxCount, K * sparseQ);
                } else {
                    int nextEndStop = curEnd - 1;
                    int nextEndPrefix = curEnd;
                    while (nextEndStop >= 0 && smemCount[nextEndStop] < max_smem_cont) nextEndStop--;
                    if (nextEndStop >= 0) nextEndPrefix = beginLong[nextEndStop];
                    collect_xmi_mli(X, P, iv[i], out, true, nextEndStop, nextEndPrefix, minMin_len, maxCount, K * sparseQ);
                }
            }
            i++;
        }
    }
}

// collectIntervals  sparseSA.cpp:968-990
void collect_intervals(const Idx& X, const Pat& P, int prefix, std::vector<Ival>& st, MatchVec& out, bool calcMli,
                       int nextEndStop, int min_len, int maxCount, int maxLeft) {
    const int K = (int)X.v->K;
    Ival xmi = st.back();
    if (xmi.depth + prefix < min_len) return;
    for (int64_t i = xmi.start; i <= xmi.end; i++) lmax_strict(X, P, prefix, X.sa(i), xmi.depth, out, min_len, maxLeft);
    if (!calcMli || st.size() == 1) return;
    int index = (int)st.size() - 2;
    while (index >= 0 && st[index].size() <= (maxCount) / K && prefix + st[index].depth > std::max((int)(nextEndStop - K), min_len)) {
        Ival mli = st[index];
        int64_t prevStart = st[index + 1].start, prevStop = st[index + 1].end;
        for (int64_t i = mli.start; i < prevStart; i++) lmax_strict(X, P, prefix, X.sa(i), mli.depth, out, min_len, maxLeft);
        for (int64_t i = prevStop + 1; i <= xmi.end; i++) lmax_strict(X, P, prefix, X.sa(i), mli.depth, out, min_len, maxLeft);
        index--;
    }
}

// sparseSA::SMEMP  sparseSA.cpp:1211-1254 (unreachable from the reference CLI)
void seed_smemp(const Idx& X, const Pat& P, MatchVec& out, int min_len, int maxCount, int sparseQ, int max_smem_cont) {
    const int K = (int)X.v->K;
    const int64_t n = X.n;
    if (min_len < K) return;
    int min_lenK = min_len - (sparseQ * K - 1);
    Ival mli = {0, n - 1, 0};
    std::vector<Ival> st;
    int prevEnd = 0, prevEndPrefix = 0;
    int64_t prevSize = -1;
    int minMin_len = sparseQ > 1 ? min_len - K : min_len;
    for (int prefix = 0; prefix <= P.len - min_lenK; prefix++) {
        st.clear();
        traverse_child(X, P, prefix, mli, min_lenK, -1);
        if (mli.depth <= 1) { mli = {0, n - 1, 0}; continue; }
        if (mli.depth >= min_lenK) {
            st.push_back(mli);
            traverse_stack(X, P, prefix, st, P.len, maxCount);
            if (!st.empty()) {
                bool extendXmi = (prefix + st.back().depth) > prevEnd || prevSize <= max_smem_cont || prefix <= prevEndPrefix + K ||
                                 st.back().size() <= max_smem_cont;
                bool extendMli = extendXmi && st.back().size() <= max_smem_cont &&
                                 ((prefix + st.back().depth) > prevEnd || prevSize <= max_smem_cont);
                if (extendXmi) collect_intervals(X, P, prefix, st, out, extendMli, prevEnd, minMin_len, maxCount, K * sparseQ);
                if (prevEnd < prefix + st.back().depth) {
                    prevEnd = prefix + st.back().depth;
                    prevSize = st.back().size();
                    prevEndPrefix = prevSize > max_smem_cont ? prefix : -1;
                } else {
                    prevSize += st.back().size();
                    if (prevEndPrefix < 0 && prevSize > max_smem_cont) prevEndPrefix = prefix;
                }
            }
        }
        mli = {0, n - 1, 0};
        if (prefix % K == K - 1) prefix += (sparseQ - 1) * K;
    }
}

void seed_dispatch(const alfa_index_view* v, int type, const char* Pp, int plen, int min_len, int max_count, int sparse_q,
                   int max_smem_cont, MatchVec& out) {
    Idx X; X.v = v; X.n = v->n;
    Pat P; P.p = Pp; P.len = plen;
    const int K = (int)v->K;
    switch (type) {
        case ALFA_SEED_SMAM:  // sparseSA::SMAM sparseSA.cpp:1121-1124
            if (min_len < K) return;
            for (int k = 0; k < K; k++) find_smam(X, k, P, out, min_len, max_count, sparse_q);
            break;
        case ALFA_SEED_MEM:   // sparseSA::MEM sparseSA.cpp:1116-1119
            if (min_len < K) return;
            for (int k = 0; k < K; k++) find_mem(X, k, P, out, min_len, max_count, sparse_q);
            break;
        case ALFA_SEED_SMEM: seed_smem(X, P, out, min_len, max_count, sparse_q, max_smem_cont); break;
        case ALFA_SEED_SMEMP: seed_smemp(X, P, out, min_len, max_count, sparse_q, max_smem_cont); break;
    }
}

struct EH { int32_t h, e; };  // eh_t ksw.c:358-360
const int32_t NEG_INF = -0x40000000;  // ksw.c:454

// query profile ksw.c:372-375: qp[k*qlen + j] = mat[k*m + query[j]]
void build_profile(std::vector<int8_t>& qp, int qlen, const uint8_t* query, int m, const int8_t* mat) {
    qp.resize((size_t)std::max(1, qlen * m));
    for (int k = 0, i = 0; k < m; ++k)
        for (int j = 0; j < qlen; ++j) qp[i++] = mat[k * m + query[j]];
}

// push_cigar ksw.c:456-466
void cig_push(std::vector<uint32_t>& c, int op, int len) {
    if (c.empty() || op != (int)(c.back() & 0xf)) c.push_back((uint32_t)len << 4 | (uint32_t)op);
    else c.back() += (uint32_t)len << 4;
}

}  // namespace

extern "C" {

void orc_counters_reset(void) { memset(&g_cnt, 0, sizeof(g_cnt)); }
void orc_counters_get(orc_counters* out) { *out = g_cnt; }
void orc_counters_add_dpband(int64_t cells) { g_cnt.cells_dpband += cells; }
int64_t orc_sa_bytes(const orc_counters* c) {
    return 8 * c->kmr_reads + 4 * c->child_reads + c->lcp_reads + 4 * c->sa_reads + c->s_bytes + c->p_bytes + 16 * c->matches;
}
void orc_free(void* p) { free(p); }

int64_t orc_seed(const alfa_index_view* v, int mem_type, const char* P, int plen, int min_len, int max_count, int sparse_q,
                 int max_smem_cont, alfa_match** out) {
    MatchVec m;
    seed_dispatch(v, mem_type, P, plen, min_len, max_count, sparse_q, max_smem_cont, m);
    alfa_match* r = (alfa_match*)malloc(sizeof(alfa_match) * (m.size() + 1));
    if (!m.empty()) memcpy(r, m.data(), sizeof(alfa_match) * m.size());
    *out = r;
    return (int64_t)m.size();
}

int orc_seed_batch(const alfa_index_view* v, int mem_type, const alfa_seed_query* q, int nq, alfa_match** matches, int64_t* offsets) {
    MatchVec m;
    offsets[0] = 0;
    for (int i = 0; i < nq; i++) {
        seed_dispatch(v, mem_type, q[i].P, q[i].plen, q[i].min_len, q[i].max_count, q[i].sparse_q, q[i].max_smem_cont, m);
        offsets[i + 1] = (int64_t)m.size();
    }
    alfa_match* r = (alfa_match*)malloc(sizeof(alfa_match) * (m.size() + 1));
    if (!m.empty()) memcpy(r, m.data(), sizeof(alfa_match) * m.size());
    *matches = r;
    return 0;
}

// ksw_extend  ksw.c:362-448
int orc_ksw_extend(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int m, const int8_t* mat, int gapo, int gape,
                   int w, int zdrop, int h0, int* _qle, int* _tle, int* _gtle, int* _gscore, int* _max_off) {
    const int gapoe = gapo + gape;
    if (h0 < 0) h0 = 0;
    std::vector<int8_t> qp;
    build_profile(qp, qlen, query, m, mat);
    std::vector<EH> eh((size_t)qlen + 2, EH{0, 0});
    // first row (ksw.c:377-379)
    eh[0].h = h0;
    if (qlen >= 1) eh[1].h = h0 > gapoe ? h0 - gapoe : 0;  // the reference writes eh[1] even when qlen==0 (calloc(1) slack)
    for (int j = 2; j <= qlen && eh[j - 1].h > gape; ++j) eh[j].h = eh[j - 1].h - gape;
    // band clamp through double arithmetic (ksw.c:381-386)
    int maxs = 0;
    for (int i = 0; i < m * m; ++i) maxs = maxs > mat[i] ? maxs : mat[i];
    int max_gap = (int)((double)(qlen * maxs - gapo) / gape + 1.);
    max_gap = max_gap > 1 ? max_gap : 1;
    w = w < max_gap ? w : max_gap;
    int max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0;
    int beg = 0, end = qlen;
    for (int i = 0; i < tlen; ++i) {
        int f = 0, h1, mrow = 0, mj = -1, j;
        const int8_t* q = &qp[(size_t)target[i] * qlen];
        h1 = h0 - (gapo + gape * (i + 1));   // regardless of beg (ksw.c:395-396)
        if (h1 < 0) h1 = 0;
        if (beg < i - w) beg = i - w;
        if (end > i + w + 1) end = i + w + 1;
        if (end > qlen) end = qlen;
        for (j = beg; j < end; ++j) {
            g_cnt.cells_extend++;
            int h = eh[j].h, e = eh[j].e;
            eh[j].h = h1;
            h += q[j];
            h = h > e ? h : e;
            h = h > f ? h : f;
            h1 = h;
            mj = mrow > h ? mj : j;      // ties -> the later column (ksw.c:414)
            mrow = mrow > h ? mrow : h;
            h -= gapoe;
            h = h > 0 ? h : 0;
            e -= gape;
            e = e > h ? e : h;
            eh[j].e = e;
            f -= gape;
            f = f > h ? f : h;
        }
        eh[end].h = h1; eh[end].e = 0;
        if (j == qlen) {
            max_ie = gscore > h1 ? max_ie : i;
            gscore = gscore > h1 ? gscore : h1;
        }
        // Z-drop test uses the loop variable j (== end), not mj (ksw.c:429)
        if (mrow == 0 || (zdrop > 0 && max - mrow - std::abs((i - max_i) - (j - max_j)) * gape > zdrop)) break;
        if (mrow > max) {
            max = mrow; max_i = i; max_j = mj;
            max_off = max_off > std::abs(mj - i) ? max_off : std::abs(mj - i);
        }
        for (j = mj; j >= beg && eh[j].h; --j) {}
        beg = j + 1;
        for (j = mj + 2; j <= end && eh[j].h; ++j) {}
        end = j;
    }
    if (_qle) *_qle = max_j + 1;
    if (_tle) *_tle = max_i + 1;
    if (_gtle) *_gtle = max_ie + 1;
    if (_gscore) *_gscore = gscore;
    if (_max_off) *_max_off = max_off;
    return max;
}

// shared fill of ksw_nw (ksw.c:544-591) and ksw_global (ksw.c:468-523); z==NULL -> score only
static int global_fill(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int m, const int8_t* mat, int gapo,
                       int gape, int w, uint8_t* z, int n_col, int64_t* cells) {
    const int gapoe = gapo + gape;
    std::vector<int8_t> qp;
    build_profile(qp, qlen, query, m, mat);
    std::vector<EH> eh((size_t)qlen + 2);
    eh[0].h = 0; eh[0].e = NEG_INF;
    int j;
    for (j = 1; j <= qlen && j <= w; ++j) { eh[j].h = -(gapo + gape * j); eh[j].e = NEG_INF; }
    for (; j <= qlen; ++j) eh[j].h = eh[j].e = NEG_INF;
    for (int i = 0; i < tlen; ++i) {
        int32_t f = NEG_INF, h1;
        const int8_t* q = &qp[(size_t)target[i] * qlen];
        int beg = i > w ? i - w : 0;
        int end = i + w + 1 < qlen ? i + w + 1 : qlen;
        h1 = beg == 0 ? -(gapo + gape * (i + 1)) : NEG_INF;
        uint8_t* zi = z ? z + (size_t)i * n_col : 0;
        for (j = beg; j < end; ++j) {
            (*cells)++;
            int32_t h = eh[j].h, e = eh[j].e;
            uint8_t d;
            eh[j].h = h1;
            h += q[j];
            d = h >= e ? 0 : 1;
            h = h >= e ? h : e;
            d = h >= f ? d : 2;
            h = h >= f ? h : f;
            h1 = h;
            h -= gapoe;
            e -= gape;
            d |= e > h ? 1 << 2 : 0;
            e = e > h ? e : h;
            eh[j].e = e;
            f -= gape;
            d |= f > h ? 2 << 4 : 0;
            f = f > h ? f : h;
            if (zi) zi[j - beg] = d;
        }
        // when beg >= end the reference still writes eh[end] (end==qlen here because beg<=i)
        eh[end].h = h1; eh[end].e = NEG_INF;
    }
    return eh[qlen].h;
}

int orc_ksw_nw(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int m, const int8_t* mat, int gapo, int gape, int w) {
    return global_fill(qlen, query, tlen, target, m, mat, gapo, gape, w, 0, 0, &g_cnt.cells_nw);
}

// ksw_global  ksw.c:468-542
int orc_ksw_global(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int m, const int8_t* mat, int gapo, int gape,
                   int w, int* n_cigar_, uint32_t** cigar_) {
    if (n_cigar_) *n_cigar_ = 0;
    int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
    std::vector<uint8_t> z((size_t)std::max(1, n_col) * std::max(1, tlen) + 1);
    int score = global_fill(qlen, query, tlen, target, m, mat, gapo, gape, w, z.data(), n_col, &g_cnt.cells_global);
    if (n_cigar_ && cigar_) {
        std::vector<uint32_t> cg;
        int which = 0;
        int i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
        while (i >= 0 && k >= 0) {
            // when the band misses the corner (w < |qlen-tlen|) the reference indexes z outside the row and may
            // read uninitialised heap: unspecified there; here out-of-buffer reads are 0 and z is zero-filled.
            int64_t zi = (int64_t)i * n_col + (k - (i > w ? i - w : 0));
            uint8_t zv = (zi >= 0 && zi < (int64_t)z.size()) ? z[(size_t)zi] : 0;
            which = zv >> (which << 1) & 3;
            if (which == 0) { cig_push(cg, 0, 1); --i; --k; }
            else if (which == 1) { cig_push(cg, 2, 1); --i; }
            else { cig_push(cg, 1, 1); --k; }
        }
        if (i >= 0) cig_push(cg, 2, i + 1);
        if (k >= 0) cig_push(cg, 1, k + 1);
        std::reverse(cg.begin(), cg.end());
        uint32_t* out = (uint32_t*)malloc(sizeof(uint32_t) * (cg.size() + 1));
        if (!cg.empty()) memcpy(out, cg.data(), sizeof(uint32_t) * cg.size());
        *n_cigar_ = (int)cg.size();
        *cigar_ = out;
    }
    return score;
}

int orc_ksw_extend_batch(const alfa_dp_params* p, const alfa_dp_task* t, int n, alfa_extend_result* r) {
    for (int i = 0; i < n; i++)
        r[i].score = orc_ksw_extend(t[i].qlen, t[i].query, t[i].tlen, t[i].target, p->m, p->mat, p->gapo, p->gape, t[i].w, p->zdrop,
                                    t[i].h0, &r[i].qle, &r[i].tle, &r[i].gtle, &r[i].gscore, &r[i].max_off);
    return 0;
}
int orc_ksw_nw_batch(const alfa_dp_params* p, const alfa_dp_task* t, int n, int32_t* score) {
    for (int i = 0; i < n; i++) score[i] = orc_ksw_nw(t[i].qlen, t[i].query, t[i].tlen, t[i].target, p->m, p->mat, p->gapo, p->gape, t[i].w);
    return 0;
}
int orc_ksw_global_batch(const alfa_dp_params* p, const alfa_dp_task* t, int n, int32_t* score, uint32_t** cigar, int32_t* n_cigar,
                         int64_t* cig_off) {
    std::vector<uint32_t> all;
    cig_off[0] = 0;
    for (int i = 0; i < n; i++) {
        uint32_t* c = 0; int nc = 0;
        score[i] = orc_ksw_global(t[i].qlen, t[i].query, t[i].tlen, t[i].target, p->m, p->mat, p->gapo, p->gape, t[i].w, &nc, &c);
        n_cigar[i] = nc;
        all.insert(all.end(), c, c + nc);
        free(c);
        cig_off[i + 1] = (int64_t)all.size();
    }
    uint32_t* out = (uint32_t*)malloc(sizeof(uint32_t) * (all.size() + 1));
    if (!all.empty()) memcpy(out, all.data(), sizeof(uint32_t) * all.size());
    *cigar = out;
    return 0;
}

}  // extern "C"
