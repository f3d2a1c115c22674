// alfa_compat.hpp -- the reference's own call signatures on top of libalfa_hot (C++; header only).
//
// For a maintainer who wants to switch call sites one at a time before batching them: every function below has the
// signature, argument meaning and ownership rules of its counterpart in the reference and forwards to the batch entry
// point of include/alfa_hot.h with a batch of one.  (A batch of one pays a launch and a PCIe round trip per call: these
// wrappers are for bring-up and for tests, the batch forms are the product.)
//
//   alfa_compat::ksw_extend / ksw_nw / ksw_global   == src/ksw.h:63-106 (ksw_global's CIGAR is malloc'd: free() it)
//   alfa_compat::ksw_align / kswr_t / KSW_X*        == src/ksw.h:6-19,63 (the query-profile cache argument must be null)
//   alfa_compat::sparseSA::SMAM / MEM / SMEM / SMEMP == src/sparseSA.h:274-283 (append to `matches`, never clear it)
//   alfa_compat::match_t                            == src/sparseSA.h:103-110
//   alfa_compat::dynProg (+ boundaries, dp_scores, dp_type, dp_output, outputType) == src/dp.h:29-74,76-137: the constructor
//       and dpBandStatic / dpBandFull / dpBandScore / dpBasic with the reference's arguments; they mutate `boundaries` and fill
//       `dp_output` as the reference does.  Where the reference calls exit(1) (a failed traceback, src/dp.cpp:402) or reads
//       matrix cells it never wrote, the wrapper throws std::runtime_error instead.
#pragma once
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "alfa_hot.h"

namespace alfa_compat {

struct match_t {
    match_t() : ref(0), query(0), len(0) {}
    match_t(long r, int q, int l) : ref(r), query(q), len(l) {}
    long ref;
    int query, len;
};

inline void fill_params(alfa_dp_params& p, int m, const int8_t* mat, int gapo, int gape, int zdrop) {
    std::memset(&p, 0, sizeof(p));
    p.m = m;
    if (m >= 1 && m <= 8) std::memcpy(p.mat, mat, (size_t)m * m);     // the library rejects m outside 1..8 (mat holds 64 entries)
    p.gapo = gapo; p.gape = gape; p.zdrop = zdrop;
}
inline void must(int rc) {
    if (rc != ALFA_OK) throw std::runtime_error(std::string("libalfa_hot: ") + alfa_last_error());
}

inline int ksw_extend(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int m, const int8_t* mat, int gapo, int gape, int w,
                      int zdrop, int h0, int* qle, int* tle, int* gtle, int* gscore, int* max_off) {
    alfa_dp_params p;
    fill_params(p, m, mat, gapo, gape, zdrop);
    alfa_dp_task t = {query, target, qlen, tlen, w, h0};
    alfa_extend_result r;
    must(alfa_ksw_extend_batch(&p, &t, 1, &r));
    if (qle) *qle = r.qle;
    if (tle) *tle = r.tle;
    if (gtle) *gtle = r.gtle;
    if (gscore) *gscore = r.gscore;
    if (max_off) *max_off = r.max_off;
    return r.score;
}

inline int ksw_nw(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int m, const int8_t* mat, int gapo, int gape, int w) {
    alfa_dp_params p;
    fill_params(p, m, mat, gapo, gape, 0);
    alfa_dp_task t = {query, target, qlen, tlen, w, 0};
    int32_t s = 0;
    must(alfa_ksw_nw_batch(&p, &t, 1, &s));
    return s;
}

inline int ksw_global(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int m, const int8_t* mat, int gapo, int gape, int w,
                      int* n_cigar, uint32_t** cigar) {
    alfa_dp_params p;
    fill_params(p, m, mat, gapo, gape, 0);
    alfa_dp_task t = {query, target, qlen, tlen, w, 0};
    int32_t s = 0, nc = 0;
    int64_t off[2] = {0, 0};
    uint32_t* c = nullptr;
    must(alfa_ksw_global_batch(&p, &t, 1, &s, &c, &nc, off));
    if (n_cigar) *n_cigar = nc;
    if (cigar) *cigar = c;          // malloc'd by the library: the caller's free() releases it, as in the reference
    else alfa_free(c);
    return s;
}

typedef alfa_kswr kswr_t;          // same seven ints (src/ksw.h:14-19)
enum { KSW_XBYTE = ALFA_KSW_XBYTE, KSW_XSTOP = ALFA_KSW_XSTOP, KSW_XSUBO = ALFA_KSW_XSUBO, KSW_XSTART = ALFA_KSW_XSTART };

// gapo / gape go through untouched: the reference's callers hand ksw_align negative numbers (src/mapper.cpp:1572) and the
// library reproduces what its unsigned arithmetic makes of them.  `qry` (the cached query profile) must be null.
inline kswr_t ksw_align(int qlen, uint8_t* query, int tlen, uint8_t* target, int m, const int8_t* mat, int gapo, int gape, int xtra, void** qry) {
    if (qry) throw std::runtime_error("alfa_compat::ksw_align: the query profile cache is not supported, pass 0");
    alfa_dp_params p;
    fill_params(p, m, mat, gapo, gape, 0);
    alfa_align_task t = {query, target, qlen, tlen, xtra};
    kswr_t r;
    must(alfa_ksw_align_batch(&p, &t, 1, &r));
    return r;
}

// the HBM-resident index with the reference's query methods
class sparseSA {
   public:
    sparseSA(const alfa_index_view* v, int device = 0) : K(v->K) { must(alfa_index_upload(v, device, &ix_)); }
    ~sparseSA() { alfa_index_free(ix_); }
    sparseSA(const sparseSA&) = delete;
    sparseSA& operator=(const sparseSA&) = delete;
    void SMAM(const std::string& P, std::vector<match_t>& matches, int min_len, int maxCount, int sparseQ) const {
        run(ALFA_SEED_SMAM, P, matches, min_len, maxCount, sparseQ, 0);
    }
    void MEM(const std::string& P, std::vector<match_t>& matches, int min_len, int maxCount, int sparseQ) const {
        run(ALFA_SEED_MEM, P, matches, min_len, maxCount, sparseQ, 0);
    }
    void SMEM(const std::string& P, std::vector<match_t>& matches, int min_len, int maxCount, int sparseQ, int max_smem_cont) const {
        run(ALFA_SEED_SMEM, P, matches, min_len, maxCount, sparseQ, max_smem_cont);
    }
    void SMEMP(const std::string& P, std::vector<match_t>& matches, int min_len, int maxCount, int sparseQ, int max_smem_cont) const {
        run(ALFA_SEED_SMEMP, P, matches, min_len, maxCount, sparseQ, max_smem_cont);
    }
    const long K;

   private:
    void run(int type, const std::string& P, std::vector<match_t>& matches, int min_len, int maxCount, int sparseQ, int max_smem) const {
        alfa_seed_query q = {P.data(), (int32_t)P.size(), min_len, maxCount, sparseQ, max_smem};
        alfa_match* m = nullptr;
        int64_t off[2] = {0, 0};
        must(alfa_seed_batch(ix_, type, &q, 1, &m, off));
        for (int64_t k = off[0]; k < off[1]; ++k) matches.push_back(match_t((long)m[k].ref, m[k].query, m[k].len));
        alfa_free(m);
    }
    alfa_index* ix_ = nullptr;
};

// ---- dynProg (src/dp.h).  The reference's object owns the three DP matrices (one object per thread, src/dp.h:78); here the
// matrices live on the device for the duration of a call, so the object only carries the scoring scheme.
struct boundaries {
    boundaries() : refB(0), refE(0), queryB(0), queryE(0) {}
    boundaries(long refBegin, long refEnd, int queryBegin, int queryEnd) : refB(refBegin), refE(refEnd), queryB(queryBegin), queryE(queryEnd) {}
    long refB, refE;
    int queryB, queryE;
};
struct dp_scores {          // (the reference also holds the 127 x 127 matrix it derives from these four: updateScoreMatrixDna)
    dp_scores() : match(0), mismatch(-1), openGap(0), extendGap(-1) {}
    dp_scores(int m, int mm, int oG, int eG) : match(m), mismatch(mm), openGap(oG), extendGap(eG) {}
    int match, mismatch, openGap, extendGap;
    void updateScoreMatrixDna() {}
};
struct dp_type {
    dp_type() : freeRefB(false), freeRefE(false), freeQueryB(false), freeQueryE(false), local(false) {}
    dp_type(bool refB, bool refE, bool qB, bool qE, bool loc) : freeRefB(refB), freeRefE(refE), freeQueryB(qB), freeQueryE(qE), local(loc) {}
    bool freeRefB, freeRefE, freeQueryB, freeQueryE, local;
};
struct dp_output {
    dp_output() : editDist(0), dpScore(0) {}
    int editDist, dpScore;
    std::vector<char> cigarChars;
    std::vector<int> cigarLengths;
    void clear() { cigarChars.clear(); cigarLengths.clear(); editDist = dpScore = 0; }
};
enum outputType { DPSCORE, SCORE, ALIGNMENT, ERRORSTRING, ALL };

struct dynProg {
    dp_scores& scores;
    const bool affine;
    // `dimension` sized the reference's matrices; the device allocates per call
    dynProg(int /*dimension*/, bool affine_, dp_scores& scoreFunction) : scores(scoreFunction), affine(affine_) {}
    int dpBandStatic(const std::string& ref, const std::string& query, boundaries& offset, const dp_type& type, const outputType& oType,
                     dp_output& output, int minbandSize, int maxbandSize, bool /*print*/ = false) {
        return call(ALFA_DPBAND_STATIC, ref, query, offset, type, oType, output, minbandSize, maxbandSize);
    }
    int dpBandFull(const std::string& ref, const std::string& query, boundaries& offset, const dp_type& type, const outputType& oType,
                   dp_output& output, int bandLeft, int bandRight, bool /*print*/ = false) {
        return call(ALFA_DPBAND_FULL, ref, query, offset, type, oType, output, bandLeft, bandRight);
    }
    int dpBandScore(const std::string& ref, const std::string& query, boundaries& offset, const dp_type& type, const outputType& oType,
                    dp_output& output, int bandLeft, int bandRight, bool /*print*/ = false) {
        return call(ALFA_DPBAND_SCORE, ref, query, offset, type, oType, output, bandLeft, bandRight);
    }
    int dpBasic(const std::string& ref, const std::string& query, boundaries& offset, const dp_type& type, const outputType& oType, dp_output& output) {
        return call(ALFA_DPBAND_BASIC, ref, query, offset, type, oType, output, 0, 0);
    }

   private:
    int call(int variant, const std::string& ref, const std::string& query, boundaries& offset, const dp_type& type, const outputType& oType,
             dp_output& output, int b1, int b2) {
        alfa_dpband_scores sc = {scores.match, scores.mismatch, scores.openGap, scores.extendGap, affine ? 1 : 0};
        alfa_dpband_task t;
        std::memset(&t, 0, sizeof(t));
        t.ref = ref.data(); t.ref_len = (int64_t)ref.size(); t.query = query.data(); t.query_len = (int64_t)query.size();
        t.refB = offset.refB; t.refE = offset.refE; t.queryB = offset.queryB; t.queryE = offset.queryE;
        t.freeRefB = type.freeRefB; t.freeRefE = type.freeRefE; t.freeQueryB = type.freeQueryB; t.freeQueryE = type.freeQueryE; t.local = type.local;
        t.out_type = (int32_t)oType; t.band1 = b1; t.band2 = b2;
        alfa_dpband_result r;
        char* cc = nullptr;
        int32_t* cl = nullptr;
        must(alfa_dp_band_batch(variant, &sc, &t, 1, &r, &cc, &cl));
        if (r.ret >= 2) {
            alfa_free(cc); alfa_free(cl);
            throw std::runtime_error(r.ret == 3 ? "alfa_compat::dynProg: the reference exits here (Error in dynamic programming, src/dp.cpp:402)"
                                                : "alfa_compat::dynProg: the reference reads matrix cells it never wrote for these arguments");
        }
        output.dpScore = r.dpScore; output.editDist = r.editDist;
        for (int32_t k = 0; k < r.n_cigar; ++k) { output.cigarChars.push_back(cc[r.cig_off + k]); output.cigarLengths.push_back(cl[r.cig_off + k]); }
        offset.refB = (long)r.refB; offset.refE = (long)r.refE; offset.queryB = r.queryB; offset.queryE = r.queryE;
        alfa_free(cc); alfa_free(cl);
        return r.ret;
    }
};

}  // namespace alfa_compat
