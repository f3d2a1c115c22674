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

}  // namespace alfa_compat
