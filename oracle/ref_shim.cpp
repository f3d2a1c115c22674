// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" wrapper around the *real* reference classes so that Python
// (ctypes) can (a) validate the CPU restatement in oracle/ and (b) generate the
// golden vectors committed under tests/golden/.  It is compiled by
// oracle/Makefile together with the reference's own sources *where they lie*
// under /root/reference/src (sparseSA.cpp qsufsort.c ksw.c dp.cpp) into
// oracle/_ref/libalfalfa_ref.so.  Nothing from the reference is copied here:
// this file only calls the reference's public interface
// (sparseSA.h:142-285, ksw.h:63-106, dp.h:93-135).
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <string>
#include <vector>

#include "sparseSA.h"
#include "dp.h"
extern "C" {
#include "ksw.h"
}

struct ref_match { long ref; int query; int len; };

extern "C" {

void* ref_sa_new() { return new sparseSA(); }
void ref_sa_free(void* h) { delete (sparseSA*)h; }

// loadRef + init + construct  (main.cpp:335-340 order)
int ref_sa_build(void* h, const char* fasta, long K, int suflink, int child, int kmer) {
    sparseSA* sa = (sparseSA*)h;
    sa->loadRef(fasta);
    sa->init(K, suflink != 0, child != 0, kmer != 0);
    sa->construct();
    return 0;
}
// Build from an in-memory sequence set (names/seqs), mimicking loadRef's
// concatenation rule (sparseSA.cpp:140-153) without going through a file.
int ref_sa_build_mem(void* h, int nseq, const char** names, const char** seqs, long K, int suflink, int child, int kmer) {
    sparseSA* sa = (sparseSA*)h;
    for (int i = 0; i < nseq; i++) {
        std::string s(seqs[i]);
        for (size_t j = 0; j < s.size(); j++) s[j] = std::tolower(s[j]);
        if (sa->S.length() > 0) sa->S += '`';
        sa->startpos.push_back(sa->S.length());
        sa->S.append(s);
        sa->descr.push_back(names[i]);
    }
    sa->init(K, suflink != 0, child != 0, kmer != 0);
    sa->construct();
    return 0;
}
void ref_sa_save(void* h, const char* prefix) { ((sparseSA*)h)->save(prefix); }
int ref_sa_load(void* h, const char* prefix) { return ((sparseSA*)h)->load(prefix) ? 1 : 0; }

long ref_sa_N(void* h) { return ((sparseSA*)h)->N; }
long ref_sa_K(void* h) { return ((sparseSA*)h)->K; }
long ref_sa_logN(void* h) { return ((sparseSA*)h)->logN; }
long ref_sa_NKm1(void* h) { return ((sparseSA*)h)->NKm1; }
long ref_sa_n(void* h) { return (long)((sparseSA*)h)->SA.size(); }
const char* ref_sa_S(void* h) { return ((sparseSA*)h)->S.data(); }
const unsigned* ref_sa_SA(void* h) { return ((sparseSA*)h)->SA.data(); }
const unsigned char* ref_sa_LCPvec(void* h) { return ((sparseSA*)h)->LCP.vec.data(); }
long ref_sa_LCP_M_size(void* h) { return (long)((sparseSA*)h)->LCP.M.size(); }
void ref_sa_LCP_M(void* h, long* idx, int* val) {
    sparseSA* sa = (sparseSA*)h;
    for (size_t i = 0; i < sa->LCP.M.size(); i++) { idx[i] = (long)sa->LCP.M[i].idx; val[i] = sa->LCP.M[i].val; }
}
const int* ref_sa_CHILD(void* h) { return ((sparseSA*)h)->CHILD.data(); }
long ref_sa_CHILD_size(void* h) { return (long)((sparseSA*)h)->CHILD.size(); }
const int* ref_sa_ISA(void* h) { return ((sparseSA*)h)->ISA.data(); }
long ref_sa_ISA_size(void* h) { return (long)((sparseSA*)h)->ISA.size(); }
const unsigned* ref_sa_KMR(void* h) { return (const unsigned*)((sparseSA*)h)->KMR.data(); }
long ref_sa_KMR_size(void* h) { return (long)((sparseSA*)h)->KMR.size(); }
long ref_sa_nseq(void* h) { return (long)((sparseSA*)h)->startpos.size(); }
long ref_sa_startpos(void* h, long i) { return ((sparseSA*)h)->startpos[i]; }
const char* ref_sa_descr(void* h, long i) { return ((sparseSA*)h)->descr[i].c_str(); }

// type: 0 SMAM, 1 MEM, 2 SMEM, 3 SMEMP  (sparseSA.h:274-283)
long ref_seed(void* h, int type, const char* P, int plen, int min_len, int maxCount, int sparseQ,
              int maxSmemCont, ref_match** out) {
    const sparseSA* sa = (const sparseSA*)h;
    std::string p(P, plen);
    std::vector<match_t> m;
    switch (type) {
        case 0: sa->SMAM(p, m, min_len, maxCount, sparseQ); break;
        case 1: sa->MEM(p, m, min_len, maxCount, sparseQ); break;
        case 2: sa->SMEM(p, m, min_len, maxCount, sparseQ, maxSmemCont); break;
        case 3: sa->SMEMP(p, m, min_len, maxCount, sparseQ, maxSmemCont); break;
        default: return -1;
    }
    ref_match* r = (ref_match*)malloc(sizeof(ref_match) * (m.size() + 1));
    for (size_t i = 0; i < m.size(); i++) { r[i].ref = m[i].ref; r[i].query = m[i].query; r[i].len = m[i].len; }
    *out = r;
    return (long)m.size();
}
void ref_free(void* p) { free(p); }

// ---- dynProg (dp.h:93-135).  variant: 0 dpBandStatic, 1 dpBandFull, 2 dpBandScore, 3 dpBasic
// out_i: [ret, dpScore, editDist, refB, refE, queryB, queryE, ncig]; cigar chars/lengths copied out.
int ref_dynprog(int variant, int affine, int match, int mismatch, int openGap, int extendGap,
                const char* ref, long reflen, const char* query, long querylen,
                long refB, long refE, int queryB, int queryE,
                int freeRefB, int freeRefE, int freeQueryB, int freeQueryE, int local,
                int outType, int band1, int band2,
                long* out_i, char* cig_chars, int* cig_lens, int cig_cap) {
    dp_scores scores(match, mismatch, openGap, extendGap);
    scores.updateScoreMatrixDna();
    dynProg dp(2048, affine != 0, scores);
    std::string R(ref, reflen), Q(query, querylen);
    boundaries b(refB, refE, queryB, queryE);
    dp_type t;
    t.freeRefB = freeRefB; t.freeRefE = freeRefE; t.freeQueryB = freeQueryB; t.freeQueryE = freeQueryE; t.local = local;
    dp_output o;
    int ret = -1;
    outputType ot = (outputType)outType;
    switch (variant) {
        case 0: ret = dp.dpBandStatic(R, Q, b, t, ot, o, band1, band2, false); break;
        case 1: ret = dp.dpBandFull(R, Q, b, t, ot, o, band1, band2, false); break;
        case 2: ret = dp.dpBandScore(R, Q, b, t, ot, o, band1, band2, false); break;
        case 3: ret = dp.dpBasic(R, Q, b, t, ot, o); break;
    }
    out_i[0] = ret; out_i[1] = o.dpScore; out_i[2] = o.editDist;
    out_i[3] = b.refB; out_i[4] = b.refE; out_i[5] = b.queryB; out_i[6] = b.queryE;
    int n = (int)o.cigarChars.size();
    out_i[7] = n;
    for (int i = 0; i < n && i < cig_cap; i++) { cig_chars[i] = o.cigarChars[i]; cig_lens[i] = o.cigarLengths[i]; }
    return ret;
}

}  // extern "C"
