/* oracle/alfa_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement of the reference's seed-retrieval and banded-DP
 * algorithms, used exclusively as the parity checker by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
 * library (libalfa_hot) never includes, links or calls anything in oracle/.
 *
 * Pinning: the oracle is itself checked (tests/test_oracle_vs_golden.py) against
 * golden vectors generated from the compiled reference (oracle/_ref, built from
 * /root/reference/src in place by oracle/Makefile; generator tools/make_goldens.py)
 * and committed under tests/golden/.
 */
#ifndef ALFA_ORACLE_H
#define ALFA_ORACLE_H
#include "../include/alfa_hot.h"

#ifdef __cplusplus
extern "C" {
#endif

/* algorithmic work counters accumulated by the calling thread */
typedef struct {
    int64_t kmr_reads;      /* 8 B each  */
    int64_t child_reads;    /* 4 B each  */
    int64_t lcp_reads;      /* 1 B each  */
    int64_t sa_reads;       /* 4 B each  */
    int64_t s_bytes;        /* 1 B each  */
    int64_t p_bytes;        /* 1 B each  */
    int64_t matches;        /* 16 B each */
    int64_t queries;        /* traverse_faster invocations */
    int64_t cells_extend, cells_nw, cells_global, cells_dpband;
} orc_counters;
void    orc_counters_reset(void);
void    orc_counters_get(orc_counters* out);
int64_t orc_sa_bytes(const orc_counters* c);   /* SURVEY 8d byte formula */

/* one sparseSA::{SMAM,MEM,SMEM,SMEMP} call; *out is malloc'd (orc_free) */
int64_t orc_seed(const alfa_index_view* v, int mem_type, const char* P, int plen, int min_len,
                 int max_count, int sparse_q, int max_smem_cont, alfa_match** out);
/* same contract as alfa_seed_batch, index given as a host view */
int orc_seed_batch(const alfa_index_view* v, int mem_type, const alfa_seed_query* q, int nq,
                   alfa_match** matches, int64_t* offsets);

int orc_ksw_extend(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int m, const int8_t* mat,
                   int gapo, int gape, int w, int zdrop, int h0, int* qle, int* tle, int* gtle, int* gscore, int* max_off);
int orc_ksw_nw(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int m, const int8_t* mat,
               int gapo, int gape, int w);
int orc_ksw_global(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int m, const int8_t* mat,
                   int gapo, int gape, int w, int* n_cigar, uint32_t** cigar);
int orc_ksw_extend_batch(const alfa_dp_params* p, const alfa_dp_task* t, int n, alfa_extend_result* r);
int orc_ksw_nw_batch(const alfa_dp_params* p, const alfa_dp_task* t, int n, int32_t* score);
int orc_ksw_global_batch(const alfa_dp_params* p, const alfa_dp_task* t, int n, int32_t* score,
                         uint32_t** cigar, int32_t* n_cigar, int64_t* cig_off);
/* ksw_align (align_oracle.cpp): the striped SSE2 kernel restated lane by lane */
void orc_ksw_align(int qlen, const uint8_t* query, int tlen, const uint8_t* target, int m, const int8_t* mat,
                   int gapo, int gape, int xtra, alfa_kswr* out);
int orc_ksw_align_batch(const alfa_dp_params* p, const alfa_align_task* t, int n, alfa_kswr* r);
int64_t orc_cells_align_take(void);   /* query length x rows processed since the last call (this thread) */
/* dynProg (dp_oracle.cpp).  r[k].ret: 0/1 = the reference's return value, 2 = the reference reads a matrix cell it never
 * wrote (result undefined), 3 = the reference prints "Error in dynamic programming" and exits.  Returns -22 for arguments
 * the reference asserts on or indexes out of bounds with (orc_dp_band_check). */
int orc_dp_band_batch(int variant, const alfa_dpband_scores* s, const alfa_dpband_task* t, int n,
                      alfa_dpband_result* r, char** cig_chars, int32_t** cig_lens);
int orc_dp_band_check(int variant, const alfa_dpband_scores* s, const alfa_dpband_task* t);
void orc_counters_add_dpband(int64_t cells);
void orc_free(void*);

#ifdef __cplusplus
}
#endif
#endif
