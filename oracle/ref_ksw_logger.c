/* oracle/ref_ksw_logger.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Interposes the three live DP entry points of the reference so that every call the REAL
 * mapper makes while mapping the example data can be captured as a known-answer vector.
 * oracle/Makefile compiles the reference's ksw.c in place with
 *   -Dksw_extend=ref_real_ksw_extend -Dksw_nw=ref_real_ksw_nw -Dksw_global=ref_real_ksw_global
 * and links this file, which provides the original names, forwards to the real code and appends
 * one binary record per call to the file named by $ALFA_KSW_LOG.  Nothing of the reference is
 * copied; signatures are those of src/ksw.h:82-83,106.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <pthread.h>

int ref_real_ksw_extend(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat, int gapo, int gape, int w, int zdrop, int h0, int *qle, int *tle, int *gtle, int *gscore, int *max_off);
int ref_real_ksw_nw(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat, int gapo, int gape, int w);
int ref_real_ksw_global(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat, int gapo, int gape, int w, int *n_cigar, uint32_t **cigar);

static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
static FILE *logf;

/* record: int32 kind(0 nw,1 global,2 extend), qlen, tlen, m, gapo, gape, w, zdrop, h0, n_out; int8 mat[m*m]; query; target; int32 out[n_out] */
static void put(int kind, int qlen, const uint8_t *q, int tlen, const uint8_t *t, int m, const int8_t *mat, int gapo, int gape, int w,
                int zdrop, int h0, int n_out, const int32_t *out)
{
	pthread_mutex_lock(&mu);
	if (!logf) {
		const char *p = getenv("ALFA_KSW_LOG");
		if (p) logf = fopen(p, "wb");
	}
	if (logf) {
		int32_t hdr[10] = {kind, qlen, tlen, m, gapo, gape, w, zdrop, h0, n_out};
		fwrite(hdr, 4, 10, logf);
		fwrite(mat, 1, (size_t)(m * m), logf);
		fwrite(q, 1, (size_t)qlen, logf);
		fwrite(t, 1, (size_t)tlen, logf);
		fwrite(out, 4, (size_t)n_out, logf);
		fflush(logf);
	}
	pthread_mutex_unlock(&mu);
}

int ksw_extend(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat, int gapo, int gape, int w, int zdrop, int h0, int *qle, int *tle, int *gtle, int *gscore, int *max_off)
{
	int32_t o[6];
	int a, b, c, d, e;
	o[0] = ref_real_ksw_extend(qlen, query, tlen, target, m, mat, gapo, gape, w, zdrop, h0, &a, &b, &c, &d, &e);
	o[1] = a; o[2] = b; o[3] = c; o[4] = d; o[5] = e;
	put(2, qlen, query, tlen, target, m, mat, gapo, gape, w, zdrop, h0, 6, o);
	if (qle) *qle = a; if (tle) *tle = b; if (gtle) *gtle = c; if (gscore) *gscore = d; if (max_off) *max_off = e;
	return o[0];
}

int ksw_nw(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat, int gapo, int gape, int w)
{
	int32_t s = ref_real_ksw_nw(qlen, query, tlen, target, m, mat, gapo, gape, w);
	put(0, qlen, query, tlen, target, m, mat, gapo, gape, w, 0, 0, 1, &s);
	return s;
}

int ksw_global(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat, int gapo, int gape, int w, int *n_cigar, uint32_t **cigar)
{
	int nc = 0, i;
	uint32_t *cg = 0;
	int32_t s = ref_real_ksw_global(qlen, query, tlen, target, m, mat, gapo, gape, w, &nc, &cg);
	int32_t *o = (int32_t*)malloc(4 * (size_t)(nc + 2));
	o[0] = s; o[1] = nc;
	for (i = 0; i < nc; ++i) o[2 + i] = (int32_t)cg[i];
	put(1, qlen, query, tlen, target, m, mat, gapo, gape, w, 0, 0, nc + 2, o);
	free(o);
	if (n_cigar) *n_cigar = nc;
	if (cigar) *cigar = cg; else free(cg);
	return s;
}
