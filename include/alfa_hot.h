/* alfa_hot.h -- C ABI of libalfa_hot: the MI355X-native seed-and-extend hot path.
 *
 * Every entry point below is the batch form of one call the reference ALFALFA
 * mapper makes synchronously, per read, inside its kt_for worker
 * (reference src/main.cpp:102,126).  The reference has no FFI/plugin layer:
 * its "interface" for this path is the in-process C/C++ call surface
 *
 *   sparseSA::SMAM / MEM / SMEM / SMEMP        src/sparseSA.h:274-283
 *   ksw_extend / ksw_nw / ksw_global           src/ksw.h:82-83,106
 *   dynProg::dpBandStatic/Full/Score/dpBasic   src/dp.h:122-135
 *
 * and each declaration cites the exact reference signature it replaces.
 * Conventions: plain pointers and sizes only; return 0 on success or a negative
 * errno-style code (never exit()); inputs are borrowed for the duration of the
 * call; outputs documented as "library-allocated" are released with alfa_free().
 * All entry points are thread-safe; one alfa_index may be shared by any number
 * of host threads (the reference shares one `const sparseSA&` the same way).
 *
 * There is NO CPU fallback in this library: without a usable HIP device every
 * compute entry point fails with ALFA_ENODEV and alfa_last_error() says why.
 */
#ifndef ALFA_HOT_H
#define ALFA_HOT_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALFA_OK        0
#define ALFA_EINVAL  (-22)
#define ALFA_ENOMEM  (-12)
#define ALFA_ENODEV  (-19)
#define ALFA_ENOTSUP (-95)
#define ALFA_EHIP    (-5)

/* seed types == the reference's method names (NOT the shifted CLI names,
 * src/options.cpp:155-172): the CLI default "--seed SMEM" is ALFA_SEED_SMAM. */
#define ALFA_SEED_SMAM  0   /* sparseSA::SMAM   src/sparseSA.cpp:1121 */
#define ALFA_SEED_MEM   1   /* sparseSA::MEM    src/sparseSA.cpp:1116 */
#define ALFA_SEED_SMEM  2   /* sparseSA::SMEM   src/sparseSA.cpp:1127 */
#define ALFA_SEED_SMEMP 3   /* sparseSA::SMEMP  src/sparseSA.cpp:1211 */

/* == match_t (src/sparseSA.h:103-110): 16 bytes */
typedef struct { int64_t ref; int32_t query; int32_t len; } alfa_match;

/* Host view of the enhanced sparse suffix array, i.e. the public data members
 * of `struct sparseSA` (src/sparseSA.h:142-163).  All arrays are borrowed.
 * S must have at least 16 readable bytes after S[N-1] (the reference reads
 * S[SA+depth] un-guarded into the terminator, src/sparseSA.cpp:775). */
typedef struct {
    const uint8_t*  S;          /* concatenated lower-case text + '`' separators + '$' padding */
    int64_t         N;          /* S length (multiple of K) */
    int64_t         K;          /* sparseness */
    int64_t         logN;
    int64_t         NKm1;       /* N/K - 1 */
    const uint32_t* SA;         /* n = N/K sampled suffixes */
    int64_t         n;
    const uint8_t*  LCP;        /* vec_uchar::vec, 255 == look up overflow table */
    const int64_t*  LCP_M_idx;  /* vec_uchar::M (sorted by idx), split into two arrays */
    const int32_t*  LCP_M_val;
    int64_t         nM;
    const int32_t*  CHILD;      /* may be NULL when !has_child */
    const int32_t*  ISA;        /* may be NULL when !has_suflink */
    const uint32_t* KMR;        /* 2*4^10 words: (left,right) per 10-mer; NULL when !has_kmer */
    int32_t         has_child, has_suflink, has_kmer;
    int32_t         n_seq;      /* number of concatenated sequences */
    const int64_t*  startpos;   /* n_seq start offsets in S */
} alfa_index_view;

/* one sparseSA::{SMAM,MEM,SMEM,SMEMP}(P, matches, min_len, maxCount, sparseQ[, max_smem_cont]) call */
typedef struct {
    const char* P;              /* read bases exactly as the reference sees them (lower-case ASCII) */
    int32_t plen;
    int32_t min_len;
    int32_t max_count;
    int32_t sparse_q;
    int32_t max_smem_cont;      /* SMEM/SMEMP only */
} alfa_seed_query;

/* one ksw_{extend,nw,global}(qlen, query, tlen, target, m, mat, gapo, gape, w, ...) call */
typedef struct {
    const uint8_t* query;       /* nt4 codes 0..m-1 */
    const uint8_t* target;
    int32_t qlen, tlen;
    int32_t w;                  /* band */
    int32_t h0;                 /* ksw_extend only */
} alfa_dp_task;

typedef struct {
    int32_t m;                  /* residue types (the reference always passes 5) */
    int8_t  mat[64];            /* m*m scoring matrix, row-major (src/options.cpp:33-41) */
    int32_t gapo, gape;         /* POSITIVE penalties, as the reference passes them (-1*openGap) */
    int32_t zdrop;              /* ksw_extend only (100 at src/mapper.cpp:787,829) */
} alfa_dp_params;

/* outputs of ksw_extend: return value + *_qle,*_tle,*_gtle,*_gscore,*_max_off (src/ksw.c:442-447) */
typedef struct { int32_t score, qle, tle, gtle, gscore, max_off; } alfa_extend_result;

typedef struct alfa_index alfa_index;   /* opaque; owns the HBM-resident copy on one device */

/* --- residency ---------------------------------------------------------- */
/* replaces: sparseSA::load / construct leaving the arrays in host vectors
 * (src/sparseSA.cpp:390-476, 478-558); here they are copied once into HBM
 * (plus derived device layouts) and stay resident. */
int  alfa_index_upload(const alfa_index_view* host_arrays, int device, alfa_index** out);
/* replaces: sparseSA::load (src/sparseSA.cpp:390-476) + the upload above for a caller that maps on the GPU (SURVEY 8f item 3,
 * "direct-to-HBM index loader"): <prefix>.sa / .lcp / .child / .isa go from the files through two pinned staging buffers to
 * device memory, file reads overlapped with the copies, never through host vectors; the 10-mer table's device companions are
 * derived on the device.  *host == NULL: the host's share is loaded first (alfa_host_index_load_text) and returned there;
 * *host != NULL (a second GPU): it is used as it is.  The device copy is the one alfa_index_upload makes, byte for byte. */
struct alfa_host_index;                  /* include/alfa_index.h */
int  alfa_index_load_to_device(const char* prefix, int device, struct alfa_host_index** host, alfa_index** out);
void alfa_index_free(alfa_index*);
int  alfa_index_device(const alfa_index*);

/* --- seeds -------------------------------------------------------------- */
/* replaces: executeSeedType -> sa.SMAM/MEM/SMEM/SMEMP(P, matches, ...)
 * (src/mapper.cpp:519-528, src/sparseSA.h:274-283) for nq independent calls.
 * *matches is library-allocated; query i owns (*matches)[offsets[i] .. offsets[i+1])
 * in exactly the order the reference appends them; offsets has nq+1 entries
 * (caller-allocated).  min_len < K is the reference's silent no-op. */
int  alfa_seed_batch(alfa_index*, int mem_type, const alfa_seed_query* q, int nq,
                     alfa_match** matches, int64_t* offsets);

/* --- seeds -> candidate regions ----------------------- */
/* replaces, for nq independent read strands: executeSeedType followed by identifyAndFilter (src/mapper.cpp:578-704) and
 * coveredBases (:706-719, the score calculateCandRegions gives every region, :1262).  The matches stay on the device
 * between the two steps.  For strand i (qLength = q[i].plen, editDist = edit_dist[i]):
 *   (*matches)[match_off[i] .. match_off[i+1])      the seeds, as alfa_seed_batch returns them
 *   status[i] == 0: (*regions)[region_off[i] .. region_off[i+1]) are its regions in the order identifyAndFilter creates
 *                   them; region r owns (*members)[member_off[i] + r.first .. + r.count), sorted as coveredBases leaves them
 *   status[i] == 1: the strand is left to the caller (more matches or regions than the kernel holds, or matches that compare
 *                   equal under the reference's comparators but differ: std::sort's order of those cannot be reproduced)
 * match_off, region_off, member_off have nq+1 entries, status nq (caller-allocated); the four arrays are library-allocated
 * (alfa_free); *chains runs parallel to *members.  p: maxSeedCandidates, maxRegionMems, minMemRegLength, errorPercent (alnOptions,
 * src/options.h). */
typedef struct {
    int64_t ref_begin, ref_end;
    int32_t score, first, count;
    /* the chain extendChain builds first in this region (nextChain src/mapper.cpp:721-782 from the anchor it tries first, the
     * longest member, :885-893): member indices (*chains)[member_off[i] + first .. + chain_count) in chain order, its edit
     * estimate (nextChain's return value) and coverage; chain_anchor = the anchor's member index */
    int32_t chain_count, chain_anchor, chain_edit, chain_coverage, reserved;
} alfa_region;
typedef struct { int32_t max_seed_candidates, max_region_mems, min_mem_reg_length; int32_t reserved; double error_percent; } alfa_region_params;
int  alfa_seed_regions_batch(alfa_index*, int mem_type, const alfa_seed_query* q, const int32_t* edit_dist, int nq,
                             const alfa_region_params* p, alfa_match** matches, int64_t* match_off,
                             alfa_region** regions, int64_t* region_off, alfa_match** members, int64_t* member_off, int32_t** chains,
                             int32_t* status);

/* --- whole reads: seeds -> regions -> the trial loop with its DP, on the device ---------------------- */
/* replaces, for n independent reads: unpairedMatch (src/mapper.cpp:1269-1302) up to, and not including, its re-seeding
 * rescue: calculateCandRegions on both strands (:1220-1267: executeSeedType, identifyAndFilter, coveredBases, the sort by
 * score), processLisIntervals (:1118-1204) with extendAlignmentChain (:869-1081: nextChain, the closed-form gap cases, ksw_nw
 * per remaining gap, the two ksw_extend calls, duplicate suppression) and the looser second pass (:1283-1289).  Nothing
 * leaves the GPU between the steps: the DP results of one chain decide the next chain inside the kernel.
 *   status[i] == 0: read i is done.  Its alignments, in the order the reference appends them to read.alignments, are
 *                   (*alns)[aln_off[i] .. aln_off[i+1]); alignment a's chain (the seeds constructCIGAR walks, :172-271) is
 *                   (*seeds)[seed_off[i] + a.chain_first .. + a.chain_count).  No alignments = the reference finds none
 *                   without re-seeding either (possible only with rescue == 0, or when the read has seeds but no region).
 *   status[i] == 1: the caller maps read i itself through the entry points above (a capacity of the kernel was exceeded:
 *                   more than 160 seeds in a region, 128 regions, 8 alignments or 24 chains per region, a band beyond 256
 *                   diagonals; or a strand alfa_seed_regions_batch would hand back).  Results are then empty for that read.
 *   status[i] == 2: (only with rescue != 0) no seed on either strand.  The reference's next step is the seed rescue
 *                   (calculateSeedsBothStrands :542-555): the caller makes the same call for these reads with that round's
 *                   seed parameters (min_len = (max(20, K + 10) + min_len) / 2, then max(20, K + 10); sparse_q = 1,
 *                   max_seed_candidates = 100000); after the last round the read is unmapped.
 *   status[i] == 3: (only with rescue != 0) seeds and regions, but no alignment after both passes.  The reference's next
 *                   step is RESCUE2 (:1290-1300): calculateCandRegions with twice the error rate and only the looser pass --
 *                   the same call with rescue_only = 1, min_len / sparse_q from 2 x errorPercent and region_edit_dist =
 *                   (int)(2 x errorPercent x plen) + 1 (edit_dist and the thresholds stay as they were, :1280).  With
 *                   rescue_only a read without alignment is done (status 0): the reference gives up there as well.
 *                   (alfalfa_amd align runs these rounds as further calls of this entry point: csrc/mapper/mapper.cpp map_raw.)
 * The float arithmetic that depends only on the read length stays on the host and arrives per read (alfa_map_read), so
 * that the device never rounds differently from the reference.  aln_off, seed_off: n+1 entries, status: n (caller-allocated);
 * *alns and *seeds are library-allocated (alfa_free).  dp: the parameters the mapper passes to ksw_extend / ksw_nw. */
typedef struct {
    int32_t match, mismatch, open_gap, extend_gap;     /* alnOptions values; the two gap penalties NEGATIVE as alnOptions holds them */
    int32_t global, fixed_band, max_trial, max_trial_best, max_aln_count, rescue;
    int32_t mem_type, max_seed_candidates, max_smem_start, max_region_mems, min_mem_reg_length;
    int32_t cigar_alignments;           /* alnOptions.alignmentCount: extended CIGARs for that many alignments per read (0 = none) */
    int32_t rescue_only;                /* 1: only the looser second pass (:1295-1298), a read without alignment is done -- the RESCUE2 round */
    int32_t reserved;
    double error_percent;
} alfa_map_options;
typedef struct {
    const char* P; const char* Prc;     /* the read and its reverse complement as the mapper holds them (lower-case ASCII) */
    int32_t plen;
    int32_t min_len, sparse_q;          /* seed parameters: src/mapper.cpp:1226-1231 and :533-536 */
    int32_t edit_dist;                  /* (int)(errorPercent * plen) + 1                                   :1224 */
    int32_t min_score, min_coverage;    /* first pass                                                        :1281, :1282 */
    int32_t min_score2, min_coverage2;  /* second pass (looser thresholds, at most 20 alignments)            :1285-1287 */
    int32_t region_edit_dist;           /* identifyAndFilter's editDist when it differs from edit_dist (RESCUE2: :1293 -> :1224); 0 = edit_dist */
    int32_t reserved;
} alfa_map_read;
enum { ALFA_MAP_DONE = 0, ALFA_MAP_CALLER = 1, ALFA_MAP_NO_SEED = 2, ALFA_MAP_NO_ALIGNMENT = 3 };
typedef struct {
    int64_t ref_begin, ref_end; int32_t q_begin, q_end, score, forward, chain_first, chain_count;
    int32_t op_first, op_count, edit_dist, reserved;   /* the extended CIGAR, when it was made with the alignment: see below */
} alfa_map_alignment;
/* CIGAR recovery (read_t::postprocess src/mapper.cpp:322-358 -> constructCIGAR :172-271 -> ksw_global + recoverCIGAR4 :80-117)
 * for the o->cigar_alignments best-scoring alignments of a read (ties as std::sort leaves them), in the same kernel: alignment a
 * with op_count >= 0 owns (*ops)[op_off[i] + a.op_first .. + a.op_count), the operations of its X0:Z tag in order -- `code` is
 * the character ('=', 'X', 'I', 'D', 'S'), `len` the number in front of it exactly as the reference prints it (the closed-form
 * gap cases can print 0 or negative lengths) -- and edit_dist is its NM.  op_count == -1: not made here (not among the best, a
 * degenerate ksw_global call, a capacity); the caller runs constructCIGAR through alfa_ksw_global_batch as before. */
typedef struct { int32_t len, code; } alfa_map_cigar_op;
int  alfa_map_reads_batch(alfa_index*, const alfa_map_options* o, const alfa_dp_params* dp, const alfa_map_read* reads, int n,
                          int32_t* status, int64_t* aln_off, alfa_map_alignment** alns, int64_t* seed_off, alfa_match** seeds,
                          int64_t* op_off, alfa_map_cigar_op** ops);

/* --- banded DP ---------------------------------------------------------- */
/* replaces: ksw_extend(...)  src/ksw.h:106 / src/ksw.c:362 */
int  alfa_ksw_extend_batch(const alfa_dp_params* p, const alfa_dp_task* t, int n, alfa_extend_result* r);
/* replaces: ksw_nw(...)      src/ksw.h:83  / src/ksw.c:544 */
int  alfa_ksw_nw_batch(const alfa_dp_params* p, const alfa_dp_task* t, int n, int32_t* score);
/* replaces: ksw_global(...)  src/ksw.h:82  / src/ksw.c:468.  *cigar is library-allocated
 * (BAM encoding len<<4|op, already in forward order); task i owns
 * (*cigar)[cig_off[i] .. cig_off[i]+n_cigar[i]); cig_off has n+1 entries. */
int  alfa_ksw_global_batch(const alfa_dp_params* p, const alfa_dp_task* t, int n, int32_t* score,
                           uint32_t** cigar, int32_t* n_cigar, int64_t* cig_off);

/* --- local alignment (striped Smith-Waterman) ---------------------------- */
/* replaces: ksw_align(qlen, query, tlen, target, m, mat, gapo, gape, xtra, 0)  src/ksw.h:63 / src/ksw.c:330
 * (= ksw_qinit :58 + ksw_u8 :106 / ksw_i16 :223, second pass for the start positions), the mate-rescue call of
 * src/mapper.cpp:1571 and :2196.  `xtra` carries the reference's KSW_X* flags and threshold (src/ksw.h:6-9).
 * p->gapo / p->gape are used AS GIVEN: the reference's call sites pass alnOptions.openGap / extendGap, which are
 * negative, and its unsigned saturating arithmetic then never lets a gap pay; that behaviour is reproduced
 * (results are bit-identical for either sign).  p->zdrop is ignored.  EINVAL: qlen < 0, tlen < 0,
 * qlen * max(mat) > 32767, or KSW_XSUBO with max(mat) <= 0 (the reference divides by zero). */
#define ALFA_KSW_XBYTE  0x10000
#define ALFA_KSW_XSTOP  0x20000
#define ALFA_KSW_XSUBO  0x40000
#define ALFA_KSW_XSTART 0x80000
typedef struct {
    const uint8_t* query;       /* nt4 codes 0..m-1 */
    const uint8_t* target;
    int32_t qlen, tlen;
    int32_t xtra;
} alfa_align_task;
typedef struct { int32_t score, te, qe, score2, te2, tb, qb; } alfa_kswr;   /* = kswr_t, src/ksw.h:14-19 */
int  alfa_ksw_align_batch(const alfa_dp_params* p, const alfa_align_task* t, int n, alfa_kswr* r);

/* --- dynProg (dp.cpp) ---------------------------------------------------- */
#define ALFA_DPBAND_STATIC 0   /* dynProg::dpBandStatic src/dp.cpp:648 */
#define ALFA_DPBAND_FULL   1   /* dynProg::dpBandFull   src/dp.cpp:761 */
#define ALFA_DPBAND_SCORE  2   /* dynProg::dpBandScore  src/dp.cpp:872 */
#define ALFA_DPBAND_BASIC  3   /* dynProg::dpBasic      src/dp.cpp:926 */
typedef struct {
    const char* ref;  int64_t ref_len;       /* ASCII, as the reference's std::string */
    const char* query; int64_t query_len;
    int64_t refB, refE; int32_t queryB, queryE;   /* boundaries (src/dp.h:29-36) */
    uint8_t freeRefB, freeRefE, freeQueryB, freeQueryE, local;  /* dp_type (src/dp.h:49-57) */
    int32_t out_type;                        /* outputType (src/dp.h:68-74) */
    int32_t band1, band2;                    /* minband,maxband | bandLeft,bandRight */
} alfa_dpband_task;
typedef struct { int32_t match, mismatch, openGap, extendGap; int32_t affine; } alfa_dpband_scores;
typedef struct {
    int32_t ret, dpScore, editDist;
    int64_t refB, refE; int32_t queryB, queryE;   /* boundaries after the call */
    int32_t n_cigar; int64_t cig_off;        /* run-length =XID ops in (*cig_chars,*cig_lens) */
} alfa_dpband_result;
int  alfa_dp_band_batch(int variant, const alfa_dpband_scores* s, const alfa_dpband_task* t, int n,
                        alfa_dpband_result* r, char** cig_chars, int32_t** cig_lens);

/* --- resident batches ---------------------------------------------------- */
/* The synchronous entry points above are create + launch + fetch of the objects below.
 * A staged caller (the batch mapper, bench.py) keeps batches resident in HBM and
 * separates the three phases: create() packs the inputs and copies them to the device
 * (synchronous), launch() only enqueues the kernels on the calling thread's stream for
 * that device (asynchronous, no allocation, no host synchronisation) and fetch() waits,
 * copies the results back and unpacks them exactly like the synchronous call.
 * No counterpart in the reference (it has no device); semantics are those of the
 * corresponding synchronous call.
 * Threading: a batch / plan belongs to the host thread that created it -- create, launch and fetch are ordered on that
 * thread's streams, so launch and fetch from any other thread are refused with ALFA_EINVAL (free may come from any thread
 * once the owner has fetched or synchronised).  Different threads work on different objects concurrently.  Launching an
 * object again before the previous launch has been fetched or synchronised (possible with alfa_set_lazy_join) overwrites
 * results that are still being produced: synchronise first. */
typedef struct alfa_dp_batch alfa_dp_batch;
#define ALFA_DP_NW     0
#define ALFA_DP_GLOBAL 1
#define ALFA_DP_EXTEND 2
int     alfa_dp_batch_create(int mode, const alfa_dp_params* p, const alfa_dp_task* t, int n, int device, alfa_dp_batch** out);
int     alfa_dp_batch_launch(alfa_dp_batch*);
/* out: alfa_extend_result[n] (EXTEND) or int32 score[n]; cigar/n_cigar/cig_off as alfa_ksw_global_batch (GLOBAL only, else NULL) */
int     alfa_dp_batch_fetch(alfa_dp_batch*, void* out, uint32_t** cigar, int32_t* n_cigar, int64_t* cig_off);
int64_t alfa_dp_batch_cells(const alfa_dp_batch*);       /* inner-loop cells of the static band envelope */
/* diagnostics: *n_fast = tasks of the batch that run in the band-in-registers kernel; returns how many ksw_extend tasks of a
 * launch gave up there because their band started to narrow and were finished by the moving-band code (synchronises; -1 on error) */
int64_t alfa_dp_batch_giveups(alfa_dp_batch*, int64_t* n_fast);
/* HIP-event time of the last launch's kernels (after fetch/sync).  Under alfa_set_lazy_join the kernels run on side streams
 * that are joined at fetch time: the closing event is recorded there, so the figure spans launch -> fetch-time join on the
 * calling thread's stream (other work enqueued in between included), and it is 0 until the batch has been fetched. */
double  alfa_dp_batch_kernel_ms(const alfa_dp_batch*);
void    alfa_dp_batch_free(alfa_dp_batch*);

typedef struct alfa_seed_plan alfa_seed_plan;
int     alfa_seed_plan_create(alfa_index*, int mem_type, const alfa_seed_query* q, int nq, alfa_seed_plan** out);
int     alfa_seed_plan_launch(alfa_seed_plan*);
int     alfa_seed_plan_fetch(alfa_seed_plan*, alfa_match** matches, int64_t* offsets);
int64_t alfa_seed_plan_queries(const alfa_seed_plan*);   /* number of (read,prefix) traversals */
/* 1 when the plan evaluates its queries in 10-mer order (SMAM on an index whose 10-mer table is >= 90 % full and that has
 * >= 2^21 sampled suffixes: far beyond the caches; sorted_*_kernel in seed_kernels.hip), else 0.  Results are identical. */
int     alfa_seed_plan_sorted(const alfa_seed_plan*);
double  alfa_seed_plan_kernel_ms(const alfa_seed_plan*); /* HIP-event time of the traversal kernel of the last launch */
void    alfa_seed_plan_free(alfa_seed_plan*);

/* wait for everything enqueued on the calling thread's stream for `device` */
int     alfa_sync(int device);
/* DP batches run their kernel bins as independent chains on side streams.  By default a launch joins them back into the
 * calling thread's stream before it returns (so alfa_dp_batch_kernel_ms brackets exactly that batch); with lazy join on,
 * the join is deferred to the next fetch / alfa_sync / alfa_timer_stop so that successive batches overlap.  No counterpart
 * in the reference. */
int     alfa_set_lazy_join(int device, int on);
/* HIP-event stopwatch on that same stream: start, (launch...), stop -> elapsed device milliseconds */
int     alfa_timer_start(int device);
int     alfa_timer_stop(int device, double* ms);

/* --- ksw_global for callers that cannot allocate ---------------------------------------------------------
 * NO counterpart in the reference.  For callers that cannot call hipMalloc or any other allocating routine: code
 * under HIP graph capture, tight serving loops, harnesses that forbid allocation.  Same results as
 * alfa_ksw_global_batch (same kernels: the wave-per-task fill + tiled backtrack), but every pointer below except `p`
 * is a DEVICE pointer on `device` and nothing is allocated, copied to the host or synchronised: the call only
 * enqueues kernels on `stream` (a hipStream_t; NULL = the default stream) and returns.
 *   d_tasks[i]   qoff/toff = byte offsets of the nt4 query/target of task i in d_seq; w = band
 *   d_score[i]   ksw_global's return value
 *   d_cigar      n slots of cig_stride words; the CIGAR of task i (BAM encoding, forward order) is the LAST
 *                d_n_cigar[i] words of slot i: d_cigar[i*cig_stride + cig_stride - d_n_cigar[i] .. i*cig_stride + cig_stride)
 *   d_n_cigar[i] number of CIGAR operations; -1 when tlen > qlen + w (the reference indexes outside its matrix there,
 *                as alfa_ksw_global_batch); -2 (and score INT32_MIN) for a task outside this entry point's envelope:
 *                qlen < 1, tlen < 1, tlen > max_tlen, 2*min(w, max(qlen,tlen))+1 > 256 or qlen+tlen+2 > cig_stride
 * What the allocating entry point allocates per batch (parameters, internal task records, direction bytes) lives in
 * one fixed device array of ALFA_NOALLOC_SCRATCH_BYTES per GPU, held until the process ends.  Limit, checked on the
 * host: 256 + roundup(64*n, 256) + 256*n*max_tlen <= ALFA_NOALLOC_SCRATCH_BYTES, else ALFA_EINVAL (split the batch).
 * Every call on a GPU uses that GPU's one array: calls that overlap in time on the same GPU (different streams or
 * threads) race; calls on one stream are ordered.  Sets the current device to `device`. */
#define ALFA_NOALLOC_SCRATCH_BYTES (32u << 20)
typedef struct { int64_t qoff, toff; int32_t qlen, tlen, w, pad; } alfa_dev_task;
int  alfa_ksw_global_noalloc(const alfa_dp_params* p, const alfa_dev_task* d_tasks, int n, int max_tlen, const uint8_t* d_seq,
                             int32_t* d_score, uint32_t* d_cigar, int32_t* d_n_cigar, int cig_stride, int device, void* stream);

/* --- misc --------------------------------------------------------------- */
void        alfa_free(void*);            /* every library-allocated host buffer */
const char* alfa_last_error(void);       /* thread-local, human readable */
int         alfa_device_count(void);     /* 0 when no HIP device is usable */
/* Hardware queues of the HIP runtime (GPU_MAX_HW_QUEUES): the host application chooses.  Loading the library leaves a
 * value that is already in the environment alone; only when the variable is absent does it install 16 (the DP chains of
 * large batches run side by side on that many queues), and ALFA_HW_QUEUES=<n> is the explicit request that wins (0 = hands
 * off).  Returns 1 when the library installed its default at load time, else 0.  No counterpart in the reference. */
int         alfa_hw_queues_defaulted(void);
/* diagnostics: lowest / highest HIP stream priority among the calling thread's streams for `device`.  The library creates all
 * of them with the default priority and relies on it (cross-stream event waits between priority levels can starve: DESIGN.md
 * section 5); a test checks *lo == *hi.  No counterpart in the reference. */
int         alfa_stream_priority_span(int device, int* lo, int* hi);

/* per-kernel timing of the calling thread's most recent batch call, measured with
 * HIP events on the library's own stream: kernel milliseconds, algorithmic work
 * units (SA: bytes per SURVEY 8d; DP: inner-loop cells) -- used by bench.py. */
typedef struct { double kernel_ms; double h2d_ms, d2h_ms; int64_t work_units; int32_t launches; } alfa_timing;
int  alfa_last_timing(alfa_timing* out);

#ifdef __cplusplus
}
#endif
#endif /* ALFA_HOT_H */
