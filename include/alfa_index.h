/* alfa_index.h -- host-side index construction / persistence (C ABI).
 *
 * Plumbing around the hot path: produces, loads and saves exactly the arrays the
 * reference's `sparseSA` object holds, bit for bit, so that both sides search
 * the same index and so that index files written by either side are
 * interchangeable.  Replaces
 *
 *   sparseSA::loadRef / init / construct     src/sparseSA.cpp:125,102,478
 *   sparseSA::computeLCP/Child/Kmer          src/sparseSA.cpp:161,176,222
 *   sparseSA::save / load                    src/sparseSA.cpp:316,390
 *   sparseSA::from_set / getChromBounds      src/sparseSA.h:173-190
 *
 * The arrays stay on the host here; alfa_index_upload() (alfa_hot.h) makes them
 * resident in HBM.
 */
#ifndef ALFA_INDEX_H
#define ALFA_INDEX_H
#include "alfa_hot.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct alfa_host_index alfa_host_index;   /* opaque, owns the host arrays */

/* loadRef(fasta) + init(K, suflink, child, kmer) + construct()   (src/main.cpp:335-340).
 * FASTA/FASTQ, plain or gzip; record name = header up to the first whitespace;
 * bases are lower-cased; records are joined with '`' (src/sparseSA.cpp:140-153). */
int alfa_host_index_build_fasta(const char* fasta_path, int64_t K, int suflink, int child, int kmer,
                                alfa_host_index** out);
/* same, from in-memory sequences */
int alfa_host_index_build_mem(int n_seq, const char* const* names, const char* const* seqs, const int64_t* seq_lens,
                              int64_t K, int suflink, int child, int kmer, alfa_host_index** out);
/* sparseSA::load: reads <prefix>.aux/.ref/.sa/.lcp/.child/.kmer/.isa as written by the reference.
 * Returns ALFA_EINVAL if <prefix>.aux is missing (the reference returns false -> caller rebuilds). */
int alfa_host_index_load(const char* prefix, alfa_host_index** out);
/* the same for a caller whose arrays go straight to device memory (alfa_index_load_to_device, alfa_hot.h): only <prefix>.aux and
 * .ref are read -- text, sequence names and offsets, what the mapper's host side needs; the view's SA / LCP / CHILD / KMR are null */
int alfa_host_index_load_text(const char* prefix, alfa_host_index** out);
/* sparseSA::save: byte-identical files */
int alfa_host_index_save(const alfa_host_index*, const char* prefix);

/* Index construction on the GPU (libalfa_hot only; replaces sparseSA::construct / computeLCP / computeChild / computeKmer,
 * src/sparseSA.cpp:478-558, 161-313, with data-parallel kernels): after alfa_index_build_on_device(device >= 0) the two build
 * calls above compute SA, LCP (+ overflow table), CHILD, KMR and ISA on that device -- the same arrays, bit for bit -- and
 * keep doing so until alfa_index_build_on_device(-1).  Process-wide switch; not meant to be flipped while builds run. */
int alfa_index_build_on_device(int device);

const alfa_index_view* alfa_host_index_view(const alfa_host_index*);
const char* alfa_host_index_descr(const alfa_host_index*, int seq);
/* sparseSA::from_set (src/sparseSA.h:173-180) */
void alfa_host_index_from_set(const alfa_host_index*, int64_t hit, int64_t* seq, int64_t* seqpos);
/* sparseSA::getChromBounds (src/sparseSA.h:182-190) */
void alfa_host_index_chrom_bounds(const alfa_host_index*, int64_t hit, int64_t* chr_start, int64_t* chr_end);
void alfa_host_index_free(alfa_host_index*);

#ifdef __cplusplus
}
#endif
#endif
