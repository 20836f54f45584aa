/*
 * snekmer_hip.h — C ABI of libsnekmer_hip.so, the MI355X (gfx950) implementation of Snekmer's
 * AAR-kmer vectorize + cosine hot path.
 *
 * The reference (PNNL-CompBio/Snekmer v1.3.0) is pure Python and has no FFI of its own; its
 * boundary for this path is the Python module surface.  Each entry point below therefore cites
 * the reference *Python* code whose per-sequence loop it replaces with one batched device call.
 * snekmer_amd/_hip.py is the ctypes binding; INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *   - Every function returns an int status: SKM_OK or a negative SKM_E_* code;
 *     skm_last_error() returns a thread-local message for the last failure.
 *   - A context is bound to one device and one HIP stream.  Calls on one context are issued in
 *     order on that stream and are asynchronous unless stated ("host-synchronous"); use
 *     skm_sync() before reading results through anything but skm_memcpy_d2h.  A context is not
 *     re-entrant; distinct contexts may be used from distinct host threads.
 *   - Pointers named d_* are device pointers (from skm_malloc or any hipMalloc), h_* are host
 *     pointers.  The caller owns every buffer; the library keeps only grow-only scratch inside
 *     the context.
 *   - Sequences are passed packed: d_seq holds all residues back to back, d_off[i]..d_off[i+1]
 *     delimits sequence i (int64, n+1 entries).
 *   - h_translate / h_rank are the 256-byte tables of snekmer_amd.alphabet.build_lut():
 *     translate[b] = recoded byte, rank[b] = class rank (ASCII order of class letters) or 0xFF
 *     when the recoded byte is not a class letter.
 *   - k-mer code = sum_i rank(c_i) * nsym^(k-1-i); code_bits is 32 or 64 and the all-ones word of
 *     that width is the "invalid window" sentinel.  nsym^k must be < 2^code_bits.
 */
#ifndef SNEKMER_HIP_H
#define SNEKMER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKM_ABI_VERSION 7

#define SKM_OK 0
#define SKM_E_BADARG (-1)
#define SKM_E_NOMEM (-2)
#define SKM_E_HIP (-3)
#define SKM_E_OVERFLOW (-4)
#define SKM_E_UNSUPPORTED (-5)
#define SKM_E_COMM (-6)
#define SKM_E_STALE (-7) /* skm_graph_launch: the graph holds a scratch address that has since been reallocated */

typedef struct skm_ctx skm_ctx;

/* ---- library / context ------------------------------------------------------------------- */
int skm_abi_version(void);
const char *skm_last_error(void);
int skm_device_count(int *h_count);
/* Switches between EXACT kernels, for tests and A/B timing; process-wide.  Each starts from the environment variable of the
 * same name, read ONCE when the library is first used (round 5: no getenv on any call path), and can be set afterwards:
 *   SKM_SORT            "rocprim" | "onesweep": the basis stage's sort whatever the size (default: by size)
 *   SKM_COSINE_PATH     "lists" | "cursor": skm_cosine_csr's routing (default: by shape)
 *   SKM_HEAVY_PANEL     "0" | "1": heavy-row panels off / on (default: by the previous call's heavy-row count)
 *   SKM_HEAVY_PACK      "0" | "1": 16-bit packed tiles for heavy rows off / on (default: by the same count)
 *   SKM_COSINE_OVERLAP  "1": the blocked two-stream schedule inside skm_cosine_csr
 *   SKM_GRAM_SHAPE      "1".."4": lane-group shape of k_gram_sparse
 *   SKM_DENSE_VARIANT   "1" | "2" | "6" | "7" | "10" | "11": the int8 GEMM kernel (see skm_cosine_dense_i8)
 * value NULL: back to what the environment said.  An unknown name or a value outside the list: SKM_E_BADARG.
 * skm_get_option copies the current value ("" when unset) into buf.  No reference counterpart (the reference has one
 * code path per stage). */
int skm_set_option(const char *name, const char *value);
int skm_get_option(const char *name, char *buf, int cap);
int skm_create(int device_id, skm_ctx **out_ctx);
/* A context whose stream may only use the compute-unit groups first..last of 0..7 (group of CU i: (i / 8) % 8, i.e. eight
 * CUs of every XCD per group).  For a SIDE context: engine.OverlappedPipeline vectorizes the next batch on one confined
 * to groups 0-3 while the main context's store-bound writer runs, so that neither fills the other's wave slots (9.66
 * against 10.5 ms per step unconfined and 10.8 on one stream; no reference counterpart: one process per FASTA file,
 * snekmer/rules/kmerize.smk:57-65).  SKM_E_UNSUPPORTED on devices with fewer than 64 or more than 512 CUs. */
int skm_create_confined(int device_id, int first_cu_group, int last_cu_group, skm_ctx **out_ctx);
int skm_destroy(skm_ctx *ctx);
/* Ordering between two contexts of one process on one device (a context is one stream): skm_event_record marks the
 * current end of ctx's stream in slot [0, SKM_EVENT_SLOTS); skm_stream_wait makes everything queued on ctx from now
 * on wait for the mark last recorded in src's slot (no-op if none was).  Neither blocks the host.  Used by
 * engine.OverlappedPipeline: the next batch is vectorized on a second context while this one's Gram runs. */
#define SKM_EVENT_SLOTS 8
int skm_event_record(skm_ctx *ctx, int slot);
int skm_stream_wait(skm_ctx *ctx, skm_ctx *src, int slot);
int skm_sync(skm_ctx *ctx); /* host-synchronous */
/* Replay of a fixed sequence of calls as one HIP graph (no reference counterpart; the reference's unit of work is one
 * Snakemake job per FASTA file of 50-3700 records, snekmer/rules/kmerize.smk:57-65, where a vectorize + cosine step is a
 * few dozen launches of microseconds each and the host's launch cost is the step).  skm_graph_begin opens a capture on
 * the context's stream; every call made on the context until skm_graph_end is RECORDED, NOT RUN (so only calls that do
 * not wait for the device may be made: skm_vectorize_csr with max_seq_len > 0, skm_cosine_csr, skm_csr_to_dense_i8,
 * skm_cosine_dense_i8, skm_cosine_fixup_rows, ...; the same calls must have run once before the capture, so that the
 * context's scratch has its size: scratch may not grow inside a capture).  skm_graph_launch queues the recorded work: the
 * same kernels on the same buffers with the same sizes, reading whatever those buffers hold now.  It returns SKM_E_STALE
 * when a scratch buffer of the context has been reallocated since the capture (capture again).  Choices the library
 * makes on the host between kernels (which of two exact kernels runs) are frozen into the graph. */
typedef struct skm_graph skm_graph;
int skm_graph_begin(skm_ctx *ctx);
int skm_graph_end(skm_ctx *ctx, skm_graph **out_graph);
int skm_graph_launch(skm_ctx *ctx, skm_graph *graph);
int skm_graph_nodes(skm_graph *graph, int64_t *h_nodes); /* kernels + fills + copies in the capture: the GPU operations of a replay */
int skm_graph_destroy(skm_ctx *ctx, skm_graph *graph);
int skm_device_info(skm_ctx *ctx, char *h_name, int name_cap, int *h_cus, int64_t *h_mem_bytes);

/* ---- device memory ----------------------------------------------------------------------- *
 * The arrays of a job (numpy arrays in the reference: snekmer/rules/kmerize.smk:112,132-139) come from a per-device
 * pool of size classes.  skm_free never waits and never calls hipFree: the block is parked behind an event on every
 * stream of the device that still has work queued, and skm_malloc hands it out again only once those have completed -
 * freeing an array while kernels that read it are queued is safe by construction, on any context of the device (the
 * context argument of skm_free only names the device).  Memory goes back to the runtime when the device runs out of it
 * or skm_mem_trim asks, after an explicit wait for every stream.  Environment SKM_GUARD=1 (read once): 512 canary bytes
 * behind every array, checked by skm_free (SKM_E_HIP naming the array's size and the first overwritten byte).
 * skm_mem_stats: h_out8 = {bytes in live arrays, bytes parked, hipMalloc calls, hipFree calls, allocations served from
 * parked blocks, parked blocks, cached streams, candidates skipped because their events had not completed}. */
int skm_malloc(skm_ctx *ctx, size_t bytes, void **out_dptr);
int skm_free(skm_ctx *ctx, void *dptr);
int skm_mem_trim(skm_ctx *ctx, int64_t *h_released_bytes); /* host-sync (every stream of the device) */
int skm_mem_stats(skm_ctx *ctx, int64_t *h_out8);
/* For a watchdog thread while another thread does not return from a call: per context whether its streams are idle or
 * busy, the first timed kernels (skm_profile_enable) that started and did not finish, and the pool's counters, as text.
 * No context argument: it reports every context of the process. */
int skm_debug_report(char *h_buf, int cap);
/* Pinned host memory that kernels of this context's device can read and write at the same address (zero-copy,
 * coherent): the staging of the per-record API, where a launch + skm_sync beats two explicit copies.  The pointer
 * may be passed wherever a d_* argument is expected. */
int skm_host_alloc(skm_ctx *ctx, size_t bytes, void **out_hptr);
int skm_host_free(skm_ctx *ctx, void *hptr);
int skm_memcpy_h2d(skm_ctx *ctx, void *d_dst, const void *h_src, size_t bytes); /* host-sync */
/* The same copy queued on the context's stream without waiting: h_src must be pinned memory (skm_host_alloc) that stays
 * untouched until the copy has run (skm_event_record behind it + skm_event_query, or skm_sync).  This is how a stream of
 * batches arrives from the host (every job of the reference starts from a file, snekmer/rules/kmerize.smk:89-129):
 * batch i + 1 is uploaded into recycled device buffers while batch i is computed. */
int skm_memcpy_h2d_async(skm_ctx *ctx, void *d_dst, const void *h_src_pinned, size_t bytes);
/* SKM_OK and *h_done = 1 once everything queued on the context before skm_event_record(ctx, slot) has run, 0 before
 * (and 1 when nothing was ever recorded in the slot).  Never waits. */
int skm_event_query(skm_ctx *ctx, int slot, int *h_done);
int skm_memcpy_d2h(skm_ctx *ctx, void *h_dst, const void *d_src, size_t bytes); /* host-sync */
int skm_memcpy_d2d(skm_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
int skm_memset(skm_ctx *ctx, void *d_dst, int byte_value, size_t bytes);

/* ---- per-kernel timing (HIP events on the context's stream) ------------------------------- */
/* While enabled, every kernel launch made by the library is bracketed by hipEvents on the
 * context's stream.  skm_profile_read synchronises, then returns for the kernel whose name
 * starts with `h_prefix` the number of launches and the summed device time since the last
 * skm_profile_reset. */
int skm_profile_enable(skm_ctx *ctx, int on);
int skm_profile_reset(skm_ctx *ctx);
int skm_profile_read(skm_ctx *ctx, const char *h_prefix, int64_t *h_launches, double *h_total_ms);
/* Writes "name\tlaunches\ttotal_ms\n" lines for every kernel seen; returns bytes needed. */
int skm_profile_dump(skm_ctx *ctx, char *h_buf, int cap, int *h_needed);

/* ---- a3: recode -------------------------------------------------------------------------- */
/* Replaces the per-sequence `reduce()` of snekmer/vectorize.py:173-195
 * (str.rstrip("*") + str.translate) as called from rules/kmerize.smk:122-126.
 * d_out has the layout of d_seq; d_outlen[i] = length after stripping trailing '*'.  Bytes of
 * d_out past d_outlen[i] within a record are unspecified. */
int skm_recode(skm_ctx *ctx, const uint8_t *h_translate, const uint8_t *d_seq, const int64_t *d_off,
               int64_t n, uint8_t *d_out, int32_t *d_outlen);

/* ---- a5/a6: k-mer windows in window order ------------------------------------------------- */
/* Replaces KmerVec._kmer_gen + KmerVec.reduce_vectorize (snekmer/vectorize.py:239-249, :292-328)
 * as called per record from rules/kmerize.smk:95,118.  For sequence i with stripped length L_i,
 * d_nwin[i] = max(L_i-k+1, 0) and d_codes[d_off[i] + p] (p < d_nwin[i]) is the code of window p
 * or the sentinel when the window holds a non-class character.  d_codes needs one word of
 * code_bits per residue. */
int skm_kmer_codes(skm_ctx *ctx, const uint8_t *h_rank, int nsym, int k, int code_bits,
                   const uint8_t *d_seq, const int64_t *d_off, int64_t n, void *d_codes,
                   int32_t *d_nwin);

/* ---- a12: per-sequence k-mer counts (CSR) -------------------------------------------------- */
/* Replaces the count loops of rules/learn.smk:359-383 and rules/apply.smk:188-206 (dict count of
 * every window of the reduced string, then projection on the basis) with one pass that yields,
 * per sequence, its distinct valid codes in ascending order with multiplicities.
 * Outputs: d_rowptr[n+1]; d_codes / d_counts (/ d_firstpos, optional: index of the first window
 * carrying that code, needed for the reference's first-seen basis order) with capacity
 * cap_entries >= total residues.  *h_nnz receives the number of entries (host-synchronous).
 * max_seq_len: an upper bound on the longest sequence of the batch (max of d_off[i+1] - d_off[i]; the offsets are host
 * data wherever a batch is packed), or 0 when the caller has none.  With the bound every count kernel is launched
 * with a worst-case grid and reads its work list on the device; without it the call also fetches the size classes
 * of the sequences to learn whether any has more than 8192 windows (one more host wait).  The bound is CHECKED on
 * the device: a sequence with more windows than max_seq_len - k + 1 gets an empty row (it is never counted into scratch
 * sized by the bound) and the context remembers it: this call (which waits for its entry count), or after
 * skm_vectorize_csr the next skm_sync / skm_memcpy_d2h on the context, returns SKM_E_BADARG once. */
int skm_count_csr(skm_ctx *ctx, const uint8_t *h_rank, int nsym, int k, int code_bits,
                  const uint8_t *d_seq, const int64_t *d_off, int64_t n, int64_t total_residues,
                  int64_t max_seq_len, int64_t cap_entries, int64_t *d_rowptr, void *d_codes, uint32_t *d_counts,
                  uint32_t *d_firstpos, int64_t *h_nnz);

/* ---- a11: observed basis ------------------------------------------------------------------ */
/* Replaces the dict-of-observed-k-mers pass of rules/kmerize.smk:89-104 and the np.unique of
 * scripts/cluster_cluster.py:67.  Input: the CSR of skm_count_csr over n sequences; key_bits =
 * number of significant low bits in a code (ceil(log2(nsym^k)); 0 means code_bits).
 * Outputs (each optional unless noted, capacity nnz unless noted):
 *   *h_ncols          number of distinct codes B (required; host-synchronous)
 *   d_basis[B]        distinct codes ascending (== lexicographic k-mer order), code_bits wide
 *   d_colidx[nnz]     basis column of every CSR entry (required)
 *   d_df[B]           sequences containing the k-mer
 *   d_total[B]        total occurrences (what `min_filter` tests, kmerize.smk:96-104)
 *   d_firstkey[B]     (row << 32 | first window); ascending order == the reference's first-seen order
 *   d_fs_order[B]     basis columns listed in first-seen order (needs d_firstpos)
 *   d_colptr[nnz+1], d_post[nnz]   the same matrix column-major (postings: for each basis column
 *                     the rows holding it, ascending; entry = row | (uint64)count << 32);
 *                     only d_colptr[0..B] is meaningful.
 * flags: SKM_BASIS_ELIDE_SINGLETONS (only with postings and without df/total/first-seen outputs):
 *   k-mers found in a single sequence can only ever contribute to that row's own norm, so their
 *   CSR entries get d_colidx = 0xFFFFFFFF and no posting is written for them; skm_cosine_csr treats
 *   such an entry as "pairs with its own row only".  Saves a third of the random traffic.
 *        SKM_BASIS_POST32 (n < 2^24, no d_total): d_post is written as 32-bit words
 *   row | min(count, 255) << 24 instead of 64-bit ones, which halves the bytes the cosine kernels
 *   gather (their dominant read).  A stored 255 is an escape: the real count of that posting is
 *   d_postcnt[its index] (uint32[nnz], touched for such postings only: a k-mer repeated >= 255 times
 *   within one sequence).  Without the flag d_postcnt is ignored and may be NULL. */
#define SKM_BASIS_ELIDE_SINGLETONS 1
#define SKM_BASIS_POST32 2
int skm_basis_build(skm_ctx *ctx, int code_bits, int key_bits, int flags, int64_t n, int64_t nnz, const int64_t *d_rowptr,
                    const void *d_codes, const uint32_t *d_counts, const uint32_t *d_firstpos,
                    int64_t *h_ncols, void *d_basis, uint32_t *d_colidx, uint32_t *d_df,
                    uint64_t *d_total, uint64_t *d_firstkey, uint32_t *d_fs_order,
                    uint32_t *d_colptr, void *d_post, uint32_t *d_postcnt);

/* The whole pre-cosine path of the square cosine in ONE call with no data-dependent read-back: skm_count_csr +
 * skm_basis_build(SKM_BASIS_ELIDE_SINGLETONS, postings) + skm_row_norms_csr, i.e. the body of
 * rules/kmerize.smk:89-104 + rules/learn.smk:359-383 as the cosine stage needs it.  Sizes that depend on the data stay
 * on the device: the entry count is d_rowptr[n], the number of basis columns *d_ncols (device int64); every launch is
 * sized by total_residues.  Outputs as documented for the three calls it replaces, capacity cap_entries >
 * total_residues each (d_colptr: cap_entries + 1); d_codes past the entry count is unspecified (all-ones under
 * SKM_SORT=rocprim, whose capacity-sized sort runs over that fill);
 * d_rnorm / d_normsq are optional.  No result is read back (read d_rowptr[n] / *d_ncols with skm_memcpy_d2h when the
 * host needs them).  With max_seq_len > 0 (see skm_count_csr) the call never waits for the device: the host may queue
 * any number of steps ahead.  With max_seq_len == 0 one wait remains: the size-class histogram of the sequences is
 * awaited on an event recorded behind the classification kernel, after the common-case count kernels have been
 * queued.  n >= 1 and total_residues >= 1.  With d_basis, d_colidx, d_colptr, d_post and d_ncols ALL null the call stops after the count
 * stage (CSR + norms, d_codes not padded): what the dense route of engine.Pipeline needs. */
int skm_vectorize_csr(skm_ctx *ctx, const uint8_t *h_rank, int nsym, int k, int code_bits, const uint8_t *d_seq,
                      const int64_t *d_off, int64_t n, int64_t total_residues, int64_t max_seq_len, int64_t cap_entries,
                      int64_t *d_rowptr, void *d_codes, uint32_t *d_counts, void *d_basis, uint32_t *d_colidx, uint32_t *d_colptr,
                      uint64_t *d_post, float *d_rnorm, uint64_t *d_normsq, int64_t *d_ncols);

/* Column-major copy (postings) of any CSR with column ids < ncols; rows ascending per column. */
int skm_csr_transpose(skm_ctx *ctx, int64_t n, int64_t nnz, int64_t ncols, const int64_t *d_rowptr,
                      const uint32_t *d_colidx, const uint32_t *d_counts, uint32_t *d_colptr,
                      uint64_t *d_post);

/* Row pointers of `nparts` CSR pieces laid end to end: d_local holds the pieces' own rowptr arrays
 * back to back (piece p has h_nrows[p]+1 entries starting at 0); d_rowptr receives the
 * sum(h_nrows)+1 entries of the concatenated matrix.  Used after the all-gather of CSR shards. */
int skm_csr_concat_rowptr(skm_ctx *ctx, int nparts, const int64_t *h_nrows, const int64_t *h_nnz,
                          const int64_t *d_local, int64_t *d_rowptr);

/* Dense rows from CSR (the `vecs` array of rules/kmerize.smk:112-119 and the count matrix of
 * rules/learn.smk:376-383).  d_colmap (optional, [ncols_in]) renumbers columns; 0xFFFFFFFF drops
 * one.  mode 0 = counts, 1 = presence (0/1).  dtype: 0 = float64, 1 = float32, 2 = int8
 * (saturation reported as SKM_E_OVERFLOW is NOT checked here; see skm_csr_max_count).
 * d_out is [n x ld] row-major and is fully overwritten. */
int skm_csr_to_dense(skm_ctx *ctx, int64_t n, const int64_t *d_rowptr, const uint32_t *d_colidx,
                     const uint32_t *d_counts, const uint32_t *d_colmap, int64_t ncols_out,
                     int mode, int dtype, void *d_out, int64_t ld);

/* Column re-indexing of a dense row-major matrix, the data movement of KmerBasis.transform
 * (snekmer/vectorize.py:107-119): out[i, p] = in[i, d_src[p]], or 0 where d_src[p] == 0xFFFFFFFF.
 * elem_bytes is 1, 2, 4 or 8; d_in is [rows x ld_in], d_out [rows x ncols_out] (tight). */
int skm_gather_columns(skm_ctx *ctx, int64_t rows, int64_t ncols_out, int elem_bytes, const void *d_in, int64_t ld_in,
                       const uint32_t *d_src, void *d_out);

/* Counts travel as bytes in the multi-GPU exchange when every count fits (skm_csr_max_count <= 255):
 * d_out[i] = (uint8) d_in[i] and back. */
int skm_narrow_u32_u8(skm_ctx *ctx, int64_t count, const uint32_t *d_in, uint8_t *d_out);
int skm_widen_u8_u32(skm_ctx *ctx, int64_t count, const uint8_t *d_in, uint32_t *d_out);

/* d_out[i] = (uint32) d_in[i] for non-negative int8 counts (lets the CSR norm kernel serve dense
 * int8 operands). */
int skm_widen_i8_u32(skm_ctx *ctx, int64_t count, const int8_t *d_in, uint32_t *d_out);

/* Largest count in a CSR (host-synchronous): the int8 dense path needs it to be <= 127. */
int skm_csr_max_count(skm_ctx *ctx, int64_t nnz, const uint32_t *d_counts, uint32_t *h_max);
/* The same without a host wait: the largest count among the entries [0, d_rowptr[n]) is left in *d_out_max (device);
 * cap_entries bounds the launch.  For callers that must not stall between stages (engine.DensePipeline). */
int skm_csr_max_count_dev(skm_ctx *ctx, int64_t n, int64_t cap_entries, const int64_t *d_rowptr, const uint32_t *d_counts,
                          uint32_t *d_out_max);

/* ---- a13/a14: cosine ----------------------------------------------------------------------- */
/* 1/||row|| (float32; 1.0 for an all-zero row, as sklearn's normalize does) and optionally the
 * exact squared norms. */
int skm_row_norms_csr(skm_ctx *ctx, int64_t n, const int64_t *d_rowptr, const uint32_t *d_counts,
                      float *d_rnorm, uint64_t *d_normsq);

/* Replaces sklearn.metrics.pairwise.cosine_similarity(X, Y) at rules/apply.smk:282-284,
 * rules/learn.smk:821-823, rules/evaluate.smk:434-436 and the metric="cosine" branch of
 * snekmer/score.py:149-172, for count rows held sparse over a shared column space.
 * X: CSR rows [row0,row1) of an n-row matrix; Y: postings (column-major) of an m-row matrix
 * (pass X's own postings for the square N x N case).  Writes
 *   out[(i-row0)*ld + j] = <x_i, y_j> * x_rnorm[i] * y_rnorm[j]   for i in [row0,row1), j in [0,m)
 * with the integer dot product exact and the scaling in float32.  Dot products are summed in int32 cells wherever they
 * provably fit: by Cauchy-Schwarz <x_i, y_j> <= 1 / (x_rnorm[i] * y_rnorm[j]), so row i takes the 32-bit kernels when
 * x_rnorm[i] * min_j y_rnorm[j] > 2^-31 (decided on the device, no host wait) and otherwise an exact wide form of the
 * cursor kernel with float64 accumulators - sklearn's own arithmetic: exact below 2^53, float64 rounding beyond, no upper
 * limit (counts use all 32 bits).  The test presumes the norms ARE the rows' norms (skm_row_norms_csr); callers that
 * pass other scalings (unit norms to obtain set sizes) must keep their dot products below 2^31 themselves.
 * mode 0 = similarity; mode 1 = cosine distance as sklearn's pairwise_distances(X) / cosine_distances(X, X)
 * gives it (1 - s clipped to [0,2]; exact 0 where i == j: Y IS X); mode 2 = cosine distance between two
 * different matrices, sklearn's cosine_distances(X, Y): the same values without the diagonal rule.
 * post_bits: 64 (d_ypost = uint64 words row | count << 32; d_ypostcnt ignored) or 32 (the
 * SKM_BASIS_POST32 form with its d_ypostcnt side array).
 * An X entry with d_xcolidx == 0xFFFFFFFF means "k-mer of this row only" (SKM_BASIS_ELIDE_SINGLETONS):
 * its count squared is added to out[i][i], which is only meaningful in the square case (Y is X). */
int skm_cosine_csr(skm_ctx *ctx, int64_t n, const int64_t *d_xrowptr, const uint32_t *d_xcolidx,
                   const uint32_t *d_xcounts, const float *d_xrnorm, int64_t m, int64_t ncols,
                   const uint32_t *d_ycolptr, const void *d_ypost, int post_bits, const uint32_t *d_ypostcnt,
                   const float *d_yrnorm, int64_t row0, int64_t row1, int mode, float *d_out,
                   int64_t ld);
/* skm_cosine_csr in two calls, for a stream of batches (engine.OverlappedPipeline; no reference counterpart: the
 * reference handles one FASTA file per process, snekmer/rules/kmerize.smk:57-65).  phase 1: prologue + sparse Gram of the
 * row block on ctx; the neighbour lists stay in ctx's scratch; d_out may be NULL.  phase 2: everything behind them (heavy
 * rows, streaming writer, cursor kernel, the mode-2 epilogue) on exec_ctx's STREAM (same device; NULL = ctx's own) with
 * ctx's scratch.  The caller orders phase 2 behind phase 1 (skm_event_record / skm_stream_wait) and starts no new phase 1
 * on ctx before the phase 2 reading its lists has finished.  Same arguments in both calls.  phase 0 = skm_cosine_csr. */
int skm_cosine_csr_phase(skm_ctx *ctx, skm_ctx *exec_ctx, int phase, int64_t n, const int64_t *d_xrowptr, const uint32_t *d_xcolidx,
                         const uint32_t *d_xcounts, const float *d_xrnorm, int64_t m, int64_t ncols, const uint32_t *d_ycolptr,
                         const void *d_ypost, int post_bits, const uint32_t *d_ypostcnt, const float *d_yrnorm, int64_t row0,
                         int64_t row1, int mode, float *d_out, int64_t ld);

/* Reporting only: how the last neighbour-list skm_cosine_csr call of this context distributed its rows.
 * h_out4[0] = rows whose neighbours overflowed the first pass's table (handed to the fused heavy-row kernel), [1] =
 * 8-row strips left to the 32-bit cursor kernel, [2] = neighbour-list entries allocated behind the rows' fixed slots,
 * [3] = 8-row strips that held a wide row (float64 cursor kernel).  Synchronises the stream. */
int skm_cosine_csr_stats(skm_ctx *ctx, int64_t *h_out4);

/* Reporting only: the heavy-row panel pipeline of the last list-path skm_cosine_csr call that ran it (rows the first
 * pass handed on are clustered, blocks of 256 of them get int8 panels over their shared long-list columns and an MFMA
 * GEMM; skm_cosine_csr.hip).  h_out6 = heavy rows, blocks, blocks without a panel, sum of panel columns, sum of panel
 * rows, multiply-accumulates of the GEMMs.  Synchronises the stream. */
int skm_heavy_panel_stats(skm_ctx *ctx, int64_t *h_out6);

/* The reference's metric="jaccard" branch (snekmer/score.py:166-168) is 1 - hamming distance on the
 * binary presence matrix: 1 - (|a| + |b| - 2|a&b|) / ncols.  Given d_out holding the exact
 * intersection sizes |a&b| (skm_cosine_csr on a 0/1 CSR with all norms = 1), rewrite it in place.
 * d_xcount[i], d_ycount[j] are the rows' set sizes as float. */
int skm_hamming_similarity_from_gram(skm_ctx *ctx, int64_t n, int64_t m, int64_t ncols, const float *d_xcount,
                                     const float *d_ycount, float *d_out, int64_t ld);

/* Jaccard distance on binary rows as scipy's pdist(X, "jaccard") + squareform gives it
 * (snekmer/scripts/cluster_cluster.py:189-190, the non-BSF branch): (|a|+|b|-2|a&b|) / (|a|+|b|-|a&b|),
 * 0 when both rows are empty.  Same in-place convention as skm_hamming_similarity_from_gram. */
int skm_jaccard_distance_from_gram(skm_ctx *ctx, int64_t n, int64_t m, const float *d_xcount, const float *d_ycount,
                                   float *d_out, int64_t ld);

/* The same two measures for ANY matrix, in float64 (what the reference returns).  snekmer/score.py:149-172 is
 * called by default with metric="jaccard" on a k-mer COUNT matrix (its docstring), i.e.
 * 1 - scipy hamming(x, y) = 1 - #{c: x_ic != y_jc} / ncols; scripts/cluster_cluster.py:189-190 calls scipy's
 * Jaccard distance, |a xor b| / |a or b| on the non-zero patterns (pass the 0/1 pattern, d_equal NULL).  Inputs are two
 * exact sparse Grams (skm_cosine_csr with unit norms, float32 cells holding integers < 2^24):
 *   d_both[i][j]  = #{c: x_ic != 0 and y_jc != 0}   (Gram of the 0/1 pattern)
 *   d_equal[i][j] = #{c: x_ic == y_jc != 0}          (Gram of the 0/1 matrix whose columns are the distinct
 *                                                    (column, value) pairs); NULL for 0/1 input (equal == both)
 * and the non-zeros per row d_xnnz[n], d_ynnz[m]: the rows differ on xnnz + ynnz - both - equal columns.
 * kind 0: out = 1 - differ / ncols; kind 1: out = differ / (xnnz + ynnz - both), 0 when both rows are empty;
 * kind 2: out = differ / ncols (the hamming distance itself). */
int skm_setsim_f64(skm_ctx *ctx, int kind, int64_t n, int64_t m, int64_t ncols, const uint32_t *d_xnnz,
                   const uint32_t *d_ynnz, const float *d_both, const float *d_equal, int64_t ld, double *d_out,
                   int64_t ld_out);

/* The `else` branch of snekmer/score.py:169-171, pairwise_distances(X, metric=m), for dense float64 matrices
 * X [n x k] (row stride ldx) and Y [m x k] (pass X itself for the square case: the diagonal is then an exact zero,
 * as squareform(pdist(X)) / euclidean_distances(X) give).  metric: 0 cityblock (manhattan, l1), 1 sqeuclidean,
 * 2 euclidean (l2), 3 chebyshev, 4 canberra, 5 braycurtis, 6 minkowski (exponent p); scipy's boolean
 * dissimilarities on x != 0, y != 0: 10 dice, 11 rogerstanimoto, 12 russellrao, 13 sokalmichener, 14 sokalsneath,
 * 15 yule (0/0 gives nan, as scipy's C implementation does).  cosine, hamming and jaccard have their own entry
 * points above.  No Snekmer rule passes these metrics; tutorial-scale matrices. */
int skm_pairwise_f64(skm_ctx *ctx, int metric, double p, int64_t n, int64_t m, int64_t k, const double *d_x, int64_t ldx,
                     const double *d_y, int64_t ldy, double *d_out, int64_t ld);

/* Apply epilogue (snekmer/rules/apply.smk:312-328, rules/learn.smk:831-849): for every row of a
 * score matrix the two largest entries and their columns, i.e. np.argsort(-S, axis=1)[:, :2] with
 * ties broken towards the lower column.  d_idx[2*i+{0,1}], d_val[2*i+{0,1}]; with m == 1 the
 * second slot holds index 0xFFFFFFFF and value 0. */
int skm_row_top2(skm_ctx *ctx, int64_t n, int64_t m, const float *d_scores, int64_t ld, uint32_t *d_idx,
                 float *d_val);

/* Learn aggregation (snekmer/rules/learn.smk:385-408): sum the count rows of each group
 * (annotation).  Input CSR rows carry a group id < ngroups (d_group[n]); output is the CSR of the
 * [ngroups x ncols] totals matrix with columns ascending per row: d_out_rowptr[ngroups+1],
 * d_out_col / d_out_val with capacity nnz.  *h_out_nnz is host-synchronous.  Since round 6 this is
 * skm_csr_transpose + skm_group_postings + skm_postings_to_csr (below); a caller that already holds the
 * count matrix by column (the postings the vectorize stage builds) calls the last two itself. */
int skm_csr_group_sum(skm_ctx *ctx, int64_t n, int64_t nnz, const int64_t *d_rowptr, const uint32_t *d_colidx,
                      const uint32_t *d_counts, const uint32_t *d_group, int64_t ngroups, int64_t *d_out_rowptr,
                      uint32_t *d_out_col, uint32_t *d_out_val, int64_t *h_out_nnz);
/* The same sums taken where the count matrix already lies by column.  Input: postings of the n x ncols count matrix
 * (d_colptr[ncols+1], d_post[nnz] = row | count << 32, as skm_basis_build / skm_vectorize_csr / skm_csr_transpose
 * write them) and a group id per row.  Output, by column as well - the layout skm_apply_top2 reads -:
 * d_out_colptr[ncols+1], d_out_post[<= nnz] = group | total << 32 with groups ascending inside a column; optionally
 * d_out_normsq[ngroups] (exact squared norm of every group's total row, uint64) and d_out_rowcount[ngroups] (entries
 * per group).  *h_out_nnz (entries written) is host-synchronous.  Totals are uint32 like the counts.
 * Replaces the pandas groupby-sum of snekmer/rules/learn.smk:385-408 on a dense N x B table. */
int skm_group_postings(skm_ctx *ctx, int64_t n, int64_t ncols, const uint32_t *d_colptr, const uint64_t *d_post, int64_t nnz,
                       const uint32_t *d_group, int64_t ngroups, uint32_t *d_out_colptr, uint64_t *d_out_post,
                       uint64_t *d_out_normsq, uint32_t *d_out_rowcount, int64_t *h_out_nnz);
/* A matrix held by column (d_colptr[ncols+1], d_post[nnz] = row | value << 32) as CSR with ascending columns:
 * d_out_rowptr[nrows+1], d_out_col[nnz], d_out_val[nnz] (one stable counting sort by row).  The family-total table of
 * snekmer/rules/learn.smk:385-408 in row order, from skm_group_postings' output. */
int skm_postings_to_csr(skm_ctx *ctx, int64_t ncols, int64_t nnz, const uint32_t *d_colptr, const uint64_t *d_post,
                        int64_t nrows, int64_t *d_out_rowptr, uint32_t *d_out_col, uint32_t *d_out_val);

/* Exact sparse Gram rows ("neighbour lists"): for rows [row0,row1) of X, every row j of Y sharing at
 * least one column, with the exact int32 dot product.  This is the reduced output for problem sizes
 * whose dense N x M matrix cannot be stored (BASELINE configs[3], 1M x 1M).
 *   d_start[row1-row0], d_len[row1-row0]: position and length of each row's list in d_ent;
 *   d_ent[cap_ent]: entries (j << 32 | dot), order within a row unspecified;
 *   d_len == 0xFFFFFFFF marks a row the kernels could not hold (> 65536 neighbours, cap_ent
 *   exhausted, or a dot product that may not fit the 32-bit entry: d_xrnorm[i] * min_j d_yrnorm[j] <= 2^-31, the
 *   test skm_cosine_csr applies); *h_overflow_rows counts them, *h_total_entries is the
 *   number of entries written (both host-synchronous).  d_xrnorm[n] / d_yrnorm[m]: skm_row_norms_csr of X and Y. */
int skm_gram_neighbors(skm_ctx *ctx, int64_t n, const int64_t *d_xrowptr, const uint32_t *d_xcolidx,
                       const uint32_t *d_xcounts, int64_t m, int64_t ncols, const uint32_t *d_ycolptr,
                       const void *d_ypost, int post_bits, const uint32_t *d_ypostcnt, const float *d_xrnorm,
                       const float *d_yrnorm, int64_t row0, int64_t row1, int64_t cap_ent, uint64_t *d_start,
                       uint32_t *d_len, uint64_t *d_ent, int64_t *h_total_entries, int64_t *h_overflow_rows);

/* k best cosine neighbours per row from the lists of skm_gram_neighbors: score = dot * xrnorm[row0+r]
 * * yrnorm[j], descending, ties towards the lower j; exclude_self drops j == row0 + r.
 * d_idx / d_val are [nrows x k]; missing slots hold 0xFFFFFFFF / 0.  m = number of rows of Y (entries of d_yrnorm;
 * non-negative values): lets the k <= 16 kernel bound an entry's score by dot * xrnorm * max_j yrnorm and skip the gather of
 * its neighbour's norm when the bound is below what 64 entries already reach; m <= 0: every norm is gathered. */
int skm_neighbors_topk(skm_ctx *ctx, int64_t nrows, int64_t row0, const uint64_t *d_start, const uint32_t *d_len,
                       const uint64_t *d_ent, const float *d_xrnorm, const float *d_yrnorm, int64_t m, int k, int exclude_self,
                       uint32_t *d_idx, float *d_val);

/* Exact sum over columns of df*(df) pairs the sparse kernel will visit (cost model input). */
int skm_pair_work(skm_ctx *ctx, int64_t ncols, const uint32_t *d_colptr, uint64_t *h_pairs);

/* ---- dense small-basis path (north-star "N x |S|^k count matrix" + MFMA cosine) ------------ */
/* Atomic scatter of window counts into a dense [n x ld] matrix of uint16 (dtype 0) or
 * uint32 (dtype 1); ld >= nsym^k, nsym^k <= 2^26.  The matrix is zero-filled first. */
int skm_count_dense(skm_ctx *ctx, const uint8_t *h_rank, int nsym, int k, const uint8_t *d_seq,
                    const int64_t *d_off, int64_t n, int dtype, void *d_out, int64_t ld);

/* out[i*ld + j] = (sum_c X[i,c]*Y[j,c]) * xr[i] * yr[j]; X [n x kdim], Y [m x kdim] int8 row-major
 * with kdim a multiple of 64 and rows padded with zeros; i8 MFMA with int32 accumulation.
 * Y may equal X.  mode as in skm_cosine_csr.  int32 is safe by construction up to kdim = 133 143 (127^2 * kdim < 2^31);
 * for wider operands the call first checks on the device that the norms bound every dot product below 2^31
 * (1 / (xrnorm * yrnorm), one host wait) and returns SKM_E_OVERFLOW otherwise. */
int skm_cosine_dense_i8(skm_ctx *ctx, int64_t n, int64_t m, int64_t kdim, const int8_t *d_x,
                        const int8_t *d_y, const float *d_xrnorm, const float *d_yrnorm, int mode,
                        float *d_out, int64_t ld);

/* The dense route for small full bases (engine.Pipeline picks it when |S|^k <= 2^17 and the rows are dense enough;
 * the reference's CI runs solvacc k=8 = 6561 columns): the int8 operand of skm_cosine_dense_i8 straight from a CSR
 * whose column ids are the k-mer codes (32-bit codes; d_out [n x kdim] is zero-filled first).  Counts above 127
 * saturate and their rows are appended to d_irr_list[n] (*d_irr_count of them, any order): skm_cosine_fixup_rows then
 * recomputes the cells of those rows AND columns of a block [row0,row1) x [0,n) of the result from the CSR itself with
 * float64 accumulators (exact below 2^53, no upper limit), overwriting what the GEMM left there.  Neither call waits
 * for the device; an empty list costs one launch that exits at once. */
int skm_csr_to_dense_i8(skm_ctx *ctx, int64_t n, const int64_t *d_rowptr, const uint32_t *d_codes, const uint32_t *d_counts,
                        int64_t kdim, int8_t *d_out, uint32_t *d_irr_list, uint32_t *d_irr_count);
int skm_cosine_fixup_rows(skm_ctx *ctx, int64_t n, const int64_t *d_rowptr, const uint32_t *d_codes, const uint32_t *d_counts,
                          const float *d_rnorm, int64_t row0, int64_t row1, const uint32_t *d_irr_list,
                          const uint32_t *d_irr_count, int mode, float *d_out, int64_t ld);

/* Sparse view of a dense count matrix, the inverse of skm_count_dense / skm_csr_to_dense: per row the
 * non-zero columns in ascending order with their values, i.e. the (code, count) rows skm_count_csr
 * gives when column id == k-mer code (rules/learn.smk:359-383 keeps exactly these pairs in its
 * per-sequence dict).  dtype: 0 = uint16, 1 = uint32, 2 = int8 cells; d_in is [n x ld] row-major.
 * d_rowptr[n+1]; d_col / d_val need cap_entries slots (SKM_E_OVERFLOW if the matrix holds more
 * non-zeros); *h_nnz is host-synchronous. */
int skm_dense_to_csr(skm_ctx *ctx, int64_t n, int64_t ncols, int dtype, const void *d_in, int64_t ld,
                     int64_t cap_entries, int64_t *d_rowptr, uint32_t *d_col, uint32_t *d_val, int64_t *h_nnz);

/* 1/||row|| (float32; 1.0 for an all-zero row) and optionally the exact squared norms of an int8
 * matrix [n x kdim] (kdim a multiple of 64, rows zero padded): the norms skm_cosine_dense_i8 takes. */
int skm_row_norms_i8(skm_ctx *ctx, int64_t n, int64_t kdim, const int8_t *d_in, float *d_rnorm, uint64_t *d_normsq);

/* float64 cosine of real-valued dense feature matrices on the f64 matrix cores, with sklearn's exact
 * order of operations (rows L2-normalised in float64, zero norms replaced by 1, then the dot products):
 * sklearn.metrics.pairwise.cosine_similarity / pairwise_distances(metric="cosine") for inputs that are
 * not counts, e.g. the length-normalised rows of snekmer/utils.py:183-203 fed to
 * snekmer/score.py:149-172.  X [n x k] (row stride ldx), Y [m x k] (row stride ldy; pass X for the
 * square case); out [n x ld] float64.  mode as in skm_cosine_csr (the exact-zero diagonal of mode 1
 * applies only when Y is X). */
int skm_cosine_dense_f64(skm_ctx *ctx, int64_t n, int64_t m, int64_t k, const double *d_x, int64_t ldx,
                         const double *d_y, int64_t ldy, int mode, double *d_out, int64_t ld);

/* Checksums of a float32 result block too large to read back (the 40 GB matrix of BASELINE
 * configs[2]): d_sum[i] = float64 sum of row i, d_nnz[i] = number of its non-zero entries. */
int skm_matrix_row_stats(skm_ctx *ctx, int64_t n, int64_t m, const float *d_in, int64_t ld, double *d_sum,
                         uint32_t *d_nnz);

/* ---- apply epilogue fused with the cosine ---------------------------------------------------- */
/* Replaces rules/apply.smk:278-328 and rules/learn.smk:811-849: cosine_similarity(totals, counts).T
 * followed by np.argsort(-S, axis=1)[:, :2], without ever storing the N x A score block.
 * X: CSR rows [row0,row1) of the n query sequences; Y: postings of the m family-total rows over the
 * same columns (an X entry whose colidx is 0xFFFFFFFF is a column Y does not have and contributes
 * nothing).  d_xnormsq[n] / d_ynormsq[m]: exact squared norms (skm_row_norms_csr).
 * Per query row r = i - row0:
 *   d_idx[2r+{0,1}]    the two best families (score descending, ties towards the lower index;
 *                      0xFFFFFFFF in the second slot when m == 1)
 *   d_score[2r+{0,1}]  their cosine in float64: (double)dot / (sqrt(|x|^2) * sqrt(|y|^2)), a zero norm
 *                      counting as 1 (sklearn's rule), formed from the EXACT integer dot and norms
 *   d_dot[2r+{0,1}]    those exact dot products (int64)
 * so Score = d_score[2r], delta = round(d_score[2r] - d_score[2r+1], 2) and the confidence lookup of
 * rules/apply.smk:325-326 are reproducible on the host from integers alone.
 * d_row_order (may be NULL): a permutation of [0, row1 - row0) giving the ORDER in which the rows are processed (results
 * stay at their rows' positions).  The kernel is bound by one random 128-byte line per query entry; an order that puts
 * similar rows next to each other (the training labels in learn.smk's self-evaluation, the records of one family's FASTA
 * file) lets the rows in flight find the columns' words in L2. */
int skm_apply_top2(skm_ctx *ctx, int64_t n, const int64_t *d_xrowptr, const uint32_t *d_xcolidx,
                   const uint32_t *d_xcounts, const uint64_t *d_xnormsq, int64_t m, int64_t ncols,
                   const uint32_t *d_ycolptr, const uint64_t *d_ypost, const uint64_t *d_ynormsq, int64_t row0,
                   int64_t row1, const uint32_t *d_row_order, uint32_t *d_idx, double *d_score, int64_t *d_dot);

/* ---- multi-GPU (RCCL over xGMI; one context per rank) --------------------------------------- */
#define SKM_COMM_ID_BYTES 128
int skm_comm_unique_id(uint8_t *h_id /* SKM_COMM_ID_BYTES */);
int skm_comm_init(skm_ctx *ctx, int nranks, int rank, const uint8_t *h_id);
int skm_comm_destroy(skm_ctx *ctx);
/* Variable-size all-gather: rank r contributes h_bytes[r] bytes from d_send; every rank receives
 * all contributions back to back (rank order) in d_recv.  h_bytes has nranks entries and must be
 * identical on all ranks. */
int skm_allgatherv(skm_ctx *ctx, const void *d_send, const int64_t *h_bytes, void *d_recv);
/* Variable-size all-to-all: the bytes for rank p are the p-th segment of d_send (segments back to
 * back in rank order, h_send_bytes[p] long); the bytes from rank p arrive as the p-th segment of
 * d_recv (h_recv_bytes[p] long).  h_recv_bytes[p] on this rank must equal h_send_bytes[this rank]
 * on rank p.  Grouped ncclSend/ncclRecv; the segment a rank sends to itself is a device copy. */
int skm_alltoallv(skm_ctx *ctx, const void *d_send, const int64_t *h_send_bytes, void *d_recv,
                  const int64_t *h_recv_bytes);

/* Several parallel arrays in ONE exchange (one RCCL group = one launch; every pair of ranks uses its own
 * xGMI link; every piece lands at its final offset).  The byte ranges are computed by two pure host
 * functions that tests can execute without RCCL: out_ops[p * narrays + a] describes what this rank
 * exchanges with peer p for array a (offsets into the rank's own send / receive buffer of that array). */
#define SKM_MAX_ARRAYS 8
typedef struct skm_p2p_op {
    int32_t peer, array;
    int64_t send_off, send_bytes; /* range of d_send[array] that goes to `peer` */
    int64_t recv_off, recv_bytes; /* range of d_recv[array] that `peer`'s data lands in */
} skm_p2p_op;
/* All-to-all: h_send_counts[p] / h_recv_counts[p] elements (the same counts for every array, element size
 * h_elem_bytes[a]) go to / come from rank p; segments are back to back in rank order in every buffer. */
int skm_plan_alltoallv(int nranks, int narrays, const int64_t *h_elem_bytes, const int64_t *h_send_counts,
                       const int64_t *h_recv_counts, skm_p2p_op *out_ops);
int skm_alltoallv_multi(skm_ctx *ctx, int narrays, const void *const *d_send, void *const *d_recv,
                        const int64_t *h_elem_bytes, const int64_t *h_send_counts, const int64_t *h_recv_counts);
/* All-gather: rank r contributes h_counts[a * nranks + r] elements of array a (its whole d_send[a]); every
 * rank receives all contributions back to back in rank order in d_recv[a]. */
int skm_plan_allgatherv(int nranks, int rank, int narrays, const int64_t *h_elem_bytes, const int64_t *h_counts,
                        skm_p2p_op *out_ops);
int skm_allgatherv_multi(skm_ctx *ctx, int narrays, const void *const *d_send, void *const *d_recv,
                         const int64_t *h_elem_bytes, const int64_t *h_counts);

/* ---- multi-GPU basis: postings of a row-sharded count matrix, built in parallel ------------- *
 * Sharded form of skm_basis_build for the cosine pipeline (singletons elided); there is no
 * reference counterpart (one process per FASTA file, snekmer/rules/kmerize.smk:57-65).  A code's
 * OWNER rank is a fixed hash of the code, so ranks agree on it without a dictionary
 * (snekmer_amd/csrc/skm_shard.hip has the data flow; snekmer_amd/dist.py drives it). */
#define SKM_MAX_RANKS 64
/* Entries of a local CSR shard (n rows, first global row = row_base) grouped by owner, original
 * order kept inside a group: d_out_codes / d_out_rowcount (global row | count << 32), capacity cap_entries each.  The
 * entry count is read on the device (d_rowptr[n] <= cap_entries < 2^30), so the call does not wait for it:
 * d_out_counts[nbuckets] (device int64) = entries per owner, which the exchange can gather device to device
 * (one host round trip for the whole [src, dst] matrix).  h_counts (optional): the same on the host; passing it makes
 * the call host-synchronous.
 * d_out_index (optional, capacity cap_entries): grouped position -> entry of the CSR it came from, for
 * skm_colidx_from_owners. */
int skm_bucket_partition(skm_ctx *ctx, int code_bits, int nbuckets, int64_t n, int64_t cap_entries, const int64_t *d_rowptr,
                         const void *d_codes, const uint32_t *d_counts, int64_t row_base, void *d_out_codes,
                         uint64_t *d_out_rowcount, int64_t *d_out_counts, int64_t *h_counts, uint32_t *d_out_index);
/* Owner side: from the nrecv (code, row | count << 32) entries an owner received (ascending rows
 * within equal codes once stably sorted, which holds when sources are concatenated in rank order)
 * build d_post[npost] (postings of k-mers found in >= 2 rows, column after column, rows ascending),
 * d_cols_start[ncols] (index of each such column's first posting in d_post) and an open-addressing
 * hash table code -> column: d_tab_vals[tsize] (0xFFFFFFFF = empty) / d_tab_keys[tsize], tsize a
 * power of two >= 2 * ncols.  d_out4 (device int64[4]) = {distinct k-mers incl. single-row ones, ncols, npost, tsize};
 * the call does not wait for them (gather them device to device with the other owners' sizes).  h_out4 (optional):
 * the same on the host, which makes the call host-synchronous.  d_cols_start / d_post need room for nrecv elements, the
 * table for skm_bucket_table_capacity(nrecv) slots (all of its value words are cleared).
 * d_ret (optional, nrecv words): the owner's ANSWER to every entry it received, in the order received: the owner-local
 * column of the entry's k-mer, 0xFFFFFFFF for a k-mer of one row.  Sent back through the reverse of the all-to-all it
 * replaces the table (d_tab_keys = d_tab_vals = NULL is then allowed): skm_colidx_from_owners. */
int64_t skm_bucket_table_capacity(int64_t nrecv);
int skm_bucket_postings(skm_ctx *ctx, int code_bits, int key_bits, int64_t nrecv, const void *d_codes,
                        const uint64_t *d_rowcount, int64_t *d_out4, int64_t *h_out4, uint32_t *d_cols_start, uint64_t *d_post,
                        void *d_tab_keys, uint32_t *d_tab_vals, uint32_t *d_ret);
/* Global column starts from the gathered per-owner arrays: d_starts holds the owners' d_cols_start
 * arrays back to back (h_ncols[p] entries each); d_colptr[sum(h_ncols) + 1] gets them rebased by
 * the owners' posting offsets (prefix sums of h_npost), closed by the total. */
int skm_concat_colptr(skm_ctx *ctx, int nparts, const int64_t *h_ncols, const int64_t *h_npost, const uint32_t *d_starts,
                      uint32_t *d_colptr);
/* Column id of every entry of a CSR shard, from the gathered owner tables (back to back in owner
 * order, h_tab_sizes[b] slots each; owner b's columns start at global id sum(h_ncols[:b])):
 * 0xFFFFFFFF if the code is in no table (k-mer of one row). */
int skm_colidx_lookup(skm_ctx *ctx, int code_bits, int nbuckets, int64_t nnz, const void *d_codes,
                      const int64_t *h_tab_sizes, const int64_t *h_ncols, const void *d_tab_keys,
                      const uint32_t *d_tab_vals, uint32_t *d_colidx);
/* Column id of every entry of a CSR shard from the owners' answers: d_back[nnz] = what came back through the reverse
 * all-to-all, i.e. in the grouped order skm_bucket_partition produced (h_group_counts[b] entries for owner b, back to
 * back); d_index = that call's d_out_index; owner b's columns start at global id sum(h_ncols[:b]).
 * d_colidx[d_index[g]] = d_back[g] + that base, 0xFFFFFFFF kept. */
int skm_colidx_from_owners(skm_ctx *ctx, int nbuckets, int64_t nnz, const uint32_t *d_back, const uint32_t *d_index,
                           const int64_t *h_group_counts, const int64_t *h_ncols, uint32_t *d_colidx);
/* d_rowptr[n_total + 1] of an n_total-row matrix that is empty except rows [lo, lo + nloc), which
 * are the local shard (d_local[nloc + 1]): lets skm_cosine_csr / skm_gram_neighbors run on a shard
 * with global row numbers. */
int skm_embed_rowptr(skm_ctx *ctx, int64_t n_total, int64_t lo, int64_t nloc, const int64_t *d_local, int64_t *d_rowptr);

/* ---- data formats either side of the path ---------------------------- *
 * HOST functions, no GPU and no context: a threaded FASTA reader replacing the
 * `for f in SeqIO.parse(fasta, "fasta")` loops of snekmer/rules/kmerize.smk:90-129.  Biopython's
 * SimpleFastaParser semantics on a text-mode handle: lines end at "\n", "\r\n" or a lone "\r"; text before the
 * first line starting with '>' is skipped; id = first whitespace-delimited word of the header ("" if none); the
 * sequence = the record's lines, each without trailing whitespace, joined, with every ' ' and '\r' removed.
 * skm_fasta_index sizes the outputs; *out_flags bit 0 = the buffer holds bytes >= 0x80 (a text handle would
 * decode them as UTF-8: the caller must take its text-mode path).  nthreads < 1 = all hardware threads.
 * skm_fasta_parse fills h_out_residues[nresidues] (records back to back), h_out_offsets[nrecords + 1] and the byte
 * span of every id inside h_buf. */
int skm_fasta_index(const uint8_t *h_buf, int64_t len, int nthreads, int64_t *out_nrecords, int64_t *out_nresidues,
                    int *out_flags);
int skm_fasta_parse(const uint8_t *h_buf, int64_t len, int nthreads, int64_t nrecords, int64_t nresidues,
                    uint8_t *h_out_residues, int64_t *h_out_offsets, int64_t *h_out_id_begin, int32_t *h_out_id_len);

/* HOST function, no GPU and no context: the `.npz` writer of the rule (np.savez_compressed at
 * snekmer/rules/kmerize.smk:132-139; read back by snekmer/io.py:46-96 through np.load).  Writes a zip archive with one
 * member "<names[m]>.npy" per array = h_headers[m] (the .npy header bytes, numpy.lib.format) followed by h_data[m]
 * (the C-contiguous array bytes).  level: -1 = zlib's default (6, what numpy uses), 1-9, or 0 = stored (np.savez).
 * Every member is deflated in 512 KiB chunks by nthreads threads (< 1: all hardware threads), each chunk a run of raw
 * deflate blocks closed by a sync flush, the chunks' CRC-32s combined: one ordinary deflate stream per member that any
 * unzip reads.  Members whose .npy header says '<U' (little-endian UTF-32: kmerlist, ids, seqs - three quarters of the
 * bytes) are tokenised by the library itself (a 4-byte unit repeats the previous item, repeats the previous unit, or
 * is one literal + a 3-byte match; dynamic Huffman per chunk; `level` only says compressed or stored there): 5-7x
 * zlib's speed on such bytes, 2-13 % larger; '<u4' / '<i4' / '<u8' / '<i8' members take the same tokeniser on 4-byte
 * units (column ids: zlib's size at 5x its speed); other chunks go through zlib at `level`, or through the library's
 * Huffman coder when a probe says string matching buys nothing.  *out_file_bytes (optional) = size of the file written. */
int skm_npz_write(const char *path, int nmembers, const char *const *names, const void *const *h_headers,
                  const int64_t *header_bytes, const void *const *h_data, const int64_t *data_bytes, int level,
                  int nthreads, int64_t *out_file_bytes);

/* Ragged byte rows -> fixed-width UCS-4 rows, zero padded: d_out[i * width + j] = d_bytes[d_off[i] + j] for
 * j < d_len[i].  Viewed as numpy '<U{width}' this is the `seqs` array of reduced strings the rule stores
 * (snekmer/rules/kmerize.smk:121-127,136) without a per-record Python string (bytes are latin-1 code points). */
int skm_rows_to_utf32(skm_ctx *ctx, const uint8_t *d_bytes, const int64_t *d_off, const int32_t *d_len, int64_t n,
                      int64_t width, uint32_t *d_out);

/* Integer k-mer codes -> k class letters each, UCS-4 (numpy '<U{k}'): the `kmerlist` array of the rule
 * (snekmer/rules/kmerize.smk:102-106,134).  h_letters[nsym]: class letters in rank order; d_index (optional)
 * selects and orders the codes: d_out[i] = letters of d_codes[d_index[i]] (first-seen order, min_filter). */
int skm_decode_kmers_utf32(skm_ctx *ctx, int code_bits, int nsym, int k, const uint8_t *h_letters, const void *d_codes,
                           const uint32_t *d_index, int64_t n, uint32_t *d_out);

/* The count matrix in `kmerlist` column order (what snekmer/rules/learn.smk:359-383 rebuilds per sequence in
 * Python): every CSR entry's column id c becomes d_colmap[c]; entries whose column maps to 0xFFFFFFFF (filtered
 * by min_filter, or absent from an explicit basis) or whose id is >= ncols are dropped; order within a row is
 * kept.  d_out_rowptr[n + 1]; d_out_col / d_out_val sized for the input's entry count.  Host-synchronous. */
int skm_csr_remap_columns(skm_ctx *ctx, int64_t n, const int64_t *d_rowptr, const uint32_t *d_colidx,
                          const uint32_t *d_counts, const uint32_t *d_colmap, int64_t ncols, int64_t *d_out_rowptr,
                          uint32_t *d_out_col, uint32_t *d_out_val, int64_t *h_out_nnz);

/* The filter of snekmer/rules/kmerize.smk:102-104 on the device: of the columns in first-seen order
 * (d_fs_order[ncols], skm_basis_build) keep those whose total occurrence count d_total[column] > min_filter, order
 * kept: d_keep[0 .. *h_nkeep) = the kept columns (the order of `kmerlist`), d_colmap[column] = its position in
 * `kmerlist` or 0xFFFFFFFF.  Host-synchronous (one 8-byte read-back). */
int skm_basis_select(skm_ctx *ctx, int64_t ncols, const uint32_t *d_fs_order, const uint64_t *d_total, uint64_t min_filter,
                     uint32_t *d_keep, uint32_t *d_colmap, int64_t *h_nkeep);

#ifdef __cplusplus
}
#endif
#endif /* SNEKMER_HIP_H */
