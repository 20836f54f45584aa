// Internal declarations shared by the HIP translation units of libsnekmer_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "skm_host.h"

#define SKM_HIP(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            skm_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return SKM_E_HIP;                                                                \
        }                                                                                    \
    } while (0)

// Grow-only scratch slots.  Every entry point draws its temporaries from fixed slots so that,
// after a warm-up call, the timed path performs no hipMalloc/hipFree.
enum skm_ws_slot {
    WS_A = 0, WS_B, WS_C, WS_D, WS_E, WS_F, WS_G, WS_H, WS_I, WS_J, WS_K, WS_L, WS_ROCPRIM, WS_SMALL, WS_LUT,
    WS_COS,  // counters and partial minima of the cosine stage; zero-filled when (re)allocated (skm_ws)
    WS_ZERO, // words every user leaves at zero (zero-filled when allocated): the size-class counters of the count stage
    WS_SCAN, // temporary storage of the count stage's row-pointer scan (its own slot: WS_ROCPRIM may hold sort state
             // that the count stage's last kernel is clearing for the basis stage)
    WS_COUNT
};

struct skm_prof_entry {
    const char *name;
    hipEvent_t start, stop;
};

struct skm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 0;
    int usable_cus = 0;  // compute units the context's stream may use (num_cus unless skm_create_confined masked some out)
    int cu_first = -1, cu_last = -1;  // the CU groups of a confined context's stream (-1: unmasked): the key of the stream cache
    void *ws[WS_COUNT] = {};
    size_t ws_bytes[WS_COUNT] = {};
    void *h_pinned = nullptr;  // small pinned buffer for count read-backs
    // Sticky device-side error word: word SKM_DEVERR_WORD of h_pinned, written by kernels through its device address when an
    // argument the host cannot check turns out wrong on the device (a sequence longer than the caller's max_seq_len); read
    // and cleared by skm_check_device_error wherever the library has just waited for the stream.  nullptr: not addressable.
    uint32_t *d_err = nullptr;
    hipEvent_t ev_host = nullptr;  // marks an asynchronous read-back the host waits for while later kernels run
    bool profiling = false;
    // stream capture (skm_graph_begin .. skm_graph_end): while set, scratch slots must not grow, nothing may wait for the
    // device and no timing events are recorded.  ws_generation counts scratch reallocations: a captured graph holds scratch
    // addresses and is refused (SKM_E_STALE) once one of them has moved.
    // the count stage's size-class counters (WS_ZERO) are cleared by its last kernel; true while a call is between its
    // first and last launch, i.e. still true at the next entry if a call failed in between: that entry clears them itself
    bool count_fill_dirty = false;
    bool capturing = false;
    uint64_t ws_generation = 0;
    std::vector<struct skm_graph *> graphs;  // live captures of this context (skm_graph_end .. skm_graph_destroy / skm_destroy)
    std::vector<skm_prof_entry> prof;
    std::vector<hipEvent_t> event_pool;
    // cosine stage: streams confined to disjoint CU sets (writer 3/4, sparse Gram 1/4) and the events
    // that chain them; created on first use (skm_cosine_csr.hip), overlap_state: 0 untried, 1 ready, -1 unavailable
    hipStream_t s_writer = nullptr, s_gram = nullptr;
    std::vector<hipEvent_t> sync_events;
    int overlap_state = 0;
    std::vector<hipEvent_t> user_events;  // skm_event_record slots, created on first use
    // last run of the heavy-row panel pipeline (skm_cosine_csr.hip), for skm_heavy_panel_stats: device pointers
    const uint32_t *panel_meta = nullptr, *panel_rows = nullptr;
    int panel_nb = 0;
    // RCCL (loaded lazily with dlopen; see skm_comm.hip)
    void *rccl_lib = nullptr;
    void *comm = nullptr;
    int nranks = 1, rank = 0;
};

int skm_ws(skm_ctx *ctx, int slot, size_t bytes, void **out);
// skm_mem.hip: everything the HIP runtime hands out is taken once and recycled (device memory in size classes behind
// completed events, streams by CU mask, events, the contexts' pinned pages); no call of the library relies on the
// implicit device-wide wait of hipFree.
int skm_pool_alloc(skm_ctx *ctx, size_t bytes, void **out);
int skm_pool_free(skm_ctx *ctx, void *ptr);
void skm_registry_add(skm_ctx *ctx);
void skm_registry_remove(skm_ctx *ctx);
int skm_quiesce_device(int device);  // hipStreamSynchronize of every registered stream of the device (not of a capturing one)
hipError_t skm_stream_acquire(int device, int first_group, int last_group, const uint32_t *mask, uint32_t words, hipStream_t *out);
void skm_stream_release(int device, int first_group, int last_group, hipStream_t s);
hipEvent_t skm_event_acquire(int device, bool timing);
void skm_event_release(int device, bool timing, hipEvent_t e);
hipError_t skm_pinned_acquire(int device, void **out_page4096);
void skm_pinned_release(int device, void *page);
void skm_graph_release(struct skm_graph *graph);
constexpr int SKM_DEVERR_WORD = 768;  // byte 3072 of h_pinned
constexpr uint32_t SKM_DEVERR_SEQ_TOO_LONG = 1u;
// Call after a stream synchronise: reports (once) what kernels flagged since the last check.
int skm_check_device_error(skm_ctx *ctx, const char *who);
// similarity -> distance in place over a float32 block: out = clamp(1 - out, 0, 2), no diagonal rule (the epilogue of
// mode 2 of skm_cosine_csr / skm_cosine_dense_i8: sklearn's cosine_distances(X, Y) with Y another matrix than X);
// defined in skm_cosine_csr.hip
int skm_similarity_to_distance(skm_ctx *ctx, int64_t rows, int64_t m, float *d_out, int64_t ld);

// RAII bracket that records start/stop events around a launch when profiling is on.
struct skm_prof_scope {
    skm_ctx *ctx;
    hipStream_t st;
    hipEvent_t stop = nullptr;
    skm_prof_scope(skm_ctx *c, const char *name, hipStream_t on = nullptr);
    ~skm_prof_scope();
};
#define SKM_CAT2(a, b) a##b
#define SKM_CAT(a, b) SKM_CAT2(a, b)
#define SKM_PROF(ctx, name) skm_prof_scope SKM_CAT(_prof_scope_, __LINE__)(ctx, name)
#define SKM_PROF_ON(ctx, name, stream) skm_prof_scope SKM_CAT(_prof_scope_, __LINE__)(ctx, name, stream)

static inline int skm_check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        skm_set_error("launch of %s failed: %s", what, hipGetErrorString(e));
        return SKM_E_HIP;
    }
    return SKM_OK;
}

struct skm_lut256 {
    uint8_t b[256];
};

static inline int64_t skm_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Grid cap for grid-stride kernels: enough workgroups to fill 256 CUs several times over.
static inline int skm_grid_cap(const skm_ctx *ctx, int64_t want, int per_cu = 8)
{
    int64_t cap = (int64_t)(ctx->num_cus > 0 ? ctx->num_cus : 256) * per_cu;
    if (want < 1)
        want = 1;
    return (int)(want < cap ? want : cap);
}

// Which sort the basis stage uses.  Both are stable sorts of the same keys: identical results.  Measured on MI355X
// (tools/ab_sort_mid.sh, red6 k=12, 4 digits, ms per sort, own / rocPRIM): 0.3 M pairs 0.058 / 0.120, 1.0 M 0.108 / 0.160,
// 1.45 M 0.105 / 0.118, 2.0 M 0.127 / 0.128, 2.5 M 0.131 / 0.155, 2.9 M 0.136 / 0.161, 3.5 M 0.159 / 0.175; from 4 M on the
// grid of the own sort no longer fits the chip at once and rocPRIM's tuned ranking wins (4.1 M 0.212 / 0.193, 5.8 M 0.260 /
// 0.251, 29 M 0.89 / 0.85).  So: ours up to 4 M pairs (round 5; 2^21 in round 4) - every single-file job of the reference,
// snekmer/rules/kmerize.smk:57-65, and BASELINE configs[1] - rocPRIM above.  SKM_SORT=rocprim / onesweep force one (A/B timing).
#ifndef SKM_OS_MAX_CAP
#define SKM_OS_MAX_CAP ((int64_t)1 << 22)
#endif
// Process-wide switches between exact kernels (skm_set_option; include/snekmer_hip.h): read from the environment once.
struct skm_options {
    int sort = 0;            // SKM_SORT: 0 by size, 1 rocprim, 2 onesweep
    int cosine_path = 0;     // SKM_COSINE_PATH: 0 by shape, 1 lists, 2 cursor
    int heavy_panel = -1;    // SKM_HEAVY_PANEL: -1 by the hint, 0 off, 1 on
    int heavy_pack = -1;     // SKM_HEAVY_PACK: -1 by the hint, 0 off, 1 on
    int cosine_overlap = 0;  // SKM_COSINE_OVERLAP
    int gram_shape = 0;      // SKM_GRAM_SHAPE
    int dense_variant = 0;   // SKM_DENSE_VARIANT
    // diagnostic builds only (-DSKM_DIAG: results NOT valid for the ablations)
    int cosine_ablate = 0, gram_ablate = 0, overlap_blocks = 0, dense_split = 0, heavy_ablate = 0;
};
const skm_options &skm_opts();

static inline bool skm_use_onesweep(int64_t cap)
{
    const int forced = skm_opts().sort;
    if (forced == 1)
        return false;
    if (forced == 2)
        return cap < ((int64_t)1 << 30);
    return cap <= SKM_OS_MAX_CAP;
}

// Stage functions shared by the fused entry point skm_vectorize_csr (skm_api.hip would be the natural home; they
// live with their kernels in skm_kmer.hip / skm_basis.hip).  Neither waits for the device.
// Work the count stage does for the stage behind it, so that a fused call needs no fill operations (and no histogram pass)
// of its own: colidx_ff[e] = 0xFFFFFFFF for every entry e the last kernel (k_compact_rows) writes; zero_words 32-bit words
// cleared at `zero`.
struct skm_count_extras {
    uint32_t *colidx_ff = nullptr;
    uint32_t *zero = nullptr;  // cleared by the stage's FIRST count kernel (nothing in the stage reads it) ...
    int64_t zero_words = 0;
    // ... so that the last one can already count into it: the digit histograms of the basis stage's sort (8-bit digits of
    // the low hist_key_bits bits of every code, layout skm_onesweep::state_header::hist at the start of `zero`)
    bool hist = false;
    int hist_passes = 0, hist_key_bits = 0;
};
int skm_count_stage_async(skm_ctx *ctx, const uint8_t *h_rank, int nsym, int k, int code_bits, const uint8_t *d_seq,
                          const int64_t *d_off, int64_t n, int64_t total_residues, int64_t max_seq_len, int64_t *d_rowptr, void *d_codes,
                          uint32_t *d_counts, uint64_t *d_rowcount, float *d_rnorm, uint64_t *d_normsq,
                          const skm_count_extras &extras = skm_count_extras());
// prepared: what the count stage already did for this call (colidx pre-filled; `zero` = the sort state it cleared)
int skm_basis_stage_async(skm_ctx *ctx, int code_bits, int key_bits, int64_t cap, const int64_t *d_nnz, const void *d_codes,
                          const uint64_t *d_rowcount, void *d_basis, uint32_t *d_colidx, uint32_t *d_colptr, uint64_t *d_post,
                          int64_t *d_ncols, const skm_count_extras &prepared = skm_count_extras());
int skm_basis_sort_state(skm_ctx *ctx, int64_t cap, int key_bits, int code_bits, uint32_t **out_state, int64_t *out_words,
                         int *out_passes = nullptr, int *out_key_bits = nullptr);
