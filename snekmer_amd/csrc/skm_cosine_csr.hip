// Stage 3 of the hot path: all-pairs cosine over sparse count rows.
//
// Reference behaviour being replaced:
//   sklearn.metrics.pairwise.cosine_similarity(X, Y)     snekmer/rules/apply.smk:282-284,
//                                                        snekmer/rules/learn.smk:821-823,
//                                                        snekmer/rules/evaluate.smk:434-436
//   pairwise_distances(X, metric="cosine")               snekmer/score.py:169-171
//
// Why not a dense GEMM here: at k = 12 the observed basis has >100 columns per sequence
// (SURVEY.md D2/H1); the dense N x B operand would be terabytes and >99.6 % of the N x N Gram
// entries are zero.  The output matrix itself (N*M float32) is the irreducible traffic, so the
// stage is organised around a streaming writer bound by HBM write bandwidth:
//
//   k_gram_sparse      (skm_gram_kernel.h) workgroup per row: groups of lanes walk the posting lists
//                      of the row's non-zeros (each posting read once, next lists' loads in flight)
//                      and accumulate v*v' in an LDS hash table keyed by the neighbour row; the
//                      row's (j, exact int32 dot) entries go to the row's slot of a global list.
//   k_cosine_heavy     the rows the first pass flags (more than 1536 neighbours or 512 distinct k-mers: members
//                      of families of thousands, low-complexity k-mers, long sequences): Gram and write fused,
//                      one workgroup per row, products added with plain LDS atomics into the dense 32768-column
//                      tile the writer uses anyway; no hash table, so no neighbour capacity.
//   k_gram_sparse_big  the list-producing form for such rows (8192-slot table, 4096 non-zeros): used by
//                      skm_gram_neighbors, whose output IS the lists.
//   k_cosine_write     workgroup per output row, pure streaming writer: per 4096-column step it
//                      drops the step's neighbour entries into a zeroed LDS tile, scales to float32
//                      (mode 1: cosine distance) and stores 16 B per lane.  Dominant kernel.
//   k_cosine_strip     exact for ANY density, the fallback for strips the kernels above cannot
//                      hold (more than 2048 shared k-mers in a row) and the default for narrow outputs:
//                      a workgroup owns 8 output rows and walks the columns in chunks of 1024 with
//                      dense int32 LDS accumulators; every non-zero holds a cursor into its
//                      (sorted) posting list and adds v*v' for the postings that fall into the
//                      chunk.  One dependent memory round trip per posting: slow, but general.
//
// Environment knobs of the shipped library select between EXACT kernels only: SKM_COSINE_PATH=cursor
// forces the fallback everywhere, SKM_COSINE_PATH=lists the neighbour-list path also for outputs of at most
// 1024 columns (which the cursor kernel takes by default), SKM_COSINE_OVERLAP=1 the blocked two-stream schedule.  The
// timing-only ablations (SKM_COSINE_ABLATE / SKM_GRAM_ABLATE, results invalid by construction) and
// the phase stamps exist only in the -DSKM_DIAG build (libsnekmer_hip_diag.so, `make diag`), which
// tools/ablate_cosine.py loads; the product library does not read those variables.
// Shapes that were measured and rejected are listed in DESIGN.md.
//
// The dense small-basis case (a true dense GEMM) is served by the i8 MFMA kernels in
// skm_dense.hip instead.
#include <cstdlib>
#include <cstring>

#include <algorithm>

#include "skm_common.h"
#include "skm_onesweep.h"

namespace {

constexpr int R = 8;
constexpr int CH = 1024;
constexpr int TB = 256;
constexpr int Q = 10;  // register-resident tasks per thread -> Q*TB = 2560 tasks per strip
constexpr uint32_t NONE = 0xFFFFFFFFu;
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef SKM_FIRST_GH
#define SKM_FIRST_GH 2048
#endif
#ifndef SKM_WRITER_CH
#define SKM_WRITER_CH 32768
#endif
#ifndef SKM_WRITER_TB
#define SKM_WRITER_TB 1024
#endif
constexpr uint32_t G_DONE_ROW = 0xFFFFFFFEu;  // g_len of a row k_cosine_heavy has already written
constexpr uint32_t G_WIDE_ROW = 0xFFFFFFFDu;  // g_len of a row whose dot products may exceed int32 (skm_row_is_wide)

// ------------------------------------------------------------------------------- sparse Gram
#include "skm_gram_kernel.h"

// Small state of one skm_cosine_csr / skm_gram_neighbors call (scratch slot WS_COS, zero-filled when allocated).
struct cos_state {
    unsigned long long g_counter;  // next free neighbour-list entry            (skm_cosine_csr_stats reads the first
    uint32_t fb_count;             // strips left to the 32-bit cursor kernel     three fields)
    uint32_t over_count[16];       // rows handed to the second pass, per row block of the overlapped schedule
    uint32_t wide_count;           // strips with a wide row (float64 accumulators)
    float min_yrnorm;              // min_j yrnorm[j]: the Y side of the int32 guard (skm_row_is_wide)
    uint32_t blocks_done;          // ticket counter of k_cosine_prologue; always 0 between launches
    unsigned long long first_entry;  // g_counter's starting value (the fixed list slots in front of it)
    uint32_t pad[6];
    float partial[1024];           // per-workgroup minima of k_cosine_prologue
};
constexpr int PRO_MAXB = 1024;

// One launch in front of every cosine call: clears the strip flags and the call's counters and reduces
// min_j yrnorm[j] (workgroup minima, then the last workgroup to finish combines them: one kernel, no pre-initialised
// accumulator).  Replaces two hipMemsetAsync and a one-thread kernel.
__global__ __launch_bounds__(256) void k_cosine_prologue(uint32_t *__restrict__ flags, int64_t nflags, cos_state *st,
                                                         unsigned long long first_entry,
                                                         const float *__restrict__ yrnorm, int64_t m)
{
    __shared__ float s_w[4];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    const int64_t gid = (int64_t)blockIdx.x * 256 + tid, stride = (int64_t)gridDim.x * 256;
    for (int64_t z = gid; z < nflags; z += stride)
        flags[z] = 0u;
    float mn = INFINITY;
    for (int64_t j = gid; j < m; j += stride)
        mn = fminf(mn, yrnorm[j]);
    auto block_min = [&](float v) -> float {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            v = fminf(v, __shfl_xor(v, o));
        __syncthreads();
        if ((tid & 63) == 0)
            s_w[tid >> 6] = v;
        __syncthreads();
        return fminf(fminf(s_w[0], s_w[1]), fminf(s_w[2], s_w[3]));
    };
    mn = block_min(mn);
    if (tid == 0) {
        __hip_atomic_store(&st->partial[blockIdx.x], mn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        s_last = atomicAdd(&st->blocks_done, 1u) == gridDim.x - 1u;
    }
    __syncthreads();
    if (!s_last)
        return;
    __threadfence();
    float all = INFINITY;
    for (uint32_t b = (uint32_t)tid; b < gridDim.x; b += 256)
        all = fminf(all, __hip_atomic_load(&st->partial[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    all = block_min(all);
    if (tid == 0) {
        st->min_yrnorm = all;
        st->g_counter = first_entry;
        st->first_entry = first_entry;
        st->fb_count = 0u;
        st->wide_count = 0u;
        st->blocks_done = 0u;
    }
    if (tid < 16)
        st->over_count[tid] = 0u;
}

static inline int cosine_prologue(skm_ctx *ctx, uint32_t *flags, int64_t nflags, cos_state *state,
                                  unsigned long long first_entry, const float *d_yrnorm, int64_t m, hipStream_t st)
{
    const int64_t work = (nflags > m ? nflags : m);
    int64_t grid = skm_ceil_div(work, 256 * 8);
    grid = grid < 1 ? 1 : (grid > PRO_MAXB ? PRO_MAXB : grid);
    SKM_PROF(ctx, "k_cosine_prologue");
    k_cosine_prologue<<<(unsigned)grid, 256, 0, st>>>(flags, nflags, state, first_entry, d_yrnorm, m);
    return skm_check_launch("k_cosine_prologue");
}

// WIDE = false: exact int32 accumulators (dot products below 2^31: rows that pass skm_row_is_wide's test).
// WIDE = true: float64 accumulators, the reference's own arithmetic (sklearn works in float64): exact while a dot
// product stays below 2^53, rounded like any float64 sum beyond, no upper limit; counts use all 32 bits.
// strip_list == nullptr: one strip per workgroup, strip = blockIdx.x (the grid covers all strips).  A 32-bit launch
// that is given `wide_list` appends the strips holding a wide row to it and leaves them unwritten.
// strip_list != nullptr: workgroups stride over the first *strip_count entries of the list (any grid).
template <int MODE, bool VEC, int ABL, typename PW, bool WIDE>
__global__ __launch_bounds__(TB) void k_cosine_strip(const int64_t *__restrict__ xrowptr,
                                                     const uint32_t *__restrict__ xcolidx,
                                                     const uint32_t *__restrict__ xcounts,
                                                     const float *__restrict__ xrnorm, int64_t m,
                                                     const uint32_t *__restrict__ ycolptr,
                                                     const PW *__restrict__ ypost,
                                                     const uint32_t *__restrict__ ypostcnt,
                                                     const float *__restrict__ yrnorm, int64_t row0, int64_t row1,
                                                     float *__restrict__ out, int64_t ld,
                                                     const uint32_t *__restrict__ strip_list,
                                                     const uint32_t *__restrict__ strip_count,
                                                     const float *__restrict__ min_yrnorm,
                                                     uint32_t *__restrict__ wide_list, uint32_t *__restrict__ wide_count)
{
    using acc_t = typename std::conditional<WIDE, double, int>::type;
    __shared__ __attribute__((aligned(16))) acc_t s_acc[R][CH];
    __shared__ int64_t s_rp[R + 1];
    __shared__ float s_rni[R];
    __shared__ int s_skip;
    const int tid = threadIdx.x;
    const uint32_t nlist = strip_list ? *strip_count : 0u;
    if constexpr (WIDE) {
        // The float64 launch is the last kernel of a list-path call.  Given (wide_list, wide_count) = (the context's pinned
        // hint words, the call's per-block hand-off counters) its first workgroup copies the 16 counters to the host:
        // what a hipMemcpyAsync (one more stream operation per call) did before.  Read without waiting by the next call
        // (heavy_panels_wanted).
        if (wide_list && wide_count && blockIdx.x == 0 && tid < 16)
            __hip_atomic_store(wide_list + tid, wide_count[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int z = tid; z < (int)(R * CH * sizeof(acc_t) / 16); z += TB)
        reinterpret_cast<int4 *>(&s_acc[0][0])[z] = make_int4(0, 0, 0, 0);
    for (uint32_t it = blockIdx.x;; it += gridDim.x) {
    if (strip_list ? it >= nlist : it != blockIdx.x)  // uniform for the workgroup
        break;
    const int64_t strip = strip_list ? strip_list[it] : blockIdx.x;
    const int64_t i0 = row0 + strip * R;
    const int rows = (int)min((int64_t)R, row1 - i0);

    __syncthreads();  // the previous strip's row pointers and accumulators are no longer in use
    if (tid <= R) {
        int64_t r = tid <= rows ? tid : rows;
        s_rp[tid] = xrowptr[i0 + r];
    }
    if (tid < R)
        s_rni[tid] = tid < rows ? xrnorm[i0 + tid] : 0.0f;
    if (tid == 0)
        s_skip = 0;
    __syncthreads();
    if (!WIDE && wide_list && !strip_list) {
        if (tid < rows && skm_row_is_wide(s_rni[tid], *min_yrnorm))
            s_skip = 1;
        __syncthreads();
        if (s_skip) {  // uniform: a row of this strip may exceed int32; the float64 launch behind this one takes it
            if (tid == 0)
                wide_list[atomicAdd(wide_count, 1u)] = (uint32_t)strip;
            continue;
        }
    }

    const int64_t e0 = s_rp[0];
    const int64_t ntasks = s_rp[R] - e0;

    uint32_t cur[Q], rem[Q], nj[Q], nv[Q], liv[Q];
    uint32_t xv[WIDE ? Q : 1];  // WIDE: the full 32-bit count (liv keeps 28 bits beside the row index)
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int64_t t = (int64_t)tid + (int64_t)q * TB;
        nj[q] = NONE;
        cur[q] = rem[q] = nv[q] = liv[q] = 0;
        if (WIDE)
            xv[q] = 0;
        if (ABL != 1 && t < ntasks) {
            const int64_t e = e0 + t;
            int li = 0;
#pragma unroll
            for (int r = 1; r < R; ++r)
                li += (e >= s_rp[r]) ? 1 : 0;
            const uint32_t c = xcolidx[e];
            const uint32_t v = xcounts[e];
            liv[q] = ((uint32_t)li << 28) | (v & 0x0FFFFFFFu);
            if (WIDE)
                xv[q] = v;
            if (c == 0xFFFFFFFFu) {  // singleton k-mer (ELIDE_SINGLETONS): pairs with its own row only
                rem[q] = 1;
                nj[q] = (uint32_t)(i0 + li);
                nv[q] = v;
            } else {
                const uint32_t pb = ycolptr[c], pe = ycolptr[c + 1];
                cur[q] = pb;
                rem[q] = pe - pb;
                if (pe > pb) {
                    const PW pw = ypost[pb];
                    nj[q] = posting<PW>::row(pw);
                    nv[q] = posting<PW>::count(pw, ypostcnt, pb);
                }
            }
        }
    }
    auto product = [](uint32_t a, uint32_t b) -> acc_t {
        if constexpr (WIDE)
            return (double)a * (double)b;
        else
            return (int)a * (int)b;
    };

    for (int64_t j0 = 0; j0 < m; j0 += CH) {
        const int64_t j1 = min(j0 + (int64_t)CH, m);
        const uint32_t j1u = (uint32_t)j1, j0u = (uint32_t)j0;
        // ---- accumulate: advance every cursor through [j0, j1).  One pass handles at most one
        // posting per task; the loads that fetch each task's next posting are all issued before
        // any of them is waited for, so a pass costs one memory round trip, not Q of them.
        bool more;
        do {
            uint32_t tj[Q], tv[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                tj[q] = nj[q];
                tv[q] = nv[q];
                if (nj[q] < j1u) {
                    const int li = (int)(liv[q] >> 28);
                    const uint32_t v = WIDE ? xv[WIDE ? q : 0] : (liv[q] & 0x0FFFFFFFu);
                    atomicAdd(&s_acc[li][nj[q] - j0u], product(v, nv[q]));
                    ++cur[q];
                    --rem[q];
                    tj[q] = NONE;
                    if (rem[q]) {
                        const PW pw = ypost[cur[q]];
                        tj[q] = posting<PW>::row(pw);
                        tv[q] = posting<PW>::count(pw, ypostcnt, cur[q]);
                    }
                }
            }
            more = false;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                nj[q] = tj[q];
                nv[q] = tv[q];
                more |= nj[q] < j1u;
            }
        } while (__any(more));
        // ---- strips with more than Q*TB non-zeros (very long sequences): stateless tasks
        for (int64_t t = (int64_t)Q * TB + tid; ABL != 1 && t < ntasks; t += TB) {
            const int64_t e = e0 + t;
            int li = 0;
#pragma unroll
            for (int r = 1; r < R; ++r)
                li += (e >= s_rp[r]) ? 1 : 0;
            const uint32_t c = xcolidx[e];
            const uint32_t v = xcounts[e];
            if (c == 0xFFFFFFFFu) {
                const uint32_t j = (uint32_t)(i0 + li);
                if (j >= j0u && j < j1u)
                    atomicAdd(&s_acc[li][j - j0u], product(v, v));
                continue;
            }
            uint32_t lo = ycolptr[c], hi = ycolptr[c + 1];
            const uint32_t pe = hi;
            while (lo < hi) {
                uint32_t mid = lo + ((hi - lo) >> 1);
                if (posting<PW>::row(ypost[mid]) < j0u)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            for (; lo < pe; ++lo) {
                const PW pw = ypost[lo];
                const uint32_t j = posting<PW>::row(pw);
                if (j >= j1u)
                    break;
                atomicAdd(&s_acc[li][j - j0u], product(v, posting<PW>::count(pw, ypostcnt, lo)));
            }
        }
        __syncthreads();
        // ---- epilogue: scale, store, clear
        const int64_t jc = j0 + 4 * tid;
        float rj[4] = {0.f, 0.f, 0.f, 0.f};
        if (VEC && jc + 3 < m) {
            const float4 t4 = *reinterpret_cast<const float4 *>(yrnorm + jc);
            rj[0] = t4.x, rj[1] = t4.y, rj[2] = t4.z, rj[3] = t4.w;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (jc + u < m)
                    rj[u] = yrnorm[jc + u];
        }
#pragma unroll
        for (int li = 0; li < R; ++li) {
            acc_t a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = s_acc[li][4 * tid + u];
                s_acc[li][4 * tid + u] = acc_t(0);
            }
            if (li < rows) {
                const float ri = s_rni[li];
                float o[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if constexpr (WIDE)
                        o[u] = (float)(a[u] * (double)ri * (double)rj[u]);
                    else
                        o[u] = (float)a[u] * ri * rj[u];
                }
                const int64_t i = i0 + li;
                if (MODE == 1) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        float d = 1.0f - o[u];
                        d = fminf(fmaxf(d, 0.0f), 2.0f);
                        o[u] = (jc + u == i) ? 0.0f : d;
                    }
                }
                float *dst = out + (i - row0) * ld + jc;
                if (ABL == 2) {
                    asm volatile("" ::"v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]));
                } else if (VEC && jc + 3 < m) {
                    f32x4 pack = {o[0], o[1], o[2], o[3]};
                    if (ABL == 3)
                        *reinterpret_cast<f32x4 *>(dst) = pack;
                    else
                        __builtin_nontemporal_store(pack, reinterpret_cast<f32x4 *>(dst));
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (jc + u < m)
                            dst[u] = o[u];
                }
            }
        }
        __syncthreads();
    }
    }
}


// ------------------------------------------------------------------------------- streaming writer
// One output row per workgroup, WCH columns per step, WCH/4 threads (4 columns per thread and step).
// The row's neighbour list (any order) is held in registers, WREG entries per thread; per step
// a thread drops its entries that fall into the step's column range into the LDS tile.  Longer
// lists re-read the tail from global memory (L2) on every step.
template <int MODE, bool VEC, int WCH, int WTB>
__global__ __launch_bounds__(WTB) void k_cosine_write(const uint64_t *__restrict__ g_ent,
                                                     const uint64_t *__restrict__ g_start,
                                                     const uint32_t *__restrict__ g_len,
                                                     const float *__restrict__ xrnorm,
                                                     const float *__restrict__ yrnorm, int64_t m, int64_t row0,
                                                     int64_t rbase, float *__restrict__ out, int64_t ld,
                                                     uint32_t *__restrict__ fb_list, uint32_t *__restrict__ fb_count,
                                                     uint32_t *__restrict__ fb_flag, uint32_t *__restrict__ wide_list,
                                                     uint32_t *__restrict__ wide_count)
{
    constexpr int CH = WCH, TB = WTB, NV = WCH / 4 / WTB, WREG = 2;  // NV 16-byte vectors per thread and step
    __shared__ __attribute__((aligned(16))) float s_acc[CH];
    const int tid = threadIdx.x;
    const int64_t r = rbase + blockIdx.x;  // row of the workgroup, counted from row0 (a launch covers rows rbase...)
    const int64_t i = row0 + r;
    const uint32_t len = g_len[r];
    if (len == G_DONE_ROW)  // written by k_cosine_heavy (Gram and write fused for rows with thousands of neighbours)
        return;
    if (len == G_OVERFLOW || len == G_WIDE_ROW) {
        // the row exceeded the sparse kernels' capacities (or the int32 range: G_WIDE_ROW): leave its strip to the
        // cursor kernel, 32-bit or float64 form (cursor strips are 8 rows; a bit of fb_flag per list makes sure a strip
        // is listed once).  A strip on both lists is written by both launches, the float64 one last.
        constexpr int CURSOR_R = 8;
        if (tid == 0) {
            const uint32_t strip = (uint32_t)(r / CURSOR_R), bit = len == G_WIDE_ROW ? 2u : 1u;
            if ((atomicOr(&fb_flag[strip], bit) & bit) == 0u) {
                if (len == G_WIDE_ROW)
                    wide_list[atomicAdd(wide_count, 1u)] = strip;
                else
                    fb_list[atomicAdd(fb_count, 1u)] = strip;
            }
        }
        return;
    }
    // Entries become (column, final float) pairs BEFORE the column loop, so that the loop itself
    // issues no loads: a wave never waits on vmcnt there and its stores stay in flight across
    // steps (with a load per step, the wait for it also drains the previous step's stores).
    const uint64_t *ent = g_ent + g_start[r];
    const float ri = xrnorm[i];
    const float background = MODE == 1 ? 1.0f : 0.0f;
    auto value = [&](uint64_t x, float rj) -> float {
        float o = (float)(int)(uint32_t)x * ri * rj;
        if (MODE == 1) {
            o = fminf(fmaxf(1.0f - o, 0.0f), 2.0f);
            if ((int64_t)(uint32_t)(x >> 32) == i)
                o = 0.0f;
        }
        return o;
    };
    // both entry loads, then both norm gathers, are issued back to back (the slot memory is always
    // readable, so lanes past the end of the list load entry 0 and discard it)
    uint64_t wx[WREG];
    float wr[WREG];
#pragma unroll
    for (int u = 0; u < WREG; ++u)
        wx[u] = ent[(uint32_t)(tid + u * TB) < len ? tid + u * TB : 0];
#pragma unroll
    for (int u = 0; u < WREG; ++u) {
        const uint32_t j = (uint32_t)(wx[u] >> 32);
        wr[u] = yrnorm[(uint32_t)(tid + u * TB) < len && (int64_t)j < m ? j : 0u];
    }
    uint32_t wj[WREG];
    float wv[WREG];
#pragma unroll
    for (int u = 0; u < WREG; ++u) {
        const bool have = (uint32_t)(tid + u * TB) < len;
        wj[u] = have ? (uint32_t)(wx[u] >> 32) : 0xFFFFFFFFu;  // 0xFFFFFFFF never matches a column
        wv[u] = value(wx[u], wr[u]);
    }
    const f32x4 bg4 = {background, background, background, background};
#pragma unroll
    for (int q = 0; q < NV; ++q)
        reinterpret_cast<f32x4 *>(s_acc)[tid + q * TB] = bg4;
    __syncthreads();

    for (int64_t j0 = 0; j0 < m; j0 += CH) {
        const uint32_t j0u = (uint32_t)j0;
#pragma unroll
        for (int u = 0; u < WREG; ++u) {
            const uint32_t dj = wj[u] - j0u;
            if (dj < (uint32_t)CH)
                s_acc[dj] = wv[u];
        }
        for (uint32_t e = (uint32_t)tid + WREG * TB; e < len; e += TB) {  // long lists: tail from L2 every step
            const uint64_t x = ent[e];
            const uint32_t dj = (uint32_t)(x >> 32) - j0u;
            if (dj < (uint32_t)CH)
                s_acc[dj] = value(x, yrnorm[(uint32_t)(x >> 32)]);
        }
        if (MODE == 1) {  // the diagonal is an exact zero even when the row has no entry for itself
            const uint32_t dj = (uint32_t)i - j0u;
            if (tid == 0 && dj < (uint32_t)CH && i < m)
                s_acc[dj] = 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int64_t jc = j0 + 4 * (tid + q * TB);
            const f32x4 o = reinterpret_cast<f32x4 *>(s_acc)[tid + q * TB];
            reinterpret_cast<f32x4 *>(s_acc)[tid + q * TB] = bg4;
            float *dst = out + r * ld + jc;
            if (VEC && jc + 3 < m) {
                __builtin_nontemporal_store(o, reinterpret_cast<f32x4 *>(dst));
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (jc + u < m)
                        dst[u] = o[u];
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------- heavy rows: Gram + write fused
// Rows the first sparse pass cannot hold (more than 512 non-zeros, or more than 1536 neighbours: members of families
// of thousands, low-complexity k-mers shared by thousands of rows, long sequences) used to go through an 8192-slot
// hash pass with one workgroup per CU that walks one posting list at a time (measured on synth_skewed at 100 k rows:
// 1.6 us per row, and again 1.6 us per row in the cursor kernel for what overflowed that table or the list space:
// 295 ms per step against 11 ms on uniform families).  Such a row has thousands of non-zero cells, so its products
// go straight into the dense LDS tile the streaming writer uses anyway: one workgroup per row, 32768 columns per step;
// per step every wave takes posting lists in turn and walks the part of each list that falls into the step's column
// range with lane-consecutive loads (postings are sorted by row, a cursor per list remembers where the previous step
// stopped; 256 postings per wave and round for lists of more than 64, four lists per wave and round for the short
// ones), adding v * v' with plain LDS atomics: no hash, no probe, no capacity other than the number of the row's own
// shared non-zeros (HEAVY_EMAX; rows beyond that stay flagged for the cursor kernel).  The step's tile is then scaled
// and stored like the writer's.  A row done here gets g_len = G_DONE_ROW: the writer's workgroup for it exits.
constexpr int HEAVY_CH = 32768, HEAVY_TB = 1024, HEAVY_EMAX = 2048, HEAVY_U = 4, HEAVY_NG = 4;
// 128 KiB tile + 24 KiB of list state: this kernel (like the writer's 128 KiB tile) needs gfx950's 160 KiB of LDS per
// workgroup; the library is built for gfx950 only (csrc/Makefile).
static_assert(HEAVY_CH * 4 + 3 * HEAVY_EMAX * 4 + 64 <= 160 * 1024, "k_cosine_heavy needs 160 KiB of LDS (gfx950)");
#ifndef SKM_HEAVYK_TB
#define SKM_HEAVYK_TB 1024  // (512: two workgroups per CU, no room for the cache: A/B builds)
#endif
constexpr int HEAVYK_TB = SKM_HEAVYK_TB, HEAVYK_EMAX = SKM_HEAVYK_TB == 1024 ? 2048 : 1024;  // the PACK forms: 64 KiB tile (two columns per word) + 24 KiB of list state + 64 KiB of cached postings
constexpr int HEAVYP_CH = 32768, HEAVYP_TB = 1024, HEAVYP_EMAX = 2048;  // the PANEL form (16384 / 512 / 1024, two workgroups per CU, measured slower: 15.3 vs 13.8 ms)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libsnekmer_hip is written for gfx950 (MI355X): 160 KiB LDS tiles, gfx950 MFMA shapes"
#endif

#include "skm_heavy_panel.h"

// PANEL: the heavy rows come in min-hash order (pb.perm), blocks of PB_ROWS of them own int8 panels whose product G
// already holds the dot products over the block's long-list columns (skm_heavy_panel.h): those columns are skipped
// here and the row of G is added through the block's row list J.
// HC columns per step, HT threads, HE list entries: (32768, 1024, 2048) = one workgroup per CU, the general form;
// (16384, 512, 1024) = the PANEL form: two workgroups per CU (what remains to walk beside a panel is a few short lists:
// the row is bound by the latency of its steps, which a second row in flight hides)
// PACK: TWO columns per 32-bit accumulator word (16 bits each), for rows whose dot products are all below 2^16 by
// Cauchy-Schwarz (rnorm_x * min_j rnorm_y > 2^-16: every row of ordinary proteins; the sums are of non-negative terms, so no
// partial sum carries into the neighbour's half either).  The tile of a 32768-column step is then 64 KiB, and the 64 KiB it
// frees hold a CACHE of the row's postings: (neighbour row, v * v') of every list the panel does not cover, fetched ONCE per
// row in one round of loads - all short lists if their postings fit, the long ones too if there is room - and scanned
// out of LDS in every column step.  Without it every list is visited in every step (a cursor, a round trip to its
// postings, mostly to find that none falls into the step's columns): the row's time was those rounds - with 8 instead of
// 16 waves per row and two rows per CU (a 512-thread form of PACK) a row took exactly twice as long.  Rows that fail
// the 16-bit test are left to the launch of the unpacked form behind this one; lists that do not fit the cache are walked
// per step as before.
#ifdef SKM_DIAG
// timing-only ablations of k_cosine_heavy (results NOT valid; SKM_HEAVY_ABLATE bits: 1 no global stores, 2 no walk of long
// lists, 4 no panel row, 8 no short lists / cache, 16 nothing after the row's set-up)
__device__ int g_heavy_abl;
#define SKM_HEAVY_ABL(bit) (heavy_abl & (bit))
#else
#define SKM_HEAVY_ABL(bit) false
#endif
template <int MODE, bool VEC, typename PW, bool PANEL, int HC, int HT, int HE, bool PACK = false>
__global__ __launch_bounds__(HT) void k_cosine_heavy(const int64_t *__restrict__ xrowptr,
                                                           const uint32_t *__restrict__ xcolidx,
                                                           const uint32_t *__restrict__ xcounts,
                                                           const float *__restrict__ xrnorm, int64_t m,
                                                           const uint32_t *__restrict__ ycolptr,
                                                           const PW *__restrict__ ypost,
                                                           const uint32_t *__restrict__ ypostcnt,
                                                           const float *__restrict__ yrnorm, int64_t row0, int64_t rbase,
                                                           const uint32_t *__restrict__ row_list,
                                                           const uint32_t *__restrict__ row_count,
                                                           uint32_t *__restrict__ g_len, float *__restrict__ out, int64_t ld,
                                                           panel_bufs pnl, const float *__restrict__ min_yrnorm)
{
    constexpr int CHH = HC, TBH = HT, NW = HT / 64, U = HEAVY_U;
    constexpr int ACCW = PACK ? HC / 2 : HC;  // accumulator words per step
    constexpr bool CACHE = PACK && HT == 1024;
    constexpr int CN = CACHE ? 8192 : 1;  // cached postings per row
    static_assert(ACCW * 4 + 3 * HE * 4 + (CACHE ? CN * 8 : 0) + 1024 <= (HT == 1024 ? 160 : 80) * 1024, "LDS budget of k_cosine_heavy");
    __shared__ __attribute__((aligned(16))) int s_acc[ACCW];
    __shared__ uint2 s_cache[CN];                     // (neighbour row, v * v')
    __shared__ uint32_t s_ncache, s_tshort, s_tlong;  // cache fill; postings behind the row's short / long lists
    __shared__ uint32_t s_cur[HE], s_end[HE], s_val[HE];
    __shared__ uint32_t s_nlong, s_nshort, s_self;
    __shared__ uint32_t s_jb[PB_STEPS + 2];  // PANEL: first entry of the block's row list J at or after every column step's start
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t cnt = *row_count;
    for (int z = tid; z < ACCW / 4; z += TBH)
        reinterpret_cast<int4 *>(s_acc)[z] = make_int4(0, 0, 0, 0);
    auto acc_add = [&](uint32_t off, int val) {
        if (PACK)
            atomicAdd(&s_acc[off >> 1], val << ((off & 1u) * 16u));
        else
            atomicAdd(&s_acc[off], val);
    };
    const float min_yr = PACK ? *min_yrnorm : 0.0f;
#ifdef SKM_DIAG
    const int heavy_abl = g_heavy_abl;
#endif
    for (uint32_t idx = blockIdx.x; idx < cnt; idx += gridDim.x) {
        const int64_t r = rbase + row_list[PANEL ? pnl.perm[idx] : idx];  // row counted from row0 (g_len, out)
        const int64_t i = row0 + r;
        // the row's panel block (sorted position idx), if it has one with columns in it
        const uint32_t pblk = idx / (uint32_t)PB_ROWS, prow = idx % (uint32_t)PB_ROWS;
        const uint32_t pK = (PANEL && pblk < (uint32_t)pnl.nb) ? pnl.meta[pblk * 4 + 0] : 0u;
        const uint32_t pJ = pK ? pnl.meta[pblk * 4 + 1] : 0u;
        if (g_len[r] == G_DONE_ROW || (PANEL && pK == 0u))  // uniform.  Not again; PANEL form: only rows with a panel (the
            continue;                                        // general form, launched behind it, takes the others)
        if (PACK && !((double)xrnorm[i] * (double)min_yr > 0x1p-16 * (1.0 + 1e-3)))  // a dot product may need more than 16 bits
            continue;
        __syncthreads();  // the previous row's tile and lists are no longer in use
        if (tid == 0) {
            s_nlong = 0;
            s_nshort = 0;
            s_self = 0;
            s_ncache = 0;
            s_tshort = 0;
            s_tlong = 0;
        }
        static_assert(!PANEL || CHH == PB_STEP_COLS, "k_panel_rows cuts J at multiples of PB_STEP_COLS");
        if (PANEL && pJ && tid <= PB_STEPS)  // J is sorted: the part of it inside column step t is [s_jb[t], s_jb[t + 1])
            s_jb[tid] = pnl.jbound[(size_t)pblk * (PB_STEPS + 1) + tid];
        __syncthreads();
        // ---- the row's shared non-zeros -> (cursor, end, count); lists of more than 64 postings are stored from the
        // front of the arrays, the others from the back
        const int64_t e0 = xrowptr[i], e1 = xrowptr[i + 1];
        uint32_t self = 0;
        for (int64_t t0 = e0; t0 < e1; t0 += TBH) {  // whole-wave iterations (ballots)
            const int64_t t = t0 + tid;
            uint32_t pb = 0, pe = 0, v = 0;
            if (t < e1) {
                const uint32_t c = xcolidx[t];
                v = xcounts[t] & 0x0FFFFFFFu;
                if (c == NONE)
                    self += v * v;  // a k-mer of this row only (ELIDE_SINGLETONS)
                else {
                    pb = ycolptr[c];
                    pe = ycolptr[c + 1];
                    if (PANEL && pK && pe - pb > PB_DF_LONG && pe - pb <= PB_DF_MAX && panel_lookup(pnl, pblk, c) != PB_NOSLOT)
                        pb = pe = 0;  // this column's products are in the block's G
                }
            }
            const bool is_long = pe - pb > 64u, is_short = pe > pb && !is_long;
            const unsigned long long bl = __ballot(is_long), bs = __ballot(is_short);
            uint32_t basel = 0, bases = 0;
            if (lane == 0) {
                if (bl)
                    basel = atomicAdd(&s_nlong, (uint32_t)__popcll(bl));
                if (bs)
                    bases = atomicAdd(&s_nshort, (uint32_t)__popcll(bs));
            }
            basel = __shfl(basel, 0);
            bases = __shfl(bases, 0);
            const unsigned long long below = (1ull << lane) - 1ull;
            uint32_t slot = 0xFFFFFFFFu;
            if (is_long)
                slot = basel + (uint32_t)__popcll(bl & below);
            else if (is_short)
                slot = (uint32_t)HE - 1u - (bases + (uint32_t)__popcll(bs & below));
            // (on overflow the counters keep counting, nothing is stored out of range, and the row is skipped below)
            if (slot < (uint32_t)HE) {
                s_cur[slot] = pb;
                s_end[slot] = pe;
                s_val[slot] = v;
            }
            if (CACHE) {  // postings behind the short and the long lists (wave sums)
                uint32_t ts = is_short ? pe - pb : 0u, tl = is_long ? min(pe - pb, 1u << 20) : 0u;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    ts += __shfl_xor(ts, o);
                    tl += __shfl_xor(tl, o);
                }
                if (lane == 0 && ts)
                    atomicAdd(&s_tshort, ts);
                if (lane == 0 && tl)
                    atomicAdd(&s_tlong, min(tl, 1u << 20));
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            self += __shfl_xor(self, o);
        if (lane == 0 && self)
            atomicAdd(&s_self, self);
        __syncthreads();
        const uint32_t nlong = s_nlong, nshort = s_nshort;
        if (nlong + nshort > (uint32_t)HE)  // uniform: the row stays flagged (cursor kernel)
            continue;
        const float ri = xrnorm[i];
        // ---- CACHE: the row's postings into LDS, one round of loads (see the comment in front of the kernel)
        if (SKM_HEAVY_ABL(16)) {
            if (tid == 0)
                g_len[r] = G_DONE_ROW;
            continue;
        }
        const bool cache_short = CACHE && s_tshort <= (uint32_t)CN && !SKM_HEAVY_ABL(8);
        const bool cache_long = cache_short && nlong && s_tlong <= (uint32_t)CN - s_tshort;
        if (CACHE) {
            if (cache_short) {
                for (uint32_t l0 = (uint32_t)wid * 4u; l0 < nshort; l0 += NW * 4u) {
                    const uint32_t l = l0 + (uint32_t)(lane >> 4), gl = (uint32_t)(lane & 15);
                    const bool have = l < nshort;
                    const uint32_t slot = (uint32_t)HE - 1u - (have ? l : 0u);
                    const uint32_t p = s_cur[slot], pe = have ? s_end[slot] : 0u, v = s_val[slot];
                    uint32_t base = 0;
                    if (have && gl == 0)
                        base = atomicAdd(&s_ncache, pe - p);
                    base = __shfl(base, lane & 48);
                    PW pw[4];
#pragma unroll
                    for (int round = 0; round < 4; ++round) {  // 4 x 16 >= 64 postings
                        const uint32_t at = p + (uint32_t)round * 16u + gl;
                        pw[round] = ypost[have && at < pe ? at : 0u];  // posting 0 exists: the row has a non-empty list
                    }
#pragma unroll
                    for (int round = 0; round < 4; ++round) {
                        const uint32_t at = p + (uint32_t)round * 16u + gl;
                        if (have && at < pe)
                            s_cache[base + (at - p)] = make_uint2(posting<PW>::row(pw[round]), v * posting<PW>::count(pw[round], ypostcnt, at));
                    }
                }
            }
            if (cache_long) {
                for (uint32_t l = (uint32_t)wid; l < nlong; l += NW) {  // wave-uniform
                    const uint32_t p = s_cur[l], pe = s_end[l], v = s_val[l];
                    uint32_t base = 0;
                    if (lane == 0)
                        base = atomicAdd(&s_ncache, pe - p);
                    base = __shfl(base, 0);
                    for (uint32_t at0 = p; at0 < pe; at0 += (uint32_t)(U * 64)) {
                        PW pw[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const uint32_t at = at0 + (uint32_t)(u * 64 + lane);
                            pw[u] = ypost[at < pe ? at : 0u];
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const uint32_t at = at0 + (uint32_t)(u * 64 + lane);
                            if (at < pe)
                                s_cache[base + (at - p)] = make_uint2(posting<PW>::row(pw[u]), v * posting<PW>::count(pw[u], ypostcnt, at));
                        }
                    }
                }
            }
            __syncthreads();
        }
        const uint32_t ncache = CACHE ? s_ncache : 0u;
        // PANEL: the row of G reaches the tile through registers that are loaded one step ahead, in front of the previous
        // step's stores: a wait for a load also waits for every store issued before it, so loads issued behind the
        // stores would expose the stores' latency in every step
        constexpr int GP = PANEL ? 2048 / TBH : 1;
        int gv[GP];
        uint32_t gj[GP];
        const int *grow = PANEL ? pnl.G + ((size_t)pblk * PB_ROWS + prow) * PB_JMAX : nullptr;
        const uint32_t *gjl = PANEL ? pnl.jlist + (size_t)pblk * PB_JMAX : nullptr;
        auto g_prefetch = [&](uint32_t step) {
            if (!(PANEL && pJ))
                return;
#pragma unroll
            for (int u = 0; u < GP; ++u) {
                const uint32_t q = s_jb[step] + (uint32_t)(tid + u * TBH);
                const bool ok = q < s_jb[step + 1];
                gv[u] = ok ? grow[q] : 0;
                gj[u] = gjl[ok ? q : 0u];
            }
        };
        g_prefetch(0);
        for (int64_t j0 = 0; j0 < m; j0 += CHH) {
            const uint32_t j0u = (uint32_t)j0;
            const uint32_t j1u = (uint32_t)min(j0 + (int64_t)CHH, m);
            // ---- long lists: a wave takes NG lists at a time; every round it issues U loads per lane for EACH list of the
            // group that still has postings in this column range, then adds them: one memory round trip per round of
            // the whole group (with one list at a time, the second and later rounds of every list were a round trip
            // each and set the kernel's rate)
            if (nlong && !cache_long && !SKM_HEAVY_ABL(2)) {
                constexpr int NG = HEAVY_NG;
                for (uint32_t l0 = (uint32_t)wid * NG; l0 < nlong; l0 += NW * NG) {  // wave-uniform
                    uint32_t lp[NG], lpe[NG], lv[NG];
                    bool act[NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const uint32_t l = l0 + (uint32_t)g;
                        const bool have = l < nlong;
                        const uint32_t ll = have ? l : 0u;
                        lp[g] = s_cur[ll];
                        lpe[g] = have ? s_end[ll] : 0u;
                        lv[g] = s_val[ll];
                        act[g] = lp[g] < lpe[g];
                    }
                    bool any = false;
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        any |= act[g];
                    while (any) {  // all values here are wave-uniform
                        PW buf[NG][U];
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                const uint32_t at = lp[g] + (uint32_t)(u * 64 + lane);
                                buf[g][u] = ypost[act[g] && at < lpe[g] ? at : 0u];  // posting 0 exists: nlong > 0
                            }
                        }
                        any = false;
#pragma unroll
                        for (int g = 0; g < NG; ++g) {
                            uint32_t took = 0;
#pragma unroll
                            for (int u = 0; u < U; ++u) {
                                const uint32_t at = lp[g] + (uint32_t)(u * 64 + lane);
                                const uint32_t j = posting<PW>::row(buf[g][u]);
                                const bool in = act[g] && at < lpe[g] && j < j1u;
                                if (in)
                                    acc_add(j - j0u, (int)(lv[g] * posting<PW>::count(buf[g][u], ypostcnt, at)));
                                took += (uint32_t)__popcll(__ballot(in));
                            }
                            lp[g] += took;
                            act[g] = act[g] && took == (uint32_t)(U * 64) && lp[g] < lpe[g];
                            any |= act[g];
                        }
                    }
#pragma unroll
                    for (int g = 0; g < NG; ++g)
                        if (lane == 0 && l0 + (uint32_t)g < nlong)
                            s_cur[l0 + g] = lp[g];
                }
            }
            // ---- the cached postings: those of this step's columns
            if (CACHE) {
                for (uint32_t q = (uint32_t)tid; q < ncache; q += (uint32_t)TBH) {
                    const uint2 e = s_cache[q];
                    if (e.x >= j0u && e.x < j1u)
                        acc_add(e.x - j0u, (int)e.y);
                }
            }
            // ---- short lists (at most 64 postings): four lists per wave and round, 16 lanes each
            for (uint32_t l0 = (uint32_t)wid * 4u; l0 < (cache_short || SKM_HEAVY_ABL(8) ? 0u : nshort); l0 += NW * 4u) {
                const uint32_t l = l0 + (uint32_t)(lane >> 4), gl = (uint32_t)(lane & 15);
                const bool have = l < nshort;
                const uint32_t slot = (uint32_t)HE - 1u - (have ? l : 0u);
                uint32_t p = s_cur[slot];
                const uint32_t pe = have ? s_end[slot] : 0u, v = s_val[slot];
                bool active = have;
                for (int round = 0; round < 4; ++round) {  // 4 x 16 >= 64 postings; every lane runs the same rounds
                    const uint32_t at = p + gl;
                    const bool ok = active && at < pe;
                    const PW pw = ypost[ok ? at : 0u];  // posting 0 exists: the row has a non-empty list
                    const uint32_t j = posting<PW>::row(pw);
                    const bool in = ok && j < j1u;
                    if (in)
                        acc_add(j - j0u, (int)(v * posting<PW>::count(pw, ypostcnt, at)));
                    const unsigned long long bal = __ballot(in);
                    const uint32_t took = (uint32_t)__popcll((bal >> (lane & 48)) & 0xFFFFull);
                    p += took;
                    active = active && took == 16u && p < pe;
                    if (!__any(active))
                        break;
                }
                if (have && gl == 0)
                    s_cur[slot] = p;
            }
            if (tid == 0 && (uint32_t)i >= j0u && (uint32_t)i < j1u && s_self)
                acc_add((uint32_t)i - j0u, (int)s_self);
            const uint32_t step = (uint32_t)(j0 / CHH);  // PANEL: m <= 2^20, at most PB_STEPS steps
            if (PANEL && pJ && !SKM_HEAVY_ABL(4)) {  // the panel's share: row prow of G, scattered through the block's row list (ascending)
#pragma unroll
                for (int u = 0; u < GP; ++u)
                    if (gv[u])
                        acc_add(gj[u] - j0u, gv[u]);
                for (uint32_t q = s_jb[step] + (uint32_t)(tid + GP * TBH); q < s_jb[step + 1]; q += TBH) {  // more than 2048 rows of J in one step
                    const int g = grow[q];
                    if (g)
                        acc_add(gjl[q] - j0u, g);
                }
            }
            __syncthreads();
            if (j0 + CHH < m)
                g_prefetch(step + 1);  // in front of this step's stores
            // ---- scale, store, clear (the writer's store shape: 16-byte non-temporal stores, 1 KiB per wave instruction).
            // The neighbours' norms are only fetched for pieces that hold a dot product (even a heavy row is mostly zeros),
            // but of the 256 columns of a wave's piece some nearly always do: with the load beside the piece's store a
            // step cost NQ load latencies in series.  Two sweeps: the norms of all NQ pieces are requested first (the
            // tile is only looked at), then every piece is read again, cleared, scaled and stored.
            constexpr int NQ = CHH / 4 / TBH;
            float4 rjv[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int z = tid + q * TBH;
                const int64_t jc = j0 + 4 * (int64_t)z;
                rjv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                bool nz = false;
                if (jc < m) {
                    if (PACK) {
                        const int2 w = reinterpret_cast<const int2 *>(s_acc)[z];
                        nz = (w.x | w.y) != 0;
                    } else {
                        const int4 w = reinterpret_cast<const int4 *>(s_acc)[z];
                        nz = (w.x | w.y | w.z | w.w) != 0;
                    }
                }
                if (nz) {
                    if (VEC && jc + 3 < m) {
                        rjv[q] = *reinterpret_cast<const float4 *>(yrnorm + jc);
                    } else {
                        rjv[q].x = yrnorm[jc];
                        if (jc + 1 < m)
                            rjv[q].y = yrnorm[jc + 1];
                        if (jc + 2 < m)
                            rjv[q].z = yrnorm[jc + 2];
                        if (jc + 3 < m)
                            rjv[q].w = yrnorm[jc + 3];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int z = tid + q * TBH;
                const int64_t jc = j0 + 4 * (int64_t)z;
                if (jc >= m)
                    continue;
                int4 a;
                if (PACK) {
                    const int2 w = reinterpret_cast<int2 *>(s_acc)[z];
                    reinterpret_cast<int2 *>(s_acc)[z] = make_int2(0, 0);
                    a = make_int4(w.x & 0xFFFF, (int)((uint32_t)w.x >> 16), w.y & 0xFFFF, (int)((uint32_t)w.y >> 16));
                } else {
                    a = reinterpret_cast<int4 *>(s_acc)[z];
                    reinterpret_cast<int4 *>(s_acc)[z] = make_int4(0, 0, 0, 0);
                }
                float o[4] = {(float)a.x * ri * rjv[q].x, (float)a.y * ri * rjv[q].y, (float)a.z * ri * rjv[q].z, (float)a.w * ri * rjv[q].w};
                if (MODE == 1) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float d = fminf(fmaxf(1.0f - o[u], 0.0f), 2.0f);
                        o[u] = (jc + u == i) ? 0.0f : d;
                    }
                }
                float *dst = out + r * ld + jc;
                if (SKM_HEAVY_ABL(1))
                    continue;
                if (VEC && jc + 3 < m) {
                    const f32x4 pack = {o[0], o[1], o[2], o[3]};
                    __builtin_nontemporal_store(pack, reinterpret_cast<f32x4 *>(dst));
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (jc + u < m)
                            dst[u] = o[u];
                }
            }
            __syncthreads();
        }
        if (tid == 0)
            g_len[r] = G_DONE_ROW;
    }
}

// wave per row: k rounds of "largest entry below the previous pick" in (score desc, j asc) order
__global__ __launch_bounds__(256) void k_neighbors_topk(int64_t nrows, int64_t row0, const uint64_t *__restrict__ g_start,
                                                        const uint32_t *__restrict__ g_len,
                                                        const uint64_t *__restrict__ g_ent,
                                                        const float *__restrict__ xr, const float *__restrict__ yr, int k,
                                                        int exclude_self, uint32_t min_len, uint32_t *__restrict__ idx,
                                                        float *__restrict__ val)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < nrows; r += nwaves) {
        const uint32_t len = g_len[r] == G_OVERFLOW ? 0u : g_len[r];
        if (len < min_len && len != 0u)  // rows the LDS kernel takes (empty rows are written here)
            continue;
        const uint64_t st = g_start[r];
        const float ri = xr[row0 + r];
        const uint32_t self = (uint32_t)(row0 + r);
        float pv = INFINITY;  // previous pick
        uint32_t pj = 0;
        bool first = true;
        for (int t = 0; t < k; ++t) {
            float bv = -INFINITY;
            uint32_t bj = 0xFFFFFFFFu;
            for (uint32_t e = lane; e < len; e += 64) {
                const uint64_t w = g_ent[st + e];
                const uint32_t j = (uint32_t)(w >> 32);
                if (exclude_self && j == self)
                    continue;
                const float v = (float)(int)(uint32_t)w * ri * yr[j];
                const bool below = first || v < pv || (v == pv && j > pj);
                if (below && (v > bv || (v == bv && j < bj))) {
                    bv = v;
                    bj = j;
                }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o);
                const uint32_t oj = __shfl_xor(bj, o);
                if (ov > bv || (ov == bv && oj < bj)) {
                    bv = ov;
                    bj = oj;
                }
            }
            if (lane == 0) {
                idx[r * k + t] = bj;
                val[r * k + t] = bj == 0xFFFFFFFFu ? 0.0f : bv;
            }
            if (bj == 0xFFFFFFFFu) {
                for (int u = t + 1 + lane; u < k; u += 64) {
                    idx[r * k + u] = 0xFFFFFFFFu;
                    val[r * k + u] = 0.0f;
                }
                break;
            }
            pv = bv;
            pj = bj;
            first = false;
        }
    }
}

// k <= TOPK_KREG (the top-10 of BASELINE configs[3]): WAVE per row, ONE pass over the list whatever its length.  Every lane
// keeps the TOPK_KREG best of the entries it has seen, sorted, in registers (the list is read once, coalesced, four loads
// and four norm gathers in flight); the k results are then k rounds of a wave-wide arg-max over the lanes' current heads,
// the winner popping its own.  No LDS, no barrier, eight rows per SIMD in flight.  (Until round 6 every row went through
// the workgroup-per-row kernel below - 64 KiB of LDS, two rows per CU, two barriers and a full scan of the cached list per
// result: 14 ms for the 125 k rows x 3264 neighbours of one rank's share of BASELINE configs[3].)  Order and arithmetic are
// the other kernels': (score desc, j asc), score = (float)dot * xr[i] * yr[j].
constexpr int TOPK_KREG = 16;

// max_j yr[j] (non-negative floats: their bit patterns order like unsigned integers)
__global__ __launch_bounds__(256) void k_max_f32(int64_t m, const float *__restrict__ yr, uint32_t *__restrict__ out)
{
    float mx = 0.0f;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < m; j += stride)
        mx = fmaxf(mx, yr[j]);
    for (int o = 32; o > 0; o >>= 1)
        mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0)
        atomicMax(out, __float_as_uint(mx));
}

template <int KR>
__global__ __launch_bounds__(256) void k_neighbors_topk_stream(int64_t nrows, int64_t row0, const uint64_t *__restrict__ g_start,
                                                               const uint32_t *__restrict__ g_len,
                                                               const uint64_t *__restrict__ g_ent,
                                                               const float *__restrict__ xr, const float *__restrict__ yr,
                                                               const float *__restrict__ yr_max, int k, int exclude_self,
                                                               uint32_t *__restrict__ idx, float *__restrict__ val)
{
    constexpr int UNR = 4;
    const float rmax = *yr_max;  // max_j yr[j]: dot * xr[i] * rmax bounds an entry's score before its norm is gathered
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < nrows; r += nwaves) {
        const uint32_t raw = g_len[r];
        const uint32_t len = raw == G_OVERFLOW ? 0u : raw;  // (a row the list kernels could not hold: no neighbours reported)
        const uint64_t st = g_start[r];
        const float ri = xr[row0 + r];
        const uint32_t self = (uint32_t)(row0 + r);
        float tv[KR];
        uint32_t tj[KR];
#pragma unroll
        for (int q = 0; q < KR; ++q) {
            tv[q] = -INFINITY;
            tj[q] = 0xFFFFFFFFu;
        }
        // thr: the smallest of the 64 lanes' best scores so far.  64 >= k entries are at least that good, so an entry whose
        // UPPER BOUND dot * xr[i] * max_j yr[j] is below it is in nobody's top k and its neighbour's norm is not even
        // gathered: the kernel is bound by those random 4-byte gathers (one per entry: 4.5 ms for 4.1e8 entries), and
        // behind the first few hundred entries of a row nearly every chance neighbour is dropped by this compare
        float thr = -INFINITY;
        for (uint32_t e0 = 0; e0 < len; e0 += 64 * UNR) {  // wave-uniform trip count
            uint64_t w[UNR];
            float rj[UNR];
            bool need[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const uint32_t e = e0 + (uint32_t)(u * 64 + lane);
                w[u] = __builtin_nontemporal_load(&g_ent[st + (e < len ? e : 0u)]);  // (streamed once)
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const uint32_t e = e0 + (uint32_t)(u * 64 + lane);
                const uint32_t j = (uint32_t)(w[u] >> 32);
                need[u] = e < len && !(exclude_self && j == self) && (float)(int)(uint32_t)w[u] * ri * rmax >= thr;
                rj[u] = yr[need[u] ? j : 0u];  // (no branch around the load: the four gathers stay in flight together; a
                                               // dropped entry reads norm 0, a line every lane shares)
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (!need[u])
                    continue;
                const uint32_t j = (uint32_t)(w[u] >> 32);
                const float v = (float)(int)(uint32_t)w[u] * ri * rj[u];
                if (v < thr || !(v > tv[KR - 1] || (v == tv[KR - 1] && j < tj[KR - 1])))
                    continue;
                // sorted insertion: every slot takes the better of (its left neighbour, the newcomer) once the newcomer
                // beats the slot itself
                float cv = v;
                uint32_t cj = j;
#pragma unroll
                for (int q = 0; q < KR; ++q) {
                    const bool ahead = cv > tv[q] || (cv == tv[q] && cj < tj[q]);
                    if (ahead) {
                        const float ov = tv[q];
                        const uint32_t oj = tj[q];
                        tv[q] = cv;
                        tj[q] = cj;
                        cv = ov;
                        cj = oj;
                    }
                }
            }
            thr = tv[0];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
                thr = fminf(thr, __shfl_xor(thr, o));
        }
        for (int t = 0; t < k; ++t) {
            float rv = tv[0];
            uint32_t rj2 = tj[0];
            int rl = lane;
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(rv, o);
                const uint32_t oj = __shfl_xor(rj2, o);
                const int ol = __shfl_xor(rl, o);
                if (ov > rv || (ov == rv && oj < rj2)) {
                    rv = ov;
                    rj2 = oj;
                    rl = ol;
                }
            }
            const bool none = rj2 == 0xFFFFFFFFu;  // fewer than k neighbours: the rest stays empty
            if (lane == 0) {
                idx[r * k + t] = rj2;
                val[r * k + t] = none ? 0.0f : rv;
            }
            if (none) {
                for (int u = t + 1 + lane; u < k; u += 64) {
                    idx[r * k + u] = 0xFFFFFFFFu;
                    val[r * k + u] = 0.0f;
                }
                break;
            }
            if (lane == rl) {  // the winner pops its head
#pragma unroll
                for (int q = 0; q + 1 < KR; ++q) {
                    tv[q] = tv[q + 1];
                    tj[q] = tj[q + 1];
                }
                tv[KR - 1] = -INFINITY;
                tj[KR - 1] = 0xFFFFFFFFu;
            }
        }
    }
}

// k > TOPK_KREG.  Rows whose list fits LDS (fewer than TOPK_CAP entries): workgroup per row.  Scores are
// computed once (one gather of the neighbour's norm per entry) and cached with their row numbers;
// each of the k rounds is then an arg-max over LDS in (score desc, j asc) order that retires its
// pick.  The wave-per-row kernel above re-reads the list and the norms in every round.
constexpr int TOPK_CAP = 8192;

__global__ __launch_bounds__(256) void k_neighbors_topk_lds(int64_t nrows, int64_t row0, const uint64_t *__restrict__ g_start,
                                                            const uint32_t *__restrict__ g_len,
                                                            const uint64_t *__restrict__ g_ent,
                                                            const float *__restrict__ xr, const float *__restrict__ yr,
                                                            int k, int exclude_self, uint32_t min_len,
                                                            uint32_t *__restrict__ idx, float *__restrict__ val)
{
    __shared__ float s_v[TOPK_CAP];
    __shared__ uint32_t s_j[TOPK_CAP];
    __shared__ float s_bv[4];
    __shared__ uint32_t s_bj[4], s_bp[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int64_t r = blockIdx.x; r < nrows; r += gridDim.x) {
        const uint32_t len = g_len[r];
        if (len == G_OVERFLOW || len < min_len || len >= (uint32_t)TOPK_CAP)
            continue;  // uniform for the workgroup
        const uint64_t st = g_start[r];
        const float ri = xr[row0 + r];
        const uint32_t self = (uint32_t)(row0 + r);
        for (uint32_t e = (uint32_t)tid; e < len; e += 256) {
            const uint64_t w = g_ent[st + e];
            const uint32_t j = (uint32_t)(w >> 32);
            s_j[e] = j;
            s_v[e] = (exclude_self && j == self) ? -INFINITY : (float)(int)(uint32_t)w * ri * yr[j];
        }
        __syncthreads();
        for (int t = 0; t < k; ++t) {
            float bv = -INFINITY;
            uint32_t bj = 0xFFFFFFFFu, bp = 0;
            for (uint32_t e = (uint32_t)tid; e < len; e += 256) {
                const float v = s_v[e];
                const uint32_t j = s_j[e];
                if (v > bv || (v == bv && j < bj && v != -INFINITY)) {
                    bv = v;
                    bj = j;
                    bp = e;
                }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o);
                const uint32_t oj = __shfl_xor(bj, o), op = __shfl_xor(bp, o);
                if (ov > bv || (ov == bv && oj < bj)) {
                    bv = ov;
                    bj = oj;
                    bp = op;
                }
            }
            if (lane == 0) {
                s_bv[wid] = bv;
                s_bj[wid] = bj;
                s_bp[wid] = bp;
            }
            __syncthreads();
            bv = s_bv[0];
            bj = s_bj[0];
            bp = s_bp[0];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                if (s_bv[w] > bv || (s_bv[w] == bv && s_bj[w] < bj)) {
                    bv = s_bv[w];
                    bj = s_bj[w];
                    bp = s_bp[w];
                }
            }
            const bool none = bj == 0xFFFFFFFFu;  // fewer than k neighbours: the rest stays empty
            if (tid == 0) {
                idx[r * k + t] = bj;
                val[r * k + t] = none ? 0.0f : bv;
                if (!none)
                    s_v[bp] = -INFINITY;  // retired
            }
            if (none) {
                for (int u = t + 1 + tid; u < k; u += 256) {
                    idx[r * k + u] = 0xFFFFFFFFu;
                    val[r * k + u] = 0.0f;
                }
            }
            __syncthreads();
            if (none)
                break;
        }
        __syncthreads();
    }
}

// Last pass of skm_gram_neighbors: rows with more neighbours (or distinct k-mers) than the LDS
// tables hold.  One row at a time per workgroup, hash table (HS slots) in a per-workgroup slice of
// global scratch; a wave walks one posting list at a time.  Few rows take this path (3 % at
// N = 1 M), so it is written for capacity, not speed.  Entries are NOT grouped by chunk.
constexpr int HS_BITS = 17;
constexpr int HS = 1 << HS_BITS;  // 131072 slots -> up to 65536 neighbours per row

template <typename PW>
__global__ __launch_bounds__(1024) void k_gram_sparse_huge(const int64_t *__restrict__ xrowptr,
                                                           const uint32_t *__restrict__ xcolidx,
                                                           const uint32_t *__restrict__ xcounts,
                                                           const uint32_t *__restrict__ ycolptr,
                                                           const PW *__restrict__ ypost,
                                                           const uint32_t *__restrict__ ypostcnt, int64_t row0,
                                                           uint64_t *__restrict__ g_ent, unsigned long long cap_ent,
                                                           unsigned long long *__restrict__ g_counter,
                                                           uint64_t *__restrict__ g_start, uint32_t *__restrict__ g_len,
                                                           const uint32_t *__restrict__ row_list,
                                                           const uint32_t *__restrict__ row_count,
                                                           uint32_t *__restrict__ scratch)
{
    __shared__ unsigned int s_distinct, s_fill;
    __shared__ unsigned long long s_off;
    __shared__ int s_over;
    uint32_t *hkeys = scratch + (size_t)blockIdx.x * 2 * HS;
    int *hvals = reinterpret_cast<int *>(hkeys + HS);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
    const uint32_t cnt = *row_count;
    for (uint32_t idx = blockIdx.x; idx < cnt; idx += gridDim.x) {
        const int64_t orow = row_list[idx];
        const int64_t i = row0 + orow;
        for (int z = tid; z < 2 * HS; z += blockDim.x)
            hkeys[z] = 0u;  // keys and values are adjacent
        if (tid == 0) {
            s_distinct = 0;
            s_fill = 0;
            s_over = 0;
        }
        __syncthreads();
        const int64_t b = xrowptr[i], e = xrowptr[i + 1];
        for (int64_t t = b + wid; t < e; t += nw) {
            const uint32_t c = xcolidx[t];
            const int v = (int)xcounts[t];
            uint32_t pb = 0, pe = 1;
            const bool single = c == 0xFFFFFFFFu;  // k-mer of this row only
            if (!single) {
                pb = ycolptr[c];
                pe = ycolptr[c + 1];
            }
            for (uint32_t p = pb + lane; p < pe; p += 64) {
                uint32_t j;
                int prod;
                if (single) {
                    j = (uint32_t)i;
                    prod = v * v;
                } else {
                    const PW pw = ypost[p];
                    j = posting<PW>::row(pw);
                    prod = v * (int)posting<PW>::count(pw, ypostcnt, p);
                }
                const uint32_t key = j + 1u;
                uint32_t h = (j * 2654435761u) >> (32 - HS_BITS);
                for (int probe = 0; probe < HS; ++probe) {
                    uint32_t seen = __hip_atomic_load(&hkeys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (seen == 0u) {
                        seen = atomicCAS(&hkeys[h], 0u, key);
                        if (seen == 0u) {
                            seen = key;
                            if (atomicAdd(&s_distinct, 1u) >= (unsigned)(HS / 2))
                                s_over = 1;
                        }
                    }
                    if (seen == key) {
                        atomicAdd(&hvals[h], prod);
                        break;
                    }
                    h = (h + 1) & (HS - 1);
                }
            }
            if (s_over)
                break;
        }
        __threadfence_block();
        __syncthreads();
        if (tid == 0) {
            unsigned long long off = 0;
            if (!s_over) {
                off = atomicAdd(g_counter, (unsigned long long)s_distinct);
                if (off + s_distinct > cap_ent)
                    s_over = 1;
            }
            s_off = off;
        }
        __syncthreads();
        if (!s_over) {
            for (int z = tid; z < HS; z += blockDim.x) {
                const uint32_t key = hkeys[z];
                if (key) {
                    const unsigned int pos = atomicAdd(&s_fill, 1u);
                    g_ent[s_off + pos] = ((uint64_t)(key - 1u) << 32) | (uint32_t)hvals[z];
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            g_start[orow] = s_off;
            g_len[orow] = s_over ? G_OVERFLOW : s_distinct;
        }
        __syncthreads();
    }
}

__global__ void k_count_overflow(int64_t nrows, const uint32_t *__restrict__ g_len, unsigned int *out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ov = i < nrows && g_len[i] == G_OVERFLOW;
    const unsigned long long bal = __ballot(ov);
    if ((threadIdx.x & 63) == 0 && bal)
        atomicAdd(out, (unsigned int)__popcll(bal));
}

}  // namespace

namespace {
template <typename PW>
int gram_neighbors_impl(skm_ctx *ctx, int64_t n, const int64_t *d_xrowptr, const uint32_t *d_xcolidx,
                        const uint32_t *d_xcounts, int64_t m, int64_t ncols, const uint32_t *d_ycolptr, const PW *d_ypost,
                        const uint32_t *d_ypostcnt, const float *d_xrnorm, const float *d_yrnorm, int64_t row0, int64_t row1,
                        int64_t cap_ent, uint64_t *d_start, uint32_t *d_len, uint64_t *d_ent, int64_t *h_total_entries,
                        int64_t *h_overflow_rows)
{
    SKM_REQUIRE(ctx && n >= 0 && m >= 0 && ncols >= 0 && cap_ent >= 0 && h_total_entries && h_overflow_rows, SKM_E_BADARG,
                "skm_gram_neighbors: bad argument");
    SKM_REQUIRE(row0 >= 0 && row0 <= row1 && row1 <= n, SKM_E_BADARG, "skm_gram_neighbors: bad row range");
    SKM_REQUIRE(m < ((int64_t)1 << 32) - 1, SKM_E_OVERFLOW, "skm_gram_neighbors: m >= 2^32");
    *h_total_entries = 0;
    *h_overflow_rows = 0;
    const int64_t nrows = row1 - row0;
    if (nrows == 0)
        return SKM_OK;
    SKM_REQUIRE(d_xrowptr && d_ycolptr && d_start && d_len && d_xrnorm && d_yrnorm && (cap_ent == 0 || d_ent), SKM_E_BADARG,
                "skm_gram_neighbors: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    void *p;
    SKM_TRY(skm_ws(ctx, WS_E, sizeof(uint32_t) * (size_t)(nrows + 8), &p));
    uint32_t *list1 = (uint32_t *)p;
    SKM_TRY(skm_ws(ctx, WS_F, sizeof(uint32_t) * (size_t)(nrows + 8), &p));
    uint32_t *list2 = (uint32_t *)p;
    SKM_TRY(skm_ws(ctx, WS_COS, sizeof(cos_state), &p));
    cos_state *state = (cos_state *)p;
    unsigned long long *g_counter = &state->g_counter;
    uint32_t *cnt1 = &state->over_count[0], *cnt2 = cnt1 + 1, *novf = cnt1 + 2;
    SKM_TRY(cosine_prologue(ctx, nullptr, 0, state, 0ull, d_yrnorm, m, st));
    const unsigned long long cap = (unsigned long long)cap_ent;
    {
        // rows whose dot products may not fit the 32-bit entries (skm_row_is_wide) are marked G_OVERFLOW at once.
        // (Round 6 measured 4096-slot tables for this pass against >= 2^19 rows - BASELINE configs[3], 3264 neighbours per
        // row -: first pass 10.8 -> 16.0 ms, second 25.5 -> 19.2, i.e. 38.7 -> 37.3 ms for a rank's 125 k rows: not kept.)
        SKM_PROF(ctx, "k_gram_sparse");
        k_gram_sparse<0, 1, 2048, 256, 2, 32, 2, PW><<<(unsigned)nrows, 256, 0, st>>>(
            d_xrowptr, d_xcolidx, d_xcounts, d_ycolptr, d_ypost, d_ypostcnt, row0, row1, 0ull, 0, d_ent, cap, g_counter, d_start,
            d_len, list1, cnt1, d_xrnorm, &state->min_yrnorm, G_OVERFLOW);
    }
    SKM_TRY(skm_check_launch("k_gram_sparse"));
    {
        // Rows the 2048-slot tables cannot hold, one row per workgroup at a time, smallest shape that
        // fits first (the lists ping-pong): 8192 slots with 256 threads and 512 non-zeros (74 KB of LDS:
        // two workgroups per CU), then 16384 slots / 1024 non-zeros, then 8192 slots / 4096 non-zeros.
        SKM_PROF(ctx, "k_gram_sparse_big");
        k_gram_sparse_big<8192, 256, 2, 16, 4, PW><<<skm_grid_cap(ctx, nrows, 2), 256, 0, st>>>(
            d_xrowptr, d_xcolidx, d_xcounts, d_ycolptr, d_ypost, d_ypostcnt, row0, row1, d_ent, cap, g_counter, d_start, d_len,
            list1, cnt1, list2, cnt2);
        SKM_HIP(hipMemsetAsync(cnt1, 0, 4, st));
        k_gram_sparse_big<16384, 512, 2, 16, 4, PW><<<skm_grid_cap(ctx, nrows, 1), 512, 0, st>>>(
            d_xrowptr, d_xcolidx, d_xcounts, d_ycolptr, d_ypost, d_ypostcnt, row0, row1, d_ent, cap, g_counter, d_start, d_len,
            list2, cnt2, list1, cnt1);
        SKM_HIP(hipMemsetAsync(cnt2, 0, 4, st));
        k_gram_sparse_big<8192, 512, 8, 16, 4, PW><<<skm_grid_cap(ctx, nrows, 1), 512, 0, st>>>(
            d_xrowptr, d_xcolidx, d_xcounts, d_ycolptr, d_ypost, d_ypostcnt, row0, row1, d_ent, cap, g_counter, d_start, d_len,
            list1, cnt1, list2, cnt2);
    }
    SKM_TRY(skm_check_launch("k_gram_sparse_big"));
    {
        // what is left: table of 131072 slots per workgroup in global scratch
        const int huge_grid = ctx->num_cus > 0 ? ctx->num_cus : 256;
        SKM_TRY(skm_ws(ctx, WS_G, sizeof(uint32_t) * 2 * (size_t)HS * (size_t)huge_grid, &p));
        SKM_PROF(ctx, "k_gram_sparse_huge");
        k_gram_sparse_huge<PW><<<huge_grid, 1024, 0, st>>>(d_xrowptr, d_xcolidx, d_xcounts, d_ycolptr, d_ypost, d_ypostcnt, row0,
                                                           d_ent, cap, g_counter, d_start, d_len, list2, cnt2, (uint32_t *)p);
    }
    SKM_TRY(skm_check_launch("k_gram_sparse_huge"));
    k_count_overflow<<<(unsigned)skm_ceil_div(nrows, 256), 256, 0, st>>>(nrows, d_len, novf);
    SKM_TRY(skm_check_launch("k_count_overflow"));
    unsigned long long *h = (unsigned long long *)ctx->h_pinned;
    SKM_HIP(hipMemcpyAsync(h, g_counter, 8, hipMemcpyDeviceToHost, st));
    SKM_HIP(hipMemcpyAsync(h + 1, novf, 4, hipMemcpyDeviceToHost, st));
    SKM_HIP(hipStreamSynchronize(st));
    *h_total_entries = (int64_t)(h[0] < cap ? h[0] : cap);
    *h_overflow_rows = (int64_t)*(uint32_t *)(h + 1);
    return SKM_OK;
}
}  // namespace

extern "C" int skm_gram_neighbors(skm_ctx *ctx, int64_t n, const int64_t *d_xrowptr, const uint32_t *d_xcolidx,
                                  const uint32_t *d_xcounts, int64_t m, int64_t ncols, const uint32_t *d_ycolptr,
                                  const void *d_ypost, int post_bits, const uint32_t *d_ypostcnt, const float *d_xrnorm,
                                  const float *d_yrnorm, int64_t row0, int64_t row1, int64_t cap_ent, uint64_t *d_start,
                                  uint32_t *d_len, uint64_t *d_ent, int64_t *h_total_entries, int64_t *h_overflow_rows)
{
    SKM_REQUIRE(post_bits == 64 || post_bits == 32, SKM_E_BADARG, "skm_gram_neighbors: post_bits must be 32 or 64");
    if (post_bits == 32) {
        SKM_REQUIRE(m <= ((int64_t)1 << 24), SKM_E_BADARG, "skm_gram_neighbors: 32-bit postings hold rows < 2^24");
        return gram_neighbors_impl<uint32_t>(ctx, n, d_xrowptr, d_xcolidx, d_xcounts, m, ncols, d_ycolptr, (const uint32_t *)d_ypost,
                                             d_ypostcnt, d_xrnorm, d_yrnorm, row0, row1, cap_ent, d_start, d_len, d_ent,
                                             h_total_entries, h_overflow_rows);
    }
    return gram_neighbors_impl<uint64_t>(ctx, n, d_xrowptr, d_xcolidx, d_xcounts, m, ncols, d_ycolptr, (const uint64_t *)d_ypost,
                                         nullptr, d_xrnorm, d_yrnorm, row0, row1, cap_ent, d_start, d_len, d_ent,
                                         h_total_entries, h_overflow_rows);
}

extern "C" int skm_neighbors_topk(skm_ctx *ctx, int64_t nrows, int64_t row0, const uint64_t *d_start, const uint32_t *d_len,
                                  const uint64_t *d_ent, const float *d_xrnorm, const float *d_yrnorm, int64_t m, int k,
                                  int exclude_self, uint32_t *d_idx, float *d_val)
{
    SKM_REQUIRE(ctx && nrows >= 0 && row0 >= 0 && k >= 1 && k <= 1024, SKM_E_BADARG, "skm_neighbors_topk: bad argument");
    if (nrows == 0)
        return SKM_OK;
    SKM_REQUIRE(d_start && d_len && d_xrnorm && d_yrnorm && d_idx && d_val, SKM_E_BADARG, "skm_neighbors_topk: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    if (k <= TOPK_KREG) {  // one pass per row, the lanes' best in registers
        // max_j yrnorm[j] for the bound that spares the gather of hopeless entries' norms (m <= 0: the caller did not say
        // how many norms there are: +infinity, every norm is gathered)
        void *p;
        SKM_TRY(skm_ws(ctx, WS_SMALL, 4096, &p));
        uint32_t *rmax = (uint32_t *)((uint8_t *)p + 3072);
        if (m > 0) {
            SKM_HIP(hipMemsetAsync(rmax, 0, 4, ctx->stream));
            k_max_f32<<<skm_grid_cap(ctx, skm_ceil_div(m, 256 * 8), 4), 256, 0, ctx->stream>>>(m, d_yrnorm, rmax);
            SKM_TRY(skm_check_launch("k_max_f32"));
        } else {
            const uint32_t inf_bits = 0x7F800000u;
            SKM_HIP(hipMemcpyAsync(rmax, &inf_bits, 4, hipMemcpyHostToDevice, ctx->stream));
            SKM_HIP(hipStreamSynchronize(ctx->stream));  // (the source is on this function's stack)
        }
        SKM_PROF(ctx, "k_neighbors_topk_stream");
        const int grid = skm_grid_cap(ctx, skm_ceil_div(nrows, 4), 32);
#define SKM_TOPK_STREAM(KR)                                                                                                   \
    k_neighbors_topk_stream<KR><<<grid, 256, 0, ctx->stream>>>(nrows, row0, d_start, d_len, d_ent, d_xrnorm, d_yrnorm,        \
                                                               (const float *)rmax, k, exclude_self, d_idx, d_val)
        if (k <= 4)  // (the lanes keep KR >= k candidates each: fewer registers and a shorter insertion for a small k)
            SKM_TOPK_STREAM(4);
        else if (k <= 8)
            SKM_TOPK_STREAM(8);
        else if (k <= 12)
            SKM_TOPK_STREAM(12);
        else
            SKM_TOPK_STREAM(16);
#undef SKM_TOPK_STREAM
        return skm_check_launch("k_neighbors_topk_stream");
    }
    {
        // lists that fit LDS: scores cached once
        SKM_PROF(ctx, "k_neighbors_topk_lds");
        k_neighbors_topk_lds<<<skm_grid_cap(ctx, nrows, 8), 256, 0, ctx->stream>>>(nrows, row0, d_start, d_len, d_ent, d_xrnorm,
                                                                                  d_yrnorm, k, exclude_self, 1u, d_idx, d_val);
    }
    SKM_TRY(skm_check_launch("k_neighbors_topk_lds"));
    SKM_PROF(ctx, "k_neighbors_topk");  // longer lists, empty and flagged rows
    k_neighbors_topk<<<skm_grid_cap(ctx, skm_ceil_div(nrows, 4), 16), 256, 0, ctx->stream>>>(
        nrows, row0, d_start, d_len, d_ent, d_xrnorm, d_yrnorm, k, exclude_self, (uint32_t)TOPK_CAP, d_idx, d_val);
    return skm_check_launch("k_neighbors_topk");
}

namespace {
// The two streams of the blocked schedule and the events that chain them.  The Gram stream is confined to half of the
// compute units (mask bit i: CU group (i / 8) % 8; groups 0-3); the writer's stream is not confined: it needs every
// CU's wave slots to keep the stores flowing, and what it must be protected from is the Gram's workgroups taking LDS
// and wave slots on ALL of them.  Measured at config 3, one batch per step: back to back 10.90 ms, Gram on groups
// 0-2 / writer on 3-7 (round 2's split) 10.48, Gram on 0-2 / writer anywhere 10.35, Gram on 0-3 / writer anywhere
// 10.24-10.33, Gram on 0-4 11.7.
int overlap_streams(skm_ctx *ctx)
{
    if (ctx->overlap_state != 0)
        return ctx->overlap_state > 0 ? SKM_OK : SKM_E_UNSUPPORTED;
    ctx->overlap_state = -1;
    const int ncu = ctx->num_cus;
    if (ncu < 64 || ncu > 512)
        return SKM_E_UNSUPPORTED;
    uint32_t mg[16] = {};
    for (int i = 0; i < ncu; ++i)
        if ((i / 8) % 8 < 4)
            mg[i / 32] |= 1u << (i % 32);
    const uint32_t words = (uint32_t)((ncu + 31) / 32);
    // (streams and events come from the device's caches and go back there with the context: skm_mem.hip)
    if (skm_stream_acquire(ctx->device, -1, -1, nullptr, 0, &ctx->s_writer) != hipSuccess ||
        skm_stream_acquire(ctx->device, 0, 3, mg, words, &ctx->s_gram) != hipSuccess) {
        (void)hipGetLastError();
        return SKM_E_UNSUPPORTED;
    }
    for (int i = 0; i < 16 + 3; ++i) {
        hipEvent_t e = skm_event_acquire(ctx->device, false);
        if (!e)
            return SKM_E_UNSUPPORTED;
        ctx->sync_events.push_back(e);
    }
    ctx->overlap_state = 1;
    return SKM_OK;
}
}  // namespace

namespace {
// The panel pipeline is a dozen launches: worth it when thousands of rows are heavy, pure overhead on a batch of small
// families (a few hundred heavy rows).  The number of heavy rows is only known on the device, so the decision uses the
// count the PREVIOUS list-path call on this context left in pinned memory (copied asynchronously, never waited for):
// streams of similar batches adapt after one step, and the choice only moves work between two exact kernels.
// SKM_HEAVY_PANEL=1 / 0 force it on / off (tests, A/B timing).
bool heavy_panels_wanted(skm_ctx *ctx, int64_t nrows, int64_t m)
{
    if (m > ((int64_t)1 << 20) || nrows < PB_ROWS)
        return false;
    if (skm_opts().heavy_panel >= 0)
        return skm_opts().heavy_panel != 0;
    // (one counter per row block of the blocked schedule, 16 words: the whole-call form uses the first only)
    const volatile uint32_t *last = (const volatile uint32_t *)((uint8_t *)ctx->h_pinned + 2048);
    uint64_t heavy = 0;
    for (int b = 0; b < 16; ++b)
        heavy += last[b];
    return heavy >= 4096u;
}

// The packed form of k_cosine_heavy (two columns per accumulator word, the row's postings cached in LDS) is one or two more
// launches in front of the unpacked one: taken when the previous call handed on a thousand rows or more (same stale
// hint), or as SKM_HEAVY_PACK=1 / 0 says.
bool heavy_pack_wanted(skm_ctx *ctx)
{
    if (skm_opts().heavy_pack >= 0)
        return skm_opts().heavy_pack != 0;
    const volatile uint32_t *last = (const volatile uint32_t *)((uint8_t *)ctx->h_pinned + 2048);
    uint64_t heavy = 0;
    for (int b = 0; b < 16; ++b)
        heavy += last[b];
    return heavy >= 1024u;
}

template <typename PW>
int heavy_panels_run(skm_ctx *ctx, const int64_t *d_xrowptr, const uint32_t *d_xcolidx, const uint32_t *d_xcounts,
                     const uint32_t *d_ycolptr, const PW *d_ypost, const uint32_t *d_ypostcnt, int64_t m, int64_t row0,
                     int64_t rbase, int64_t nrows, const uint32_t *row_list, const uint32_t *row_count, panel_bufs *out,
                     hipStream_t st)
{
    panel_bufs pb = {};
    // Blocks: sized from the number of rows the PREVIOUS call on this context handed on (the same stale hint that switched
    // the panels on: pinned memory, never waited for) plus a quarter, not from the call's row count - G alone is 16 MiB per
    // block, and a 100 k-row call whose 4 096 heavy rows need 16 blocks reserved 391 of them (6.1 GiB per context, three
    // contexts in engine.OverlappedPipeline).  Heavy rows beyond the blocks are walked in full: exact either way.
    int64_t want_rows = nrows;
    {
        const volatile uint32_t *last = (const volatile uint32_t *)((uint8_t *)ctx->h_pinned + 2048);
        uint64_t heavy = 0;
        for (int b = 0; b < 16; ++b)
            heavy += last[b];
        if (heavy >= 4096u)  // (forced on without a hint - SKM_HEAVY_PANEL=1 on a first call -: the call's row count)
            want_rows = std::min<int64_t>(nrows, (int64_t)(heavy + heavy / 4 + PB_ROWS));
    }
    pb.nb = (int)std::min<int64_t>(skm_ceil_div(want_rows, PB_ROWS), PB_MAXBLOCKS);
    pb.mwords = (uint32_t)((m + 31) / 32);
    const size_t hcap = (size_t)nrows + 8, nb = (size_t)pb.nb;
    // one scratch slot, carved up (every piece 256-byte aligned)
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    };
    const size_t o_key = take(4 * hcap), o_skey = take(4 * hcap), o_perm = take(4 * hcap), o_ktmp = take(4 * hcap),
                 o_vtmp = take(4 * hcap), o_cnt = take(8), o_dk = take(8 * nb * PB_DICT),
                 o_cols = take(4 * nb * PB_KMAX), o_bad = take(nb * PB_KMAX), o_meta = take(16 * nb),
                 o_jl = take(4 * nb * PB_JMAX), o_jb = take(4 * nb * (PB_STEPS + 1)), o_A = take(nb * PB_ROWS * PB_KMAX),
                 o_G = take(4 * nb * (size_t)PB_ROWS * PB_JMAX),
                 o_state = take(skm_onesweep::state_bytes((int64_t)hcap, 8192, 4) + skm_onesweep::state_bytes((int64_t)hcap, 2048, 4));
    void *p;
    {
        const int rc_ws = skm_ws(ctx, WS_G, off, &p);
        if (rc_ws == SKM_E_NOMEM) {  // the panels are an optimisation: without room for them every heavy row is walked
            *out = panel_bufs{};
            return SKM_E_NOMEM;
        }
        SKM_TRY(rc_ws);
    }
    uint8_t *base = (uint8_t *)p;
    pb.key = (uint32_t *)(base + o_key);
    pb.skey = (uint32_t *)(base + o_skey);
    pb.perm = (uint32_t *)(base + o_perm);
    pb.count64 = (int64_t *)(base + o_cnt);
    pb.dict = (uint2 *)(base + o_dk);
    pb.cols = (uint32_t *)(base + o_cols);
    pb.bad = base + o_bad;
    pb.meta = (uint32_t *)(base + o_meta);
    pb.jlist = (uint32_t *)(base + o_jl);
    pb.jbound = (uint32_t *)(base + o_jb);
    pb.A = (int8_t *)(base + o_A);
    pb.G = (int *)(base + o_G);
    hipStream_t saved = ctx->stream;
    ctx->stream = st;  // the sort and the profiling scopes follow the context's stream
    int rc = SKM_OK;
    do {
        {
            SKM_PROF(ctx, "k_panel_key");
            k_panel_key<PW><<<skm_grid_cap(ctx, skm_ceil_div(nrows, 4), 4), 256, 0, st>>>(d_xrowptr, d_xcolidx, d_ycolptr, d_ypost, row0, rbase, row_list,
                                                                                   row_count, pb);
        }
        if ((rc = skm_check_launch("k_panel_key")) != SKM_OK)
            break;
        if ((rc = skm_onesweep::sort_pairs_dev<uint32_t>(ctx, pb.count64, (int64_t)hcap, pb.key, pb.skey, pb.perm,
                                                         (uint32_t *)(base + o_ktmp), (uint32_t *)(base + o_vtmp), base + o_state, 32,
                                                         "onesweep_sort_heavy_rows")) != SKM_OK)
            break;
        {
            SKM_PROF(ctx, "k_panel_dict");
            k_panel_dict<<<pb.nb, 1024, 0, st>>>(d_xrowptr, d_xcolidx, d_xcounts, d_ycolptr, row0, rbase, row_list, row_count, pb);
        }
        {
            SKM_PROF(ctx, "k_panel_rows");
            auto kern = k_panel_rows<PW>;
            const size_t lds = sizeof(uint32_t) * (size_t)pb.mwords;
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) {
                skm_set_error("k_panel_rows: %s", hipGetErrorString(e));
                rc = SKM_E_HIP;
                break;
            }
            kern<<<pb.nb, 1024, lds, st>>>(d_ycolptr, d_ypost, d_ypostcnt, row_count, pb);
        }
        {
            SKM_PROF(ctx, "k_panel_gemm");
            auto kern = k_panel_gemm<PW>;
            constexpr size_t lds = (size_t)128 * PG_KP;
            hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) {
                skm_set_error("k_panel_gemm: %s", hipGetErrorString(e));
                rc = SKM_E_HIP;
                break;
            }
            kern<<<dim3(PG_CHUNKS, pb.nb), PG_TB, lds, st>>>(d_ycolptr, d_ypost, d_ypostcnt, row_count, pb);
        }
        rc = skm_check_launch("k_panel_gemm");
    } while (0);
    ctx->stream = saved;
    ctx->panel_meta = pb.meta;
    ctx->panel_rows = row_count;
    ctx->panel_nb = pb.nb;
    *out = pb;
    return rc;
}
}  // namespace

namespace {
// phase 0: the whole call.  Phases 1 and 2 cut the list path in two (skm_cosine_csr_phase): 1 = prologue + sparse Gram
// (the neighbour lists stay in the context's scratch), 2 = everything behind them (heavy rows, writer, cursor kernel).
// The other routes (cursor kernel only) have no lists: phase 1 does nothing and phase 2 is the whole call.
template <typename PW>
int cosine_csr_impl(skm_ctx *ctx, int64_t n, const int64_t *d_xrowptr, const uint32_t *d_xcolidx, const uint32_t *d_xcounts,
                    const float *d_xrnorm, int64_t m, int64_t ncols, const uint32_t *d_ycolptr, const PW *d_ypost,
                    const uint32_t *d_ypostcnt, const float *d_yrnorm, int64_t row0, int64_t row1, int mode, float *d_out,
                    int64_t ld, int phase = 0)
{
    SKM_REQUIRE(ctx && n >= 0 && m >= 0 && ncols >= 0, SKM_E_BADARG, "skm_cosine_csr: bad argument");
    SKM_REQUIRE(row0 >= 0 && row0 <= row1 && row1 <= n, SKM_E_BADARG, "skm_cosine_csr: bad row range [%lld,%lld) of %lld",
                (long long)row0, (long long)row1, (long long)n);
    SKM_REQUIRE(ld >= m, SKM_E_BADARG, "skm_cosine_csr: ld (%lld) < m (%lld)", (long long)ld, (long long)m);
    SKM_REQUIRE(mode == 0 || mode == 1, SKM_E_BADARG, "skm_cosine_csr: mode must be 0 or 1");
    SKM_REQUIRE(m < ((int64_t)1 << 32) - 1, SKM_E_OVERFLOW, "skm_cosine_csr: m >= 2^32");
    if (row1 == row0 || m == 0)
        return SKM_OK;
    SKM_REQUIRE(d_xrowptr && d_xrnorm && d_ycolptr && d_yrnorm && (d_out || phase == 1), SKM_E_BADARG, "skm_cosine_csr: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    const int64_t nrows = row1 - row0;
    const int64_t strips = skm_ceil_div(nrows, R);
    SKM_REQUIRE(strips < ((int64_t)1 << 31), SKM_E_OVERFLOW, "skm_cosine_csr: too many rows in one call");
    const bool vec = (ld % 4 == 0) && (((uintptr_t)d_out & 15) == 0) && (((uintptr_t)d_yrnorm & 15) == 0);
    hipStream_t st = ctx->stream;

// expands CALL(MODE, VEC) for the runtime (mode, vec) pair
#define SKM_BY_MODE_VEC(CALL)  \
    do {                       \
        if (mode == 0) {       \
            if (vec)           \
                CALL(0, true); \
            else               \
                CALL(0, false); \
        } else {               \
            if (vec)           \
                CALL(1, true); \
            else               \
                CALL(1, false); \
        }                      \
    } while (0)
    // 32-bit launch over fb_list (nullptr: every strip, wide ones skipped and listed), then the float64 launch over
    // the strips with a wide row: a small grid that strides over a list which is empty in all but pathological batches
#define SKM_CURSOR(MODE, VEC)                                                                                        \
    do {                                                                                                             \
        k_cosine_strip<MODE, VEC, 0, PW, false><<<(unsigned)strips, TB, 0, st>>>(                                    \
            d_xrowptr, d_xcolidx, d_xcounts, d_xrnorm, m, d_ycolptr, d_ypost, d_ypostcnt, d_yrnorm, row0, row1, d_out, ld, fb_list, \
            fb_count, &state->min_yrnorm, wide_list, &state->wide_count);                                            \
        k_cosine_strip<MODE, VEC, 0, PW, true><<<skm_grid_cap(ctx, strips, 2), TB, 0, st>>>(                         \
            d_xrowptr, d_xcolidx, d_xcounts, d_xrnorm, m, d_ycolptr, d_ypost, d_ypostcnt, d_yrnorm, row0, row1, d_out, ld,          \
            wide_list, &state->wide_count, nullptr, hint_dst, hint_src);                                             \
    } while (0)

    // (set on the list path: the last kernel of the call then copies the hand-off counters to the pinned hint words)
    uint32_t *hint_dst = nullptr, *hint_src = nullptr;
    void *p;
    SKM_TRY(skm_ws(ctx, WS_COS, sizeof(cos_state), &p));
    cos_state *state = (cos_state *)p;
    SKM_TRY(skm_ws(ctx, WS_D, sizeof(uint32_t) * 2 * (size_t)(strips + 8), &p));  // each strip is listed at most once per list
    uint32_t *fb_list = nullptr, *fb_count = nullptr, *wide_list = (uint32_t *)p + (strips + 8);
#ifdef SKM_DIAG
    // Diagnostic builds of the cursor kernel (tools/ablate_cosine.py): results are NOT valid.
    const int abl = skm_opts().cosine_ablate;
    if (abl >= 1 && abl <= 3 && mode == 0 && vec) {
        SKM_PROF(ctx, "k_cosine_strip");
#define SKM_CURSOR_ABL(ABL)                                                                                          \
    k_cosine_strip<0, true, ABL, PW, false><<<(unsigned)strips, TB, 0, st>>>(d_xrowptr, d_xcolidx, d_xcounts, d_xrnorm, m, d_ycolptr, \
                                                                       d_ypost, d_ypostcnt, d_yrnorm, row0, row1, d_out, ld,   \
                                                                       nullptr, nullptr, nullptr, nullptr, nullptr)
        if (abl == 1)
            SKM_CURSOR_ABL(1);
        else if (abl == 2)
            SKM_CURSOR_ABL(2);
        else
            SKM_CURSOR_ABL(3);
#undef SKM_CURSOR_ABL
        return skm_check_launch("k_cosine_strip");
    }
#endif
    const int forced_path = skm_opts().cosine_path;  // 2 ("cursor") forces the fallback kernel everywhere
    if (forced_path == 2) {
        if (phase == 1)
            return SKM_OK;
        SKM_TRY(cosine_prologue(ctx, nullptr, 0, state, 0ull, d_yrnorm, m, st));
        SKM_PROF(ctx, "k_cosine_strip");
        SKM_BY_MODE_VEC(SKM_CURSOR);
        return skm_check_launch("k_cosine_strip");
    }

    // Few columns and many rows (the tall-skinny apply case: many query rows against a handful of family totals): the
    // cursor kernel's dense per-strip accumulators hold every column, no neighbour lists are needed
    // (a list path would pin SLOT entries of scratch per row for a few-MB output).  Small SQUARE outputs (a FASTA file
    // of a few hundred records against itself) take the list kernels like large ones: the cursor kernel advances one
    // posting per memory round trip and list, so a k-mer shared by all rows costs it m round trips per strip
    // (measured, 200 - 1000 rows: 0.12 ms whatever the size, against 0.04 - 0.05 ms for the four list-path launches).
    if (m <= CH && nrows >= 8 * m && forced_path != 1) {  // SKM_COSINE_PATH=lists keeps the list path (tests)
        if (phase == 1)
            return SKM_OK;
        SKM_TRY(cosine_prologue(ctx, nullptr, 0, state, 0ull, d_yrnorm, m, st));
        SKM_PROF(ctx, "k_cosine_strip");
        SKM_BY_MODE_VEC(SKM_CURSOR);
        return skm_check_launch("k_cosine_strip");
    }

    // ---- fast path: sparse Gram (+ large-table pass) -> streaming writer -> cursor kernel for what is left
    // neighbour lists: row r of the block owns SLOT entries at g_ent[r * SLOT] (all the first pass
    // can produce, and never more than the m neighbours a row can have); the lists of the large-table
    // pass are allocated behind that region
    constexpr int FIRST_GH = SKM_FIRST_GH;                 // hash slots of the first pass (a row holds 3/4 of them)
    constexpr long long FIRST_CAP = FIRST_GH / 4 * 3;
    const unsigned long long SLOT = (unsigned long long)(m < FIRST_CAP ? m : FIRST_CAP);
    const unsigned long long fixed_ent = (unsigned long long)nrows * SLOT;
    const unsigned long long cap_ent = fixed_ent + (unsigned long long)max((int64_t)(1 << 20), nrows * 256);
    SKM_TRY(skm_ws(ctx, WS_A, sizeof(uint64_t) * (size_t)cap_ent, &p));
    uint64_t *g_ent = (uint64_t *)p;
    SKM_TRY(skm_ws(ctx, WS_B, sizeof(uint64_t) * (size_t)(nrows + 8), &p));
    uint64_t *g_start = (uint64_t *)p;
    SKM_TRY(skm_ws(ctx, WS_C, sizeof(uint32_t) * (size_t)(nrows + 8), &p));
    uint32_t *g_len = (uint32_t *)p;
    fb_list = wide_list - (strips + 8);
    SKM_TRY(skm_ws(ctx, WS_E, sizeof(uint32_t) * (size_t)(nrows + 8), &p));
    uint32_t *over_list = (uint32_t *)p;
    SKM_TRY(skm_ws(ctx, WS_F, sizeof(uint32_t) * (size_t)(strips + 8), &p));
    uint32_t *fb_flag = (uint32_t *)p;
    unsigned long long *g_counter = &state->g_counter;
    fb_count = &state->fb_count;
    uint32_t *over_count = state->over_count;
    constexpr int MAXB = 16;  // row blocks of the overlapped schedule (one overflow counter each)
    // strip flags and counters cleared, list allocation behind the fixed slots, min_j yrnorm[j]: one launch
    if (phase != 2)
        SKM_TRY(cosine_prologue(ctx, fb_flag, strips + 8, state, fixed_ent, d_yrnorm, m, st));
#ifdef SKM_DIAG
    const int gabl = skm_opts().gram_ablate;  // diagnostic builds of k_gram_sparse (1, 2, 4: results NOT valid)
#else
    constexpr int gabl = 0;
#endif

    // Schedule.  By default the kernels run back to back on the context's stream.  SKM_COSINE_OVERLAP=1 selects a
    // blocked schedule for outputs of 2 GB and more: the rows are cut into 8 blocks; block b's lists are built on a
    // stream confined to half of the CUs while block b-1 is written on an unconfined one (overlap_streams above).
    // Measured at config 3 (round 4): 10.45 against 10.98 ms per step, 10.70 against 11.06 with the per-stage events of
    // the profiler on - and inside bench.py's timed region (events on, a box whose single stream took 10.6) 10.59
    // against 10.61.  Both kernels run slower side by side (the Gram's 4.4 GB of gathers share HBM with the stores:
    // the writer's launches sum to 7.5-7.7 ms instead of 6.6-7.1), so the writer's own roofline fraction drops from 0.72-0.76
    // to 0.66 for a gain inside the box-to-box spread: it stays opt-in.  A batch whose previous call handed thousands of
    // rows to the heavy kernel takes the panel pipeline (written for the whole call) whatever the variable says.
    const bool ov_wanted = skm_opts().cosine_overlap == 1;
    bool panels_first = false;
    if constexpr (sizeof(PW) == 8)
        panels_first = heavy_panels_wanted(ctx, nrows, m);
    int nblk = 1;
    if (phase == 0 && gabl == 0 && ov_wanted && !panels_first && nrows >= 4096 &&
        (double)nrows * (double)ld * 4.0 >= 2e9 && overlap_streams(ctx) == SKM_OK)
        nblk = 8;
#ifdef SKM_DIAG
    if (nblk > 1 && skm_opts().overlap_blocks)  // diagnostic: block count of the overlapped schedule (2..16)
        nblk = max(2, min(16, skm_opts().overlap_blocks));
#endif
    const int64_t brows = skm_ceil_div(skm_ceil_div(nrows, nblk), 8) * 8;  // whole cursor strips per block
    hipStream_t s_g = nblk > 1 ? ctx->s_gram : st, s_w = nblk > 1 ? ctx->s_writer : st;
    if (nblk > 1) {
        SKM_HIP(hipEventRecord(ctx->sync_events[0], st));
        SKM_HIP(hipStreamWaitEvent(s_g, ctx->sync_events[0], 0));
        SKM_HIP(hipStreamWaitEvent(s_w, ctx->sync_events[0], 0));
    }
    for (int b = 0; b < nblk; ++b) {
        const int64_t b0 = (int64_t)b * brows, b1 = min(nrows, b0 + brows);
        if (b0 >= b1)
            break;
        const int64_t bn = b1 - b0;
        hipStream_t gs = b == 0 ? st : s_g;
        uint32_t *b_over_list = over_list + b0, *b_over_count = over_count + b;
        if (phase != 2) {
            // one row per workgroup, 2048 slots: 26 KB of LDS -> 6 workgroups per CU
            SKM_PROF_ON(ctx, "k_gram_sparse", gs);
#define SKM_GRAM(GABL)                                                                                               \
    k_gram_sparse<GABL, 1, FIRST_GH, 256, 2, 32, 2, PW><<<(unsigned)bn, 256, 0, gs>>>(                               \
        d_xrowptr, d_xcolidx, d_xcounts, d_ycolptr, d_ypost, d_ypostcnt, row0 + b0, row0 + b1, SLOT, b0, g_ent, cap_ent, \
        g_counter, g_start + b0, g_len + b0, b_over_list, b_over_count, d_xrnorm, &state->min_yrnorm, G_WIDE_ROW)
#ifdef SKM_DIAG
            // diagnostic: other lane-group shapes for the lists of 17+ postings (exact results)
#define SKM_GRAM_SHAPE(GG, UU)                                                                                       \
    k_gram_sparse<0, 1, FIRST_GH, 256, 2, GG, UU, PW><<<(unsigned)bn, 256, 0, gs>>>(                                 \
        d_xrowptr, d_xcolidx, d_xcounts, d_ycolptr, d_ypost, d_ypostcnt, row0 + b0, row0 + b1, SLOT, b0, g_ent, cap_ent, \
        g_counter, g_start + b0, g_len + b0, b_over_list, b_over_count, d_xrnorm, &state->min_yrnorm, G_WIDE_ROW)
            const int shape = skm_opts().gram_shape;
            if (gabl == 0 && shape == 1)
                SKM_GRAM_SHAPE(16, 2);
            else if (gabl == 0 && shape == 2)
                SKM_GRAM_SHAPE(16, 1);
            else if (gabl == 0 && shape == 3)
                SKM_GRAM_SHAPE(32, 1);
            else if (gabl == 0 && shape == 4)
                SKM_GRAM_SHAPE(64, 1);
            else
#undef SKM_GRAM_SHAPE
            if (gabl == 1)
                SKM_GRAM(1);
            else if (gabl == 2)
                SKM_GRAM(2);
            else if (gabl == 4)
                SKM_GRAM(4);
            else if (gabl == 5)
                SKM_GRAM(5);
            else if (gabl == 6)
                SKM_GRAM(6);
            else if (gabl == 7)
                SKM_GRAM(7);
            else if (gabl == 8)
                SKM_GRAM(8);
            else if (gabl == 9)
                SKM_GRAM(9);
            else if (gabl == 3) {  // phase stamps (exact results, slower): read with skm_debug_gram_phases
                unsigned long long zeros[8] = {};
                SKM_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_gram_phase_ticks), zeros, sizeof(zeros), 0, hipMemcpyHostToDevice, st));
                SKM_GRAM(3);
            } else
#endif
                SKM_GRAM(0);
#undef SKM_GRAM
        }
        SKM_TRY(skm_check_launch("k_gram_sparse"));
        if (phase == 1)
            return SKM_OK;
        if (nblk > 1) {
            SKM_HIP(hipEventRecord(ctx->sync_events[1 + b], gs));
            SKM_HIP(hipStreamWaitEvent(s_w, ctx->sync_events[1 + b], 0));
        }
        panel_bufs pnl = {};
        bool use_panels = false;
        const bool pack_heavy = heavy_pack_wanted(ctx);
        if constexpr (sizeof(PW) == 8) {
            if (nblk == 1 && panels_first) {
                const int prc = heavy_panels_run<PW>(ctx, d_xrowptr, d_xcolidx, d_xcounts, d_ycolptr, d_ypost, d_ypostcnt, m, row0, b0, bn,
                                                     b_over_list, b_over_count, &pnl, s_w);
                if (prc != SKM_E_NOMEM)  // (no room for the panels: every heavy row is walked, as exact as with them)
                    SKM_TRY(prc);
                use_panels = prc == SKM_OK;
            }
        }
        {
            // rows the first pass could not hold: Gram and write fused, one row per workgroup, dense LDS tile
#ifdef SKM_DIAG
            {
                const int abl_h = skm_opts().heavy_ablate;
                SKM_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_heavy_abl), &abl_h, sizeof(int), 0, hipMemcpyHostToDevice, s_w));
            }
#endif
            SKM_PROF_ON(ctx, "k_cosine_heavy", s_w);
#define SKM_HEAVY_ARGS                                                                                               \
    d_xrowptr, d_xcolidx, d_xcounts, d_xrnorm, m, d_ycolptr, d_ypost, d_ypostcnt, d_yrnorm, row0, b0, b_over_list, b_over_count, \
        g_len, d_out, ld, pnl, &state->min_yrnorm
#define SKM_HEAVY(MODE, VEC)                                                                                         \
    do {                                                                                                             \
        if (use_panels) { /* rows with a panel first (the PANEL forms skip the others); packed tiles, then what they left */ \
            if (pack_heavy)                                                                                          \
                k_cosine_heavy<MODE, VEC, PW, true, HEAVYP_CH, HEAVYK_TB, HEAVYK_EMAX, true>                         \
                    <<<skm_grid_cap(ctx, bn, HEAVYK_TB == 1024 ? 1 : 2), HEAVYK_TB, 0, s_w>>>(SKM_HEAVY_ARGS);                               \
            k_cosine_heavy<MODE, VEC, PW, true, HEAVYP_CH, HEAVYP_TB, HEAVYP_EMAX><<<skm_grid_cap(ctx, bn, 1), HEAVYP_TB, 0, s_w>>>( \
                SKM_HEAVY_ARGS);                                                                                     \
        }                                                                                                            \
        /* every row (left): the general form */                                                                     \
        if (pack_heavy)                                                                                              \
            k_cosine_heavy<MODE, VEC, PW, false, HEAVY_CH, HEAVYK_TB, HEAVYK_EMAX, true>                             \
                <<<skm_grid_cap(ctx, bn, HEAVYK_TB == 1024 ? 1 : 2), HEAVYK_TB, 0, s_w>>>(SKM_HEAVY_ARGS);                                   \
        k_cosine_heavy<MODE, VEC, PW, false, HEAVY_CH, HEAVY_TB, HEAVY_EMAX><<<skm_grid_cap(ctx, bn, 1), HEAVY_TB, 0, s_w>>>( \
            SKM_HEAVY_ARGS);                                                                                         \
    } while (0)
            SKM_BY_MODE_VEC(SKM_HEAVY);
#undef SKM_HEAVY
#undef SKM_HEAVY_ARGS
        }
        SKM_TRY(skm_check_launch("k_cosine_heavy"));
        {
            // one output row per workgroup of 1024 threads
            SKM_PROF_ON(ctx, "k_cosine_write", s_w);
#define SKM_WRITE(MODE, VEC)                                                                                         \
    do {                                                                                                             \
        if (m >= 65536) /* wide rows: 128 KiB per step (fewer barriers, longer bursts: 6.9 vs 7.2 ms at m = 100k) */ \
            k_cosine_write<MODE, VEC, SKM_WRITER_CH, SKM_WRITER_TB><<<(unsigned)bn, SKM_WRITER_TB, 0, s_w>>>(        \
                g_ent, g_start, g_len, d_xrnorm, d_yrnorm, m, row0, b0, d_out, ld, fb_list, fb_count, fb_flag,       \
                wide_list, &state->wide_count);                                                                      \
        else                                                                                                         \
            k_cosine_write<MODE, VEC, 4096, 1024><<<(unsigned)bn, 1024, 0, s_w>>>(                                   \
                g_ent, g_start, g_len, d_xrnorm, d_yrnorm, m, row0, b0, d_out, ld, fb_list, fb_count, fb_flag,       \
                wide_list, &state->wide_count);                                                                      \
    } while (0)
            SKM_BY_MODE_VEC(SKM_WRITE);
#undef SKM_WRITE
        }
        SKM_TRY(skm_check_launch("k_cosine_write"));
    }
    if (nblk > 1) {  // the context's stream continues after both side streams
        SKM_HIP(hipEventRecord(ctx->sync_events[1 + MAXB], s_w));
        SKM_HIP(hipEventRecord(ctx->sync_events[2 + MAXB], s_g));
        SKM_HIP(hipStreamWaitEvent(st, ctx->sync_events[1 + MAXB], 0));
        SKM_HIP(hipStreamWaitEvent(st, ctx->sync_events[2 + MAXB], 0));
    }
    // how many rows this call handed on (one counter per row block): the next call on this context reads them (without
    // waiting) to decide whether the panel pipeline is worth its dozen launches (heavy_panels_wanted)
    static_assert(MAXB == 16, "k_cosine_strip<WIDE> copies 16 counters");
    if (ctx->d_err) {  // pinned memory is device-addressable: the last kernel stores the counters itself
        hint_dst = ctx->d_err - SKM_DEVERR_WORD + 2048 / 4;
        hint_src = over_count;
    } else {
        SKM_HIP(hipMemcpyAsync((uint8_t *)ctx->h_pinned + 2048, over_count, sizeof(uint32_t) * MAXB, hipMemcpyDeviceToHost, st));
    }
    {
        // strips with a row still flagged; worst-case grid, surplus workgroups exit at once
        SKM_PROF(ctx, "k_cosine_strip");
        SKM_BY_MODE_VEC(SKM_CURSOR);
    }
#undef SKM_CURSOR
#undef SKM_BY_MODE_VEC
    return skm_check_launch("k_cosine_strip");
}
}  // namespace

// How the last list-path skm_cosine_csr call on this context distributed its rows (reporting only: bench.py's skewed
// workload): h_out[0] = rows whose neighbours did not fit the first pass's table (sent to the large-table pass),
// h_out[1] = 8-row strips left to the cursor kernel (a row overflowed the large table too), h_out[2] = neighbour-list
// entries written.  Synchronises the stream.
extern "C" int skm_cosine_csr_stats(skm_ctx *ctx, int64_t *h_out4)
{
    SKM_REQUIRE(ctx && h_out4, SKM_E_BADARG, "skm_cosine_csr_stats: bad argument");
    h_out4[0] = h_out4[1] = h_out4[2] = h_out4[3] = 0;
    if (!ctx->ws[WS_COS] || ctx->ws_bytes[WS_COS] < sizeof(cos_state))
        return SKM_OK;
    SKM_HIP(hipSetDevice(ctx->device));
    cos_state host;
    SKM_HIP(hipMemcpyAsync(&host, ctx->ws[WS_COS], offsetof(cos_state, pad), hipMemcpyDeviceToHost, ctx->stream));
    SKM_HIP(hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < 16; ++b)
        h_out4[0] += host.over_count[b];
    h_out4[1] = host.fb_count;
    h_out4[2] = (int64_t)(host.g_counter - host.first_entry);
    h_out4[3] = host.wide_count;
    return SKM_OK;
}

// Reporting only (bench.py's skewed workload): what the last panel pipeline of this context did.  h_out6 = heavy rows,
// blocks of 256 of them, blocks without a panel (their rows were walked in full), sum of panel columns K, sum of panel
// rows |J|, multiply-accumulates of the panel GEMMs (256 x |J| x K rounded up to 64, per block).  Synchronises the stream.
extern "C" int skm_heavy_panel_stats(skm_ctx *ctx, int64_t *h_out6)
{
    SKM_REQUIRE(ctx && h_out6, SKM_E_BADARG, "skm_heavy_panel_stats: bad argument");
    for (int q = 0; q < 6; ++q)
        h_out6[q] = 0;
    if (!ctx->panel_meta || ctx->panel_nb <= 0)
        return SKM_OK;
    SKM_HIP(hipSetDevice(ctx->device));
    SKM_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<uint32_t> meta(4 * (size_t)ctx->panel_nb);
    uint32_t rows = 0;
    // on the context's own stream: a plain hipMemcpy goes through the null stream, which a CU-masked (blocking) stream of
    // another context would synchronise with
    SKM_HIP(hipMemcpyAsync(meta.data(), ctx->panel_meta, sizeof(uint32_t) * meta.size(), hipMemcpyDeviceToHost, ctx->stream));
    SKM_HIP(hipMemcpyAsync(&rows, ctx->panel_rows, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    SKM_HIP(hipStreamSynchronize(ctx->stream));
    const int64_t used = std::min<int64_t>(ctx->panel_nb, ((int64_t)rows + PB_ROWS - 1) / PB_ROWS);
    h_out6[0] = rows;
    h_out6[1] = used;
    for (int64_t b = 0; b < used; ++b) {
        const int64_t K = meta[4 * b], J = meta[4 * b + 1];
        if (K == 0) {
            ++h_out6[2];
            continue;
        }
        h_out6[3] += K;
        h_out6[4] += J;
        h_out6[5] += (int64_t)PB_ROWS * J * ((K + 63) / 64 * 64);
    }
    return SKM_OK;
}

namespace {
__global__ __launch_bounds__(256) void k_similarity_to_distance(int64_t rows, int64_t m, float *__restrict__ out, int64_t ld)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m)
        return;
    for (int64_t i = blockIdx.y; i < rows; i += gridDim.y)
        out[i * ld + j] = fminf(fmaxf(1.0f - out[i * ld + j], 0.0f), 2.0f);
}
}  // namespace

int skm_similarity_to_distance(skm_ctx *ctx, int64_t rows, int64_t m, float *d_out, int64_t ld)
{
    if (rows <= 0 || m <= 0)
        return SKM_OK;
    dim3 grid((unsigned)skm_ceil_div(m, 256), (unsigned)(rows < 4096 ? rows : 4096));
    SKM_PROF(ctx, "k_similarity_to_distance");
    k_similarity_to_distance<<<grid, 256, 0, ctx->stream>>>(rows, m, d_out, ld);
    return skm_check_launch("k_similarity_to_distance");
}

extern "C" int skm_cosine_csr(skm_ctx *ctx, int64_t n, const int64_t *d_xrowptr, const uint32_t *d_xcolidx,
                              const uint32_t *d_xcounts, const float *d_xrnorm, int64_t m, int64_t ncols,
                              const uint32_t *d_ycolptr, const void *d_ypost, int post_bits, const uint32_t *d_ypostcnt,
                              const float *d_yrnorm, int64_t row0, int64_t row1, int mode, float *d_out, int64_t ld)
{
    SKM_REQUIRE(post_bits == 64 || post_bits == 32, SKM_E_BADARG, "skm_cosine_csr: post_bits must be 32 or 64");
    SKM_REQUIRE(mode >= 0 && mode <= 2, SKM_E_BADARG, "skm_cosine_csr: mode must be 0, 1 or 2");
    // mode 2 (distance between two DIFFERENT matrices: no diagonal rule) = the similarity block, then 1 - s clamped, the
    // same float operations the fused mode-1 epilogue applies
    const int kmode = mode == 2 ? 0 : mode;
    int rc;
    if (post_bits == 32) {
        SKM_REQUIRE(m <= ((int64_t)1 << 24), SKM_E_BADARG, "skm_cosine_csr: 32-bit postings hold rows < 2^24");
        rc = cosine_csr_impl<uint32_t>(ctx, n, d_xrowptr, d_xcolidx, d_xcounts, d_xrnorm, m, ncols, d_ycolptr,
                                       (const uint32_t *)d_ypost, d_ypostcnt, d_yrnorm, row0, row1, kmode, d_out, ld);
    } else {
        rc = cosine_csr_impl<uint64_t>(ctx, n, d_xrowptr, d_xcolidx, d_xcounts, d_xrnorm, m, ncols, d_ycolptr,
                                       (const uint64_t *)d_ypost, nullptr, d_yrnorm, row0, row1, kmode, d_out, ld);
    }
    if (rc == SKM_OK && mode == 2)
        rc = skm_similarity_to_distance(ctx, row1 - row0, m, d_out, ld);
    return rc;
}

// skm_cosine_csr in two calls, for a stream of batches (engine.OverlappedPipeline): phase 1 builds the neighbour lists of
// the row block on ctx (prologue + sparse Gram; they stay in ctx's scratch), phase 2 runs everything behind them - heavy
// rows, streaming writer, cursor kernel, the mode-2 epilogue - on exec_ctx's STREAM, still with ctx's scratch.  The
// caller orders the two (skm_event_record / skm_stream_wait) and does not start the next phase 1 on ctx before the
// phase 2 that reads its lists has finished.  phase 0 = skm_cosine_csr on ctx.
extern "C" int skm_cosine_csr_phase(skm_ctx *ctx, skm_ctx *exec_ctx, int phase, int64_t n, const int64_t *d_xrowptr,
                                    const uint32_t *d_xcolidx, const uint32_t *d_xcounts, const float *d_xrnorm, int64_t m,
                                    int64_t ncols, const uint32_t *d_ycolptr, const void *d_ypost, int post_bits,
                                    const uint32_t *d_ypostcnt, const float *d_yrnorm, int64_t row0, int64_t row1, int mode,
                                    float *d_out, int64_t ld)
{
    SKM_REQUIRE(ctx && phase >= 0 && phase <= 2, SKM_E_BADARG, "skm_cosine_csr_phase: phase must be 0, 1 or 2");
    SKM_REQUIRE(post_bits == 64 || post_bits == 32, SKM_E_BADARG, "skm_cosine_csr_phase: post_bits must be 32 or 64");
    SKM_REQUIRE(mode >= 0 && mode <= 2, SKM_E_BADARG, "skm_cosine_csr_phase: mode must be 0, 1 or 2");
    SKM_REQUIRE(!exec_ctx || exec_ctx->device == ctx->device, SKM_E_BADARG, "skm_cosine_csr_phase: the two contexts are on different devices");
    if (post_bits == 32)
        SKM_REQUIRE(m <= ((int64_t)1 << 24), SKM_E_BADARG, "skm_cosine_csr_phase: 32-bit postings hold rows < 2^24");
    struct stream_swap {  // phase 2: ctx's kernels, scratch and profiling scopes, on exec_ctx's stream
        skm_ctx *c;
        hipStream_t saved;
        stream_swap(skm_ctx *c_, hipStream_t s) : c(c_), saved(c_->stream) { c->stream = s; }
        ~stream_swap() { c->stream = saved; }
    } swap(ctx, phase == 2 && exec_ctx ? exec_ctx->stream : ctx->stream);
    const int kmode = mode == 2 ? 0 : mode;
    int rc;
    if (post_bits == 32)
        rc = cosine_csr_impl<uint32_t>(ctx, n, d_xrowptr, d_xcolidx, d_xcounts, d_xrnorm, m, ncols, d_ycolptr, (const uint32_t *)d_ypost,
                                       d_ypostcnt, d_yrnorm, row0, row1, kmode, d_out, ld, phase);
    else
        rc = cosine_csr_impl<uint64_t>(ctx, n, d_xrowptr, d_xcolidx, d_xcounts, d_xrnorm, m, ncols, d_ycolptr, (const uint64_t *)d_ypost,
                                       nullptr, d_yrnorm, row0, row1, kmode, d_out, ld, phase);
    if (rc == SKM_OK && mode == 2 && phase != 1)
        rc = skm_similarity_to_distance(ctx, row1 - row0, m, d_out, ld);
    return rc;
}

#ifdef SKM_DIAG
// Diagnostic: per-phase shader-clock ticks summed over the workgroups of the last k_gram_sparse
// launch made with SKM_GRAM_ABLATE=3 (tools/ablate_cosine.py).  Diagnostic library only.
extern "C" int skm_debug_gram_phases(skm_ctx *ctx, unsigned long long *h_ticks8)
{
    SKM_REQUIRE(ctx && h_ticks8, SKM_E_BADARG, "skm_debug_gram_phases: bad argument");
    SKM_HIP(hipStreamSynchronize(ctx->stream));
    SKM_HIP(hipMemcpyFromSymbol(h_ticks8, HIP_SYMBOL(g_gram_phase_ticks), 8 * sizeof(unsigned long long)));
    return SKM_OK;
}
#endif
