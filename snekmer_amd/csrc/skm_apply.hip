// Consumers of the cosine stage that never need the score block itself, and the real-valued form of
// the cosine.
//
// Reference behaviour being replaced:
//   cosine_similarity(totals, counts).T -> np.argsort(-S, axis=1)[:, :2] -> Score / delta / Prediction
//                                          snekmer/rules/apply.smk:278-328, rules/learn.smk:811-849
//   sklearn cosine_similarity / pairwise_distances(metric="cosine") on arbitrary float matrices
//                                          snekmer/score.py:149-172 (e.g. length-normalised rows of
//                                          snekmer/utils.py:183-203)
//
//   k_apply_columns       two words per column of the family-total matrix: up to two (family, total) postings inline,
//                         or a marker and the posting range.  Built per call (one pass over the column starts).
//   k_apply_top2          WAVE per query row: a lane reads an entry's column id and count (coalesced), then ONE
//                         16-byte gather - the column's words - gives it the families and totals; columns shared by
//                         three or more families go on to the posting list.  The exact integer dot products are summed
//                         in a 512-slot LDS hash table per wave keyed by the family (a query touches a handful
//                         of families, whatever their number), the float64 scores dot / (|x| |y|) are formed
//                         from those exact integers, and only the two best (score desc, column asc) leave the
//                         kernel: the N x A block of rules/apply.smk:282-289 is never stored.  (Until round 5: a
//                         workgroup per row with dense accumulators over all families, three dependent gathers
//                         per entry and a scan of every family per row: 1.23 ms for 100 k queries x 1000 families.)
//   k_apply_top2_dense    that workgroup-per-row form, for the rows that touch more than 384 families.
//   k_cosine_dense_f64    dense float64 GEMM of row-normalised operands on the f64 matrix cores
//                         (v_mfma_f64_16x16x4_f64): sklearn semantics (normalise in float64, zero
//                         norms -> 1, then dot) for feature matrices that are not counts.
#include "skm_common.h"

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;

// ------------------------------------------------------------------------------- apply epilogue
constexpr int ACH = 8192;  // families per LDS pass at most (64 KiB of int64 accumulators)
constexpr int ATB = 256;

struct top2 {
    double v1, v2;
    long long d1, d2;
    uint32_t i1, i2;
};

__device__ __forceinline__ bool better(double a, uint32_t ia, double b, uint32_t ib)
{
    return a > b || (a == b && ia < ib);
}

__device__ __forceinline__ void top2_push(top2 &t, double v, uint32_t i, long long d)
{
    if (better(v, i, t.v1, t.i1)) {
        t.v2 = t.v1, t.i2 = t.i1, t.d2 = t.d1;
        t.v1 = v, t.i1 = i, t.d1 = d;
    } else if (better(v, i, t.v2, t.i2)) {
        t.v2 = v, t.i2 = i, t.d2 = d;
    }
}

// d_ynorm[j] = sqrt(normsq) (1 for an all-zero row: sklearn's zero-norm rule), float64
__global__ void k_norms_f64(int64_t m, const uint64_t *__restrict__ normsq, double *__restrict__ out)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m)
        out[j] = normsq[j] ? sqrt((double)normsq[j]) : 1.0;
}

// rows: those listed in row_list[0 .. *row_count) (positions in the launch's row range)
__global__ __launch_bounds__(ATB) void k_apply_top2_dense(const int64_t *__restrict__ xrowptr,
                                                          const uint32_t *__restrict__ xcolidx,
                                                          const uint32_t *__restrict__ xcounts,
                                                          const uint64_t *__restrict__ xnormsq, int64_t m,
                                                          const uint32_t *__restrict__ ycolptr,
                                                          const uint64_t *__restrict__ ypost,
                                                          const double *__restrict__ ynorm, int64_t row0,
                                                          const uint32_t *__restrict__ row_list,
                                                          const uint32_t *__restrict__ row_count,
                                                          int ach, uint32_t *__restrict__ out_idx,
                                                          double *__restrict__ out_score, long long *__restrict__ out_dot)
{
    // `ach` accumulators (families per pass) of dynamic LDS: sized to the problem so that a handful of
    // families does not cost the occupancy of 8192
    extern __shared__ __attribute__((aligned(16))) unsigned long long s_acc[];
    __shared__ top2 s_t[ATB / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t nlisted = *row_count;
    for (uint32_t it = blockIdx.x; it < nlisted; it += gridDim.x) {
        const int64_t r = row_list[it];
        const int64_t i = row0 + r;
        const int64_t b = xrowptr[i], e = xrowptr[i + 1];
        const uint64_t nx = xnormsq[i];
        const double sx = nx ? sqrt((double)nx) : 1.0;
        top2 t = {-INFINITY, -INFINITY, 0, 0, NONE, NONE};
        for (int64_t a0 = 0; a0 < m; a0 += ach) {
            const int span = (int)min((int64_t)ach, m - a0);
            for (int z = tid; z < span; z += ATB)
                s_acc[z] = 0ull;
            __syncthreads();
            // one THREAD per entry of the query row: a family-total matrix has few families per k-mer (1.1 on
            // average at 1000 families), so a posting list is a handful of words; the three dependent loads
            // (column id, list bounds, postings) of all entries are in flight together
            for (int64_t q = b + tid; q < e; q += ATB) {
                const uint32_t c = xcolidx[q];
                if (c == NONE)  // a column Y does not have
                    continue;
                const unsigned long long v = xcounts[q];
                const uint32_t pb = ycolptr[c], pe = ycolptr[c + 1];
                for (uint32_t p = pb; p < pe; ++p) {
                    const uint64_t pw = ypost[p];
                    const int64_t a = (int64_t)(uint32_t)pw - a0;
                    if (a >= 0 && a < span)
                        atomicAdd(&s_acc[a], v * (unsigned long long)(pw >> 32));
                }
            }
            __syncthreads();
            for (int a = tid; a < span; a += ATB) {
                const long long d = (long long)s_acc[a];
                const double s = d ? (double)d / (sx * ynorm[a0 + a]) : 0.0;
                top2_push(t, s, (uint32_t)(a0 + a), d);
            }
            __syncthreads();
        }
        // wave merge, then the workgroup's four waves
        for (int o = 32; o > 0; o >>= 1) {
            top2 u;
            u.v1 = __shfl_down(t.v1, o), u.v2 = __shfl_down(t.v2, o);
            u.d1 = __shfl_down(t.d1, o), u.d2 = __shfl_down(t.d2, o);
            u.i1 = __shfl_down(t.i1, o), u.i2 = __shfl_down(t.i2, o);
            if (u.i1 != NONE)
                top2_push(t, u.v1, u.i1, u.d1);
            if (u.i2 != NONE)
                top2_push(t, u.v2, u.i2, u.d2);
        }
        if (lane == 0)
            s_t[wid] = t;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < ATB / 64; ++w) {
                const top2 u = s_t[w];
                if (u.i1 != NONE)
                    top2_push(t, u.v1, u.i1, u.d1);
                if (u.i2 != NONE)
                    top2_push(t, u.v2, u.i2, u.d2);
            }
            out_idx[2 * r] = t.i1;
            out_idx[2 * r + 1] = t.i2;
            out_score[2 * r] = t.i1 == NONE ? 0.0 : t.v1;
            out_score[2 * r + 1] = t.i2 == NONE ? 0.0 : t.v2;
            out_dot[2 * r] = t.i1 == NONE ? 0 : t.d1;
            out_dot[2 * r + 1] = t.i2 == NONE ? 0 : t.d2;
        }
        __syncthreads();
    }
}

// ---- the wave-per-row form
constexpr uint32_t COL_MULTI = 0xFFFFFFFEu;  // family field of a column word: several families hold the k-mer
constexpr int AHS = 512, AHCAP = 384;        // hash slots per wave / families a wave holds

// Two words per column: up to two (family | total << 32) postings inline (0xFFFFFFFF... = none), or the marker and the
// column's posting range (start | end << 32) when three or more families hold the k-mer.  One 16-byte gather per query
// entry then covers every column with at most two families, and longer ones need no look at the column starts.
__global__ __launch_bounds__(256) void k_apply_columns(int64_t ncols, const uint32_t *__restrict__ ycolptr,
                                                       const uint64_t *__restrict__ ypost, ulonglong2 *__restrict__ desc)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= ncols)
        return;
    const uint32_t pb = ycolptr[c], pe = ycolptr[c + 1];
    ulonglong2 w = make_ulonglong2(~0ull, ~0ull);  // no family holds it
    if (pe - pb == 1u) {
        w.x = ypost[pb];
    } else if (pe - pb == 2u) {
        w.x = ypost[pb];
        w.y = ypost[pb + 1];
    } else if (pe > pb) {
        w.x = (uint64_t)COL_MULTI;
        w.y = (uint64_t)pb | ((uint64_t)pe << 32);
    }
    desc[c] = w;
}

__global__ __launch_bounds__(ATB) void k_apply_top2(const int64_t *__restrict__ xrowptr,
                                                    const uint32_t *__restrict__ xcolidx,
                                                    const uint32_t *__restrict__ xcounts,
                                                    const uint64_t *__restrict__ xnormsq, int64_t m,
                                                    const ulonglong2 *__restrict__ ydesc,
                                                    const uint64_t *__restrict__ ypost,
                                                    const double *__restrict__ ynorm, int64_t row0, int64_t nrows,
                                                    const uint32_t *__restrict__ row_order,
                                                    uint32_t *__restrict__ over_list, uint32_t *__restrict__ over_count,
                                                    uint32_t *__restrict__ out_idx,
                                                    double *__restrict__ out_score, long long *__restrict__ out_dot)
{
    constexpr int NW = ATB / 64;
    __shared__ uint32_t s_key[NW][AHS];
    __shared__ unsigned long long s_val[NW][AHS];
    __shared__ uint32_t s_distinct[NW];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    uint32_t *keys = s_key[wid];
    unsigned long long *vals = s_val[wid];
    for (int z = lane; z < AHS; z += 64) {
        keys[z] = NONE;
        vals[z] = 0ull;
    }
    if (lane == 0)
        s_distinct[wid] = 0u;
    __threadfence_block();
    auto add = [&](uint32_t fam, unsigned long long prod) {
        uint32_t h = (fam * 2654435761u) >> (32 - 9);
        for (int probe = 0; probe < AHS; ++probe) {
            uint32_t seen = __atomic_load_n(&keys[h], __ATOMIC_RELAXED);
            if (seen == NONE) {
                seen = atomicCAS(&keys[h], NONE, fam);
                if (seen == NONE) {
                    seen = fam;
                    atomicAdd(&s_distinct[wid], 1u);
                }
            }
            if (seen == fam) {
                atomicAdd(&vals[h], prod);
                break;
            }
            h = (h + 1) & (AHS - 1);
        }
    };
    // Position p of the launch is row row_order[p] (or p).  Every XCD (workgroups go to the eight XCDs round-robin) owns
    // one contiguous eighth of the positions and its resident waves walk it side by side: with an order that puts
    // similar rows next to each other - the caller's labels, the records of one FASTA file - the rows in flight on an
    // XCD read the same columns' words and find them in that XCD's L2.
    const int64_t nx = gridDim.x >= 8 ? 8 : (int64_t)gridDim.x;  // (a launch of fewer than 8 workgroups: one share per workgroup)
    const int64_t xcd = blockIdx.x % nx, wg_in = blockIdx.x / nx;
    const int64_t wgs_here = ((int64_t)gridDim.x + nx - 1 - xcd) / nx;  // workgroups of this launch on this XCD
    const int64_t share = (nrows + nx - 1) / nx, pos_end = min(nrows, (xcd + 1) * share);
    for (int64_t pos = xcd * share + wg_in * NW + wid; pos < pos_end; pos += wgs_here * NW) {
        const int64_t r = row_order ? (int64_t)row_order[pos] : pos;
        const int64_t i = row0 + r;
        const int64_t b = xrowptr[i], e = xrowptr[i + 1];
        // four entries per lane and step: the coalesced loads, then the four gathers, are issued together
        for (int64_t q0 = b; q0 < e; q0 += 256) {
            uint32_t c[4], v[4];
            ulonglong2 w[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t q = q0 + u * 64 + lane;
                c[u] = q < e ? xcolidx[q] : NONE;
                v[u] = q < e ? xcounts[q] : 0u;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)  // (ydesc == nullptr: Y has no column or no row)
                w[u] = (ydesc && c[u] != NONE) ? ydesc[c[u]] : make_ulonglong2(~0ull, ~0ull);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t fam = (uint32_t)w[u].x;
                if (fam == NONE)
                    continue;
                if (fam != COL_MULTI) {
                    add(fam, (unsigned long long)v[u] * (unsigned long long)(w[u].x >> 32));
                    if ((uint32_t)w[u].y != NONE)
                        add((uint32_t)w[u].y, (unsigned long long)v[u] * (unsigned long long)(w[u].y >> 32));
                    continue;
                }
                const uint32_t pb = (uint32_t)w[u].y, pe = (uint32_t)(w[u].y >> 32);
                for (uint32_t p = pb; p < pe; ++p) {
                    const uint64_t pw = ypost[p];
                    add((uint32_t)pw, (unsigned long long)v[u] * (unsigned long long)(pw >> 32));
                }
            }
        }
        __threadfence_block();
        const uint32_t distinct = __atomic_load_n(&s_distinct[wid], __ATOMIC_RELAXED);  // wave-uniform
        const uint64_t nsq = xnormsq[i];
        const double sx = nsq ? sqrt((double)nsq) : 1.0;
        top2 t = {-INFINITY, -INFINITY, 0, 0, NONE, NONE};
        // the table's entries (and the table cleared for the next row): all keys first, then all norm gathers together
        uint32_t fk[AHS / 64];
        double fn[AHS / 64];
#pragma unroll
        for (int z = 0; z < AHS / 64; ++z)
            fk[z] = keys[z * 64 + lane];
#pragma unroll
        for (int z = 0; z < AHS / 64; ++z)
            fn[z] = (fk[z] != NONE && distinct <= (uint32_t)AHCAP) ? ynorm[fk[z]] : 1.0;
#pragma unroll
        for (int z = 0; z < AHS / 64; ++z) {
            if (fk[z] != NONE) {
                const int slot = z * 64 + lane;
                const long long d = (long long)vals[slot];
                if (distinct <= (uint32_t)AHCAP)
                    top2_push(t, (double)d / (sx * fn[z]), fk[z], d);
                keys[slot] = NONE;
                vals[slot] = 0ull;
            }
        }
        if (lane == 0)
            s_distinct[wid] = 0u;
        __threadfence_block();
        if (distinct > (uint32_t)AHCAP) {  // more families than the table may hold: the dense form takes the row
            if (lane == 0)
                over_list[atomicAdd(over_count, 1u)] = (uint32_t)r;
            continue;
        }
        for (int o = 32; o > 0; o >>= 1) {
            top2 u;
            u.v1 = __shfl_down(t.v1, o), u.v2 = __shfl_down(t.v2, o);
            u.d1 = __shfl_down(t.d1, o), u.d2 = __shfl_down(t.d2, o);
            u.i1 = __shfl_down(t.i1, o), u.i2 = __shfl_down(t.i2, o);
            if (u.i1 != NONE)
                top2_push(t, u.v1, u.i1, u.d1);
            if (u.i2 != NONE)
                top2_push(t, u.v2, u.i2, u.d2);
        }
        if (lane == 0) {
            // families the row shares nothing with score 0.0 and rank by their index (np.argsort(-S) on equal scores,
            // ties towards the lower column): the smallest indices not taken yet fill what is left of the two slots
            if (t.i1 == NONE) {
                t.i1 = m > 0 ? 0u : NONE;
                t.v1 = 0.0, t.d1 = 0;
            }
            if (t.i2 == NONE && t.i1 != NONE) {
                const uint32_t cand = t.i1 == 0u ? 1u : 0u;
                if ((int64_t)cand < m) {
                    t.i2 = cand;
                    t.v2 = 0.0, t.d2 = 0;
                }
            }
            out_idx[2 * r] = t.i1;
            out_idx[2 * r + 1] = t.i2;
            out_score[2 * r] = t.i1 == NONE ? 0.0 : t.v1;
            out_score[2 * r + 1] = t.i2 == NONE ? 0.0 : t.v2;
            out_dot[2 * r] = t.i1 == NONE ? 0 : t.d1;
            out_dot[2 * r + 1] = t.i2 == NONE ? 0 : t.d2;
        }
    }
}

// ------------------------------------------------------------------------------- f64 dense cosine
typedef double f64x4 __attribute__((ext_vector_type(4)));

// rows scaled to unit L2 norm in float64 (zero norm -> divide by 1), as sklearn.preprocessing.normalize
__global__ __launch_bounds__(256) void k_normalize_rows_f64(int64_t n, int64_t k, const double *__restrict__ in,
                                                            int64_t ld_in, double *__restrict__ out, int64_t ld_out)
{
    __shared__ double s_part[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const double *src = in + i * ld_in;
        double s = 0.0;
        for (int64_t c = tid; c < k; c += 256)
            s += src[c] * src[c];
        for (int o = 32; o > 0; o >>= 1)
            s += __shfl_down(s, o);
        if (lane == 0)
            s_part[wid] = s;
        __syncthreads();
        const double tot = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
        double nrm = sqrt(tot);
        if (nrm == 0.0)
            nrm = 1.0;
        double *dst = out + i * ld_out;
        for (int64_t c = tid; c < ld_out; c += 256)
            dst[c] = c < k ? src[c] / nrm : 0.0;
        __syncthreads();
    }
}

// out[i][j] = sum_c X[i][c] * Y[j][c]; X [n x kp], Y [m x kp] row-major, kp a multiple of FK (zero padded).
// 64 x 64 tile per workgroup, 4 waves of 32 x 32 (2 x 2 tiles of v_mfma_f64_16x16x4_f64).
constexpr int FM = 64, FN = 64, FK = 16, FROW = FK + 1;  // 17-double rows: conflict-free ds_read_b64 of 16 rows

template <int MODE>
__global__ __launch_bounds__(256) void k_cosine_dense_f64(int64_t n, int64_t m, int64_t kp, const double *__restrict__ X,
                                                          const double *__restrict__ Y, double *__restrict__ out,
                                                          int64_t ld, int square)
{
    __shared__ double s_a[FM * FROW];
    __shared__ double s_b[FN * FROW];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int64_t row0 = (int64_t)blockIdx.y * FM, col0 = (int64_t)blockIdx.x * FN;
    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
            acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const int sr = tid >> 2, sc = (tid & 3) * 4;  // staging: 64 rows x 16 doubles, 4 doubles per thread
    const int fr = lane & 15, fk = lane >> 4;
    for (int64_t k0 = 0; k0 < kp; k0 += FK) {
        double va[4], vb[4];
        const int64_t gi = row0 + sr, gj = col0 + sr;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            va[u] = gi < n ? X[gi * kp + k0 + sc + u] : 0.0;
            vb[u] = gj < m ? Y[gj * kp + k0 + sc + u] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s_a[sr * FROW + sc + u] = va[u];
            s_b[sr * FROW + sc + u] = vb[u];
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < FK / 4; ++ks) {
            double fa[2], fb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[t] = s_a[(wr * 32 + t * 16 + fr) * FROW + ks * 4 + fk];
                fb[t] = s_b[(wc * 32 + t * 16 + fr) * FROW + ks * 4 + fk];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
    }
    // C/D layout of the f64 shape: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int64_t j = col0 + wc * 32 + b * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t i = row0 + wr * 32 + a * 16 + (lane >> 4) + 4 * r;
                if (i < n && j < m) {
                    double o = acc[a][b][r];
                    if (MODE == 1) {
                        o = fmin(fmax(1.0 - o, 0.0), 2.0);
                        if (square && i == j)  // sklearn zeroes the diagonal only for cosine_distances(X) / (X, X)
                            o = 0.0;
                    }
                    out[i * ld + j] = o;
                }
            }
        }
}

// Per-row float64 sum and non-zero count of a float32 block: a checksum of a result too large to read
// back (40 GB at BASELINE configs[2]).  Workgroup per row, 16-byte loads.
__global__ __launch_bounds__(256) void k_matrix_row_stats(int64_t n, int64_t m, const float *__restrict__ in, int64_t ld,
                                                          int vec_ok, double *__restrict__ out_sum,
                                                          uint32_t *__restrict__ out_nnz)
{
    __shared__ double s_sum[4];
    __shared__ uint32_t s_cnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const float *row = in + i * ld;
        double s = 0.0;
        uint32_t c = 0;
        if (vec_ok) {
            const int64_t nv = m >> 2;
            for (int64_t v = tid; v < nv; v += 256) {
                const float4 w = *reinterpret_cast<const float4 *>(row + 4 * v);
                s += ((double)w.x + (double)w.y) + ((double)w.z + (double)w.w);
                c += (w.x != 0.0f) + (w.y != 0.0f) + (w.z != 0.0f) + (w.w != 0.0f);
            }
            for (int64_t j = (nv << 2) + tid; j < m; j += 256) {
                s += (double)row[j];
                c += row[j] != 0.0f;
            }
        } else {
            for (int64_t j = tid; j < m; j += 256) {
                s += (double)row[j];
                c += row[j] != 0.0f;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            s += __shfl_down(s, o);
            c += __shfl_down(c, o);
        }
        if (lane == 0) {
            s_sum[wid] = s;
            s_cnt[wid] = c;
        }
        __syncthreads();
        if (tid == 0) {
            out_sum[i] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
            out_nnz[i] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int skm_matrix_row_stats(skm_ctx *ctx, int64_t n, int64_t m, const float *d_in, int64_t ld, double *d_sum,
                                    uint32_t *d_nnz)
{
    SKM_REQUIRE(ctx && n >= 0 && m >= 0 && ld >= m, SKM_E_BADARG, "skm_matrix_row_stats: bad argument");
    if (n == 0)
        return SKM_OK;
    SKM_REQUIRE(d_sum && d_nnz && (m == 0 || d_in), SKM_E_BADARG, "skm_matrix_row_stats: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    const int vec_ok = (((uintptr_t)d_in & 15) == 0 && ld % 4 == 0) ? 1 : 0;
    SKM_PROF(ctx, "k_matrix_row_stats");
    k_matrix_row_stats<<<skm_grid_cap(ctx, n, 16), 256, 0, ctx->stream>>>(n, m, d_in, ld, vec_ok, d_sum, d_nnz);
    return skm_check_launch("k_matrix_row_stats");
}

extern "C" int skm_apply_top2(skm_ctx *ctx, int64_t n, const int64_t *d_xrowptr, const uint32_t *d_xcolidx,
                              const uint32_t *d_xcounts, const uint64_t *d_xnormsq, int64_t m, int64_t ncols,
                              const uint32_t *d_ycolptr, const uint64_t *d_ypost, const uint64_t *d_ynormsq, int64_t row0,
                              int64_t row1, const uint32_t *d_row_order, uint32_t *d_idx, double *d_score, int64_t *d_dot)
{
    SKM_REQUIRE(ctx && n >= 0 && m >= 0 && ncols >= 0, SKM_E_BADARG, "skm_apply_top2: bad argument");
    SKM_REQUIRE(row0 >= 0 && row0 <= row1 && row1 <= n, SKM_E_BADARG, "skm_apply_top2: bad row range");
    SKM_REQUIRE(m < ((int64_t)1 << 32) - 1, SKM_E_OVERFLOW, "skm_apply_top2: m >= 2^32");
    const int64_t nrows = row1 - row0;
    if (nrows == 0)
        return SKM_OK;
    SKM_REQUIRE(d_xrowptr && d_xnormsq && d_idx && d_score && d_dot, SKM_E_BADARG, "skm_apply_top2: null array");
    SKM_REQUIRE(m == 0 || (d_ycolptr && d_ynormsq), SKM_E_BADARG, "skm_apply_top2: null Y array");
    SKM_HIP(hipSetDevice(ctx->device));
    void *p;
    SKM_TRY(skm_ws(ctx, WS_J, sizeof(double) * (size_t)(m + 1), &p));
    double *ynorm = (double *)p;
    if (m) {
        k_norms_f64<<<(unsigned)skm_ceil_div(m, 256), 256, 0, ctx->stream>>>(m, d_ynormsq, ynorm);
        SKM_TRY(skm_check_launch("k_norms_f64"));
    }
    SKM_TRY(skm_ws(ctx, WS_I, sizeof(ulonglong2) * (size_t)(ncols + 1), &p));
    ulonglong2 *desc = (ulonglong2 *)p;
    SKM_TRY(skm_ws(ctx, WS_H, sizeof(uint32_t) * (size_t)(nrows + 8), &p));
    uint32_t *over_list = (uint32_t *)p + 8, *over_count = (uint32_t *)p;
    SKM_HIP(hipMemsetAsync(over_count, 0, 4, ctx->stream));
    if (ncols && m) {
        SKM_REQUIRE(d_ypost, SKM_E_BADARG, "skm_apply_top2: null postings");
        SKM_PROF(ctx, "k_apply_columns");
        k_apply_columns<<<(unsigned)skm_ceil_div(ncols, 256), 256, 0, ctx->stream>>>(ncols, d_ycolptr, d_ypost, desc);
        SKM_TRY(skm_check_launch("k_apply_columns"));
    }
    {
        // (with no column or no family every entry is skipped: xcolidx may then only hold 0xFFFFFFFF)
        SKM_PROF(ctx, "k_apply_top2");
        k_apply_top2<<<skm_grid_cap(ctx, skm_ceil_div(nrows, ATB / 64), 16), ATB, 0, ctx->stream>>>(
            d_xrowptr, d_xcolidx, d_xcounts, d_xnormsq, m, (ncols && m) ? desc : nullptr, d_ypost, ynorm, row0, nrows, d_row_order, over_list,
            over_count, d_idx, d_score, (long long *)d_dot);
        SKM_TRY(skm_check_launch("k_apply_top2"));
    }
    int ach = 256;  // power of two >= m, at most ACH
    while (ach < ACH && ach < m)
        ach <<= 1;
    const size_t lds = sizeof(unsigned long long) * (size_t)ach;
    SKM_HIP(hipFuncSetAttribute((const void *)k_apply_top2_dense, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    SKM_PROF(ctx, "k_apply_top2_dense");
    // rows that touch more than 384 families (listed on the device; usually none: the workgroups find an empty list)
    k_apply_top2_dense<<<skm_grid_cap(ctx, nrows, 4), ATB, lds, ctx->stream>>>(d_xrowptr, d_xcolidx, d_xcounts, d_xnormsq, m, d_ycolptr,
                                                                              d_ypost, ynorm, row0, over_list, over_count, ach,
                                                                              d_idx, d_score, (long long *)d_dot);
    return skm_check_launch("k_apply_top2_dense");
}

extern "C" int skm_cosine_dense_f64(skm_ctx *ctx, int64_t n, int64_t m, int64_t k, const double *d_x, int64_t ldx,
                                    const double *d_y, int64_t ldy, int mode, double *d_out, int64_t ld)
{
    SKM_REQUIRE(ctx && n >= 0 && m >= 0 && k >= 0 && ldx >= k && ldy >= k, SKM_E_BADARG, "skm_cosine_dense_f64: bad argument");
    SKM_REQUIRE(ld >= m, SKM_E_BADARG, "skm_cosine_dense_f64: ld < m");
    SKM_REQUIRE(mode == 0 || mode == 1, SKM_E_BADARG, "skm_cosine_dense_f64: mode must be 0 or 1");
    if (n == 0 || m == 0)
        return SKM_OK;
    SKM_REQUIRE(d_x && d_y && d_out, SKM_E_BADARG, "skm_cosine_dense_f64: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    const int64_t kp = skm_ceil_div(k > 0 ? k : 1, FK) * FK;
    const bool same = d_x == d_y && ldx == ldy && n == m;
    void *p;
    SKM_TRY(skm_ws(ctx, WS_A, sizeof(double) * (size_t)n * (size_t)kp, &p));
    double *xn = (double *)p, *yn = xn;
    {
        SKM_PROF(ctx, "k_normalize_rows_f64");
        k_normalize_rows_f64<<<skm_grid_cap(ctx, n, 16), 256, 0, ctx->stream>>>(n, k, d_x, ldx, xn, kp);
        if (!same) {
            SKM_TRY(skm_ws(ctx, WS_B, sizeof(double) * (size_t)m * (size_t)kp, &p));
            yn = (double *)p;
            k_normalize_rows_f64<<<skm_grid_cap(ctx, m, 16), 256, 0, ctx->stream>>>(m, k, d_y, ldy, yn, kp);
        }
    }
    SKM_TRY(skm_check_launch("k_normalize_rows_f64"));
    dim3 grid((unsigned)skm_ceil_div(m, FN), (unsigned)skm_ceil_div(n, FM));
    SKM_PROF(ctx, "k_cosine_dense_f64");
    if (mode == 0)
        k_cosine_dense_f64<0><<<grid, 256, 0, ctx->stream>>>(n, m, kp, xn, yn, d_out, ld, same ? 1 : 0);
    else
        k_cosine_dense_f64<1><<<grid, 256, 0, ctx->stream>>>(n, m, kp, xn, yn, d_out, ld, same ? 1 : 0);
    return skm_check_launch("k_cosine_dense_f64");
}
