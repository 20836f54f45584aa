// Stage 2 of the hot path: observed basis, column ids, postings (column-major copy), dense views.
//
// Reference behaviour being replaced:
//   dict of observed k-mers, first-seen order, min_filter    snekmer/rules/kmerize.smk:89-104
//   np.unique over all k-mer lists                           snekmer/scripts/cluster_cluster.py:67
//   vecs[n][np.isin(kmerbasis, addvec)] = 1                  snekmer/rules/kmerize.smk:112-119
//   store = [k_counts.get(item, 0) for item in kmerlist]     snekmer/rules/learn.smk:376-383
//
// Device design: because a code is a pure function of (alphabet, k, window), columns are found by
// one stable LSD radix sort of the CSR's codes (rocPRIM building block, key bits = log2 |S|^k) with
// the entry index as payload.  The sorted order *is* the column-major copy of the matrix: one
// scatter pass then emits column ids, postings, document frequencies and first-seen keys.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "skm_common.h"
#include "skm_onesweep.h"
#include "skm_sort.h"

namespace {

constexpr int BLK = 256;

__global__ void k_set_u32(uint32_t *p, uint32_t v) { *p = v; }

// rowcount[e] = the posting word of CSR entry e (row | count << 32, or the 32-bit form row |
// min(count, 255) << 24 described in skm_gram_kernel.h), so that the pass over the sorted order fetches
// row and count with one gather; one wave per row.
template <typename PW>
__device__ __forceinline__ PW make_posting(uint32_t row, uint32_t count);
template <>
__device__ __forceinline__ uint64_t make_posting<uint64_t>(uint32_t row, uint32_t count)
{
    return (uint64_t)row | ((uint64_t)count << 32);
}
template <>
__device__ __forceinline__ uint32_t make_posting<uint32_t>(uint32_t row, uint32_t count)
{
    return row | ((count < 255u ? count : 255u) << 24);
}

template <typename PW>
__global__ __launch_bounds__(BLK) void k_expand_rowid(const int64_t *__restrict__ rowptr, int64_t n,
                                                      const uint32_t *__restrict__ counts, PW *__restrict__ rowcount)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += nwaves) {
        const int64_t b = rowptr[i], e = rowptr[i + 1];
        for (int64_t t = b + lane; t < e; t += 64)
            rowcount[t] = make_posting<PW>((uint32_t)i, counts[t]);
    }
}

template <typename K>
struct head_flag {
    const K *keys;
    __device__ uint32_t operator()(uint32_t t) const { return (t == 0 || keys[t] != keys[t - 1]) ? 1u : 0u; }
};

// One pass over the sorted order: column ids back to CSR order, postings, basis codes,
// column starts and first-seen keys.
template <typename K, typename PW>
__global__ __launch_bounds__(BLK) void k_basis_scatter(int64_t nnz, const K *__restrict__ skeys,
                                                       const uint32_t *__restrict__ sidx,
                                                       const uint32_t *__restrict__ colid1,
                                                       const PW *__restrict__ rowcount,
                                                       const uint32_t *__restrict__ counts,
                                                       const uint32_t *__restrict__ firstpos,
                                                       K *__restrict__ basis, uint32_t *__restrict__ colidx,
                                                       uint32_t *__restrict__ colptr, PW *__restrict__ post,
                                                       uint32_t *__restrict__ postcnt,
                                                       uint64_t *__restrict__ firstkey, int elide)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nnz; t += stride) {
        const uint32_t c = colid1[t] - 1u;
        const uint32_t e = sidx[t];
        const K key = skeys[t];
        const bool head = t == 0 || skeys[t - 1] != key;
        if (elide && head && (t == nnz - 1 || skeys[t + 1] != key)) {
            // a k-mer seen in one sequence only: no posting, colidx stays 0xFFFFFFFF (pre-filled)
            if (basis)
                basis[c] = key;
            if (colptr)
                colptr[c] = (uint32_t)t;
            continue;
        }
        colidx[e] = c;
        PW rc = 0;
        if (post || (head && firstkey))
            rc = rowcount[e];
        const uint32_t row = sizeof(PW) == 8 ? (uint32_t)rc : ((uint32_t)rc & 0x00FFFFFFu);
        if (post) {
            post[t] = rc;
            if (sizeof(PW) == 4 && ((uint32_t)rc >> 24) == 255u)  // escape: the real count goes to the side array
                postcnt[t] = counts[e];
        }
        if (head) {
            if (basis)
                basis[c] = key;
            if (colptr)
                colptr[c] = (uint32_t)t;
            if (firstkey)
                firstkey[c] = ((uint64_t)row << 32) | (firstpos ? firstpos[e] : 0u);
        }
    }
}

__global__ void k_col_stats(int64_t ncols, const uint32_t *__restrict__ colptr, const uint64_t *__restrict__ post,
                            uint32_t *__restrict__ df, uint64_t *__restrict__ total)
{
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols)
        return;
    uint32_t b = colptr[c], e = colptr[c + 1];
    if (df)
        df[c] = e - b;
    if (total) {
        uint64_t s = 0;
        for (uint32_t t = b; t < e; ++t)
            s += post[t] >> 32;
        total[c] = s;
    }
}

// wave per row: two largest scores and their columns (ties -> lower column)
__global__ __launch_bounds__(BLK) void k_row_top2(int64_t n, int64_t m, const float *__restrict__ sc, int64_t ld,
                                                  uint32_t *__restrict__ idx, float *__restrict__ val)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    auto better = [](float a, uint32_t ia, float b, uint32_t ib) { return a > b || (a == b && ia < ib); };
    for (int64_t i = wave; i < n; i += nwaves) {
        float v1 = -INFINITY, v2 = -INFINITY;
        uint32_t i1 = 0xFFFFFFFFu, i2 = 0xFFFFFFFFu;
        for (int64_t j = lane; j < m; j += 64) {
            const float x = sc[i * ld + j];
            const uint32_t jj = (uint32_t)j;
            if (better(x, jj, v1, i1)) {
                v2 = v1, i2 = i1;
                v1 = x, i1 = jj;
            } else if (better(x, jj, v2, i2)) {
                v2 = x, i2 = jj;
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ov1 = __shfl_down(v1, o), ov2 = __shfl_down(v2, o);
            const uint32_t oi1 = __shfl_down(i1, o), oi2 = __shfl_down(i2, o);
            // merge two (best, second) pairs
            if (better(ov1, oi1, v1, i1)) {
                if (better(v1, i1, ov2, oi2))
                    v2 = v1, i2 = i1;
                else
                    v2 = ov2, i2 = oi2;
                v1 = ov1, i1 = oi1;
            } else if (better(ov1, oi1, v2, i2)) {
                v2 = ov1, i2 = oi1;
            }
        }
        if (lane == 0) {
            idx[2 * i] = i1;
            val[2 * i] = i1 == 0xFFFFFFFFu ? 0.0f : v1;
            idx[2 * i + 1] = i2;
            val[2 * i + 1] = i2 == 0xFFFFFFFFu ? 0.0f : v2;
        }
    }
}

// KIND 0: 1 - hamming distance; KIND 1: Jaccard distance.  In place over the intersection sizes.
template <int KIND>
__global__ void k_setsim_from_gram(int64_t n, int64_t m, float inv_cols, const float *__restrict__ xc,
                                   const float *__restrict__ yc, float *__restrict__ out, int64_t ld)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m)
        return;
    const float cj = yc[j];
    for (int64_t i = blockIdx.y; i < n; i += gridDim.y) {
        const float g = out[i * ld + j];
        const float sym = xc[i] + cj - 2.0f * g;  // |a xor b|
        if (KIND == 0) {
            out[i * ld + j] = 1.0f - sym * inv_cols;
        } else {
            const float uni = xc[i] + cj - g;
            out[i * ld + j] = uni > 0.0f ? sym / uni : 0.0f;
        }
    }
}

// float64 forms for ANY matrix (counts, real values): both = #{c: x_ic != 0 and y_jc != 0}, equal = #{c: x_ic == y_jc != 0}
// (null for 0/1 input, where equal == both), xn / yn = non-zeros per row.  Columns on which two rows differ:
// xn + yn - both - equal.  KIND 0: 1 - hamming distance, count / ncols in double as scipy's hamming does;
// KIND 1: differ / #{c: x_ic != 0 or y_jc != 0}, 0 for two empty rows (scipy's Jaccard distance when fed the 0/1 pattern);
// KIND 2: the hamming distance itself.
template <int KIND>
__global__ void k_setsim_f64(int64_t n, int64_t m, double ncols, const uint32_t *__restrict__ xn,
                             const uint32_t *__restrict__ yn, const float *__restrict__ both,
                             const float *__restrict__ equal, int64_t ld, double *__restrict__ out, int64_t ld_out)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m)
        return;
    const double cj = (double)yn[j];
    for (int64_t i = blockIdx.y; i < n; i += gridDim.y) {
        const double b = (double)both[i * ld + j];
        const double e = equal ? (double)equal[i * ld + j] : b;
        const double differ = (double)xn[i] + cj - b - e;
        if (KIND == 0) {
            out[i * ld_out + j] = 1.0 - differ / ncols;
        } else if (KIND == 2) {
            out[i * ld_out + j] = differ / ncols;
        } else {
            const double uni = (double)xn[i] + cj - b;
            out[i * ld_out + j] = uni > 0.0 ? differ / uni : 0.0;
        }
    }
}

template <typename T>
__global__ void k_gather_columns(int64_t rows, int64_t ncols_out, const T *__restrict__ in, int64_t ld_in,
                                 const uint32_t *__restrict__ src, T *__restrict__ out)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= ncols_out)
        return;
    const uint32_t c = src[p];
    for (int64_t i = blockIdx.y; i < rows; i += gridDim.y)
        out[i * ncols_out + p] = c == 0xFFFFFFFFu ? T(0) : in[i * ld_in + c];
}

__global__ void k_widen_i8_u32(int64_t n, const int8_t *__restrict__ in, uint32_t *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride)
        out[i] = (uint32_t)(uint8_t)in[i];
}

__global__ void k_narrow_u32_u8(int64_t n, const uint32_t *__restrict__ in, uint8_t *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride)
        out[i] = (uint8_t)in[i];
}

__global__ void k_max_u32(int64_t n, const uint32_t *__restrict__ v, unsigned int *out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned int mx = 0;
    for (; i < n; i += stride)
        mx = max(mx, v[i]);
    for (int o = 32; o > 0; o >>= 1)
        mx = max(mx, (unsigned int)__shfl_down(mx, o));
    if ((threadIdx.x & 63) == 0 && mx)
        atomicMax(out, mx);
}

// the same over the entries [0, rowptr[n]) of a CSR whose entry count only the device knows
__global__ void k_max_u32_dev(const int64_t *__restrict__ rowptr, int64_t n, const uint32_t *__restrict__ v, unsigned int *out)
{
    const int64_t nnz = rowptr[n];
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned int mx = 0;
    for (; i < nnz; i += stride)
        mx = max(mx, v[i]);
    for (int o = 32; o > 0; o >>= 1)
        mx = max(mx, (unsigned int)__shfl_down(mx, o));
    if ((threadIdx.x & 63) == 0 && mx)
        atomicMax(out, mx);
}

__global__ void k_pair_work(int64_t ncols, const uint32_t *__restrict__ colptr, unsigned long long *out)
{
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0;
    if (c < ncols) {
        unsigned long long d = colptr[c + 1] - colptr[c];
        v = d * d;
    }
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_down(v, o);
    if ((threadIdx.x & 63) == 0 && v)
        atomicAdd(out, v);
}

// colptr[c] = first sorted position whose column id is >= c (c in [0, ncols]).
__global__ void k_colptr_search(int64_t ncols, int64_t nnz, const uint32_t *__restrict__ skeys,
                                uint32_t *__restrict__ colptr)
{
    int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c > ncols)
        return;
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
        int64_t mid = (lo + hi) >> 1;
        if (skeys[mid] < (uint32_t)c)
            lo = mid + 1;
        else
            hi = mid;
    }
    colptr[c] = (uint32_t)lo;
}

__global__ void k_gather_postings(int64_t nnz, const uint32_t *__restrict__ sidx,
                                  const uint64_t *__restrict__ rowcount, uint64_t *__restrict__ post)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nnz; t += stride) {
        uint32_t e = sidx[t];
        post[t] = rowcount[e];
    }
}

__global__ void k_rebase_rowptr(const int64_t *__restrict__ local, int64_t nrows, int64_t src0, int64_t dst0,
                                int64_t base, int last, int64_t *__restrict__ out)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nrows + (last ? 1 : 0))
        out[dst0 + i] = local[src0 + i] + base;
}

template <typename T>
__device__ __forceinline__ T dense_value(uint32_t v);
template <>
__device__ __forceinline__ double dense_value<double>(uint32_t v) { return (double)v; }
template <>
__device__ __forceinline__ float dense_value<float>(uint32_t v) { return (float)v; }
template <>
__device__ __forceinline__ int8_t dense_value<int8_t>(uint32_t v) { return (int8_t)(v > 127u ? 127u : v); }

template <typename T>
__global__ __launch_bounds__(BLK) void k_csr_to_dense(int64_t n, const int64_t *__restrict__ rowptr,
                                                      const uint32_t *__restrict__ colidx,
                                                      const uint32_t *__restrict__ counts,
                                                      const uint32_t *__restrict__ colmap, int64_t ncols_out,
                                                      int mode, T *__restrict__ out, int64_t ld)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += nwaves) {
        const int64_t b = rowptr[i], e = rowptr[i + 1];
        for (int64_t t = b + lane; t < e; t += 64) {
            uint32_t c = colidx[t];
            if (colmap)
                c = colmap[c];
            if (c == 0xFFFFFFFFu || (int64_t)c >= ncols_out)
                continue;
            out[i * ld + c] = dense_value<T>(mode ? 1u : counts[t]);
        }
    }
}

__global__ __launch_bounds__(BLK) void k_row_norms(int64_t n, const int64_t *__restrict__ rowptr,
                                                   const uint32_t *__restrict__ counts, float *__restrict__ rnorm,
                                                   uint64_t *__restrict__ normsq)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += nwaves) {
        const int64_t b = rowptr[i], e = rowptr[i + 1];
        unsigned long long s = 0;
        double sd = 0.0;  // the same sum in float64: tells when the exact one has left 64 bits (counts near 2^32)
        for (int64_t t = b + lane; t < e; t += 64) {
            unsigned long long v = counts[t];
            s += v * v;
            sd += (double)v * (double)v;
        }
        for (int o = 32; o > 0; o >>= 1) {
            s += __shfl_down(s, o);
            sd += __shfl_down(sd, o);
        }
        if (lane == 0) {
            const bool wrapped = sd >= 0x1p63;
            if (normsq)
                normsq[i] = wrapped ? ~0ull : s;  // saturates
            if (rnorm)
                rnorm[i] = wrapped ? (float)(1.0 / sqrt(sd)) : (s ? (float)(1.0 / sqrt((double)s)) : 1.0f);
        }
    }
}

// ------------------------------------------------------------------------------- device-sized basis stage
// The basis stage of skm_vectorize_csr: the entry count lives on the device (d_nnz), every launch is sized by the
// host-known capacity `cap`, and the head scan is fused into the scatter (per-block head counts, one small scan
// over the blocks, then every block rebuilds its local prefix): no rocPRIM scan pass over the entries, no
// column-id array written and re-read, no host round trip.  Singletons are elided (cosine pipeline form).
constexpr int HB = 2048;  // sorted positions per block: 256 threads x 8 consecutive positions

// SCAN: the last workgroup to finish (atomic ticket in a word that is zero between launches) also turns the per-block
// counts into exclusive prefixes, the number of columns and the closing colptr entry - what k_scan_blocks does in a launch
// of its own.  Taken for small inputs, where a step is a chain of launch-bound kernels (one dependent dispatch less).
template <typename K, bool SCAN>
__global__ __launch_bounds__(256) void k_head_count(const int64_t *__restrict__ d_nnz, const K *__restrict__ skeys,
                                                    uint32_t *__restrict__ blockheads, uint32_t *__restrict__ ticket,
                                                    int64_t *__restrict__ d_ncols, uint32_t *__restrict__ colptr)
{
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_carry;
    __shared__ int s_last;
    const int64_t nnz = *d_nnz;
    const int64_t base = (int64_t)blockIdx.x * HB;
    uint32_t heads = 0;
#pragma unroll
    for (int j = 0; j < HB / 256; ++j) {  // lane-consecutive positions: both loads coalesce
        const int64_t t = base + j * 256 + threadIdx.x;
        if (t < nnz)
            heads += (t == 0 || skeys[t] != skeys[t - 1]) ? 1u : 0u;
    }
    for (int o = 32; o > 0; o >>= 1)
        heads += __shfl_down(heads, o);
    if ((threadIdx.x & 63) == 0)
        s_w[threadIdx.x >> 6] = heads;
    __syncthreads();
    if (!SCAN) {
        if (threadIdx.x == 0)
            blockheads[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        return;
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&blockheads[blockIdx.x], s_w[0] + s_w[1] + s_w[2] + s_w[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        s_last = atomicAdd(ticket, 1u) == gridDim.x - 1u;
        s_carry = 0;
    }
    __syncthreads();
    if (!s_last)
        return;
    __threadfence();
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t nblocks = gridDim.x;
    for (int64_t b0 = 0; b0 < nblocks; b0 += 256) {
        const int64_t b = b0 + tid;
        const uint32_t v = b < nblocks ? __hip_atomic_load(&blockheads[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o)
                incl += up;
        }
        __syncthreads();  // the previous round's s_w is no longer read
        if (lane == 63)
            s_w[wid] = incl;
        __syncthreads();
        uint32_t before = s_carry, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            before += w < wid ? s_w[w] : 0u;
            total += s_w[w];
        }
        if (b < nblocks)
            blockheads[b] = before + incl - v;
        __syncthreads();
        if (tid == 0)
            s_carry += total;
        __syncthreads();
    }
    if (tid == 0) {
        *d_ncols = (int64_t)s_carry;
        if (colptr)
            colptr[s_carry] = (uint32_t)nnz;
        *ticket = 0u;
    }
}

// exclusive prefix of the per-block head counts (one workgroup), the number of columns and the closing colptr entry
__global__ __launch_bounds__(1024) void k_scan_blocks(int64_t nblocks, uint32_t *__restrict__ blockheads,
                                                      const int64_t *__restrict__ d_nnz, int64_t *__restrict__ d_ncols,
                                                      uint32_t *__restrict__ colptr)
{
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0)
        s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < nblocks; base += 1024) {
        const int64_t b = base + tid;
        const uint32_t v = b < nblocks ? blockheads[b] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o)
                incl += up;
        }
        if (lane == 63)
            s_w[wid] = incl;
        __syncthreads();
        uint32_t before = s_carry, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            before += w < wid ? s_w[w] : 0u;
            total += s_w[w];
        }
        if (b < nblocks)
            blockheads[b] = before + incl - v;
        __syncthreads();
        if (tid == 0)
            s_carry += total;
        __syncthreads();
    }
    if (tid == 0) {
        *d_ncols = (int64_t)s_carry;
        if (colptr)
            colptr[s_carry] = (uint32_t)*d_nnz;
    }
}

template <typename K>
__global__ __launch_bounds__(256) void k_basis_scatter_fused(const int64_t *__restrict__ d_nnz, const K *__restrict__ skeys,
                                                             const uint32_t *__restrict__ sidx,
                                                             const uint32_t *__restrict__ blockbase,
                                                             const uint64_t *__restrict__ rowcount, K *__restrict__ basis,
                                                             uint32_t *__restrict__ colidx, uint32_t *__restrict__ colptr,
                                                             uint64_t *__restrict__ post)
{
    // Three phases so that every global access is lane-consecutive: (1) the block's keys come into LDS with
    // coalesced loads; (2) every thread ranks 8 CONSECUTIVE positions out of LDS (blocked layout: what the scan
    // wants) and leaves a column word per position in LDS; (3) positions are walked lane-consecutively again for
    // the gathers and scatters.
    __shared__ K s_key[HB + 2];      // [0] = key in front of the block, [HB + 1] = key behind it
    __shared__ uint32_t s_col[HB];
    __shared__ uint8_t s_flag[HB];  // bit 0 = first position of its column, bit 1 = also the last (singleton)
    __shared__ uint32_t s_w[4];
    const int64_t nnz = *d_nnz;
    const int64_t base = (int64_t)blockIdx.x * HB;
    if (base >= nnz)  // uniform for the workgroup
        return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int live = (int)min((int64_t)HB, nnz - base);
    for (int z = tid; z < HB + 2; z += 256) {
        const int64_t t = base + z - 1;
        s_key[z] = (t >= 0 && t < nnz) ? skeys[t] : ~K(0);
    }
    __syncthreads();
    const int p0 = tid * 8;
    uint32_t hm = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (p0 + i < live && (base + p0 + i == 0 || s_key[p0 + i + 1] != s_key[p0 + i]))
            hm |= 1u << i;
    const uint32_t mine = (uint32_t)__popc(hm);
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if (lane >= o)
            incl += up;
    }
    if (lane == 63)
        s_w[wid] = incl;
    __syncthreads();
    uint32_t heads = blockbase[blockIdx.x] + incl - mine;  // heads in front of this thread's first position
#pragma unroll
    for (int w = 0; w < 4; ++w)
        heads += w < wid ? s_w[w] : 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (p0 + i >= live)
            break;
        const bool head = (hm >> i) & 1u;
        heads += head ? 1u : 0u;
        const bool last_of_col = base + p0 + i == nnz - 1 || s_key[p0 + i + 2] != s_key[p0 + i + 1];
        s_col[p0 + i] = heads - 1u;
        s_flag[p0 + i] = (uint8_t)((head ? 1 : 0) | (head && last_of_col ? 2 : 0));
    }
    __syncthreads();
    for (int z = tid; z < live; z += 256) {
        const int64_t t = base + z;
        const uint32_t c = s_col[z];
        const uint32_t f = s_flag[z];
        if (f & 1u) {
            basis[c] = s_key[z + 1];
            colptr[c] = (uint32_t)t;
        }
        if (f & 2u)  // the only position of its column: a k-mer of one sequence, no posting; colidx stays 0xFFFFFFFF
            continue;
        const uint32_t e = sidx[t];
        colidx[e] = c;
        post[t] = rowcount[e];
    }
}

template <typename K, typename PW>
int basis_impl(skm_ctx *ctx, int key_bits, int flags, int64_t n, int64_t nnz, const int64_t *d_rowptr, const K *d_codes,
               const uint32_t *d_counts, const uint32_t *d_firstpos, int64_t *h_ncols, K *d_basis, uint32_t *d_colidx,
               uint32_t *d_df, uint64_t *d_total, uint64_t *d_firstkey, uint32_t *d_fs_order, uint32_t *d_colptr,
               PW *d_post, uint32_t *d_postcnt)
{
    hipStream_t st = ctx->stream;
    void *p;
    SKM_TRY(skm_ws(ctx, WS_A, sizeof(K) * (size_t)nnz, &p));
    K *skeys = (K *)p;
    SKM_TRY(skm_ws(ctx, WS_C, sizeof(uint32_t) * (size_t)nnz, &p));
    uint32_t *sidx = (uint32_t *)p;
    SKM_TRY(skm_ws(ctx, WS_D, sizeof(uint32_t) * (size_t)nnz, &p));
    uint32_t *colid1 = (uint32_t *)p;
    SKM_TRY(skm_ws(ctx, WS_E, sizeof(PW) * (size_t)nnz, &p));
    PW *rowcount = (PW *)p;

    const bool need_stats = d_df || d_total;
    const bool need_fs = d_fs_order != nullptr;
    uint32_t *colptr = d_colptr;
    PW *post = d_post;
    if (need_stats && !colptr) {
        SKM_TRY(skm_ws(ctx, WS_F, sizeof(uint32_t) * (size_t)(nnz + 1), &p));
        colptr = (uint32_t *)p;
    }
    if (need_stats && d_total && !post) {
        SKM_TRY(skm_ws(ctx, WS_G, sizeof(PW) * (size_t)nnz, &p));
        post = (PW *)p;
    }
    uint64_t *firstkey = d_firstkey;
    if (need_fs && !firstkey) {
        SKM_TRY(skm_ws(ctx, WS_I, sizeof(uint64_t) * (size_t)nnz, &p));
        firstkey = (uint64_t *)p;
    }

    const int g_ent = skm_grid_cap(ctx, skm_ceil_div(nnz, BLK), 16);
    {
        SKM_PROF(ctx, "k_expand_rowid");
        k_expand_rowid<PW><<<skm_grid_cap(ctx, skm_ceil_div(n, BLK / 64), 16), BLK, 0, st>>>(d_rowptr, n, d_counts, rowcount);
    }
    SKM_TRY(skm_check_launch("k_expand_rowid"));
    SKM_TRY(sort_pairs<K>(ctx, d_codes, skeys, sidx, nnz, key_bits, "rocprim_radix_sort_codes"));
    {
        size_t tmp = 0;
        auto in = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), head_flag<K>{skeys});
        SKM_HIP(rocprim::inclusive_scan(nullptr, tmp, in, colid1, (size_t)nnz, rocprim::plus<uint32_t>(), st));
        SKM_TRY(skm_ws(ctx, WS_ROCPRIM, tmp, &p));
        SKM_PROF(ctx, "rocprim_scan_heads");
        SKM_HIP(rocprim::inclusive_scan(p, tmp, in, colid1, (size_t)nnz, rocprim::plus<uint32_t>(), st));
    }
    const int elide = (flags & SKM_BASIS_ELIDE_SINGLETONS) ? 1 : 0;
    if (elide)
        SKM_HIP(hipMemsetAsync(d_colidx, 0xFF, sizeof(uint32_t) * (size_t)nnz, st));
    {
        SKM_PROF(ctx, "k_basis_scatter");
        k_basis_scatter<K, PW><<<g_ent, BLK, 0, st>>>(nnz, skeys, sidx, colid1, rowcount, d_counts, d_firstpos, d_basis,
                                                       d_colidx, colptr, post, d_postcnt, firstkey, elide);
    }
    SKM_TRY(skm_check_launch("k_basis_scatter"));
    uint32_t *h_b = (uint32_t *)ctx->h_pinned;
    SKM_HIP(hipMemcpyAsync(h_b, colid1 + (nnz - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SKM_HIP(hipStreamSynchronize(st));
    const int64_t B = (int64_t)*h_b;
    *h_ncols = B;
    if (colptr) {
        k_set_u32<<<1, 1, 0, st>>>(colptr + B, (uint32_t)nnz);
    }
    if (need_stats) {
        SKM_PROF(ctx, "k_col_stats");
        if constexpr (sizeof(PW) == 8) {
            k_col_stats<<<(unsigned)skm_ceil_div(B, BLK), BLK, 0, st>>>(B, colptr, post, d_df, d_total);
        } else {  // 32-bit postings saturate the count: totals are not available (rejected by the entry point)
            k_col_stats<<<(unsigned)skm_ceil_div(B, BLK), BLK, 0, st>>>(B, colptr, nullptr, d_df, nullptr);
        }
        SKM_TRY(skm_check_launch("k_col_stats"));
    }
    if (need_fs) {
        // first-seen order = ascending (row, first window) key
        SKM_TRY(skm_ws(ctx, WS_A, sizeof(uint64_t) * (size_t)B, &p));  // skeys no longer needed
        uint64_t *fk_sorted = (uint64_t *)p;
        SKM_TRY(sort_pairs<uint64_t>(ctx, firstkey, fk_sorted, d_fs_order, B, 64, "rocprim_radix_sort_firstseen"));
    }
    return SKM_OK;
}

}  // namespace

extern "C" int skm_basis_build(skm_ctx *ctx, int code_bits, int key_bits, int flags, int64_t n, int64_t nnz,
                               const int64_t *d_rowptr, const void *d_codes, const uint32_t *d_counts,
                               const uint32_t *d_firstpos, int64_t *h_ncols, void *d_basis, uint32_t *d_colidx,
                               uint32_t *d_df, uint64_t *d_total, uint64_t *d_firstkey, uint32_t *d_fs_order,
                               uint32_t *d_colptr, void *d_post, uint32_t *d_postcnt)
{
    SKM_REQUIRE(ctx && h_ncols && n >= 0 && nnz >= 0, SKM_E_BADARG, "skm_basis_build: bad argument");
    SKM_REQUIRE(code_bits == 32 || code_bits == 64, SKM_E_BADARG, "skm_basis_build: code_bits must be 32 or 64");
    SKM_REQUIRE(nnz < ((int64_t)1 << 32) - 1, SKM_E_OVERFLOW, "skm_basis_build: nnz >= 2^32; split the batch");
    SKM_REQUIRE(!d_fs_order || d_firstpos, SKM_E_BADARG, "skm_basis_build: d_fs_order needs d_firstpos");
    SKM_REQUIRE(!(flags & SKM_BASIS_ELIDE_SINGLETONS) || (d_post && !d_df && !d_total && !d_firstkey && !d_fs_order),
                SKM_E_BADARG, "skm_basis_build: ELIDE_SINGLETONS needs postings and no df/total/first-seen outputs");
    const bool p32 = (flags & SKM_BASIS_POST32) != 0;
    SKM_REQUIRE(!p32 || (d_post && d_postcnt && !d_total && n <= ((int64_t)1 << 24)), SKM_E_BADARG,
                "skm_basis_build: POST32 needs d_post, d_postcnt, no d_total and fewer than 2^24 rows");
    *h_ncols = 0;
    if (nnz == 0) {
        if (d_colptr)
            SKM_HIP(hipMemsetAsync(d_colptr, 0, sizeof(uint32_t), ctx->stream));
        return SKM_OK;
    }
    SKM_REQUIRE(d_rowptr && d_codes && d_counts && d_colidx, SKM_E_BADARG, "skm_basis_build: null CSR array");
    if (key_bits <= 0 || key_bits > code_bits)
        key_bits = code_bits;
    SKM_HIP(hipSetDevice(ctx->device));
#define SKM_BASIS(K, PW)                                                                                                   \
    return basis_impl<K, PW>(ctx, key_bits, flags, n, nnz, d_rowptr, (const K *)d_codes, d_counts, d_firstpos, h_ncols,  \
                             (K *)d_basis, d_colidx, d_df, d_total, d_firstkey, d_fs_order, d_colptr, (PW *)d_post, d_postcnt)
    if (code_bits == 32) {
        if (p32)
            SKM_BASIS(uint32_t, uint32_t);
        SKM_BASIS(uint32_t, uint64_t);
    }
    if (p32)
        SKM_BASIS(uint64_t, uint32_t);
    SKM_BASIS(uint64_t, uint64_t);
#undef SKM_BASIS
}

extern "C" int skm_csr_transpose(skm_ctx *ctx, int64_t n, int64_t nnz, int64_t ncols, const int64_t *d_rowptr,
                                 const uint32_t *d_colidx, const uint32_t *d_counts, uint32_t *d_colptr,
                                 uint64_t *d_post)
{
    SKM_REQUIRE(ctx && d_colptr && n >= 0 && nnz >= 0 && ncols >= 0, SKM_E_BADARG, "skm_csr_transpose: bad argument");
    SKM_REQUIRE(nnz < ((int64_t)1 << 32) - 1 && ncols < ((int64_t)1 << 32) - 1, SKM_E_OVERFLOW,
                "skm_csr_transpose: nnz or ncols >= 2^32");
    SKM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (nnz == 0) {
        SKM_HIP(hipMemsetAsync(d_colptr, 0, sizeof(uint32_t) * (size_t)(ncols + 1), st));
        return SKM_OK;
    }
    SKM_REQUIRE(d_rowptr && d_colidx && d_counts && d_post, SKM_E_BADARG, "skm_csr_transpose: null array");
    void *p;
    SKM_TRY(skm_ws(ctx, WS_A, sizeof(uint32_t) * (size_t)nnz, &p));
    uint32_t *skeys = (uint32_t *)p;
    SKM_TRY(skm_ws(ctx, WS_C, sizeof(uint32_t) * (size_t)nnz, &p));
    uint32_t *sidx = (uint32_t *)p;
    SKM_TRY(skm_ws(ctx, WS_E, sizeof(uint64_t) * (size_t)nnz, &p));
    uint64_t *rowcount = (uint64_t *)p;
    const int g_ent = skm_grid_cap(ctx, skm_ceil_div(nnz, BLK), 16);
    k_expand_rowid<uint64_t><<<skm_grid_cap(ctx, skm_ceil_div(n, BLK / 64), 16), BLK, 0, st>>>(d_rowptr, n, d_counts, rowcount);
    int bits = 1;
    while (bits < 32 && ((int64_t)1 << bits) < ncols)
        ++bits;
    SKM_TRY(sort_pairs<uint32_t>(ctx, d_colidx, skeys, sidx, nnz, bits, "rocprim_radix_sort_cols"));
    {
        SKM_PROF(ctx, "k_colptr_search");
        k_colptr_search<<<(unsigned)skm_ceil_div(ncols + 1, BLK), BLK, 0, st>>>(ncols, nnz, skeys, d_colptr);
    }
    {
        SKM_PROF(ctx, "k_gather_postings");
        k_gather_postings<<<g_ent, BLK, 0, st>>>(nnz, sidx, rowcount, d_post);
    }
    return skm_check_launch("k_gather_postings");
}

extern "C" int skm_csr_concat_rowptr(skm_ctx *ctx, int nparts, const int64_t *h_nrows, const int64_t *h_nnz,
                                     const int64_t *d_local, int64_t *d_rowptr)
{
    SKM_REQUIRE(ctx && nparts >= 1 && h_nrows && h_nnz && d_local && d_rowptr, SKM_E_BADARG,
                "skm_csr_concat_rowptr: bad argument");
    SKM_HIP(hipSetDevice(ctx->device));
    int64_t src = 0, dst = 0, base = 0;
    SKM_PROF(ctx, "k_rebase_rowptr");
    for (int p = 0; p < nparts; ++p) {
        const int last = p == nparts - 1;
        const int64_t cnt = h_nrows[p] + (last ? 1 : 0);
        if (cnt > 0)
            k_rebase_rowptr<<<(unsigned)skm_ceil_div(cnt, BLK), BLK, 0, ctx->stream>>>(d_local, h_nrows[p], src, dst, base,
                                                                                    last, d_rowptr);
        src += h_nrows[p] + 1;
        dst += h_nrows[p];
        base += h_nnz[p];
    }
    return skm_check_launch("k_rebase_rowptr");
}

extern "C" int skm_csr_to_dense(skm_ctx *ctx, int64_t n, const int64_t *d_rowptr, const uint32_t *d_colidx,
                                const uint32_t *d_counts, const uint32_t *d_colmap, int64_t ncols_out, int mode,
                                int dtype, void *d_out, int64_t ld)
{
    SKM_REQUIRE(ctx && n >= 0 && ncols_out >= 0 && ld >= ncols_out, SKM_E_BADARG, "skm_csr_to_dense: bad argument");
    SKM_REQUIRE(dtype >= 0 && dtype <= 2, SKM_E_BADARG, "skm_csr_to_dense: dtype must be 0 (f64), 1 (f32) or 2 (i8)");
    if (n == 0 || ld == 0)
        return SKM_OK;
    SKM_REQUIRE(d_rowptr && d_out, SKM_E_BADARG, "skm_csr_to_dense: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    const size_t esz = dtype == 0 ? 8 : (dtype == 1 ? 4 : 1);
    SKM_HIP(hipMemsetAsync(d_out, 0, esz * (size_t)n * (size_t)ld, ctx->stream));
    const int grid = skm_grid_cap(ctx, skm_ceil_div(n, BLK / 64), 16);
    SKM_PROF(ctx, "k_csr_to_dense");
    if (dtype == 0)
        k_csr_to_dense<double><<<grid, BLK, 0, ctx->stream>>>(n, d_rowptr, d_colidx, d_counts, d_colmap, ncols_out, mode,
                                                               (double *)d_out, ld);
    else if (dtype == 1)
        k_csr_to_dense<float><<<grid, BLK, 0, ctx->stream>>>(n, d_rowptr, d_colidx, d_counts, d_colmap, ncols_out, mode,
                                                              (float *)d_out, ld);
    else
        k_csr_to_dense<int8_t><<<grid, BLK, 0, ctx->stream>>>(n, d_rowptr, d_colidx, d_counts, d_colmap, ncols_out, mode,
                                                               (int8_t *)d_out, ld);
    return skm_check_launch("k_csr_to_dense");
}

extern "C" int skm_row_norms_csr(skm_ctx *ctx, int64_t n, const int64_t *d_rowptr, const uint32_t *d_counts,
                                 float *d_rnorm, uint64_t *d_normsq)
{
    SKM_REQUIRE(ctx && n >= 0, SKM_E_BADARG, "skm_row_norms_csr: bad argument");
    if (n == 0)
        return SKM_OK;
    SKM_REQUIRE(d_rowptr, SKM_E_BADARG, "skm_row_norms_csr: null rowptr");
    SKM_HIP(hipSetDevice(ctx->device));
    SKM_PROF(ctx, "k_row_norms");
    k_row_norms<<<skm_grid_cap(ctx, skm_ceil_div(n, BLK / 64), 16), BLK, 0, ctx->stream>>>(n, d_rowptr, d_counts, d_rnorm,
                                                                                          d_normsq);
    return skm_check_launch("k_row_norms");
}

extern "C" int skm_row_top2(skm_ctx *ctx, int64_t n, int64_t m, const float *d_scores, int64_t ld, uint32_t *d_idx,
                            float *d_val)
{
    SKM_REQUIRE(ctx && n >= 0 && m >= 0 && ld >= m, SKM_E_BADARG, "skm_row_top2: bad argument");
    if (n == 0)
        return SKM_OK;
    SKM_REQUIRE(d_idx && d_val && (m == 0 || d_scores), SKM_E_BADARG, "skm_row_top2: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    SKM_PROF(ctx, "k_row_top2");
    k_row_top2<<<skm_grid_cap(ctx, skm_ceil_div(n, BLK / 64), 16), BLK, 0, ctx->stream>>>(n, m, d_scores, ld, d_idx, d_val);
    return skm_check_launch("k_row_top2");
}

// skm_csr_group_sum (learn aggregation): skm_learn.hip

extern "C" int skm_hamming_similarity_from_gram(skm_ctx *ctx, int64_t n, int64_t m, int64_t ncols, const float *d_xcount,
                                                const float *d_ycount, float *d_out, int64_t ld)
{
    SKM_REQUIRE(ctx && n >= 0 && m >= 0 && ncols > 0 && ld >= m, SKM_E_BADARG, "skm_hamming_similarity_from_gram: bad argument");
    if (n == 0 || m == 0)
        return SKM_OK;
    SKM_REQUIRE(d_xcount && d_ycount && d_out, SKM_E_BADARG, "skm_hamming_similarity_from_gram: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    dim3 grid((unsigned)skm_ceil_div(m, BLK), (unsigned)(n < 1024 ? n : 1024));
    SKM_PROF(ctx, "k_setsim_from_gram");
    k_setsim_from_gram<0><<<grid, BLK, 0, ctx->stream>>>(n, m, 1.0f / (float)ncols, d_xcount, d_ycount, d_out, ld);
    return skm_check_launch("k_setsim_from_gram");
}

extern "C" int skm_jaccard_distance_from_gram(skm_ctx *ctx, int64_t n, int64_t m, const float *d_xcount,
                                              const float *d_ycount, float *d_out, int64_t ld)
{
    SKM_REQUIRE(ctx && n >= 0 && m >= 0 && ld >= m, SKM_E_BADARG, "skm_jaccard_distance_from_gram: bad argument");
    if (n == 0 || m == 0)
        return SKM_OK;
    SKM_REQUIRE(d_xcount && d_ycount && d_out, SKM_E_BADARG, "skm_jaccard_distance_from_gram: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    dim3 grid((unsigned)skm_ceil_div(m, BLK), (unsigned)(n < 1024 ? n : 1024));
    SKM_PROF(ctx, "k_setsim_from_gram");
    k_setsim_from_gram<1><<<grid, BLK, 0, ctx->stream>>>(n, m, 0.0f, d_xcount, d_ycount, d_out, ld);
    return skm_check_launch("k_setsim_from_gram");
}

extern "C" int skm_setsim_f64(skm_ctx *ctx, int kind, int64_t n, int64_t m, int64_t ncols, const uint32_t *d_xnnz,
                              const uint32_t *d_ynnz, const float *d_both, const float *d_equal, int64_t ld, double *d_out,
                              int64_t ld_out)
{
    SKM_REQUIRE(ctx && kind >= 0 && kind <= 2 && n >= 0 && m >= 0 && ncols > 0 && ld >= m && ld_out >= m, SKM_E_BADARG,
                "skm_setsim_f64: bad argument");
    SKM_REQUIRE(ncols < ((int64_t)1 << 24), SKM_E_OVERFLOW, "skm_setsim_f64: 2^24 columns or more (float32 Gram cells would round)");
    if (n == 0 || m == 0)
        return SKM_OK;
    SKM_REQUIRE(d_xnnz && d_ynnz && d_both && d_out, SKM_E_BADARG, "skm_setsim_f64: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    dim3 grid((unsigned)skm_ceil_div(m, BLK), (unsigned)(n < 1024 ? n : 1024));
    SKM_PROF(ctx, "k_setsim_f64");
    if (kind == 0)
        k_setsim_f64<0><<<grid, BLK, 0, ctx->stream>>>(n, m, (double)ncols, d_xnnz, d_ynnz, d_both, d_equal, ld, d_out, ld_out);
    else if (kind == 1)
        k_setsim_f64<1><<<grid, BLK, 0, ctx->stream>>>(n, m, (double)ncols, d_xnnz, d_ynnz, d_both, d_equal, ld, d_out, ld_out);
    else
        k_setsim_f64<2><<<grid, BLK, 0, ctx->stream>>>(n, m, (double)ncols, d_xnnz, d_ynnz, d_both, d_equal, ld, d_out, ld_out);
    return skm_check_launch("k_setsim_f64");
}

extern "C" int skm_gather_columns(skm_ctx *ctx, int64_t rows, int64_t ncols_out, int elem_bytes, const void *d_in,
                                  int64_t ld_in, const uint32_t *d_src, void *d_out)
{
    SKM_REQUIRE(ctx && rows >= 0 && ncols_out >= 0 && ld_in >= 0, SKM_E_BADARG, "skm_gather_columns: bad argument");
    SKM_REQUIRE(elem_bytes == 1 || elem_bytes == 2 || elem_bytes == 4 || elem_bytes == 8, SKM_E_BADARG,
                "skm_gather_columns: elem_bytes must be 1, 2, 4 or 8");
    if (rows == 0 || ncols_out == 0)
        return SKM_OK;
    SKM_REQUIRE(d_src && d_out && (ld_in == 0 || d_in), SKM_E_BADARG, "skm_gather_columns: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    dim3 grid((unsigned)skm_ceil_div(ncols_out, BLK), (unsigned)(rows < 4096 ? rows : 4096));
    SKM_PROF(ctx, "k_gather_columns");
    if (elem_bytes == 8)
        k_gather_columns<uint64_t><<<grid, BLK, 0, ctx->stream>>>(rows, ncols_out, (const uint64_t *)d_in, ld_in, d_src, (uint64_t *)d_out);
    else if (elem_bytes == 4)
        k_gather_columns<uint32_t><<<grid, BLK, 0, ctx->stream>>>(rows, ncols_out, (const uint32_t *)d_in, ld_in, d_src, (uint32_t *)d_out);
    else if (elem_bytes == 2)
        k_gather_columns<uint16_t><<<grid, BLK, 0, ctx->stream>>>(rows, ncols_out, (const uint16_t *)d_in, ld_in, d_src, (uint16_t *)d_out);
    else
        k_gather_columns<uint8_t><<<grid, BLK, 0, ctx->stream>>>(rows, ncols_out, (const uint8_t *)d_in, ld_in, d_src, (uint8_t *)d_out);
    return skm_check_launch("k_gather_columns");
}

extern "C" int skm_widen_i8_u32(skm_ctx *ctx, int64_t count, const int8_t *d_in, uint32_t *d_out)
{
    SKM_REQUIRE(ctx && count >= 0, SKM_E_BADARG, "skm_widen_i8_u32: bad argument");
    if (count == 0)
        return SKM_OK;
    SKM_REQUIRE(d_in && d_out, SKM_E_BADARG, "skm_widen_i8_u32: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    k_widen_i8_u32<<<skm_grid_cap(ctx, skm_ceil_div(count, BLK), 16), BLK, 0, ctx->stream>>>(count, d_in, d_out);
    return skm_check_launch("k_widen_i8_u32");
}

extern "C" int skm_narrow_u32_u8(skm_ctx *ctx, int64_t count, const uint32_t *d_in, uint8_t *d_out)
{
    SKM_REQUIRE(ctx && count >= 0, SKM_E_BADARG, "skm_narrow_u32_u8: bad argument");
    if (count == 0)
        return SKM_OK;
    SKM_REQUIRE(d_in && d_out, SKM_E_BADARG, "skm_narrow_u32_u8: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    k_narrow_u32_u8<<<skm_grid_cap(ctx, skm_ceil_div(count, BLK), 16), BLK, 0, ctx->stream>>>(count, d_in, d_out);
    return skm_check_launch("k_narrow_u32_u8");
}

extern "C" int skm_widen_u8_u32(skm_ctx *ctx, int64_t count, const uint8_t *d_in, uint32_t *d_out)
{
    return skm_widen_i8_u32(ctx, count, (const int8_t *)d_in, d_out);
}

extern "C" int skm_csr_max_count(skm_ctx *ctx, int64_t nnz, const uint32_t *d_counts, uint32_t *h_max)
{
    SKM_REQUIRE(ctx && h_max && nnz >= 0, SKM_E_BADARG, "skm_csr_max_count: bad argument");
    *h_max = 0;
    if (nnz == 0)
        return SKM_OK;
    SKM_HIP(hipSetDevice(ctx->device));
    void *p;
    SKM_TRY(skm_ws(ctx, WS_SMALL, 4096, &p));
    unsigned int *acc = (unsigned int *)((uint8_t *)p + 1536);
    SKM_HIP(hipMemsetAsync(acc, 0, 4, ctx->stream));
    k_max_u32<<<skm_grid_cap(ctx, skm_ceil_div(nnz, BLK), 8), BLK, 0, ctx->stream>>>(nnz, d_counts, acc);
    SKM_TRY(skm_check_launch("k_max_u32"));
    SKM_HIP(hipMemcpyAsync(ctx->h_pinned, acc, 4, hipMemcpyDeviceToHost, ctx->stream));
    SKM_HIP(hipStreamSynchronize(ctx->stream));
    *h_max = *(uint32_t *)ctx->h_pinned;
    return SKM_OK;
}

extern "C" int skm_csr_max_count_dev(skm_ctx *ctx, int64_t n, int64_t cap_entries, const int64_t *d_rowptr,
                                     const uint32_t *d_counts, uint32_t *d_out_max)
{
    SKM_REQUIRE(ctx && n >= 0 && cap_entries >= 0 && d_rowptr && d_out_max, SKM_E_BADARG, "skm_csr_max_count_dev: bad argument");
    SKM_HIP(hipSetDevice(ctx->device));
    SKM_HIP(hipMemsetAsync(d_out_max, 0, 4, ctx->stream));
    if (cap_entries == 0)
        return SKM_OK;
    SKM_REQUIRE(d_counts, SKM_E_BADARG, "skm_csr_max_count_dev: null counts");
    k_max_u32_dev<<<skm_grid_cap(ctx, skm_ceil_div(cap_entries, BLK), 8), BLK, 0, ctx->stream>>>(d_rowptr, n, d_counts, d_out_max);
    return skm_check_launch("k_max_u32_dev");
}

extern "C" int skm_pair_work(skm_ctx *ctx, int64_t ncols, const uint32_t *d_colptr, uint64_t *h_pairs)
{
    SKM_REQUIRE(ctx && h_pairs && ncols >= 0, SKM_E_BADARG, "skm_pair_work: bad argument");
    *h_pairs = 0;
    if (ncols == 0)
        return SKM_OK;
    SKM_HIP(hipSetDevice(ctx->device));
    void *p;
    SKM_TRY(skm_ws(ctx, WS_SMALL, 4096, &p));
    unsigned long long *acc = (unsigned long long *)((uint8_t *)p + 1024);
    SKM_HIP(hipMemsetAsync(acc, 0, 8, ctx->stream));
    k_pair_work<<<(unsigned)skm_ceil_div(ncols, BLK), BLK, 0, ctx->stream>>>(ncols, d_colptr, acc);
    SKM_TRY(skm_check_launch("k_pair_work"));
    SKM_HIP(hipMemcpyAsync(ctx->h_pinned, acc, 8, hipMemcpyDeviceToHost, ctx->stream));
    SKM_HIP(hipStreamSynchronize(ctx->stream));
    *h_pairs = *(uint64_t *)ctx->h_pinned;
    return SKM_OK;
}

// The sort state the basis stage will use for `cap` keys (same slot, same size as skm_basis_stage_async asks for), so that
// the stage in front can have it cleared: *out_state = nullptr when the stage uses the vendor sort (which clears its own).
int skm_basis_sort_state(skm_ctx *ctx, int64_t cap, int key_bits, int code_bits, uint32_t **out_state, int64_t *out_words,
                         int *out_passes, int *out_key_bits)
{
    *out_state = nullptr;
    *out_words = 0;
    if (key_bits <= 0 || key_bits > code_bits)
        key_bits = code_bits;
    if (!skm_use_onesweep(cap))
        return SKM_OK;
    const int passes = (key_bits + 7) / 8;
    void *p;
    SKM_TRY(skm_ws(ctx, WS_ROCPRIM, skm_onesweep::state_bytes(cap, 8192, passes) + skm_onesweep::state_bytes(cap, 2048, passes), &p));
    *out_state = (uint32_t *)p;
    *out_words = (int64_t)((skm_onesweep::sort_state_bytes(cap, key_bits, code_bits / 8) + 3) / 4);
    if (out_passes)
        *out_passes = passes;
    if (out_key_bits)
        *out_key_bits = key_bits;
    return SKM_OK;
}

int skm_basis_stage_async(skm_ctx *ctx, int code_bits, int key_bits, int64_t cap, const int64_t *d_nnz, const void *d_codes,
                          const uint64_t *d_rowcount, void *d_basis, uint32_t *d_colidx, uint32_t *d_colptr, uint64_t *d_post,
                          int64_t *d_ncols, const skm_count_extras &prepared)
{
    hipStream_t st = ctx->stream;
    const size_t kb = (size_t)(code_bits / 8);
    void *p;
    SKM_TRY(skm_ws(ctx, WS_A, kb * (size_t)cap, &p));
    void *skeys = p;
    SKM_TRY(skm_ws(ctx, WS_C, sizeof(uint32_t) * (size_t)cap, &p));
    uint32_t *sidx = (uint32_t *)p;
    const int64_t nblocks = skm_ceil_div(cap, HB);
    SKM_TRY(skm_ws(ctx, WS_L, sizeof(uint32_t) * (size_t)(nblocks + 1), &p));
    uint32_t *blockheads = (uint32_t *)p;
    if (key_bits <= 0 || key_bits > code_bits)
        key_bits = code_bits;
    if (prepared.colidx_ff != d_colidx)  // (a fused call: the count stage wrote the marker beside every entry)
        SKM_HIP(hipMemsetAsync(d_colidx, 0xFF, sizeof(uint32_t) * (size_t)cap, st));
    if (skm_use_onesweep(cap)) {
        // the library's own sort: sized by the device-side entry count, 2 + (key_bits / 8) launches (skm_onesweep.h)
        const int passes = (key_bits + 7) / 8;
        SKM_TRY(skm_ws(ctx, WS_H, kb * (size_t)cap, &p));
        void *ktmp = p;
        SKM_TRY(skm_ws(ctx, WS_I, sizeof(uint32_t) * (size_t)cap, &p));
        uint32_t *vtmp = (uint32_t *)p;
        SKM_TRY(skm_ws(ctx, WS_ROCPRIM, skm_onesweep::state_bytes(cap, 8192, passes) + skm_onesweep::state_bytes(cap, 2048, passes), &p));
        const bool zeroed = prepared.zero == (uint32_t *)p &&
                            (size_t)prepared.zero_words * 4 >= skm_onesweep::sort_state_bytes(cap, key_bits, code_bits / 8);
        if (code_bits == 32)
            SKM_TRY(skm_onesweep::sort_pairs_dev<uint32_t>(ctx, d_nnz, cap, (const uint32_t *)d_codes, (uint32_t *)skeys, sidx,
                                                            (uint32_t *)ktmp, vtmp, p, key_bits, "onesweep_sort_codes", zeroed, zeroed && prepared.hist));
        else
            SKM_TRY(skm_onesweep::sort_pairs_dev<uint64_t>(ctx, d_nnz, cap, (const uint64_t *)d_codes, (uint64_t *)skeys, sidx,
                                                            (uint64_t *)ktmp, vtmp, p, key_bits, "onesweep_sort_codes", zeroed, zeroed && prepared.hist));
    } else {
        // rocPRIM sorts the capacity: entries past the device-side count carry the all-ones sentinel
        // (skm_count_stage_async); they are the last ones of the input and the sort is stable, so positions [0, nnz) of
        // the sorted order are exactly the entries.
        if (code_bits == 32)
            SKM_TRY(sort_pairs<uint32_t>(ctx, (const uint32_t *)d_codes, (uint32_t *)skeys, sidx, cap, key_bits, "rocprim_radix_sort_codes"));
        else
            SKM_TRY(sort_pairs<uint64_t>(ctx, (const uint64_t *)d_codes, (uint64_t *)skeys, sidx, cap, key_bits, "rocprim_radix_sort_codes"));
    }
    {
        SKM_PROF(ctx, "k_head_count");
        if (nblocks <= 256) {  // small inputs (<= 0.5 M entries): the scan over the blocks rides on the last workgroup of the count:
                               // one launch less; above, the blocks' same-address ticket atomics (~5 ns each) cost more than the launch
            SKM_TRY(skm_ws(ctx, WS_ZERO, 256, &p));
            uint32_t *ticket = (uint32_t *)p + 32;  // (words 0-6: the count stage's size-class counters; zero between launches)
            if (code_bits == 32)
                k_head_count<uint32_t, true><<<(unsigned)nblocks, 256, 0, st>>>(d_nnz, (const uint32_t *)skeys, blockheads, ticket, d_ncols, d_colptr);
            else
                k_head_count<uint64_t, true><<<(unsigned)nblocks, 256, 0, st>>>(d_nnz, (const uint64_t *)skeys, blockheads, ticket, d_ncols, d_colptr);
        } else {
            if (code_bits == 32)
                k_head_count<uint32_t, false><<<(unsigned)nblocks, 256, 0, st>>>(d_nnz, (const uint32_t *)skeys, blockheads, nullptr, nullptr, nullptr);
            else
                k_head_count<uint64_t, false><<<(unsigned)nblocks, 256, 0, st>>>(d_nnz, (const uint64_t *)skeys, blockheads, nullptr, nullptr, nullptr);
            k_scan_blocks<<<1, 1024, 0, st>>>(nblocks, blockheads, d_nnz, d_ncols, d_colptr);
        }
    }
    SKM_TRY(skm_check_launch("k_head_count"));
    {
        SKM_PROF(ctx, "k_basis_scatter");
        if (code_bits == 32)
            k_basis_scatter_fused<uint32_t><<<(unsigned)nblocks, 256, 0, st>>>(d_nnz, (const uint32_t *)skeys, sidx, blockheads, d_rowcount,
                                                                             (uint32_t *)d_basis, d_colidx, d_colptr, d_post);
        else
            k_basis_scatter_fused<uint64_t><<<(unsigned)nblocks, 256, 0, st>>>(d_nnz, (const uint64_t *)skeys, sidx, blockheads, d_rowcount,
                                                                             (uint64_t *)d_basis, d_colidx, d_colptr, d_post);
    }
    return skm_check_launch("k_basis_scatter");
}
