// Learn aggregation in column-major form.
//
// Reference behaviour being replaced:
//   per-annotation sums of the count rows (pandas groupby + sum over a dense N x B table)   snekmer/rules/learn.smk:385-408
//   the family-total table every query is scored against                                  snekmer/rules/apply.smk:278-289
//
// Until round 5 the sums were one 64-bit (group, column) vendor radix sort of ALL entries (six passes over 12 bytes per
// entry: 1.5 of the 2.4 ms at 100 k sequences), and the consumer (skm_apply_top2) then sorted the totals AGAIN to get them
// by column.  The vectorize stage already leaves the count matrix by column (postings: rows holding the k-mer, ascending),
// and that is the order the consumer wants the totals in, so the aggregation is done there, list by list:
//
//   k_gp_lists    a lane per column for lists of at most 8 postings (the bulk): groups looked up, sorted and merged in
//                 registers; lists of 9 .. 4096 postings (a family's conserved k-mers: tens of postings, ONE group) are then
//                 taken by the whole wave, one after the other: summed in a 512-slot LDS hash table keyed by the group, the
//                 distinct groups ranked and written in order
//   k_gp_long     workgroup per column, anything longer (low-complexity k-mers) or with more than 384 distinct groups: dense
//                 LDS counters over 16384 groups at a time, ordered emission
//   k_gp_compact  the merged lists, left at the front of each column's own slot, packed to the exclusive scan of their
//                 lengths; per-group squared norms and entry counts summed in LDS first (a thousand hot addresses otherwise)
//
// skm_postings_to_csr turns a column-major matrix into CSR with ascending columns (one stable counting sort by row, the
// library's own one-sweep): the [groups x columns] table of learn.smk in the layout skm_csr_group_sum has always returned.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "skm_common.h"
#include "skm_onesweep.h"
#include "skm_sort.h"

namespace {

constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int BLK = 256;
constexpr int GP_SHORT = 8;
constexpr int GP_MID_MAX = 4096;
constexpr int GP_HS = 512, GP_HCAP = 384;  // hash slots per wave / distinct groups a wave holds
constexpr int GP_LCH = 16384;              // dense counters per pass of k_gp_long

// Columns in wave order: a lane owns one column of the wave's 64.  Lists of at most GP_SHORT postings are merged by their
// lane in registers; the longer ones (ballot) are then taken one after the other by the whole wave through its hash table.
// (A first version listed the longer columns with one atomic each for a second kernel: 300 k atomics on one address were
// 1.6 of its 1.9 ms.)
__global__ __launch_bounds__(BLK) void k_gp_lists(int64_t ncols, const uint32_t *__restrict__ colptr,
                                                  const uint64_t *__restrict__ post, const uint32_t *__restrict__ group,
                                                  uint64_t *__restrict__ tmp, uint32_t *__restrict__ cnt,
                                                  uint32_t *__restrict__ long_list, uint32_t *__restrict__ counters)
{
    constexpr int NW = BLK / 64;
    __shared__ uint32_t s_key[NW][GP_HS];
    __shared__ uint32_t s_val[NW][GP_HS];
    __shared__ uint64_t s_ent[NW][GP_HCAP];
    __shared__ uint32_t s_distinct[NW];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    uint32_t *keys = s_key[wid], *vals = s_val[wid];
    uint64_t *ent = s_ent[wid];
    for (int z = lane; z < GP_HS; z += 64) {
        keys[z] = NONE;
        vals[z] = 0u;
    }
    if (lane == 0)
        s_distinct[wid] = 0u;
    __threadfence_block();
    const unsigned long long lt = (1ull << lane) - 1ull;
    const int64_t wave = (int64_t)blockIdx.x * NW + wid, nwaves = (int64_t)gridDim.x * NW;
    for (int64_t cb = wave * 64; cb <= ncols; cb += nwaves * 64) {
        const int64_t c = cb + lane;
        uint32_t pb = 0, pe = 0;
        if (c < ncols) {
            pb = colptr[c];
            pe = colptr[c + 1];
        }
        const uint32_t df = pe - pb;
        if (c == ncols)
            cnt[c] = 0;  // the scan's last element
        if (c < ncols && df == 0)
            cnt[c] = 0;
        if (df >= 1 && df <= (uint32_t)GP_SHORT) {
            uint32_t g[GP_SHORT], v[GP_SHORT];
            uint64_t pw[GP_SHORT];
#pragma unroll
            for (int q = 0; q < GP_SHORT; ++q)  // all posting loads, then all group gathers, are in flight together
                pw[q] = (uint32_t)q < df ? post[pb + q] : 0ull;
#pragma unroll
            for (int q = 0; q < GP_SHORT; ++q) {
                g[q] = (uint32_t)q < df ? group[(uint32_t)pw[q]] : NONE;
                v[q] = (uint32_t)(pw[q] >> 32);
            }
            // ascending by group (unused slots hold 0xFFFFFFFF and stay behind), then equal neighbours merged
#pragma unroll
            for (int a = 0; a < GP_SHORT - 1; ++a)
#pragma unroll
                for (int j = 0; j < GP_SHORT - 1 - a; ++j)
                    if (g[j] > g[j + 1]) {
                        const uint32_t tg = g[j], tv = v[j];
                        g[j] = g[j + 1], v[j] = v[j + 1];
                        g[j + 1] = tg, v[j + 1] = tv;
                    }
            uint32_t d = 0, cur_g = NONE, cur_v = 0;
#pragma unroll
            for (int j = 0; j < GP_SHORT; ++j) {
                if (g[j] != NONE) {
                    if (g[j] == cur_g) {
                        cur_v += v[j];
                    } else {
                        if (cur_g != NONE)
                            tmp[pb + d++] = (uint64_t)cur_g | ((uint64_t)cur_v << 32);
                        cur_g = g[j];
                        cur_v = v[j];
                    }
                }
            }
            if (cur_g != NONE)
                tmp[pb + d++] = (uint64_t)cur_g | ((uint64_t)cur_v << 32);
            cnt[c] = d;
        }
        if (df > (uint32_t)GP_MID_MAX)
            long_list[atomicAdd(&counters[1], 1u)] = (uint32_t)c;
        unsigned long long mid = __ballot(df > (uint32_t)GP_SHORT && df <= (uint32_t)GP_MID_MAX);
        while (mid) {  // wave-uniform
            const int src = __ffsll((long long)mid) - 1;
            mid &= mid - 1ull;
            const uint32_t mc = (uint32_t)(cb + src);
            const uint32_t mb = __shfl(pb, src), me = __shfl(pe, src);
            for (uint32_t p = mb + (uint32_t)lane; p < me; p += 64) {
                const uint64_t pw = post[p];
                const uint32_t gg = group[(uint32_t)pw], vv = (uint32_t)(pw >> 32);
                uint32_t h = (gg * 2654435761u) >> (32 - 9);
                for (int probe = 0; probe < GP_HS; ++probe) {
                    uint32_t seen = __atomic_load_n(&keys[h], __ATOMIC_RELAXED);
                    if (seen == NONE) {
                        seen = atomicCAS(&keys[h], NONE, gg);
                        if (seen == NONE) {
                            seen = gg;
                            atomicAdd(&s_distinct[wid], 1u);
                        }
                    }
                    if (seen == gg) {
                        atomicAdd(&vals[h], vv);
                        break;
                    }
                    h = (h + 1) & (GP_HS - 1);
                }
            }
            __threadfence_block();
            const uint32_t distinct = __atomic_load_n(&s_distinct[wid], __ATOMIC_RELAXED);  // wave-uniform
            // pack the table's entries (and clear it for the next column)
            uint32_t d = 0;
            for (int z = 0; z < GP_HS / 64; ++z) {
                const int slot = z * 64 + lane;
                const uint32_t key = keys[slot];
                const bool has = key != NONE;
                const unsigned long long bal = __ballot(has);
                if (has && distinct <= (uint32_t)GP_HCAP)
                    ent[d + (uint32_t)__popcll(bal & lt)] = (uint64_t)key | ((uint64_t)vals[slot] << 32);
                if (has) {
                    keys[slot] = NONE;
                    vals[slot] = 0u;
                }
                d += (uint32_t)__popcll(bal);
            }
            if (lane == 0)
                s_distinct[wid] = 0u;
            __threadfence_block();
            if (distinct > (uint32_t)GP_HCAP) {  // more groups than the table may hold: the dense kernel takes the column
                if (lane == 0)
                    long_list[atomicAdd(&counters[1], 1u)] = mc;
                continue;
            }
            for (uint32_t e = (uint32_t)lane; e < d; e += 64) {
                const uint64_t mine = ent[e];
                uint32_t rank = 0;
                for (uint32_t j = 0; j < d; ++j)
                    rank += (uint32_t)ent[j] < (uint32_t)mine ? 1u : 0u;
                tmp[mb + rank] = mine;
            }
            if (lane == 0)
                cnt[mc] = d;
            __threadfence_block();
        }
    }
}

__global__ __launch_bounds__(BLK) void k_gp_long(const uint32_t *__restrict__ colptr, const uint64_t *__restrict__ post,
                                                 const uint32_t *__restrict__ group, int64_t ngroups, uint64_t *__restrict__ tmp,
                                                 uint32_t *__restrict__ cnt, const uint32_t *__restrict__ long_list,
                                                 const uint32_t *__restrict__ counters)
{
    __shared__ uint32_t s_c[GP_LCH];
    __shared__ uint32_t s_wcnt[BLK / 64];
    __shared__ uint32_t s_out;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t nlong = counters[1];
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t idx = blockIdx.x; idx < nlong; idx += gridDim.x) {
        const uint32_t c = long_list[idx];
        const uint32_t pb = colptr[c], pe = colptr[c + 1];
        if (tid == 0)
            s_out = 0u;
        for (int64_t g0 = 0; g0 < ngroups; g0 += GP_LCH) {
            const int span = (int)min((int64_t)GP_LCH, ngroups - g0);
            for (int z = tid; z < span; z += BLK)
                s_c[z] = 0u;
            __syncthreads();
            for (uint32_t p = pb + (uint32_t)tid; p < pe; p += BLK) {
                const uint64_t pw = post[p];
                const int64_t a = (int64_t)group[(uint32_t)pw] - g0;
                if (a >= 0 && a < span)
                    atomicAdd(&s_c[a], (uint32_t)(pw >> 32));
            }
            __syncthreads();
            for (int a0 = 0; a0 < span; a0 += BLK) {  // ordered emission, 256 groups per round
                const int a = a0 + tid;
                const uint32_t val = a < span ? s_c[a] : 0u;
                const bool has = val != 0u;
                const unsigned long long bal = __ballot(has);
                if (lane == 0)
                    s_wcnt[wid] = (uint32_t)__popcll(bal);
                __syncthreads();
                uint32_t before = s_out, total = 0;
                for (int w = 0; w < BLK / 64; ++w) {
                    before += w < wid ? s_wcnt[w] : 0u;
                    total += s_wcnt[w];
                }
                if (has)
                    tmp[pb + before + (uint32_t)__popcll(bal & lt)] = (uint64_t)(uint32_t)(g0 + a) | ((uint64_t)val << 32);
                __syncthreads();
                if (tid == 0)
                    s_out += total;
                __syncthreads();
            }
        }
        if (tid == 0)
            cnt[c] = s_out;
        __syncthreads();
    }
}

constexpr int GP_NACC = 4096;  // groups whose norms / entry counts a workgroup of k_gp_compact sums in LDS

__global__ __launch_bounds__(BLK) void k_gp_compact(int64_t ncols, int64_t cols_per_block, const uint32_t *__restrict__ colptr,
                                                    const uint64_t *__restrict__ tmp, const uint32_t *__restrict__ cnt,
                                                    const uint32_t *__restrict__ out_colptr, uint64_t *__restrict__ out_post,
                                                    int64_t ngroups, unsigned long long *__restrict__ out_normsq,
                                                    uint32_t *__restrict__ out_rowcount)
{
    __shared__ unsigned long long s_sq[GP_NACC];
    __shared__ uint32_t s_n[GP_NACC];
    const bool local = ngroups <= GP_NACC;
    const int tid = threadIdx.x;
    if (local) {
        for (int z = tid; z < (int)ngroups; z += BLK) {
            s_sq[z] = 0ull;
            s_n[z] = 0u;
        }
        __syncthreads();
    }
    const int64_t c0 = (int64_t)blockIdx.x * cols_per_block, c1 = min(ncols, c0 + cols_per_block);
    for (int64_t c = c0 + tid; c < c1; c += BLK) {
        const uint32_t d = cnt[c];
        const uint64_t *src = tmp + colptr[c];
        uint64_t *dst = out_post + out_colptr[c];
        for (uint32_t j = 0; j < d; ++j) {
            const uint64_t w = src[j];
            dst[j] = w;
            const uint32_t g = (uint32_t)w;
            const unsigned long long v = w >> 32;
            if (local) {
                if (out_normsq)
                    atomicAdd(&s_sq[g], v * v);
                if (out_rowcount)
                    atomicAdd(&s_n[g], 1u);
            } else {
                if (out_normsq)
                    atomicAdd(&out_normsq[g], v * v);
                if (out_rowcount)
                    atomicAdd(&out_rowcount[g], 1u);
            }
        }
    }
    if (local) {
        __syncthreads();
        for (int z = tid; z < (int)ngroups; z += BLK) {
            if (out_normsq && s_sq[z])
                atomicAdd(&out_normsq[z], s_sq[z]);
            if (out_rowcount && s_n[z])
                atomicAdd(&out_rowcount[z], s_n[z]);
        }
    }
}

// ---- column-major -> CSR
// rowkey[p] = the posting's row (the sort key), colval[p] = column | value << 32 (what the sorted order gathers: one
// random access per entry)
__global__ __launch_bounds__(BLK) void k_pc_keys(int64_t ncols, const uint32_t *__restrict__ colptr, const uint64_t *__restrict__ post,
                                                 uint32_t *__restrict__ rowkey, uint64_t *__restrict__ colval)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * BLK + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * BLK) >> 6;
    // 64 columns per wave step: a lane takes one (short lists are the rule); long lists are finished by the whole wave
    for (int64_t cb = wave * 64; cb < ncols; cb += nwaves * 64) {
        const int64_t c = cb + lane;
        uint32_t pb = 0, pe = 0;
        if (c < ncols) {
            pb = colptr[c];
            pe = colptr[c + 1];
        }
        const uint32_t mine = min(pe, pb + 8u);
        for (uint32_t p = pb; p < mine; ++p) {
            const uint64_t pw = post[p];
            rowkey[p] = (uint32_t)pw;
            colval[p] = (uint64_t)(uint32_t)c | (pw & 0xFFFFFFFF00000000ull);
        }
        unsigned long long more = __ballot(pe > mine);
        while (more) {
            const int src = __ffsll((long long)more) - 1;
            more &= more - 1ull;
            const uint32_t b = __shfl(mine, src), e = __shfl(pe, src);
            const uint32_t cc = (uint32_t)(cb + src);
            for (uint32_t p = b + (uint32_t)lane; p < e; p += 64) {
                const uint64_t pw = post[p];
                rowkey[p] = (uint32_t)pw;
                colval[p] = (uint64_t)cc | (pw & 0xFFFFFFFF00000000ull);
            }
        }
    }
}

__global__ void k_rowptr_search(int64_t nrows, int64_t nnz, const uint32_t *__restrict__ skeys, int64_t *__restrict__ rowptr)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > nrows)
        return;
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (skeys[mid] < (uint32_t)r)
            lo = mid + 1;
        else
            hi = mid;
    }
    rowptr[r] = lo;
}

__global__ void k_pc_gather(int64_t nnz, const uint32_t *__restrict__ sidx, const uint64_t *__restrict__ colval,
                            uint32_t *__restrict__ out_col, uint32_t *__restrict__ out_val)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nnz; t += stride) {
        const uint64_t w = colval[sidx[t]];
        out_col[t] = (uint32_t)w;
        out_val[t] = (uint32_t)(w >> 32);
    }
}

__global__ void k_set_i64(int64_t *dst, int64_t value) { *dst = value; }

__global__ void k_max_u32_plain(int64_t n, const uint32_t *__restrict__ v, unsigned int *__restrict__ acc)
{
    uint32_t m = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        m = max(m, v[i]);
    for (int o = 32; o > 0; o >>= 1)
        m = max(m, (uint32_t)__shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m)
        atomicMax(acc, m);
}

}  // namespace

extern "C" int skm_group_postings(skm_ctx *ctx, int64_t n, int64_t ncols, const uint32_t *d_colptr, const uint64_t *d_post,
                                  int64_t nnz, const uint32_t *d_group, int64_t ngroups, uint32_t *d_out_colptr,
                                  uint64_t *d_out_post, uint64_t *d_out_normsq, uint32_t *d_out_rowcount, int64_t *h_out_nnz)
{
    SKM_REQUIRE(ctx && n >= 0 && ncols >= 0 && nnz >= 0 && ngroups >= 0 && d_out_colptr && h_out_nnz, SKM_E_BADARG,
                "skm_group_postings: bad argument");
    SKM_REQUIRE(nnz < ((int64_t)1 << 32) - 1 && ncols < ((int64_t)1 << 32) - 1 && ngroups < ((int64_t)1 << 32) - 2, SKM_E_OVERFLOW,
                "skm_group_postings: too large");
    SKM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    *h_out_nnz = 0;
    if (d_out_normsq && ngroups)
        SKM_HIP(hipMemsetAsync(d_out_normsq, 0, sizeof(uint64_t) * (size_t)ngroups, st));
    if (d_out_rowcount && ngroups)
        SKM_HIP(hipMemsetAsync(d_out_rowcount, 0, sizeof(uint32_t) * (size_t)ngroups, st));
    if (nnz == 0 || ncols == 0) {
        SKM_HIP(hipMemsetAsync(d_out_colptr, 0, sizeof(uint32_t) * (size_t)(ncols + 1), st));
        return SKM_OK;
    }
    SKM_REQUIRE(d_colptr && d_post && d_group && d_out_post, SKM_E_BADARG, "skm_group_postings: null array");
    void *p;
    SKM_TRY(skm_ws(ctx, WS_A, sizeof(uint64_t) * (size_t)nnz, &p));
    uint64_t *tmp = (uint64_t *)p;
    SKM_TRY(skm_ws(ctx, WS_B, sizeof(uint32_t) * (size_t)(ncols + 1), &p));
    uint32_t *cnt = (uint32_t *)p;
    SKM_TRY(skm_ws(ctx, WS_D, sizeof(uint32_t) * (size_t)ncols, &p));
    uint32_t *long_list = (uint32_t *)p;
    SKM_TRY(skm_ws(ctx, WS_SMALL, 4096, &p));
    uint32_t *counters = (uint32_t *)((uint8_t *)p + 2048);
    SKM_HIP(hipMemsetAsync(counters, 0, 8, st));
    {
        SKM_PROF(ctx, "k_gp_lists");
        k_gp_lists<<<skm_grid_cap(ctx, skm_ceil_div(ncols + 1, BLK), 32), BLK, 0, st>>>(ncols, d_colptr, d_post, d_group, tmp, cnt, long_list, counters);
    }
    SKM_TRY(skm_check_launch("k_gp_lists"));
    {
        SKM_PROF(ctx, "k_gp_long");
        k_gp_long<<<skm_grid_cap(ctx, ncols, 2), BLK, 0, st>>>(d_colptr, d_post, d_group, ngroups, tmp, cnt, long_list, counters);
    }
    SKM_TRY(skm_check_launch("k_gp_long"));
    {
        size_t bytes = 0;
        SKM_HIP(rocprim::exclusive_scan(nullptr, bytes, cnt, d_out_colptr, 0u, (size_t)(ncols + 1), rocprim::plus<uint32_t>(), st));
        SKM_TRY(skm_ws(ctx, WS_ROCPRIM, bytes, &p));
        SKM_PROF(ctx, "rocprim_scan_group_lists");
        SKM_HIP(rocprim::exclusive_scan(p, bytes, cnt, d_out_colptr, 0u, (size_t)(ncols + 1), rocprim::plus<uint32_t>(), st));
    }
    {
        const int grid = skm_grid_cap(ctx, skm_ceil_div(ncols, BLK), 8);
        const int64_t per = skm_ceil_div(ncols, grid);
        SKM_PROF(ctx, "k_gp_compact");
        k_gp_compact<<<grid, BLK, 0, st>>>(ncols, per, d_colptr, tmp, cnt, d_out_colptr, d_out_post, ngroups,
                                          (unsigned long long *)d_out_normsq, d_out_rowcount);
    }
    SKM_TRY(skm_check_launch("k_gp_compact"));
    SKM_HIP(hipMemcpyAsync(ctx->h_pinned, d_out_colptr + ncols, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SKM_HIP(hipStreamSynchronize(st));
    *h_out_nnz = (int64_t)*(uint32_t *)ctx->h_pinned;
    return skm_check_device_error(ctx, "skm_group_postings");
}

extern "C" int skm_postings_to_csr(skm_ctx *ctx, int64_t ncols, int64_t nnz, const uint32_t *d_colptr, const uint64_t *d_post,
                                   int64_t nrows, int64_t *d_out_rowptr, uint32_t *d_out_col, uint32_t *d_out_val)
{
    SKM_REQUIRE(ctx && ncols >= 0 && nnz >= 0 && nrows >= 0 && d_out_rowptr, SKM_E_BADARG, "skm_postings_to_csr: bad argument");
    SKM_REQUIRE(nnz < ((int64_t)1 << 32) - 1 && nrows < ((int64_t)1 << 32) - 1, SKM_E_OVERFLOW, "skm_postings_to_csr: too large");
    SKM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if (nnz == 0) {
        SKM_HIP(hipMemsetAsync(d_out_rowptr, 0, sizeof(int64_t) * (size_t)(nrows + 1), st));
        return SKM_OK;
    }
    SKM_REQUIRE(d_colptr && d_post && d_out_col && d_out_val, SKM_E_BADARG, "skm_postings_to_csr: null array");
    void *p;
    SKM_TRY(skm_ws(ctx, WS_E, sizeof(uint32_t) * (size_t)nnz, &p));
    uint32_t *rowkey = (uint32_t *)p;
    SKM_TRY(skm_ws(ctx, WS_F, sizeof(uint32_t) * (size_t)nnz, &p));
    uint32_t *skeys = (uint32_t *)p;
    SKM_TRY(skm_ws(ctx, WS_G, sizeof(uint32_t) * (size_t)nnz, &p));
    uint32_t *sidx = (uint32_t *)p;
    SKM_TRY(skm_ws(ctx, WS_H, sizeof(uint64_t) * (size_t)nnz, &p));
    uint64_t *colval = (uint64_t *)p;
    {
        SKM_PROF(ctx, "k_pc_keys");
        k_pc_keys<<<skm_grid_cap(ctx, skm_ceil_div(ncols, BLK), 16), BLK, 0, st>>>(ncols, d_colptr, d_post, rowkey, colval);
    }
    SKM_TRY(skm_check_launch("k_pc_keys"));
    int bits = 1;
    while (bits < 32 && ((int64_t)1 << bits) < nrows)
        ++bits;
    if (nnz < ((int64_t)1 << 30)) {
        // the library's own one-sweep: stable, so the columns of a row stay in the ascending order the lists were walked in
        const int passes = (bits + 7) / 8;
        SKM_TRY(skm_ws(ctx, WS_I, sizeof(uint32_t) * (size_t)nnz, &p));
        uint32_t *ktmp = (uint32_t *)p;
        SKM_TRY(skm_ws(ctx, WS_J, sizeof(uint32_t) * (size_t)nnz, &p));
        uint32_t *vtmp = (uint32_t *)p;
        SKM_TRY(skm_ws(ctx, WS_K, skm_onesweep::state_bytes(nnz, 2048, passes) + 64, &p));
        int64_t *d_n = (int64_t *)p;
        void *state = (uint8_t *)p + 64;
        k_set_i64<<<1, 1, 0, st>>>(d_n, nnz);
        SKM_TRY(skm_onesweep::sort_pairs_dev<uint32_t>(ctx, d_n, nnz, rowkey, skeys, sidx, ktmp, vtmp, state, bits, "onesweep_sort_rows"));
    } else {
        SKM_TRY(sort_pairs<uint32_t>(ctx, rowkey, skeys, sidx, nnz, bits, "rocprim_radix_sort_rows"));
    }
    {
        SKM_PROF(ctx, "k_rowptr_search");
        k_rowptr_search<<<(unsigned)skm_ceil_div(nrows + 1, BLK), BLK, 0, st>>>(nrows, nnz, skeys, d_out_rowptr);
    }
    {
        SKM_PROF(ctx, "k_pc_gather");
        k_pc_gather<<<skm_grid_cap(ctx, skm_ceil_div(nnz, BLK), 16), BLK, 0, st>>>(nnz, sidx, colval, d_out_col, d_out_val);
    }
    return skm_check_launch("k_pc_gather");
}

// Learn aggregation from a CSR (snekmer/rules/learn.smk:385-408): column-major copy of the input (skm_csr_transpose),
// per-column sums (skm_group_postings), and back to CSR by group (skm_postings_to_csr).  A caller that holds the count
// matrix's postings already (engine.Basis after the vectorize stage) calls the last two itself (snekmer_amd/apply.py).
extern "C" int skm_csr_group_sum(skm_ctx *ctx, int64_t n, int64_t nnz, const int64_t *d_rowptr, const uint32_t *d_colidx,
                                 const uint32_t *d_counts, const uint32_t *d_group, int64_t ngroups,
                                 int64_t *d_out_rowptr, uint32_t *d_out_col, uint32_t *d_out_val, int64_t *h_out_nnz)
{
    SKM_REQUIRE(ctx && n >= 0 && nnz >= 0 && ngroups >= 0 && d_out_rowptr && h_out_nnz, SKM_E_BADARG,
                "skm_csr_group_sum: bad argument");
    SKM_REQUIRE(nnz < ((int64_t)1 << 32) - 1 && ngroups < ((int64_t)1 << 32) - 2, SKM_E_OVERFLOW, "skm_csr_group_sum: too large");
    SKM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    *h_out_nnz = 0;
    if (nnz == 0) {
        SKM_HIP(hipMemsetAsync(d_out_rowptr, 0, sizeof(int64_t) * (size_t)(ngroups + 1), st));
        return SKM_OK;
    }
    SKM_REQUIRE(d_rowptr && d_colidx && d_counts && d_group && d_out_col && d_out_val, SKM_E_BADARG,
                "skm_csr_group_sum: null array");
    // the number of columns: largest column id + 1 (one host wait; the totals' size is another, as before)
    void *p;
    SKM_TRY(skm_ws(ctx, WS_SMALL, 4096, &p));
    unsigned int *acc = (unsigned int *)((uint8_t *)p + 1536);
    SKM_HIP(hipMemsetAsync(acc, 0, 4, st));
    k_max_u32_plain<<<skm_grid_cap(ctx, skm_ceil_div(nnz, BLK), 8), BLK, 0, st>>>(nnz, d_colidx, acc);
    SKM_TRY(skm_check_launch("k_max_u32_plain"));
    SKM_HIP(hipMemcpyAsync(ctx->h_pinned, acc, 4, hipMemcpyDeviceToHost, st));
    SKM_HIP(hipStreamSynchronize(st));
    const uint32_t maxcol = *(uint32_t *)ctx->h_pinned;
    SKM_REQUIRE(maxcol != NONE, SKM_E_BADARG, "skm_csr_group_sum: a column id is 0xFFFFFFFF (a CSR built with elide_singletons has no real "
                                              "column ids)");
    const int64_t ncols = (int64_t)maxcol + 1;
    struct scratch {  // arrays of this call, from the library's pool (parked again when the call returns: no wait)
        skm_ctx *c;
        void *ptr[4] = {};
        ~scratch()
        {
            for (void *q : ptr)
                if (q)
                    skm_pool_free(c, q);
        }
    } tmp{ctx};
    SKM_TRY(skm_pool_alloc(ctx, sizeof(uint32_t) * (size_t)(ncols + 1), &tmp.ptr[0]));
    SKM_TRY(skm_pool_alloc(ctx, sizeof(uint64_t) * (size_t)nnz, &tmp.ptr[1]));
    SKM_TRY(skm_pool_alloc(ctx, sizeof(uint32_t) * (size_t)(ncols + 1), &tmp.ptr[2]));
    SKM_TRY(skm_pool_alloc(ctx, sizeof(uint64_t) * (size_t)nnz, &tmp.ptr[3]));
    uint32_t *colptr = (uint32_t *)tmp.ptr[0], *t_colptr = (uint32_t *)tmp.ptr[2];
    uint64_t *post = (uint64_t *)tmp.ptr[1], *t_post = (uint64_t *)tmp.ptr[3];
    SKM_TRY(skm_csr_transpose(ctx, n, nnz, ncols, d_rowptr, d_colidx, d_counts, colptr, post));
    SKM_TRY(skm_group_postings(ctx, n, ncols, colptr, post, nnz, d_group, ngroups, t_colptr, t_post, nullptr, nullptr, h_out_nnz));
    return skm_postings_to_csr(ctx, ncols, *h_out_nnz, t_colptr, t_post, ngroups, d_out_rowptr, d_out_col, d_out_val);
}
