// Multi-GPU basis: the column-major copy (postings) of a row-sharded count matrix, built in parallel.
//
// The reference has nothing to replace here (single process per FASTA file,
// snekmer/rules/kmerize.smk:57-65); this is the sharded form of the basis/postings stage that
// skm_basis_build provides on one GPU (snekmer/rules/kmerize.smk:89-104 + the count loops of
// snekmer/rules/learn.smk:359-383), used by the cosine pipeline only (singletons elided).
//
// A k-mer's code is a pure function of (alphabet, k, window), so every rank maps a code to the same
// OWNER rank (bucket_of) without a dictionary:
//
//   skm_bucket_partition   local CSR entries -> (code, row | count << 32) grouped by owner
//        ... all-to-all ...                     (skm_alltoallv)
//   skm_bucket_postings    owner: stable sort of what it received by code -> compact postings, the
//                          starts of its columns (k-mers found in >= 2 rows) and a hash table
//                          code -> column
//        ... all-gather of postings and tables ...   (skm_allgatherv)
//   skm_concat_colptr      global column starts from the per-owner tables
//   skm_colidx_from_owners (round 5) each rank: column id of its own CSR entries from the owners' ANSWERS - the owner of an
//                          entry knows its column once it has sorted its share, and sends one uint32 per entry back
//                          through the reverse of the all-to-all (0xFFFFFFFF: a k-mer of one row only); no hash table is
//                          built, gathered or probed (round 4: skm_colidx_lookup, kept: 1.0 ms of probes per rank + the
//                          tables in the all-gather)
//   skm_embed_rowptr       the local rows as rows [lo, hi) of an otherwise empty N-row matrix, so
//                          that the cosine kernels see global row numbers
//
// Every rank sorts 1/G of the entries instead of all of them (the replicated skm_basis_build was
// the Amdahl term of the strong-scaling bench).
#include "skm_common.h"
#include "skm_onesweep.h"
#include "skm_sort.h"

namespace {

constexpr int BLK = 256;

// owner of a code among nb ranks; multiplicative hash so that low-complexity code ranges spread out
__device__ __forceinline__ uint32_t bucket_of(uint32_t code, uint32_t nb)
{
    uint32_t h = code * 0x9E3779B1u;
    h ^= h >> 15;
    h *= 0x85EBCA77u;
    return (uint32_t)(((uint64_t)h * nb) >> 32);
}
__device__ __forceinline__ uint32_t bucket_of(uint64_t code, uint32_t nb)
{
    const uint64_t m = code * 0x9E3779B97F4A7C15ull;
    return bucket_of((uint32_t)(m >> 32) ^ (uint32_t)m, nb);
}

// slot of a code in its owner's column table (tsize a power of two); different bits than bucket_of
__device__ __forceinline__ uint32_t table_slot(uint32_t code, uint32_t tsize)
{
    uint32_t h = code * 0xC2B2AE35u;
    h ^= h >> 16;
    h *= 0x27D4EB2Fu;
    h ^= h >> 15;
    return h & (tsize - 1u);
}
__device__ __forceinline__ uint32_t table_slot(uint64_t code, uint32_t tsize)
{
    const uint64_t m = code * 0xD6E8FEB86659FD93ull;
    return table_slot((uint32_t)(m >> 32) ^ (uint32_t)m, tsize);
}
constexpr uint32_t TAB_EMPTY = 0xFFFFFFFFu;

// one wave per row: posting word and owner of every entry, owner histogram
template <typename K>
__global__ __launch_bounds__(BLK) void k_bucket_keys(const int64_t *__restrict__ rowptr, int64_t n, int64_t row_base,
                                                     const K *__restrict__ codes, const uint32_t *__restrict__ counts,
                                                     uint32_t nb, uint64_t *__restrict__ rowcount,
                                                     uint8_t *__restrict__ key8, unsigned int *__restrict__ hist)
{
    __shared__ unsigned int s_hist[SKM_MAX_RANKS];
    if (threadIdx.x < SKM_MAX_RANKS)
        s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += nwaves) {
        const int64_t b = rowptr[i], e = rowptr[i + 1];
        for (int64_t t = b + lane; t < e; t += 64) {
            const uint32_t bk = bucket_of(codes[t], nb);
            rowcount[t] = (uint64_t)(uint32_t)(row_base + i) | ((uint64_t)counts[t] << 32);
            key8[t] = (uint8_t)bk;
            atomicAdd(&s_hist[bk], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < nb && s_hist[threadIdx.x])
        atomicAdd(&hist[threadIdx.x], s_hist[threadIdx.x]);
}

// (the entry count lives on the device: d_rowptr[n]; the grid covers the capacity)
template <typename K>
__global__ __launch_bounds__(BLK) void k_partition_gather(const int64_t *__restrict__ d_nnz, const uint32_t *__restrict__ idx,
                                                          const K *__restrict__ codes,
                                                          const uint64_t *__restrict__ rowcount,
                                                          K *__restrict__ out_codes, uint64_t *__restrict__ out_rowcount,
                                                          const unsigned int *__restrict__ hist, int nbuckets,
                                                          int64_t *__restrict__ out_counts, uint32_t *__restrict__ out_index)
{
    const int64_t nnz = *d_nnz;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nbuckets)
        out_counts[t] = (int64_t)hist[t];  // entries per owner, as the exchange gathers them (int64)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nnz; t += stride) {
        const uint32_t e = idx[t];
        out_codes[t] = codes[e];
        out_rowcount[t] = rowcount[e];
        if (out_index)
            out_index[t] = e;  // grouped position -> entry of the shard's CSR (what the owners' answers are scattered by)
    }
}

// per sorted position: low word 1 if the entry's k-mer occurs in another row too, high word 1 if
// it is the first entry of such a k-mer
template <typename K>
struct ns_flags {
    const K *keys;
    uint32_t n;
    __device__ uint64_t operator()(uint32_t t) const
    {
        const K k = keys[t];
        const bool head = t == 0 || keys[t - 1] != k;
        const bool more = t + 1 < n && keys[t + 1] == k;
        const bool ns = !head || more;
        return (ns ? 1ull : 0ull) | ((head && more) ? (1ull << 32) : 0ull);
    }
};

// sizes of an owner's share from the closing element of the scan: (distinct k-mers, shared columns, postings, table slots)
__device__ __forceinline__ uint32_t table_size_dev(int64_t ncols)
{
    uint32_t t = 2;
    while ((int64_t)t < 2 * ncols)
        t <<= 1;
    return t;
}

__global__ void k_owner_sizes(int64_t n, const uint64_t *__restrict__ incl, int64_t *__restrict__ out4)
{
    const uint64_t last = incl[n - 1];
    const int64_t npost = (int64_t)(uint32_t)last, ncols_ns = (int64_t)(last >> 32);
    out4[0] = (n - npost) + ncols_ns;
    out4[1] = ncols_ns;
    out4[2] = npost;
    out4[3] = (int64_t)table_size_dev(ncols_ns);
}

template <typename K>
__global__ __launch_bounds__(BLK) void k_bucket_emit(int64_t n, const K *__restrict__ skeys,
                                                     const uint32_t *__restrict__ sidx,
                                                     const uint64_t *__restrict__ incl,
                                                     const uint64_t *__restrict__ rowcount,
                                                     uint32_t *__restrict__ cols_start, uint64_t *__restrict__ post,
                                                     const int64_t *__restrict__ out4, K *__restrict__ tab_keys,
                                                     uint32_t *__restrict__ tab_vals, uint32_t *__restrict__ ret)
{
    const uint32_t tsize = (uint32_t)out4[3];  // power of two >= 2 x shared columns (k_owner_sizes)
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride) {
        const K k = skeys[t];
        const bool head = t == 0 || skeys[t - 1] != k;
        const bool more = t + 1 < n && skeys[t + 1] == k;
        if (head && !more) {  // k-mer of one row only: no posting, no column
            if (ret)
                ret[sidx[t]] = TAB_EMPTY;
            continue;
        }
        const uint64_t sc = incl[t];
        const uint32_t pos = (uint32_t)sc - 1u;  // inclusive scan: this entry is counted
        post[pos] = rowcount[sidx[t]];
        // the answer to whoever sent this entry: the owner-local column of its k-mer (heads counted up to here, this
        // k-mer's included), in the order the entries were received
        if (ret)
            ret[sidx[t]] = (uint32_t)(sc >> 32) - 1u;
        if (head) {
            const uint32_t ci = (uint32_t)(sc >> 32) - 1u;
            cols_start[ci] = pos;
            if (!tab_vals)
                continue;
            // every column is inserted once, so claiming the value word is enough; the key word is
            // only read by later kernels
            for (uint32_t h = table_slot(k, tsize);; h = (h + 1u) & (tsize - 1u)) {
                if (atomicCAS(&tab_vals[h], TAB_EMPTY, ci) == TAB_EMPTY) {
                    tab_keys[h] = k;
                    break;
                }
            }
        }
    }
}

__global__ void k_rebase_u32(const uint32_t *__restrict__ src, int64_t count, uint32_t add, uint32_t *__restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count)
        dst[i] = src[i] + add;
}

__global__ void k_store_u32(uint32_t *p, uint32_t v) { *p = v; }

struct bucket_segments {
    uint32_t slot[SKM_MAX_RANKS + 1];  // first table slot of every owner (sizes are powers of two)
    uint32_t col[SKM_MAX_RANKS + 1];   // first global column id of every owner
};

template <typename K>
__global__ __launch_bounds__(BLK) void k_colidx_lookup(int64_t nnz, const K *__restrict__ codes, uint32_t nb,
                                                       bucket_segments seg, const K *__restrict__ tab_keys,
                                                       const uint32_t *__restrict__ tab_vals,
                                                       uint32_t *__restrict__ colidx)
{
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < nnz; t += stride) {
        const K k = codes[t];
        const uint32_t b = bucket_of(k, nb);
        const uint32_t base = seg.slot[b], tsize = seg.slot[b + 1] - base;
        uint32_t found = 0xFFFFFFFFu;
        if (tsize) {
            for (uint32_t h = table_slot(k, tsize);; h = (h + 1u) & (tsize - 1u)) {  // load factor <= 1/2: terminates
                const uint32_t v = tab_vals[base + h];
                if (v == TAB_EMPTY)
                    break;
                if (tab_keys[base + h] == k) {
                    found = seg.col[b] + v;
                    break;
                }
            }
        }
        colidx[t] = found;
    }
}

__global__ void k_embed_rowptr(int64_t n_total, int64_t lo, int64_t nloc, const int64_t *__restrict__ local,
                               int64_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n_total)
        out[i] = i <= lo ? 0 : (i <= lo + nloc ? local[i - lo] : local[nloc]);
}

template <typename K>
int partition_impl(skm_ctx *ctx, int nbuckets, int64_t n, int64_t cap, const int64_t *d_rowptr, const K *d_codes,
                   const uint32_t *d_counts, int64_t row_base, K *d_out_codes, uint64_t *d_out_rowcount,
                   int64_t *d_out_counts, int64_t *h_counts, uint32_t *d_out_index)
{
    // Nothing here waits for the device unless the caller asks for the counts on the host: the entry count is
    // d_rowptr[n], the launches cover the capacity, the stable grouping by owner is one digit of the library's own
    // one-sweep sort (which reads its size on the device; rocPRIM's takes it as a host argument).
    hipStream_t st = ctx->stream;
    void *p;
    SKM_TRY(skm_ws(ctx, WS_A, sizeof(uint64_t) * (size_t)cap, &p));
    uint64_t *rowcount = (uint64_t *)p;
    SKM_TRY(skm_ws(ctx, WS_B, (size_t)cap, &p));
    uint8_t *key8 = (uint8_t *)p;
    SKM_TRY(skm_ws(ctx, WS_C, (size_t)cap, &p));
    uint8_t *key8s = (uint8_t *)p;
    SKM_TRY(skm_ws(ctx, WS_D, sizeof(uint32_t) * (size_t)cap, &p));
    uint32_t *idx = (uint32_t *)p;
    SKM_TRY(skm_ws(ctx, WS_SMALL, 4096, &p));
    unsigned int *hist = (unsigned int *)p;
    SKM_HIP(hipMemsetAsync(hist, 0, sizeof(unsigned int) * SKM_MAX_RANKS, st));
    {
        SKM_PROF(ctx, "k_bucket_keys");
        k_bucket_keys<K><<<skm_grid_cap(ctx, skm_ceil_div(n, BLK / 64), 16), BLK, 0, st>>>(
            d_rowptr, n, row_base, d_codes, d_counts, (uint32_t)nbuckets, rowcount, key8, hist);
    }
    SKM_TRY(skm_check_launch("k_bucket_keys"));
    int bits = 1;
    while ((1 << bits) < nbuckets)
        ++bits;
    SKM_TRY(skm_ws(ctx, WS_ROCPRIM, skm_onesweep::state_bytes(cap, 8192, 1) + skm_onesweep::state_bytes(cap, 2048, 1), &p));
    SKM_TRY(skm_onesweep::sort_pairs_dev<uint8_t>(ctx, d_rowptr + n, cap, key8, key8s, idx, nullptr, nullptr, p, bits,
                                                  "onesweep_sort_owner"));
    {
        SKM_PROF(ctx, "k_partition_gather");
        k_partition_gather<K><<<skm_grid_cap(ctx, skm_ceil_div(cap, BLK), 16), BLK, 0, st>>>(
            d_rowptr + n, idx, d_codes, rowcount, d_out_codes, d_out_rowcount, hist, nbuckets, d_out_counts, d_out_index);
    }
    SKM_TRY(skm_check_launch("k_partition_gather"));
    if (h_counts) {
        SKM_HIP(hipMemcpyAsync(ctx->h_pinned, d_out_counts, sizeof(int64_t) * (size_t)nbuckets, hipMemcpyDeviceToHost, st));
        SKM_HIP(hipStreamSynchronize(st));
        memcpy(h_counts, ctx->h_pinned, sizeof(int64_t) * (size_t)nbuckets);
    }
    return SKM_OK;
}

static uint32_t table_size_for(int64_t ncols)
{
    uint32_t t = 2;
    while ((int64_t)t < 2 * ncols)
        t <<= 1;
    return t;
}

template <typename K>
int postings_impl(skm_ctx *ctx, int key_bits, int64_t n, const K *d_codes, const uint64_t *d_rowcount, int64_t *d_out4,
                  int64_t *h_out4, uint32_t *d_cols_start, uint64_t *d_post, K *d_tab_keys, uint32_t *d_tab_vals, uint32_t *d_ret)
{
    hipStream_t st = ctx->stream;
    void *p;
    SKM_TRY(skm_ws(ctx, WS_A, sizeof(K) * (size_t)n, &p));
    K *skeys = (K *)p;
    SKM_TRY(skm_ws(ctx, WS_C, sizeof(uint32_t) * (size_t)n, &p));
    uint32_t *sidx = (uint32_t *)p;
    SKM_TRY(skm_ws(ctx, WS_E, sizeof(uint64_t) * (size_t)n, &p));
    uint64_t *incl = (uint64_t *)p;
    SKM_TRY(sort_pairs<K>(ctx, d_codes, skeys, sidx, n, key_bits, "rocprim_radix_sort_owned_codes"));
    {
        size_t tmp = 0;
        auto in = rocprim::make_transform_iterator(rocprim::counting_iterator<uint32_t>(0), ns_flags<K>{skeys, (uint32_t)n});
        SKM_HIP(rocprim::inclusive_scan(nullptr, tmp, in, incl, (size_t)n, rocprim::plus<uint64_t>(), st));
        SKM_TRY(skm_ws(ctx, WS_ROCPRIM, tmp, &p));
        SKM_PROF(ctx, "rocprim_scan_shared_kmers");
        SKM_HIP(rocprim::inclusive_scan(p, tmp, in, incl, (size_t)n, rocprim::plus<uint64_t>(), st));
    }
    // sizes stay on the device (d_out4); the table's value words are cleared over their whole capacity because the
    // size actually used (a power of two >= 2 x shared columns) is only known there
    k_owner_sizes<<<1, 1, 0, st>>>(n, incl, d_out4);
    if (d_tab_vals)
        SKM_HIP(hipMemsetAsync(d_tab_vals, 0xFF, sizeof(uint32_t) * (size_t)table_size_for(n / 2), st));
    {
        SKM_PROF(ctx, "k_bucket_emit");
        k_bucket_emit<K><<<skm_grid_cap(ctx, skm_ceil_div(n, BLK), 16), BLK, 0, st>>>(n, skeys, sidx, incl, d_rowcount,
                                                                                   d_cols_start, d_post, d_out4, d_tab_keys,
                                                                                   d_tab_vals, d_ret);
    }
    SKM_TRY(skm_check_launch("k_bucket_emit"));
    if (h_out4) {
        SKM_HIP(hipMemcpyAsync(ctx->h_pinned, d_out4, sizeof(int64_t) * 4, hipMemcpyDeviceToHost, st));
        SKM_HIP(hipStreamSynchronize(st));
        memcpy(h_out4, ctx->h_pinned, sizeof(int64_t) * 4);
    }
    return SKM_OK;
}

}  // namespace

extern "C" int skm_bucket_partition(skm_ctx *ctx, int code_bits, int nbuckets, int64_t n, int64_t cap_entries,
                                    const int64_t *d_rowptr, const void *d_codes, const uint32_t *d_counts,
                                    int64_t row_base, void *d_out_codes, uint64_t *d_out_rowcount, int64_t *d_out_counts,
                                    int64_t *h_counts, uint32_t *d_out_index)
{
    SKM_REQUIRE(ctx && d_out_counts && n >= 0 && cap_entries >= 0 && row_base >= 0, SKM_E_BADARG, "skm_bucket_partition: bad argument");
    SKM_REQUIRE(code_bits == 32 || code_bits == 64, SKM_E_BADARG, "skm_bucket_partition: code_bits must be 32 or 64");
    SKM_REQUIRE(nbuckets >= 1 && nbuckets <= SKM_MAX_RANKS, SKM_E_BADARG, "skm_bucket_partition: 1 <= nbuckets <= %d",
                SKM_MAX_RANKS);
    SKM_REQUIRE(cap_entries < ((int64_t)1 << 30) && row_base + n < ((int64_t)1 << 32) - 1, SKM_E_OVERFLOW,
                "skm_bucket_partition: 2^30 entries or more in one shard, or a row index >= 2^32");
    SKM_HIP(hipSetDevice(ctx->device));
    if (h_counts)
        for (int b = 0; b < nbuckets; ++b)
            h_counts[b] = 0;
    if (n == 0 || cap_entries == 0) {
        SKM_HIP(hipMemsetAsync(d_out_counts, 0, sizeof(int64_t) * (size_t)nbuckets, ctx->stream));
        return SKM_OK;
    }
    SKM_REQUIRE(d_rowptr && d_codes && d_counts && d_out_codes && d_out_rowcount, SKM_E_BADARG,
                "skm_bucket_partition: null array");
    if (code_bits == 32)
        return partition_impl<uint32_t>(ctx, nbuckets, n, cap_entries, d_rowptr, (const uint32_t *)d_codes, d_counts, row_base,
                                        (uint32_t *)d_out_codes, d_out_rowcount, d_out_counts, h_counts, d_out_index);
    return partition_impl<uint64_t>(ctx, nbuckets, n, cap_entries, d_rowptr, (const uint64_t *)d_codes, d_counts, row_base,
                                    (uint64_t *)d_out_codes, d_out_rowcount, d_out_counts, h_counts, d_out_index);
}

extern "C" int64_t skm_bucket_table_capacity(int64_t nrecv)
{
    return (int64_t)table_size_for(nrecv / 2);  // a shared column has at least two entries
}

namespace {
__global__ void k_empty_owner(int64_t *out4, uint32_t *tab_vals)
{
    out4[0] = out4[1] = out4[2] = 0;
    out4[3] = 2;  // an empty table of the minimum size, so that every owner contributes one
    if (tab_vals)
        tab_vals[0] = tab_vals[1] = 0xFFFFFFFFu;
}

// colidx[index[g]] = the owner's answer for grouped position g, moved into the global column numbering (owners'
// columns back to back in rank order); group of g = the owner its entry went to
struct owner_groups {
    int64_t first[SKM_MAX_RANKS + 1];  // first grouped position of every owner's group
    uint32_t col[SKM_MAX_RANKS + 1];   // first global column id of every owner
};
__global__ __launch_bounds__(BLK) void k_colidx_from_owners(int64_t nnz, int nb, owner_groups grp, const uint32_t *__restrict__ back,
                                                            const uint32_t *__restrict__ index, uint32_t *__restrict__ colidx)
{
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; g < nnz; g += stride) {
        int b = 0;
        while (b + 1 < nb && g >= grp.first[b + 1])
            ++b;
        const uint32_t v = back[g];
        colidx[index[g]] = v == TAB_EMPTY ? TAB_EMPTY : grp.col[b] + v;
    }
}
}  // namespace

extern "C" int skm_bucket_postings(skm_ctx *ctx, int code_bits, int key_bits, int64_t nrecv, const void *d_codes,
                                   const uint64_t *d_rowcount, int64_t *d_out4, int64_t *h_out4, uint32_t *d_cols_start,
                                   uint64_t *d_post, void *d_tab_keys, uint32_t *d_tab_vals, uint32_t *d_ret)
{
    SKM_REQUIRE(ctx && d_out4 && nrecv >= 0, SKM_E_BADARG, "skm_bucket_postings: bad argument");
    SKM_REQUIRE(code_bits == 32 || code_bits == 64, SKM_E_BADARG, "skm_bucket_postings: code_bits must be 32 or 64");
    SKM_REQUIRE(nrecv < ((int64_t)1 << 31), SKM_E_OVERFLOW, "skm_bucket_postings: more than 2^31 entries for one owner");
    SKM_REQUIRE((d_tab_keys && d_tab_vals) || (!d_tab_keys && !d_tab_vals && d_ret), SKM_E_BADARG,
                "skm_bucket_postings: give the table (keys and values), the answer array d_ret, or both");
    SKM_HIP(hipSetDevice(ctx->device));
    if (nrecv == 0) {
        k_empty_owner<<<1, 1, 0, ctx->stream>>>(d_out4, d_tab_vals);
        SKM_TRY(skm_check_launch("k_empty_owner"));
        if (h_out4) {
            h_out4[0] = h_out4[1] = h_out4[2] = 0;
            h_out4[3] = 2;
        }
        return SKM_OK;
    }
    SKM_REQUIRE(d_codes && d_rowcount && d_cols_start && d_post, SKM_E_BADARG, "skm_bucket_postings: null array");
    if (key_bits <= 0 || key_bits > code_bits)
        key_bits = code_bits;
    if (code_bits == 32)
        return postings_impl<uint32_t>(ctx, key_bits, nrecv, (const uint32_t *)d_codes, d_rowcount, d_out4, h_out4, d_cols_start,
                                       d_post, (uint32_t *)d_tab_keys, d_tab_vals, d_ret);
    return postings_impl<uint64_t>(ctx, key_bits, nrecv, (const uint64_t *)d_codes, d_rowcount, d_out4, h_out4, d_cols_start,
                                   d_post, (uint64_t *)d_tab_keys, d_tab_vals, d_ret);
}

extern "C" int skm_concat_colptr(skm_ctx *ctx, int nparts, const int64_t *h_ncols, const int64_t *h_npost,
                                 const uint32_t *d_starts, uint32_t *d_colptr)
{
    SKM_REQUIRE(ctx && nparts >= 1 && h_ncols && h_npost && d_colptr, SKM_E_BADARG, "skm_concat_colptr: bad argument");
    SKM_HIP(hipSetDevice(ctx->device));
    int64_t col = 0, base = 0;
    SKM_PROF(ctx, "k_rebase_u32");
    for (int p = 0; p < nparts; ++p) {
        SKM_REQUIRE(h_ncols[p] >= 0 && h_npost[p] >= 0, SKM_E_BADARG, "skm_concat_colptr: negative size");
        if (h_ncols[p] > 0) {
            SKM_REQUIRE(d_starts, SKM_E_BADARG, "skm_concat_colptr: null starts");
            k_rebase_u32<<<(unsigned)skm_ceil_div(h_ncols[p], BLK), BLK, 0, ctx->stream>>>(d_starts + col, h_ncols[p],
                                                                                        (uint32_t)base, d_colptr + col);
        }
        col += h_ncols[p];
        base += h_npost[p];
    }
    SKM_REQUIRE(base < ((int64_t)1 << 32) - 1, SKM_E_OVERFLOW, "skm_concat_colptr: more than 2^32 postings");
    k_store_u32<<<1, 1, 0, ctx->stream>>>(d_colptr + col, (uint32_t)base);
    return skm_check_launch("k_rebase_u32");
}

extern "C" int skm_colidx_lookup(skm_ctx *ctx, int code_bits, int nbuckets, int64_t nnz, const void *d_codes,
                                 const int64_t *h_tab_sizes, const int64_t *h_ncols, const void *d_tab_keys,
                                 const uint32_t *d_tab_vals, uint32_t *d_colidx)
{
    SKM_REQUIRE(ctx && nnz >= 0 && h_tab_sizes && h_ncols, SKM_E_BADARG, "skm_colidx_lookup: bad argument");
    SKM_REQUIRE(code_bits == 32 || code_bits == 64, SKM_E_BADARG, "skm_colidx_lookup: code_bits must be 32 or 64");
    SKM_REQUIRE(nbuckets >= 1 && nbuckets <= SKM_MAX_RANKS, SKM_E_BADARG, "skm_colidx_lookup: 1 <= nbuckets <= %d",
                SKM_MAX_RANKS);
    if (nnz == 0)
        return SKM_OK;
    bucket_segments seg = {};
    int64_t slot = 0, col = 0;
    for (int b = 0; b < nbuckets; ++b) {
        const int64_t t = h_tab_sizes[b];
        SKM_REQUIRE(t >= 0 && (t & (t - 1)) == 0 && h_ncols[b] >= 0 && 2 * h_ncols[b] <= (t > 2 ? t : 2), SKM_E_BADARG,
                    "skm_colidx_lookup: table %d: size %lld must be a power of two >= 2 * columns (%lld)", b, (long long)t,
                    (long long)h_ncols[b]);
        seg.slot[b] = (uint32_t)slot;
        seg.col[b] = (uint32_t)col;
        slot += t;
        col += h_ncols[b];
    }
    SKM_REQUIRE(slot < ((int64_t)1 << 32) - 1 && col < ((int64_t)1 << 32) - 1, SKM_E_OVERFLOW, "skm_colidx_lookup: >= 2^32 slots");
    seg.slot[nbuckets] = (uint32_t)slot;
    seg.col[nbuckets] = (uint32_t)col;
    SKM_REQUIRE(d_codes && d_colidx && (slot == 0 || (d_tab_keys && d_tab_vals)), SKM_E_BADARG, "skm_colidx_lookup: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    SKM_PROF(ctx, "k_colidx_lookup");
    const int grid = skm_grid_cap(ctx, skm_ceil_div(nnz, BLK), 16);
    if (code_bits == 32)
        k_colidx_lookup<uint32_t><<<grid, BLK, 0, ctx->stream>>>(nnz, (const uint32_t *)d_codes, (uint32_t)nbuckets, seg,
                                                                 (const uint32_t *)d_tab_keys, d_tab_vals, d_colidx);
    else
        k_colidx_lookup<uint64_t><<<grid, BLK, 0, ctx->stream>>>(nnz, (const uint64_t *)d_codes, (uint32_t)nbuckets, seg,
                                                                 (const uint64_t *)d_tab_keys, d_tab_vals, d_colidx);
    return skm_check_launch("k_colidx_lookup");
}

extern "C" int skm_colidx_from_owners(skm_ctx *ctx, int nbuckets, int64_t nnz, const uint32_t *d_back, const uint32_t *d_index,
                                      const int64_t *h_group_counts, const int64_t *h_ncols, uint32_t *d_colidx)
{
    SKM_REQUIRE(ctx && nnz >= 0 && h_group_counts && h_ncols, SKM_E_BADARG, "skm_colidx_from_owners: bad argument");
    SKM_REQUIRE(nbuckets >= 1 && nbuckets <= SKM_MAX_RANKS, SKM_E_BADARG, "skm_colidx_from_owners: 1 <= nbuckets <= %d", SKM_MAX_RANKS);
    owner_groups grp = {};
    int64_t first = 0, col = 0;
    for (int b = 0; b < nbuckets; ++b) {
        SKM_REQUIRE(h_group_counts[b] >= 0 && h_ncols[b] >= 0, SKM_E_BADARG, "skm_colidx_from_owners: negative size");
        grp.first[b] = first;
        grp.col[b] = (uint32_t)col;
        first += h_group_counts[b];
        col += h_ncols[b];
    }
    grp.first[nbuckets] = first;
    SKM_REQUIRE(first == nnz, SKM_E_BADARG, "skm_colidx_from_owners: the groups hold %lld entries, nnz is %lld", (long long)first,
                (long long)nnz);
    SKM_REQUIRE(col < ((int64_t)1 << 32) - 1, SKM_E_OVERFLOW, "skm_colidx_from_owners: >= 2^32 columns");
    if (nnz == 0)
        return SKM_OK;
    SKM_REQUIRE(d_back && d_index && d_colidx, SKM_E_BADARG, "skm_colidx_from_owners: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    SKM_PROF(ctx, "k_colidx_from_owners");
    k_colidx_from_owners<<<skm_grid_cap(ctx, skm_ceil_div(nnz, BLK), 16), BLK, 0, ctx->stream>>>(nnz, nbuckets, grp, d_back, d_index,
                                                                                                 d_colidx);
    return skm_check_launch("k_colidx_from_owners");
}

extern "C" int skm_embed_rowptr(skm_ctx *ctx, int64_t n_total, int64_t lo, int64_t nloc, const int64_t *d_local,
                                int64_t *d_rowptr)
{
    SKM_REQUIRE(ctx && n_total >= 0 && lo >= 0 && nloc >= 0 && lo + nloc <= n_total && d_local && d_rowptr, SKM_E_BADARG,
                "skm_embed_rowptr: bad argument");
    SKM_HIP(hipSetDevice(ctx->device));
    SKM_PROF(ctx, "k_embed_rowptr");
    k_embed_rowptr<<<(unsigned)skm_ceil_div(n_total + 1, BLK), BLK, 0, ctx->stream>>>(n_total, lo, nloc, d_local, d_rowptr);
    return skm_check_launch("k_embed_rowptr");
}
