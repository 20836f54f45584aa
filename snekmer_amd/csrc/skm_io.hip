// The data formats either side of the hot path, so that the reference-shaped entry point
// (snekmer_amd.kmerize.vectorize_fasta = the body of snekmer/rules/kmerize.smk:67-142) is not bounded by per-record
// Python:
//
//   (skm_fasta_index / skm_fasta_parse, the threaded host-side FASTA reader, live in skm_host.cpp)
//   skm_rows_to_utf32     ragged byte rows -> fixed-width UCS-4 rows (numpy '<U{w}'): the `seqs` array of reduced
//                         strings (kmerize.smk:121-127, 136) straight from the recoded bytes.
//   skm_decode_kmers_utf32   integer k-mer codes -> '<U{k}' strings: the `kmerlist` array (kmerize.smk:102-106, 134).
//   skm_csr_remap_columns    CSR entries re-labelled through a column map and filtered: the count matrix in kmerlist
//                         order (rules/learn.smk:359-383 projects every sequence on `kmerlist`).
#include <cstring>

#include "skm_common.h"

#include <rocprim/rocprim.hpp>

namespace {

// ------------------------------------------------------------------------------------------------ device formatting
constexpr int BLK = 256;

// out[i][j] = bytes[off[i] + j] for j < len[i], 0 beyond: a numpy '<U{width}' row per record (latin-1 bytes are their
// own code points).  One wave per row.
__global__ __launch_bounds__(BLK) void k_rows_to_utf32(const uint8_t *__restrict__ bytes, const int64_t *__restrict__ off,
                                                       const int32_t *__restrict__ len, int64_t n, int64_t width,
                                                       uint32_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += nwaves) {
        const uint8_t *src = bytes + off[i];
        const int64_t l = len[i];
        uint32_t *dst = out + i * width;
        for (int64_t j = lane; j < width; j += 64)
            dst[j] = j < l ? (uint32_t)src[j] : 0u;
    }
}

struct letters64 {
    uint32_t c[64];  // code point of every rank (an alphabet has at most 30 class letters)
};

// out[i][0..k) = letters of codes[idx ? idx[i] : i], most significant symbol first.  A thread decodes one k-mer into
// LDS, then the workgroup stores the block's BLK * k words lane-consecutively.
template <typename K>
__global__ __launch_bounds__(BLK) void k_decode_kmers_utf32(const K *__restrict__ codes, const uint32_t *__restrict__ idx,
                                                            int64_t n, int k, uint32_t nsym, letters64 lt,
                                                            uint32_t *__restrict__ out)
{
    extern __shared__ uint32_t s_word[];  // BLK * k
    for (int64_t base = (int64_t)blockIdx.x * BLK; base < n; base += (int64_t)gridDim.x * BLK) {
        const int64_t i = base + threadIdx.x;
        if (i < n) {
            K c = codes[idx ? (int64_t)idx[i] : i];
            for (int j = k - 1; j >= 0; --j) {
                const K q = c / (K)nsym;
                s_word[threadIdx.x * k + j] = lt.c[(uint32_t)(c - q * (K)nsym)];
                c = q;
            }
        }
        __syncthreads();
        const int64_t words = (n - base < BLK ? n - base : BLK) * k;
        for (int64_t w = threadIdx.x; w < words; w += BLK)
            out[base * k + w] = s_word[w];
        __syncthreads();
    }
}

// kept entries per row: colmap[colidx[e]] != 0xFFFFFFFF (an entry with colidx 0xFFFFFFFF itself is dropped)
__global__ __launch_bounds__(BLK) void k_remap_count(const int64_t *__restrict__ rowptr, const uint32_t *__restrict__ colidx,
                                                     const uint32_t *__restrict__ colmap, int64_t ncols, int64_t n,
                                                     int64_t *__restrict__ kept)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += nwaves) {
        const int64_t b = rowptr[i], e = rowptr[i + 1];
        int64_t cnt = 0;
        for (int64_t t0 = b; t0 < e; t0 += 64) {  // whole-wave iterations: the ballot needs every lane
            const int64_t t = t0 + lane;
            bool keep = false;
            if (t < e) {
                const uint32_t c = colidx[t];
                keep = (int64_t)c < ncols && colmap[c] != 0xFFFFFFFFu;
            }
            cnt += __popcll(__ballot(keep));
        }
        if (lane == 0)
            kept[i] = cnt;
    }
}

__global__ __launch_bounds__(BLK) void k_remap_write(const int64_t *__restrict__ rowptr, const uint32_t *__restrict__ colidx,
                                                     const uint32_t *__restrict__ counts, const uint32_t *__restrict__ colmap,
                                                     int64_t ncols, int64_t n, const int64_t *__restrict__ out_rowptr,
                                                     uint32_t *__restrict__ out_col, uint32_t *__restrict__ out_val)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += nwaves) {
        const int64_t b = rowptr[i], e = rowptr[i + 1];
        int64_t dst = out_rowptr[i];
        for (int64_t t0 = b; t0 < e; t0 += 64) {  // whole-wave iterations: the ballot needs every lane
            const int64_t t = t0 + lane;
            uint32_t nc = 0xFFFFFFFFu;
            if (t < e) {
                const uint32_t c = colidx[t];
                if ((int64_t)c < ncols)
                    nc = colmap[c];
            }
            const unsigned long long mask = __ballot(nc != 0xFFFFFFFFu);
            if (nc != 0xFFFFFFFFu) {
                const int64_t at = dst + __popcll(mask & ((1ull << lane) - 1ull));
                out_col[at] = nc;
                out_val[at] = counts[t];
            }
            dst += __popcll(mask);
        }
    }
}

// keep flag of every position of the first-seen order: total occurrences of that column > min_filter
__global__ __launch_bounds__(BLK) void k_select_flags(int64_t ncols, const uint32_t *__restrict__ order,
                                                      const uint64_t *__restrict__ total, uint64_t min_filter,
                                                      uint32_t *__restrict__ flags)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ncols)
        flags[i] = total[order[i]] > min_filter ? 1u : 0u;
}

__global__ __launch_bounds__(BLK) void k_select_scatter(int64_t ncols, const uint32_t *__restrict__ order,
                                                        const uint32_t *__restrict__ flags, const uint32_t *__restrict__ pos,
                                                        uint32_t *__restrict__ keep, uint32_t *__restrict__ colmap)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncols)
        return;
    const uint32_t c = order[i];
    if (flags[i]) {
        keep[pos[i]] = c;
        colmap[c] = pos[i];
    } else {
        colmap[c] = 0xFFFFFFFFu;
    }
}

}  // namespace

// =============================================================================================== C ABI
extern "C" int skm_rows_to_utf32(skm_ctx *ctx, const uint8_t *d_bytes, const int64_t *d_off, const int32_t *d_len, int64_t n,
                                 int64_t width, uint32_t *d_out)
{
    SKM_REQUIRE(ctx && n >= 0 && width >= 1, SKM_E_BADARG, "skm_rows_to_utf32: bad argument");
    if (n == 0)
        return SKM_OK;
    SKM_REQUIRE(d_bytes && d_off && d_len && d_out, SKM_E_BADARG, "skm_rows_to_utf32: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    SKM_PROF(ctx, "k_rows_to_utf32");
    k_rows_to_utf32<<<skm_grid_cap(ctx, skm_ceil_div(n, BLK / 64), 16), BLK, 0, ctx->stream>>>(d_bytes, d_off, d_len, n, width, d_out);
    return skm_check_launch("k_rows_to_utf32");
}

extern "C" int skm_decode_kmers_utf32(skm_ctx *ctx, int code_bits, int nsym, int k, const uint8_t *h_letters,
                                      const void *d_codes, const uint32_t *d_index, int64_t n, uint32_t *d_out)
{
    SKM_REQUIRE(ctx && h_letters && n >= 0, SKM_E_BADARG, "skm_decode_kmers_utf32: bad argument");
    SKM_REQUIRE(code_bits == 32 || code_bits == 64, SKM_E_BADARG, "skm_decode_kmers_utf32: code_bits must be 32 or 64");
    SKM_REQUIRE(nsym >= 1 && nsym <= 64 && k >= 1 && k <= 64, SKM_E_BADARG, "skm_decode_kmers_utf32: 1 <= nsym, k <= 64");
    if (n == 0)
        return SKM_OK;
    SKM_REQUIRE(d_codes && d_out, SKM_E_BADARG, "skm_decode_kmers_utf32: null array");
    letters64 lt = {};
    for (int s = 0; s < nsym; ++s)
        lt.c[s] = h_letters[s];
    SKM_HIP(hipSetDevice(ctx->device));
    const size_t lds = sizeof(uint32_t) * BLK * (size_t)k;
    const int grid = skm_grid_cap(ctx, skm_ceil_div(n, BLK), 16);
    SKM_PROF(ctx, "k_decode_kmers_utf32");
    if (code_bits == 32)
        k_decode_kmers_utf32<uint32_t><<<grid, BLK, lds, ctx->stream>>>((const uint32_t *)d_codes, d_index, n, k, (uint32_t)nsym, lt, d_out);
    else
        k_decode_kmers_utf32<uint64_t><<<grid, BLK, lds, ctx->stream>>>((const uint64_t *)d_codes, d_index, n, k, (uint32_t)nsym, lt, d_out);
    return skm_check_launch("k_decode_kmers_utf32");
}

extern "C" int skm_csr_remap_columns(skm_ctx *ctx, int64_t n, const int64_t *d_rowptr, const uint32_t *d_colidx,
                                     const uint32_t *d_counts, const uint32_t *d_colmap, int64_t ncols, int64_t *d_out_rowptr,
                                     uint32_t *d_out_col, uint32_t *d_out_val, int64_t *h_out_nnz)
{
    SKM_REQUIRE(ctx && n >= 0 && ncols >= 0 && d_out_rowptr && h_out_nnz, SKM_E_BADARG, "skm_csr_remap_columns: bad argument");
    SKM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    *h_out_nnz = 0;
    if (n == 0) {
        SKM_HIP(hipMemsetAsync(d_out_rowptr, 0, sizeof(int64_t), st));
        return SKM_OK;
    }
    SKM_REQUIRE(d_rowptr && d_colidx && d_counts && (ncols == 0 || d_colmap) && d_out_col && d_out_val, SKM_E_BADARG,
                "skm_csr_remap_columns: null array");
    void *p;
    SKM_TRY(skm_ws(ctx, WS_A, sizeof(int64_t) * (size_t)(n + 1), &p));
    int64_t *kept = (int64_t *)p;
    SKM_HIP(hipMemsetAsync(kept + n, 0, sizeof(int64_t), st));
    const int grid = skm_grid_cap(ctx, skm_ceil_div(n, BLK / 64), 16);
    {
        SKM_PROF(ctx, "k_remap_count");
        k_remap_count<<<grid, BLK, 0, st>>>(d_rowptr, d_colidx, d_colmap, ncols, n, kept);
    }
    SKM_TRY(skm_check_launch("k_remap_count"));
    {
        size_t tmp = 0;
        SKM_HIP(rocprim::exclusive_scan(nullptr, tmp, kept, d_out_rowptr, (int64_t)0, (size_t)(n + 1), rocprim::plus<int64_t>(), st));
        SKM_TRY(skm_ws(ctx, WS_ROCPRIM, tmp, &p));
        SKM_PROF(ctx, "rocprim_scan_remap");
        SKM_HIP(rocprim::exclusive_scan(p, tmp, kept, d_out_rowptr, (int64_t)0, (size_t)(n + 1), rocprim::plus<int64_t>(), st));
    }
    {
        SKM_PROF(ctx, "k_remap_write");
        k_remap_write<<<grid, BLK, 0, st>>>(d_rowptr, d_colidx, d_counts, d_colmap, ncols, n, d_out_rowptr, d_out_col, d_out_val);
    }
    SKM_TRY(skm_check_launch("k_remap_write"));
    int64_t *h = (int64_t *)ctx->h_pinned;
    SKM_HIP(hipMemcpyAsync(h, d_out_rowptr + n, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    SKM_HIP(hipStreamSynchronize(st));
    *h_out_nnz = *h;
    return SKM_OK;
}

extern "C" int skm_basis_select(skm_ctx *ctx, int64_t ncols, const uint32_t *d_fs_order, const uint64_t *d_total,
                                uint64_t min_filter, uint32_t *d_keep, uint32_t *d_colmap, int64_t *h_nkeep)
{
    SKM_REQUIRE(ctx && ncols >= 0 && h_nkeep, SKM_E_BADARG, "skm_basis_select: bad argument");
    SKM_REQUIRE(ncols < ((int64_t)1 << 32) - 1, SKM_E_OVERFLOW, "skm_basis_select: 2^32 columns or more");
    *h_nkeep = 0;
    if (ncols == 0)
        return SKM_OK;
    SKM_REQUIRE(d_fs_order && d_total && d_keep && d_colmap, SKM_E_BADARG, "skm_basis_select: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    void *p;
    SKM_TRY(skm_ws(ctx, WS_A, sizeof(uint32_t) * (size_t)ncols, &p));
    uint32_t *flags = (uint32_t *)p;
    SKM_TRY(skm_ws(ctx, WS_B, sizeof(uint32_t) * (size_t)(ncols + 1), &p));
    uint32_t *pos = (uint32_t *)p;
    const unsigned grid = (unsigned)skm_ceil_div(ncols, BLK);
    SKM_PROF(ctx, "k_basis_select");
    k_select_flags<<<grid, BLK, 0, st>>>(ncols, d_fs_order, d_total, min_filter, flags);
    SKM_TRY(skm_check_launch("k_select_flags"));
    {
        size_t tmp = 0;
        SKM_HIP(rocprim::exclusive_scan(nullptr, tmp, flags, pos, 0u, (size_t)ncols, rocprim::plus<uint32_t>(), st));
        SKM_TRY(skm_ws(ctx, WS_ROCPRIM, tmp, &p));
        SKM_HIP(rocprim::exclusive_scan(p, tmp, flags, pos, 0u, (size_t)ncols, rocprim::plus<uint32_t>(), st));
    }
    k_select_scatter<<<grid, BLK, 0, st>>>(ncols, d_fs_order, flags, pos, d_keep, d_colmap);
    SKM_TRY(skm_check_launch("k_select_scatter"));
    uint32_t *h = (uint32_t *)ctx->h_pinned;
    SKM_HIP(hipMemcpyAsync(h, pos + (ncols - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SKM_HIP(hipMemcpyAsync(h + 1, flags + (ncols - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    SKM_HIP(hipStreamSynchronize(st));
    *h_nkeep = (int64_t)h[0] + (int64_t)h[1];
    return SKM_OK;
}
