// Dense small-basis path: the full |S|^k feature basis materialised (north_star's
// "N x |S|^k count matrix") and the cosine matrix as a true dense GEMM on the matrix cores.
//
// Reference behaviour being replaced:
//   count matrix over a full basis        snekmer/vectorize.py:259-290 (intent), rules/learn.smk:359-383
//   sklearn cosine_similarity(X, Y)       snekmer/rules/apply.smk:282-284, rules/learn.smk:821-823
//
// Only legitimate when |S|^k is small (hydro k<=20, solvacc k<=12, ...: SURVEY.md D2); at
// |S|^k = 6^12 the sparse path of skm_cosine_csr.hip is the only one that fits.
//
//   k_count_dense       wave per sequence: class ranks staged in LDS, window codes formed from LDS,
//                       one atomic increment per valid window into row i of the count matrix
//                       (uint32 words; uint16 cells are incremented through their containing word).
//   k_cosine_dense_i8   128x128 output tile per workgroup, K-step 64, int8 operands staged in LDS,
//                       v_mfma_i32_32x32x32_i8 with int32 accumulation (exact), float32 epilogue
//                       acc * rnorm_x[i] * rnorm_y[j].
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include <rocprim/rocprim.hpp>

#include "skm_common.h"

namespace {

// ------------------------------------------------------------------------------- count scatter
template <typename CELL>
__global__ __launch_bounds__(64) void k_count_dense(skm_lut256 lut, int nsym, int k,
                                                    const uint8_t *__restrict__ seq,
                                                    const int64_t *__restrict__ off, int64_t n,
                                                    CELL *__restrict__ out, int64_t ld, int *__restrict__ overflow)
{
    constexpr int TILE = 1024;
    __shared__ uint8_t s_lut[256];
    __shared__ uint8_t s_rank[TILE + 64];
    const int lane = threadIdx.x;
    reinterpret_cast<uint32_t *>(s_lut)[lane] = reinterpret_cast<const uint32_t *>(lut.b)[lane];
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const int64_t b = off[i];
        int64_t e = off[i + 1];
        while (e > b && seq[e - 1] == '*')  // every lane walks the same few bytes
            --e;
        const int len = (int)(e - b);
        const int w = len - k + 1;
        if (sizeof(CELL) == 2 && w > 65535 && lane == 0)
            *overflow = 1;
        CELL *row = out + i * ld;
        for (int t0 = 0; t0 < w; t0 += TILE) {
            __syncthreads();
            const int span = min(TILE + k - 1, len - t0);
            for (int p = lane; p < span; p += 64)
                s_rank[p] = s_lut[seq[b + t0 + p]];
            __syncthreads();
            const int wt = min(TILE, w - t0);
            for (int p = lane; p < wt; p += 64) {
                uint32_t c = 0, bad = 0;
                for (int j = 0; j < k; ++j) {
                    const uint32_t r = s_rank[p + j];
                    bad |= (r == 0xFFu);
                    c = c * (uint32_t)nsym + r;
                }
                if (!bad) {
                    if (sizeof(CELL) == 4) {
                        atomicAdd(reinterpret_cast<unsigned int *>(row) + c, 1u);
                    } else {
                        // 16-bit cell inside its aligned 32-bit word (row base is 4-byte aligned: ld even)
                        unsigned int *word = reinterpret_cast<unsigned int *>(row) + (c >> 1);
                        atomicAdd(word, (c & 1u) ? 0x10000u : 1u);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------- i8 MFMA cosine
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int LDS_ROW = BK + 16;  // 80-byte rows: 16-byte reads of 32 consecutive rows spread over the banks

template <int MODE>
__global__ __launch_bounds__(256) void k_cosine_dense_i8(int64_t n, int64_t m, int64_t kdim,
                                                         const int8_t *__restrict__ X,
                                                         const int8_t *__restrict__ Y,
                                                         const float *__restrict__ xr,
                                                         const float *__restrict__ yr, float *__restrict__ out,
                                                         int64_t ld)
{
    __shared__ __attribute__((aligned(16))) int8_t s_a[BM * LDS_ROW];
    __shared__ __attribute__((aligned(16))) int8_t s_b[BN * LDS_ROW];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;  // 2 x 2 waves, each 64 x 64 of the tile
    const int64_t row0 = (int64_t)blockIdx.y * BM, col0 = (int64_t)blockIdx.x * BN;

    i32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[a][b][r] = 0;

    // staging: 256 threads x 16 B = 64 rows x 64 B per pass, two passes per operand
    const int st_row = tid >> 2, st_chunk = tid & 3;
    for (int64_t k0 = 0; k0 < kdim; k0 += BK) {
        i32x4 va[2], vb[2];
#pragma unroll
        for (int pss = 0; pss < 2; ++pss) {
            const int r = st_row + pss * 64;
            const int64_t gi = row0 + r, gj = col0 + r;
            va[pss] = gi < n ? *reinterpret_cast<const i32x4 *>(X + gi * kdim + k0 + st_chunk * 16) : (i32x4){0, 0, 0, 0};
            vb[pss] = gj < m ? *reinterpret_cast<const i32x4 *>(Y + gj * kdim + k0 + st_chunk * 16) : (i32x4){0, 0, 0, 0};
        }
        __syncthreads();  // previous step's fragment reads are done
#pragma unroll
        for (int pss = 0; pss < 2; ++pss) {
            const int r = st_row + pss * 64;
            *reinterpret_cast<i32x4 *>(s_a + r * LDS_ROW + st_chunk * 16) = va[pss];
            *reinterpret_cast<i32x4 *>(s_b + r * LDS_ROW + st_chunk * 16) = vb[pss];
        }
        __syncthreads();
        // fragment of v_mfma_i32_32x32x32_i8: lane l holds 16 consecutive k of row (l & 31),
        // k offset 16 * (l >> 5)
        const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            i32x4 fa[2], fb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[t] = *reinterpret_cast<const i32x4 *>(s_a + (wr * 64 + t * 32 + fr) * LDS_ROW + ks * 32 + fh * 16);
                fb[t] = *reinterpret_cast<const i32x4 *>(s_b + (wc * 64 + t * 32 + fr) * LDS_ROW + ks * 32 + fh * 16);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
    }

    // epilogue: C/D layout of the 32x32 shapes: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int ccol = lane & 31, chalf = lane >> 5;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int64_t j = col0 + wc * 64 + b * 32 + ccol;
            const float rj = j < m ? yr[j] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t i = row0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * chalf;
                if (i < n && j < m) {
                    float o = (float)acc[a][b][r] * xr[i] * rj;
                    if (MODE == 1) {
                        o = fminf(fmaxf(1.0f - o, 0.0f), 2.0f);
                        if (i == j)
                            o = 0.0f;
                    }
                    out[i * ld + j] = o;
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------- i8 MFMA cosine, v2
// Same 128 x 128 tile and 32x32x32 MFMA, but the operands go global -> LDS directly
// (global_load_lds_dwordx4: 1 KiB per wave instruction, no VGPR staging), two LDS stages so the
// loads of K-step s+1 are in flight while step s is multiplied (counted vmcnt + raw s_barrier:
// __syncthreads() would drain the DMA), K-step 128, and an XOR swizzle of the 16-byte chunk index
// (chunk ^ ((row >> 1) & 7)) applied to the SOURCE address and to the fragment reads, which makes
// the ds_read_b128 fragment reads bank-conflict free on 128-byte rows.
constexpr int BK2 = 128;
constexpr int STAGE_BYTES = (BM + BN) * BK2;  // 32 KiB

// SYM (X is Y, round 5): a one-dimensional grid over the tile pairs ty <= tx; the tile below the diagonal gets its cells (the same
// integer, scaled in the order the rectangular launch would use: identical bits) through mirrored stores (4-byte stores a row apart: a small launch's output is a few tens of megabytes and stays in
// L2 until its lines are complete - this form only serves launches of a few hundred tiles, where it beats the 256 x 256
// kernels: N = 3 383, K = 6 656 (the reference's CI proteome): 105 tiles of 256 x 256 0.109 ms, 729 of 128 x 128 0.104, 378 here).
template <int MODE, bool SYM = false>
__global__ __launch_bounds__(256) void k_cosine_dense_i8_v2(int64_t n, int64_t m, int64_t kdim,
                                                            const int8_t *__restrict__ X,
                                                            const int8_t *__restrict__ Y,
                                                            const float *__restrict__ xr,
                                                            const float *__restrict__ yr, float *__restrict__ out,
                                                            int64_t ld)
{
    __shared__ __attribute__((aligned(16))) int8_t s_t[2 * STAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    int64_t ty = blockIdx.y, tx = blockIdx.x;
    if (SYM) {  // blockIdx.x enumerates the pairs row by row: row ty holds tiles tx = ty .. T - 1
        const int64_t T = (n + BM - 1) / BM;
        int64_t rest = blockIdx.x;
        ty = 0;
        while (rest >= T - ty) {
            rest -= T - ty;
            ++ty;
        }
        tx = ty + rest;
    }
    const int64_t row0 = ty * BM, col0 = tx * BN;

    i32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[a][b][r] = 0;

    // staging: wave w copies rows [32w, 32w+32) of both operands, 8 rows (1 KiB) per instruction;
    // lane l lands at (row l/8, chunk l%8) and therefore fetches source chunk (l%8) ^ ((row>>1)&7)
    const int srow = lane >> 3, schunk = lane & 7;
    auto stage = [&](int buf, int64_t k0) {
        int8_t *sa = s_t + buf * STAGE_BYTES, *sb = sa + BM * BK2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = wid * 32 + q * 8 + srow;
            const int src_chunk = schunk ^ ((r >> 1) & 7);
            const int64_t gi = min(row0 + r, n - 1), gj = min(col0 + r, m - 1);  // clamp: masked in the epilogue
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(X + gi * kdim + k0 + src_chunk * 16),
                                             (__attribute__((address_space(3))) void *)(sa + (wid * 32 + q * 8) * BK2), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Y + gj * kdim + k0 + src_chunk * 16),
                                             (__attribute__((address_space(3))) void *)(sb + (wid * 32 + q * 8) * BK2), 16, 0, 0);
        }
    };

    const int fr = lane & 31, fh = lane >> 5;
    const int64_t nsteps = kdim / BK2;
    stage(0, 0);
    for (int64_t s = 0; s < nsteps; ++s) {
        const int buf = (int)(s & 1);
        // everyone has finished reading buf^1 (step s-1) before it is overwritten
        __builtin_amdgcn_s_barrier();
        if (s + 1 < nsteps) {
            stage(buf ^ 1, (s + 1) * BK2);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // step s landed; step s+1 (8 loads) in flight
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const int8_t *sa = s_t + buf * STAGE_BYTES, *sb = sa + BM * BK2;
#pragma unroll
        for (int ks = 0; ks < BK2 / 32; ++ks) {
            i32x4 fa[2], fb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ra = wr * 64 + t * 32 + fr, rb = wc * 64 + t * 32 + fr;
                const int ca = (ks * 2 + fh) ^ ((ra >> 1) & 7), cb = (ks * 2 + fh) ^ ((rb >> 1) & 7);
                fa[t] = *reinterpret_cast<const i32x4 *>(sa + ra * BK2 + ca * 16);
                fb[t] = *reinterpret_cast<const i32x4 *>(sb + rb * BK2 + cb * 16);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
    }

    const int ccol = lane & 31, chalf = lane >> 5;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int64_t j = col0 + wc * 64 + b * 32 + ccol;
            const float rj = j < m ? yr[j] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t i = row0 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * chalf;
                if (i < n && j < m) {
                    float o = (float)acc[a][b][r] * xr[i] * rj;
                    if (MODE == 1) {
                        o = fminf(fmaxf(1.0f - o, 0.0f), 2.0f);
                        if (i == j)
                            o = 0.0f;
                    }
                    out[i * ld + j] = o;
                    if (SYM && tx != ty) {  // the cell below the diagonal, multiplied in the order its own tile would use:
                        float om = (float)acc[a][b][r] * rj * xr[i];  // (acc * r_j) * r_i, the bits of the rectangular launch
                        if (MODE == 1)
                            om = fminf(fmaxf(1.0f - om, 0.0f), 2.0f);
                        out[j * ld + i] = om;
                    }
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------- 256 x 256 tiles
// (round 1's lock-step 256 x 256 kernel, "v3", and the row-major staging of v4 were removed in round 5: v5 / v4 from tiled
// operand copies cover every shape they served; their A/B numbers are in profiles/r02_dense_mfma.json and r04_dense_mfma.json)
constexpr int BM3 = 256, BN3 = 256;
#ifndef SKM_ST_W
#define SKM_ST_W 4  // tiles per row of a supertile (4 rows x SKM_ST_W columns of tiles go to consecutive workgroups of one XCD)
#endif

// Row-major int8 [rows x kdim] -> the tiled image k_cosine_dense_i8_v4<.., TILED> stages from (see there).
// One thread per 16 bytes; rows are padded with zeros to a multiple of 256.
__global__ __launch_bounds__(256) void k_retile_i8(int64_t rows, int64_t kdim, const int8_t *__restrict__ in, int64_t nrb,
                                                   int8_t *__restrict__ out)
{
    const int64_t nst = kdim / 64;
    const int64_t total = nst * nrb * 64;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)(t & 63);
        const int64_t piece = t >> 6, rb = piece % nrb, sidx = piece / nrb;
        const int64_t r = rb * 16 + (l >> 2);
        const int src_chunk = (l & 3) ^ (int)((r >> 2) & 3);
        int4 v = make_int4(0, 0, 0, 0);
        if (r < rows)
            v = *reinterpret_cast<const int4 *>(in + r * kdim + sidx * 64 + src_chunk * 16);
        *reinterpret_cast<int4 *>(out + t * 16) = v;
    }
}

// ------------------------------------------------------------------------------- i8 MFMA cosine, v4
// The v3 tile (256 x 256, 8 waves of 128 x 64) with the K loop restructured so that the two waves that
// share a SIMD never want the same pipe at the same time.  In v3 all eight waves issue their LDS-DMA
// staging, then all read fragments, then all run MFMAs: while the staging instructions issue (60-185
// cycles each on this chip) and the fragment reads return, the matrix pipe idles (PMC: busy 48 %).
// Here a K stage is 64 bytes deep (32 KiB of LDS, a ring of four), every stage is cut into a LOAD
// interval (12 fragment reads into registers) and a COMPUTE interval (16 MFMAs on registers, with the 4
// LDS-DMA instructions for the stage three ahead issued between them), each closed by one raw
// s_barrier, and waves 4-7 run one interval behind waves 0-3 (one extra barrier up front): on every
// SIMD one wave computes while its partner loads.
//   RAW  stage s+1 is read in L(s+1).  Every wave waits (counted vmcnt) for ITS part of stage s+1 at the
//        end of its own L(s); the later of the two groups does so in the interval before group 0's
//        L(s+1), so the barrier between them orders every part.
//   WAR  stage s+3 goes to the slot of stage s-1, last read in L(s-1), whose reads are waited for
//        (lgkmcnt(0)) before the barrier that closes it; the first writer issues two intervals later.
// SYM (X is Y): only tiles on or above the diagonal are computed; a tile above it is also stored
// transposed (four consecutive rows of an accumulator column are 16 contiguous bytes of the mirror).
constexpr int BK4 = 64;
constexpr int NSLOT4 = 4;
constexpr int SLOT4_BYTES = (BM3 + BN3) * BK4;  // 32 KiB

// The operands were re-laid out by k_retile_i8 so that the 16 rows x 64 B one LDS-DMA instruction lands are
// 1 KiB of CONTIGUOUS global memory in exactly the LDS image (swizzle included), K stage major: piece (stage s, row
// block rb) at ((s * nrb + rb) * 1024).  From row-major operands the same instruction touches 16 half lines (one
// 64-byte L2 request each, 2.1e9 per launch at N = 32 768, K = 16 384: the kernel's bound, section 5.2 of
// DESIGN.md); from the tiled image it reads 8 whole lines.  Rows past n / m are zero in the image (no clamping).
template <int MODE, bool SYM>
__global__ __launch_bounds__(512) void k_cosine_dense_i8_v4(int64_t n, int64_t m, int64_t kdim,
                                                            const int8_t *__restrict__ X,
                                                            const int8_t *__restrict__ Y,
                                                            const float *__restrict__ xr,
                                                            const float *__restrict__ yr, float *__restrict__ out,
                                                            int64_t ld)
{
    __shared__ __attribute__((aligned(16))) int8_t s_t[NSLOT4 * SLOT4_BYTES];  // the ONLY shared object (128 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 2, wc = wid & 3;
    const int64_t nty = (n + BM3 - 1) / BM3, ntx = (m + BN3 - 1) / BN3;
    const int64_t nsx = (ntx + SKM_ST_W - 1) / SKM_ST_W;
    const int64_t nwg = gridDim.x, q8 = nwg / 8, r8 = nwg % 8, xcd = blockIdx.x % 8;
    const int64_t seq = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + blockIdx.x / 8;
    int64_t st = seq / (4 * SKM_ST_W);
    const int64_t within = seq % (4 * SKM_ST_W);
    int64_t sy, sx;
    if (SYM) {
        // only the 4 x 8 supertiles that reach the diagonal or lie above it are enumerated (row sy keeps
        // columns sx >= sy / 2), so that every XCD's contiguous share of the grid holds the same amount of work
        sy = 0;
        for (;; ++sy) {
            const int64_t cnt = nsx - sy / (SKM_ST_W / 4);
            if (st < cnt)
                break;
            st -= cnt;
        }
        sx = sy / (SKM_ST_W / 4) + st;
    } else {
        sy = st / nsx;
        sx = st % nsx;
    }
    const int64_t ty = sy * 4 + within / SKM_ST_W, tx = sx * SKM_ST_W + within % SKM_ST_W;
    if (ty >= nty || tx >= ntx || (SYM && ty > tx))
        return;
    const int64_t row0 = ty * BM3, col0 = tx * BN3;

    i32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[a][b][r] = 0;

    // staging: one LDS-DMA instruction lands 16 rows x 64 B linearly (lane l -> row l / 4, 16-byte chunk l % 4);
    // the XOR swizzle lives on the SOURCE chunk and on the fragment reads (same involution).  A wave stages
    // rows [32 w, 32 w + 32) of both operands: four instructions per stage, piece p = 0..3 (A q0, B q0, A q1, B q1).
    auto stage_piece = [&](int slot, int64_t k0, int p) {
        int8_t *sa = s_t + slot * SLOT4_BYTES, *sb = sa + BM3 * BK4;
        const int q = p >> 1;
        const int64_t nrbx = (n + BM3 - 1) / BM3 * (BM3 / 16), nrby = (m + BN3 - 1) / BN3 * (BN3 / 16);
        const int64_t sidx = k0 / BK4;
        if ((p & 1) == 0) {
            const int64_t rb = row0 / 16 + wid * 2 + q;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(X + ((sidx * nrbx + rb) * 64 + lane) * 16),
                                             (__attribute__((address_space(3))) void *)(sa + (wid * 32 + q * 16) * BK4), 16, 0, 0);
        } else {
            const int64_t rb = col0 / 16 + wid * 2 + q;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(Y + ((sidx * nrby + rb) * 64 + lane) * 16),
                                             (__attribute__((address_space(3))) void *)(sb + (wid * 32 + q * 16) * BK4), 16, 0, 0);
        }
    };
    auto stage = [&](int slot, int64_t k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
            stage_piece(slot, k0, p);
    };

    const int fr = lane & 31, fh = lane >> 5;
    const int64_t nst = kdim / BK4;  // >= 4 (checked by the launcher)
    stage(0, 0);
    stage(1, BK4);
    stage(2, 2 * BK4);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // this wave's part of stage 0
    __builtin_amdgcn_s_barrier();
    if (wr == 1)
        __builtin_amdgcn_s_barrier();  // waves 4-7 run one interval behind

    i32x4 fa[4][2], fb[2][2];
    for (int64_t s = 0; s < nst; ++s) {
        // ---- LOAD interval: fragments of stage s into registers
        const int8_t *sa = s_t + (int)(s & 3) * SLOT4_BYTES, *sb = sa + BM3 * BK4;
        {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int ra = wr * 128 + t * 32 + fr;
                    fa[t][ks] = *reinterpret_cast<const i32x4 *>(sa + ra * BK4 + (((ks * 2 + fh) ^ ((ra >> 2) & 3)) * 16));
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int rb = wc * 64 + t * 32 + fr;
                    fb[t][ks] = *reinterpret_cast<const i32x4 *>(sb + rb * BK4 + (((ks * 2 + fh) ^ ((rb >> 2) & 3)) * 16));
                }
            }
        }
        // this wave's part of stage s+1 has landed (stage s+2 may still fly); the partner group does the same
        // one interval later, which is still one barrier before anyone reads stage s+1
        if (s + 2 < nst)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments in registers: the slot may be restaged
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- COMPUTE interval: 16 MFMAs on registers, the four staging instructions of stage s+3 (into the
        // slot of stage s-1) issued between them: an LDS-DMA instruction costs ~60 cycles among MFMAs and
        // 100-185 next to fragment reads, and the LOAD interval is the longer one
        const bool more = s + 3 < nst;
        const int nslot = (int)((s + 3) & 3);
        const int64_t nk0 = (s + 3) * BK4;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a][ks], fb[b][ks], acc[a][b], 0, 0, 0);
                }
                if ((a & 1) == 1) {  // after every fourth MFMA
                    __builtin_amdgcn_sched_barrier(0);
                    if (more)
                        stage_piece(nslot, nk0, ks * 2 + (a >> 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (wr == 0)
        __builtin_amdgcn_s_barrier();  // pairs with the extra barrier of waves 4-7

    const int ccol = lane & 31, chalf = lane >> 5;
    const bool mirror = SYM && ty < tx;
    // the mirrored copy of a 32 x 32 sub-tile goes through LDS (free after the K loop; 4.5 KiB per wave) so
    // that every store instruction writes whole 128-byte lines: storing the accumulator columns directly
    // scatters 16-byte pieces over 32 rows per instruction and ran at 0.6 TB/s
    constexpr int TROW = 36;  // floats per transposed row: 16-byte aligned, spreads the banks
    float *tbuf = reinterpret_cast<float *>(s_t) + wid * (32 * TROW);
    const bool vec_ok = (ld & 3) == 0 && ((uintptr_t)out & 15) == 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int64_t j = col0 + wc * 64 + b * 32 + ccol;
            const float rj = j < m ? yr[j] : 0.0f;
            const int64_t ibase = row0 + wr * 128 + a * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int li = 8 * q + 4 * chalf;  // registers 4q .. 4q+3 hold rows li .. li+3 of the sub-tile
                float o[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t i = ibase + li + u;
                    float v = 0.0f, vm = 0.0f;
                    if (i < n && j < m) {
                        v = (float)acc[a][b][4 * q + u] * xr[i] * rj;
                        vm = (float)acc[a][b][4 * q + u] * rj * xr[i];  // the cell below the diagonal: (acc * r_j) * r_i, the
                        if (MODE == 1) {                                 // bits its own tile (the rectangular launch) would give
                            v = fminf(fmaxf(1.0f - v, 0.0f), 2.0f);
                            vm = fminf(fmaxf(1.0f - vm, 0.0f), 2.0f);
                            if (i == j)
                                v = vm = 0.0f;
                        }
                        out[i * ld + j] = v;
                    }
                    o[u] = vm;
                }
                if (mirror)
                    *reinterpret_cast<float4 *>(tbuf + ccol * TROW + li) = make_float4(o[0], o[1], o[2], o[3]);
            }
            if (mirror) {
                // tbuf[c][i] = value at (row ibase + i, column jbase + c); row jbase + c of the mirror gets columns ibase ..
                const int64_t jbase = col0 + wc * 64 + b * 32;
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int c = pass * 8 + (lane >> 3), i4 = (lane & 7) * 4;
                    const float4 v = *reinterpret_cast<const float4 *>(tbuf + c * TROW + i4);
                    const int64_t jj = jbase + c, ii = ibase + i4;
                    if (jj < m) {
                        float *dst = out + jj * ld + ii;
                        if (vec_ok && ii + 3 < n) {
                            *reinterpret_cast<float4 *>(dst) = v;
                        } else {
                            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (ii + u < n)
                                    dst[u] = e[u];
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------- i8 MFMA cosine, v5
// v4 stages both operands through LDS, and its LDS-DMA issue (60-185 cycles per instruction, four per wave and stage)
// and fragment reads leave the matrix pipe idle 45 % of the time (section 5.2 of profiles/HISTORY.md).  Here only A goes
// through LDS.  The eight waves sit side by side (1 x 8): wave w owns all 256 rows x columns [32 w, 32 w + 32) of the
// tile, so
//   A (256 rows x 64 B per stage) goes through LDS (two LDS-DMA instructions per wave per stage, a ring of eight
//     16 KiB slots, staged DA = 3 stages ahead) and every wave reads all of it (16 ds_read_b128 per stage);
//   B (this wave's 32 columns x 64 B per stage) is loaded by the wave itself, three stages ahead, straight into the
//     MFMA operand registers from an image laid out in LANE ORDER by k_retile_b_i8 (piece (stage s, column block cb,
//     half ks) = 1 KiB at (((s * ncb + cb) * 2 + ks) * 1024), lane l at l * 16: row cb * 32 + l % 32, K chunk
//     2 ks + l / 32): one fully coalesced 1 KiB load per MFMA K step, no LDS, no duplicate.
// Same staggered wave groups as v4 (waves 4-7 one interval behind), same barriers, same symmetric form.
// vm queue of a wave: every COMPUTE interval issues B(s+3) [2 loads] then A(s+DA) [2 LDS-DMA]; at the end of LOAD(s)
// it needs B(s) (the first two of compute(s-3)'s four) and its part of A(s+1) (the last two of compute(s+1-DA)'s):
// with DA = 3 the four operations of compute(s-1) may still fly -> vmcnt(4), on every path (past the end the last stage
// is re-fetched into dead slots / registers, so that the count never depends on s).
// Measured (profiles/r04_dense_mfma.json, N = M = 32 768, K = 16 384): 11.1-11.4 ms against v4's 12.7 rectangular,
// 6.9 against 7.4 symmetric.  Tried around it and not kept (profiles/r04_experiments/README.md): v4 with only B moved
// to registers (its 2 x 4 wave grid fetches B twice: 11.6-12.0), the memory instructions issued beside the fragment
// reads instead of among the MFMAs (12.7-12.9), DA = 4 / 6 (same), a software-pipelined K loop with one barrier per
// stage and no staggered groups (same: 11.3).  All of them stop at 137 GB / 11.2 ms = 12.2 TB/s of L2 -> CU traffic.
constexpr int NSLOT5 = 8;
constexpr int SLOT5_BYTES = BM3 * BK4;  // 16 KiB

__global__ __launch_bounds__(256) void k_retile_b_i8(int64_t rows, int64_t kdim, const int8_t *__restrict__ in, int64_t ncb,
                                                     int8_t *__restrict__ out)
{
    const int64_t total = kdim / 64 * ncb * 2 * 64;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)(t & 63);
        const int64_t piece = t >> 6;
        const int ks = (int)(piece & 1);
        const int64_t cb = (piece >> 1) % ncb, sidx = (piece >> 1) / ncb;
        const int64_t r = cb * 32 + (l & 31);
        int4 v = make_int4(0, 0, 0, 0);
        if (r < rows)
            v = *reinterpret_cast<const int4 *>(in + r * kdim + sidx * 64 + (ks * 2 + (l >> 5)) * 16);
        *reinterpret_cast<int4 *>(out + t * 16) = v;
    }
}

// SPLIT (small problems: fewer tiles than compute units): the K stages of a tile are shared by `nsplit` workgroups
// (grid = tile slots x nsplit, split-major; `sps` stages each, a multiple of 4).  Every workgroup of a tile stores its int32
// partial tile to its slab (the accumulator registers as they stand: 1 KiB per wave instruction), takes a ticket, and the
// LAST to arrive adds the others' slabs to its own registers and runs the epilogue (no workgroup ever waits for another:
// any grid size is safe).  Hand-off per MI355X_MICROARCH.md / G16: stores drained, workgroup barrier, one lane's agent-scope
// acq_rel ticket (release of the slab, acquire of the others'), barrier, plain loads.  The last arriver puts the ticket
// back to zero for the next launch.
template <int MODE, bool SYM, int DA = 3, bool SPLIT = false>
__global__ __launch_bounds__(512) void k_cosine_dense_i8_v5(int64_t n, int64_t m, int64_t kdim,
                                                            const int8_t *__restrict__ X,   // A image (k_retile_i8)
                                                            const int8_t *__restrict__ YB,  // B image (k_retile_b_i8)
                                                            const float *__restrict__ xr,
                                                            const float *__restrict__ yr, float *__restrict__ out,
                                                            int64_t ld, int nsplit = 1, int64_t sps = 0,
                                                            int32_t *__restrict__ slabs = nullptr,
                                                            uint32_t *__restrict__ tickets = nullptr)
{
    __shared__ __attribute__((aligned(16))) int8_t s_t[NSLOT5 * SLOT5_BYTES];  // the ONLY shared object (128 KiB)
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t nty = (n + BM3 - 1) / BM3, ntx = (m + BN3 - 1) / BN3;
    const int64_t nsx = (ntx + SKM_ST_W - 1) / SKM_ST_W;
    const int64_t nwg = SPLIT ? gridDim.x / nsplit : gridDim.x;  // tile slots (a multiple of 32)
    const int64_t bid = SPLIT ? blockIdx.x % nwg : blockIdx.x;
    const int split = SPLIT ? (int)(blockIdx.x / nwg) : 0;
    const int64_t q8 = nwg / 8, r8 = nwg % 8, xcd = bid % 8;
    const int64_t seq = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / 8;
    int64_t st = seq / (4 * SKM_ST_W);
    const int64_t within = seq % (4 * SKM_ST_W);
    int64_t sy, sx;
    if (SYM) {  // as v4: only the supertiles that reach the diagonal or lie above it
        sy = 0;
        for (;; ++sy) {
            const int64_t cnt = nsx - sy / (SKM_ST_W / 4);
            if (st < cnt)
                break;
            st -= cnt;
        }
        sx = sy / (SKM_ST_W / 4) + st;
    } else {
        sy = st / nsx;
        sx = st % nsx;
    }
    const int64_t ty = sy * 4 + within / SKM_ST_W, tx = sx * SKM_ST_W + within % SKM_ST_W;
    if (ty >= nty || tx >= ntx || (SYM && ty > tx))
        return;
    const int64_t row0 = ty * BM3, col0 = tx * BN3;

    i32x16 acc[8];
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[a][r] = 0;

    const int64_t nst_all = kdim / BK4;  // a multiple of 4 (the launcher's condition)
    // this workgroup's K stages [s_lo, nst): both ends multiples of 4 (the launcher makes every split non-empty)
    const int64_t s_lo = SPLIT ? (int64_t)split * sps : 0;
    const int64_t nst = SPLIT ? min(nst_all, s_lo + sps) : nst_all;
    const int64_t nrbx = nty * (BM3 / 16);
    // A staging: wave w lands rows [32 w, 32 w + 32) of the stage, two instructions of 16 rows x 64 B
    auto stage_a = [&](int64_t slot_of, int64_t sidx, int q) {  // stage sidx into the slot of stage slot_of
        const int64_t rb = row0 / 16 + wid * 2 + q;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(X + ((sidx * nrbx + rb) * 64 + lane) * 16),
                                         (__attribute__((address_space(3))) void *)(s_t + (int)(slot_of & (NSLOT5 - 1)) * SLOT5_BYTES + (wid * 32 + q * 16) * BK4),
                                         16, 0, 0);
    };
    const int64_t ncb = ntx * (BN3 / 32);
    const int8_t *bp = YB + ((col0 / 32 + wid) * 2) * 1024 + lane * 16;
    const int64_t bstride = ncb * 2048;  // bytes per K stage of the B image
    i32x4 fbq[4][2];                     // ring by stage
    auto load_b = [&](auto BUF, int64_t sidx, int ks) {
        constexpr int buf = decltype(BUF)::value;
        fbq[buf][ks] = *reinterpret_cast<const i32x4 *>(bp + sidx * bstride + ks * 1024);
    };
#pragma unroll
    for (int s0 = 0; s0 < DA; ++s0) {
        stage_a(s_lo + s0, s_lo + s0, 0);
        stage_a(s_lo + s0, s_lo + s0, 1);
    }
    load_b(std::integral_constant<int, 0>{}, s_lo + 0, 0);
    load_b(std::integral_constant<int, 0>{}, s_lo + 0, 1);
    load_b(std::integral_constant<int, 1>{}, s_lo + 1, 0);
    load_b(std::integral_constant<int, 1>{}, s_lo + 1, 1);
    load_b(std::integral_constant<int, 2>{}, s_lo + 2, 0);
    load_b(std::integral_constant<int, 2>{}, s_lo + 2, 1);
    // builtin waits (0x0F70 = vmcnt(0), 0x0F78 = vmcnt(8); the other counters untouched), not asm: the compiler's own
    // wait-count model must see them, or it waits for everything before the first MFMA of a stage
    __builtin_amdgcn_s_waitcnt(0x0F70);  // once per tile
    __builtin_amdgcn_s_barrier();
    if (wid >= 4)
        __builtin_amdgcn_s_barrier();  // waves 4-7 run one interval behind

    const int fr = lane & 31, fh = lane >> 5;
    i32x4 fa[8][2];
    auto body = [&](auto CUR, int64_t s) {
        constexpr int cur = decltype(CUR)::value;
        const int8_t *sa = s_t + (int)(s & (NSLOT5 - 1)) * SLOT5_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int ra = t * 32 + fr;
                fa[t][ks] = *reinterpret_cast<const i32x4 *>(sa + ra * BK4 + (((ks * 2 + fh) ^ ((ra >> 2) & 3)) * 16));
            }
        static_assert(DA == 3 || DA == 4, "vmcnt below");
        __builtin_amdgcn_s_waitcnt(DA == 3 ? 0x0F74 : 0x0F78);  // vmcnt(4 / 8): low four bits of the immediate
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments in registers: the slot may be restaged
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        const int64_t na = min(s + DA, nst - 1), nb = min(s + 3, nst - 1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                acc[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a][ks], fbq[cur][ks], acc[a], 0, 0, 0);
                if ((a & 3) == 3) {  // after every fourth MFMA
                    __builtin_amdgcn_sched_barrier(0);
                    const int piece = ks * 2 + (a >> 2);  // 0, 1: B(s+3) halves; 2, 3: A(s+DA) halves
                    if (piece < 2)
                        load_b(std::integral_constant<int, (cur + 3) & 3>{}, nb, piece);
                    else  // past the end: the last stage again, into the (dead) slot of stage s + DA - 8
                        stage_a(s + DA, na, piece - 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    for (int64_t s = s_lo; s < nst; s += 4) {
        body(std::integral_constant<int, 0>{}, s);
        body(std::integral_constant<int, 1>{}, s + 1);
        body(std::integral_constant<int, 2>{}, s + 2);
        body(std::integral_constant<int, 3>{}, s + 3);
    }
    if (wid < 4)
        __builtin_amdgcn_s_barrier();  // pairs with the extra barrier of waves 4-7
    __builtin_amdgcn_s_waitcnt(0x0F70);  // the dead re-fetches have landed before the epilogue reuses the LDS
    __builtin_amdgcn_s_barrier();

    if (SPLIT) {
        // Hand-off.  A workgroup takes the tile's ticket as soon as its K stages are done.  Not the last one: it stores its
        // partial tile to its slab ([wave][a][q][lane] int4 = this lane's registers 4q .. 4q+3 of sub-tile a: 1 KiB per
        // wave instruction), drains the stores and counts itself in `done`.  The last one keeps its registers, waits until
        // `done` shows every other slab complete - it only ever waits for workgroups that hold a ticket, i.e. that are
        // running and wait for nobody: no residency assumption - adds them and runs the epilogue.  The slab bytes travel
        // with write-through (sc1) stores and sc1 loads instead of a release / acquire pair: those write back / invalidate
        // the XCD's whole L2, which at this point holds the other workgroups' freshly stored output tiles.
        constexpr int64_t SLAB = (int64_t)BM3 * BN3;  // int32 words
        uint32_t *s_ticket = reinterpret_cast<uint32_t *>(s_t);
        if (tid == 0)
            *s_ticket = __hip_atomic_fetch_add(tickets + 2 * bid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const uint32_t arrived = *s_ticket;
        __syncthreads();  // (the epilogue reuses the LDS)
        if (arrived != (uint32_t)nsplit - 1u) {
            i32x4 *mine = reinterpret_cast<i32x4 *>(slabs + (bid * (nsplit - 1) + arrived) * SLAB) + (wid * 32) * 64 + lane;
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    i32x4 v;
                    v[0] = acc[a][4 * q];
                    v[1] = acc[a][4 * q + 1];
                    v[2] = acc[a][4 * q + 2];
                    v[3] = acc[a][4 * q + 3];
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(mine + (a * 4 + q) * 64), "v"(v) : "memory");
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores ...
            __syncthreads();                                   // ... before one lane reports the slab complete
            if (tid == 0)
                __hip_atomic_fetch_add(tickets + 2 * bid + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (tid == 0) {
            while (__hip_atomic_load(tickets + 2 * bid + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (uint32_t)nsplit - 1u)
                __builtin_amdgcn_s_sleep(2);
            // both counters back to zero for the next launch: every workgroup of this tile is past its last atomic
            __hip_atomic_store(tickets + 2 * bid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(tickets + 2 * bid + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        for (int o = 0; o + 1 < nsplit; ++o) {
            const i32x4 *theirs = reinterpret_cast<const i32x4 *>(slabs + (bid * (nsplit - 1) + o) * SLAB) + (wid * 32) * 64 + lane;
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // 16 loads of 1 KiB in flight per wave (the operand registers of the K loop are free)
                i32x4 v[16];
#pragma unroll
                for (int z = 0; z < 16; ++z)
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[z]) : "v"(theirs + (h * 16 + z) * 64) : "memory");
                // (the loaded registers are operands of the wait, so that no use of them can be scheduled above it)
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                               "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])::"memory");
#pragma unroll
                for (int z = 0; z < 16; ++z) {
                    const int a = (h * 16 + z) >> 2, q = (h * 16 + z) & 3;
                    acc[a][4 * q] += v[z][0];
                    acc[a][4 * q + 1] += v[z][1];
                    acc[a][4 * q + 2] += v[z][2];
                    acc[a][4 * q + 3] += v[z][3];
                }
            }
        }
    }

    // epilogue for this wave's 8 sub-tiles of 32 x 32.  A lane holds one COLUMN of a sub-tile (16 rows of it), so storing
    // the registers as they stand is 16 dword stores per sub-tile (two 128-byte row pieces per instruction) and the
    // epilogue was store-ISSUE-bound: ~35 us per tile, a fifth of a large launch's time per tile and most of a small
    // launch's.  Both images now go through LDS (free after the K loop; 2 x 4.5 KiB per wave) and leave as 16-byte
    // stores, 1 KiB per instruction: the tile itself row-major (8 rows x 128 B per instruction), its mirror transposed.
    const int ccol = lane & 31, chalf = lane >> 5;
    const bool mirror = SYM && ty < tx;
    constexpr int TROW = 36;  // floats per buffered row: 16-byte aligned, spreads the banks
    float *tdir = reinterpret_cast<float *>(s_t) + wid * (2 * 32 * TROW), *tmir = tdir + 32 * TROW;
    const bool vec_ok = (ld & 3) == 0 && ((uintptr_t)out & 15) == 0;
    const int64_t jbase = col0 + wid * 32;
    const int64_t j = jbase + ccol;
    const float rj = j < m ? yr[j] : 0.0f;
    const int prow = lane >> 3, pc4 = (lane & 7) * 4;  // a 16-byte store: row 8 * pass + prow, columns pc4 .. pc4 + 3
#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int64_t ibase = row0 + a * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int li = 8 * q + 4 * chalf;  // registers 4q .. 4q+3 hold rows li .. li+3 of the sub-tile
            float o[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = ibase + li + u;
                float v = 0.0f, vm = 0.0f;
                if (i < n && j < m) {
                    v = (float)acc[a][4 * q + u] * xr[i] * rj;
                    vm = (float)acc[a][4 * q + u] * rj * xr[i];  // the cell below the diagonal: (acc * r_j) * r_i, the bits its
                    if (MODE == 1) {                             // own tile (the rectangular launch) would give
                        v = fminf(fmaxf(1.0f - v, 0.0f), 2.0f);
                        vm = fminf(fmaxf(1.0f - vm, 0.0f), 2.0f);
                        if (i == j)
                            v = vm = 0.0f;
                    }
                }
                o[u] = vm;
                tdir[(li + u) * TROW + ccol] = v;
            }
            if (mirror)
                *reinterpret_cast<float4 *>(tmir + ccol * TROW + li) = make_float4(o[0], o[1], o[2], o[3]);
        }
        // (one wave: its LDS operations complete in order, no barrier needed between the writes above and these reads)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = pass * 8 + prow;
            const float4 v = *reinterpret_cast<const float4 *>(tdir + r * TROW + pc4);
            const int64_t ii = ibase + r, jj = jbase + pc4;
            if (ii < n) {
                float *dst = out + ii * ld + jj;
                if (vec_ok && jj + 3 < m) {
                    *reinterpret_cast<float4 *>(dst) = v;
                } else {
                    const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (jj + u < m)
                            dst[u] = e[u];
                }
            }
        }
        if (mirror) {
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int c = pass * 8 + prow;
                const float4 v = *reinterpret_cast<const float4 *>(tmir + c * TROW + pc4);
                const int64_t jj = jbase + c, ii = ibase + pc4;
                if (jj < m) {
                    float *dst = out + jj * ld + ii;
                    if (vec_ok && ii + 3 < n) {
                        *reinterpret_cast<float4 *>(dst) = v;
                    } else {
                        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (ii + u < n)
                                dst[u] = e[u];
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------- dense -> CSR
// Sparse view of a dense count matrix (the inverse of k_count_dense / k_csr_to_dense).  A workgroup of
// four waves owns a row; wave w owns the contiguous quarter [w * seg, (w + 1) * seg) of its columns, so
// the non-zeros of a row come out in ascending column order without any cross-wave ordering step:
// pass 1 counts the non-zeros of every (row, quarter), a scan over those counts gives every quarter's
// position in the output, pass 2 writes (column, value) pairs, each wave numbering its own with a
// ballot prefix.  Rows are read with 16-byte loads when the layout allows it.
template <typename CELL>
__device__ __forceinline__ uint32_t cell_value(CELL c)
{
    return (uint32_t)c;
}

template <typename CELL, bool WRITE>
__global__ __launch_bounds__(256) void k_dense_to_csr(int64_t n, int64_t ncols, const CELL *__restrict__ in, int64_t ld,
                                                      int vec_ok, int64_t *__restrict__ seg_pos,
                                                      uint32_t *__restrict__ out_col, uint32_t *__restrict__ out_val)
{
    constexpr int PER = 16 / (int)sizeof(CELL);  // cells per 16-byte load
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // quarter boundaries on multiples of PER so that vector loads stay aligned
    const int64_t seg = ((ncols + 3) / 4 + PER - 1) / PER * PER;
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const CELL *row = in + i * ld;
        const int64_t c0 = min((int64_t)wid * seg, ncols), c1 = min(c0 + seg, ncols);
        int64_t pos = WRITE ? seg_pos[i * 4 + wid] : 0;
        uint32_t count = 0;
        for (int64_t base = c0; base < c1; base += 64 * PER) {
            const int64_t c = base + (int64_t)lane * PER;
            CELL v[PER];
            if (vec_ok && c + PER <= c1) {
                const uint4 w = *reinterpret_cast<const uint4 *>(row + c);
                *reinterpret_cast<uint4 *>(v) = w;
            } else {
#pragma unroll
                for (int u = 0; u < PER; ++u)
                    v[u] = c + u < c1 ? row[c + u] : CELL(0);
            }
            uint32_t mine = 0;
#pragma unroll
            for (int u = 0; u < PER; ++u)
                mine += v[u] != CELL(0);
            if (!WRITE) {
                count += mine;
            } else if (__any(mine != 0)) {
                // exclusive wave scan of the per-lane counts
                uint32_t incl = mine;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t up = __shfl_up(incl, o);
                    if (lane >= o)
                        incl += up;
                }
                int64_t at = pos + (incl - mine);
#pragma unroll
                for (int u = 0; u < PER; ++u)
                    if (v[u] != CELL(0)) {
                        out_col[at] = (uint32_t)(c + u);
                        out_val[at] = cell_value(v[u]);
                        ++at;
                    }
                pos += __shfl(incl, 63);
            }
        }
        if (!WRITE) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1)
                count += __shfl_down(count, o);
            if (lane == 0)
                seg_pos[i * 4 + wid] = (int64_t)count;
        }
    }
}

__global__ void k_rowptr_from_segments(int64_t n, const int64_t *__restrict__ seg_pos, int64_t *__restrict__ rowptr)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n)
        rowptr[i] = seg_pos[i * 4];
}

// 1/||row|| of an int8 matrix (and optionally the exact squared norms): wave per row, 16 bytes per lane.
// out[which] = min of v[0..n) (one workgroup; the rare int32-range check of skm_cosine_dense_i8)
__global__ __launch_bounds__(1024) void k_min_f32(int64_t n, const float *__restrict__ v, float *__restrict__ out, int which)
{
    __shared__ float s_w[16];
    float mn = INFINITY;
    for (int64_t i = threadIdx.x; i < n; i += 1024)
        mn = fminf(mn, v[i]);
    for (int o = 32; o > 0; o >>= 1)
        mn = fminf(mn, __shfl_xor(mn, o));
    if ((threadIdx.x & 63) == 0)
        s_w[threadIdx.x >> 6] = mn;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            mn = fminf(mn, s_w[w]);
        out[which] = mn;
    }
}

__global__ __launch_bounds__(256) void k_row_norms_i8(int64_t n, int64_t kdim, const int8_t *__restrict__ in,
                                                      float *__restrict__ rnorm, uint64_t *__restrict__ normsq)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += nwaves) {
        const int8_t *row = in + i * kdim;
        unsigned long long s = 0;
        for (int64_t c = (int64_t)lane * 16; c < kdim; c += 64 * 16) {  // kdim is a multiple of 64
            const uint4 w = *reinterpret_cast<const uint4 *>(row + c);
            const uint32_t q[4] = {w.x, w.y, w.z, w.w};
            uint32_t acc = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int v = (int)(int8_t)(q[u] >> (8 * bb));
                    acc += (uint32_t)(v * v);
                }
            s += acc;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            s += __shfl_down(s, o);
        if (lane == 0) {
            if (normsq)
                normsq[i] = s;
            if (rnorm)
                rnorm[i] = s ? (float)(1.0 / sqrt((double)s)) : 1.0f;
        }
    }
}

}  // namespace

extern "C" int skm_count_dense(skm_ctx *ctx, const uint8_t *h_rank, int nsym, int k, const uint8_t *d_seq,
                               const int64_t *d_off, int64_t n, int dtype, void *d_out, int64_t ld)
{
    SKM_REQUIRE(ctx && h_rank && d_off && n >= 0, SKM_E_BADARG, "skm_count_dense: bad argument");
    SKM_REQUIRE(dtype == 0 || dtype == 1, SKM_E_BADARG, "skm_count_dense: dtype must be 0 (uint16) or 1 (uint32)");
    SKM_REQUIRE(nsym >= 1 && nsym <= 254 && k >= 1 && k <= 64, SKM_E_BADARG, "skm_count_dense: bad nsym/k");
    int64_t space = 1;
    for (int j = 0; j < k; ++j) {
        space *= nsym;
        SKM_REQUIRE(space <= ((int64_t)1 << 26), SKM_E_UNSUPPORTED,
                    "skm_count_dense: |S|^k = %d^%d exceeds 2^26 dense columns; use skm_count_csr", nsym, k);
    }
    SKM_REQUIRE(ld >= space && (dtype == 1 || ld % 2 == 0), SKM_E_BADARG,
                "skm_count_dense: ld (%lld) must be >= %lld (and even for uint16)", (long long)ld, (long long)space);
    if (n == 0)
        return SKM_OK;
    SKM_REQUIRE(d_out && d_seq, SKM_E_BADARG, "skm_count_dense: null buffer");
    SKM_HIP(hipSetDevice(ctx->device));
    skm_lut256 lut;
    memcpy(lut.b, h_rank, 256);
    const size_t cell = dtype == 0 ? 2 : 4;
    void *p;
    SKM_TRY(skm_ws(ctx, WS_SMALL, 4096, &p));
    int *flag = (int *)((uint8_t *)p + 3072);
    SKM_HIP(hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
    {
        SKM_PROF(ctx, "memset_count_dense");
        SKM_HIP(hipMemsetAsync(d_out, 0, cell * (size_t)n * (size_t)ld, ctx->stream));
    }
    {
        SKM_PROF(ctx, "k_count_dense");
        const int grid = skm_grid_cap(ctx, n, 64);
        if (dtype == 0)
            k_count_dense<uint16_t><<<grid, 64, 0, ctx->stream>>>(lut, nsym, k, d_seq, d_off, n, (uint16_t *)d_out, ld, flag);
        else
            k_count_dense<uint32_t><<<grid, 64, 0, ctx->stream>>>(lut, nsym, k, d_seq, d_off, n, (uint32_t *)d_out, ld, flag);
    }
    SKM_TRY(skm_check_launch("k_count_dense"));
    if (dtype == 0) {
        SKM_HIP(hipMemcpyAsync(ctx->h_pinned, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SKM_HIP(hipStreamSynchronize(ctx->stream));
        SKM_REQUIRE(*(int *)ctx->h_pinned == 0, SKM_E_OVERFLOW,
                    "skm_count_dense: a sequence has more than 65535 windows; use dtype uint32");
    }
    return SKM_OK;
}

static int cosine_dense_i8_impl(skm_ctx *ctx, int64_t n, int64_t m, int64_t kdim, const int8_t *d_x, const int8_t *d_y,
                                const float *d_xrnorm, const float *d_yrnorm, int mode, float *d_out, int64_t ld);

extern "C" int skm_cosine_dense_i8(skm_ctx *ctx, int64_t n, int64_t m, int64_t kdim, const int8_t *d_x,
                                   const int8_t *d_y, const float *d_xrnorm, const float *d_yrnorm, int mode,
                                   float *d_out, int64_t ld)
{
    SKM_REQUIRE(mode >= 0 && mode <= 2, SKM_E_BADARG, "skm_cosine_dense_i8: mode must be 0, 1 or 2");
    // mode 2: distance between two different matrices (no diagonal rule), see skm_cosine_csr
    int rc = cosine_dense_i8_impl(ctx, n, m, kdim, d_x, d_y, d_xrnorm, d_yrnorm, mode == 2 ? 0 : mode, d_out, ld);
    if (rc == SKM_OK && mode == 2)
        rc = skm_similarity_to_distance(ctx, n, m, d_out, ld);
    return rc;
}

static int cosine_dense_i8_impl(skm_ctx *ctx, int64_t n, int64_t m, int64_t kdim, const int8_t *d_x, const int8_t *d_y,
                                const float *d_xrnorm, const float *d_yrnorm, int mode, float *d_out, int64_t ld)
{
    SKM_REQUIRE(ctx && n >= 0 && m >= 0 && kdim >= 0, SKM_E_BADARG, "skm_cosine_dense_i8: bad argument");
    SKM_REQUIRE(kdim % 64 == 0, SKM_E_BADARG, "skm_cosine_dense_i8: kdim (%lld) must be a multiple of 64", (long long)kdim);
    SKM_REQUIRE(ld >= m, SKM_E_BADARG, "skm_cosine_dense_i8: ld < m");
    SKM_REQUIRE(mode == 0 || mode == 1, SKM_E_BADARG, "skm_cosine_dense_i8: mode must be 0 or 1");
    if (n == 0 || m == 0)
        return SKM_OK;
    SKM_REQUIRE(d_x && d_y && d_xrnorm && d_yrnorm && d_out, SKM_E_BADARG, "skm_cosine_dense_i8: null array");
    SKM_REQUIRE((((uintptr_t)d_x | (uintptr_t)d_y) & 15) == 0, SKM_E_BADARG, "skm_cosine_dense_i8: operands must be 16-byte aligned");
    SKM_HIP(hipSetDevice(ctx->device));
    if (kdim * 127 * 127 >= ((int64_t)1 << 31)) {
        // Wider than 133 143 columns, 127^2 * kdim no longer bounds the int32 accumulators: the norms must
        // (<x, y> <= 1 / (xrnorm * yrnorm), the test of skm_cosine_csr).  Rare shape, so this one waits for the device.
        void *p;
        SKM_TRY(skm_ws(ctx, WS_SMALL, 4096, &p));
        float *mins = (float *)((uint8_t *)p + 3584);
        k_min_f32<<<1, 1024, 0, ctx->stream>>>(n, d_xrnorm, mins, 0);
        k_min_f32<<<1, 1024, 0, ctx->stream>>>(m, d_yrnorm, mins, 1);
        SKM_TRY(skm_check_launch("k_min_f32"));
        SKM_HIP(hipMemcpyAsync(ctx->h_pinned, mins, 2 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        SKM_HIP(hipStreamSynchronize(ctx->stream));
        const float *h = (const float *)ctx->h_pinned;
        SKM_REQUIRE((double)h[0] * (double)h[1] > 0x1p-31 * (1.0 + 1e-6), SKM_E_OVERFLOW,
                    "skm_cosine_dense_i8: a dot product may reach 2^31 (row norms %.3g x %.3g); use skm_cosine_csr, which has a wide path",
                    1.0 / h[0], 1.0 / h[1]);
    }
    dim3 grid((unsigned)skm_ceil_div(m, BN), (unsigned)skm_ceil_div(n, BM));
    // SKM_DENSE_VARIANT (all variants exact; A/B timing and the tests of the kernels the default route does not pick at a
    // shape): 1 the register-staged 128 x 128 kernel, 2 the 128 x 128 LDS-DMA kernel, 6 / 7 v4 (256 x 256, staggered wave
    // groups, both operands through LDS from tiled copies) with / without the symmetric form, 10 / 11 v5 (A through LDS, B
    // straight into registers; the default where K % 256 == 0) with / without it.  Round 5 removed 3 (round 1's lock-step
    // 256 x 256 kernel) and 4 / 5 (v4 staging from row-major operands).
    const int forced0 = skm_opts().dense_variant;
    const bool v2 = kdim % BK2 == 0 && forced0 != 1;
    const bool sym = d_x == d_y && n == m && d_xrnorm == d_yrnorm && forced0 != 7 && forced0 != 11;
    // A symmetric launch of few 256 x 256 tiles leaves most of the chip idle: up to SKM_V2_MAX_TILES of them (N <= ~4 000) the
    // 128 x 128 kernel takes it in its symmetric form - four times the workgroups, two per CU, no re-tiling passes in front.
    // tools/bench_dense_small.py, K = 6 656, ms (256 x 256 v5 with split-K / 128 x 128 symmetric): N = 1 100 0.068 / 0.049,
    // 1 500 0.069 / 0.050, 2 000 0.074 / 0.053, 2 600 0.104 / 0.054, 3 383 (the reference's CI proteome) 0.109 / 0.072,
    // 4 200 0.128 / 0.126, 5 000 0.124 / 0.154, 8 000 0.300 / 0.334; N = 3 383 with K = 1 024: 0.052 / 0.032, K = 13 312: 0.180 / 0.130.
#ifndef SKM_V2_MAX_TILES
#define SKM_V2_MAX_TILES 140
#endif
    const int64_t tiles256 = skm_ceil_div(n, BM3) * (skm_ceil_div(n, BM3) + 1) / 2;
    const bool small_launch = forced0 == 0 && v2 && sym && tiles256 <= SKM_V2_MAX_TILES;
    const bool v4 = kdim % BK4 == 0 && kdim >= 4 * BK4 && n >= 1024 && m >= 1024 && !small_launch &&
                    (forced0 == 0 || forced0 == 6 || forced0 == 7 || forced0 == 10 || forced0 == 11);
    SKM_PROF(ctx, "k_cosine_dense_i8");
    if (v4) {
        const int64_t nsy4 = skm_ceil_div(skm_ceil_div(n, BM3), 4), nsx4 = skm_ceil_div(skm_ceil_div(m, BN3), SKM_ST_W);
        int64_t supertiles = nsy4 * nsx4;
        if (sym) {  // rows of supertiles shrink towards the bottom: row sy keeps columns sx >= sy / 2 (see the kernel)
            supertiles = 0;
            for (int64_t sy = 0; sy < nsy4; ++sy)
                supertiles += nsx4 - sy / (SKM_ST_W / 4) > 0 ? nsx4 - sy / (SKM_ST_W / 4) : 0;
        }
        SKM_REQUIRE(supertiles * (4 * SKM_ST_W) < ((int64_t)1 << 31), SKM_E_OVERFLOW, "skm_cosine_dense_i8: too many tiles");
        dim3 grid4((unsigned)(supertiles * (4 * SKM_ST_W)));
        // default where K is a multiple of 256: v5 (SKM_DENSE_VARIANT=10; 11 the same without symmetry)
        if ((forced0 == 0 || forced0 == 10 || forced0 == 11) && kdim % (4 * BK4) == 0) {
            const int64_t nrbx = skm_ceil_div(n, BM3) * (BM3 / 16), ncb = skm_ceil_div(m, BN3) * (BN3 / 32);
            void *p;
            SKM_TRY(skm_ws(ctx, WS_K, (size_t)nrbx * 16 * (size_t)kdim, &p));
            int8_t *xt = (int8_t *)p;
            SKM_TRY(skm_ws(ctx, WS_L, (size_t)ncb * 32 * (size_t)kdim, &p));
            int8_t *ybt = (int8_t *)p;
            k_retile_i8<<<skm_grid_cap(ctx, skm_ceil_div(nrbx * 16 * kdim / 16, 256), 8), 256, 0, ctx->stream>>>(n, kdim, d_x, nrbx, xt);
            k_retile_b_i8<<<skm_grid_cap(ctx, skm_ceil_div(ncb * 32 * kdim / 16, 256), 8), 256, 0, ctx->stream>>>(m, kdim, d_y, ncb, ybt);
            SKM_TRY(skm_check_launch("k_retile_i8"));
            // Far fewer real tiles than compute units (the reference's own job size: one FASTA file of up to a few thousand
            // records, snekmer/rules/kmerize.smk:57-65): the K loop of a tile is shared by 2-4 workgroups while tiles x
            // splits fill at most HALF of the chip.  Measured (tools/bench_dense_small.py, K = 6656): N = 1500 (21 tiles)
            // 0.099 -> 0.069 ms with 4 splits; N = 3383 (105 tiles: the CI proteome) 0.109 -> 0.119 with 2, N = 8000 0.307 ->
            // 0.329: from ~100 tiles on the launch is bound by what all tiles share (75 % L2 hits, waves waiting half of
            // their cycles on operands from beyond L2), and more workgroups per tile only add the slab traffic.
            const int64_t nty5 = skm_ceil_div(n, BM3), ntx5 = skm_ceil_div(m, BN3);
            const int64_t real_tiles = sym ? nty5 * (nty5 + 1) / 2 : nty5 * ntx5;
            const int64_t nst5 = kdim / BK4;
            int nsplit = 1;
            int64_t sps = nst5;
            for (int cand = 4; cand >= 2 && nsplit == 1; --cand) {
                const int64_t per = skm_ceil_div(skm_ceil_div(nst5, cand), 4) * 4;  // stages per split: a multiple of 4
                if (real_tiles * cand <= (int64_t)ctx->usable_cus / 2 && per >= 8 && skm_ceil_div(nst5, per) == cand) {
                    nsplit = cand;
                    sps = per;
                }
            }
#ifdef SKM_DIAG
            if (const int cand = skm_opts().dense_split) {  // A/B timing: force the split count (1 = off)
                const int64_t per = skm_ceil_div(skm_ceil_div(nst5, cand > 0 ? cand : 1), 4) * 4;
                if (cand >= 1 && cand <= 8 && per >= 4 && skm_ceil_div(nst5, per) == cand) {
                    nsplit = cand;
                    sps = per;
                }
            }
#endif
            if (nsplit > 1) {
                const int64_t slots = supertiles * (4 * SKM_ST_W);
                SKM_TRY(skm_ws(ctx, WS_J, sizeof(int32_t) * (size_t)slots * (size_t)(nsplit - 1) * BM3 * BN3, &p));
                int32_t *slabs = (int32_t *)p;
                SKM_TRY(skm_ws(ctx, WS_ZERO, 256 + sizeof(uint32_t) * 2 * (size_t)slots, &p));  // (ticket, done) per slot: left at zero by every launch
                uint32_t *tickets = (uint32_t *)p + 64;
                dim3 grid5((unsigned)(slots * nsplit));
#define SKM_V5S(MODE, SYM) \
    k_cosine_dense_i8_v5<MODE, SYM, 3, true><<<grid5, 512, 0, ctx->stream>>>(n, m, kdim, xt, ybt, d_xrnorm, d_yrnorm, d_out, ld, nsplit, sps, slabs, tickets)
                if (mode == 0) {
                    if (sym)
                        SKM_V5S(0, true);
                    else
                        SKM_V5S(0, false);
                } else {
                    if (sym)
                        SKM_V5S(1, true);
                    else
                        SKM_V5S(1, false);
                }
#undef SKM_V5S
                return skm_check_launch("k_cosine_dense_i8");
            }
#ifndef SKM_V5_DA
#define SKM_V5_DA 3
#endif
#define SKM_V5(MODE, SYM) k_cosine_dense_i8_v5<MODE, SYM, SKM_V5_DA><<<grid4, 512, 0, ctx->stream>>>(n, m, kdim, xt, ybt, d_xrnorm, d_yrnorm, d_out, ld)
            if (mode == 0) {
                if (sym)
                    SKM_V5(0, true);
                else
                    SKM_V5(0, false);
            } else {
                if (sym)
                    SKM_V5(1, true);
                else
                    SKM_V5(1, false);
            }
#undef SKM_V5
            return skm_check_launch("k_cosine_dense_i8");
        }
        // otherwise (K a multiple of 64 only, or forced): v4 staging from tiled copies of the operands
        {
            const int64_t nrbx = skm_ceil_div(n, BM3) * (BM3 / 16), nrby = skm_ceil_div(m, BN3) * (BN3 / 16);
            void *p;
            SKM_TRY(skm_ws(ctx, WS_K, (size_t)nrbx * 16 * (size_t)kdim, &p));
            int8_t *xt = (int8_t *)p, *yt = xt;
            k_retile_i8<<<skm_grid_cap(ctx, skm_ceil_div(nrbx * 16 * kdim / 16, 256), 8), 256, 0, ctx->stream>>>(n, kdim, d_x, nrbx, xt);
            if (!(d_x == d_y && n == m)) {
                SKM_TRY(skm_ws(ctx, WS_L, (size_t)nrby * 16 * (size_t)kdim, &p));
                yt = (int8_t *)p;
                k_retile_i8<<<skm_grid_cap(ctx, skm_ceil_div(nrby * 16 * kdim / 16, 256), 8), 256, 0, ctx->stream>>>(m, kdim, d_y, nrby, yt);
            }
            SKM_TRY(skm_check_launch("k_retile_i8"));
#define SKM_V4T(MODE, SYM) \
    k_cosine_dense_i8_v4<MODE, SYM><<<grid4, 512, 0, ctx->stream>>>(n, m, kdim, xt, yt, d_xrnorm, d_yrnorm, d_out, ld)
            if (mode == 0) {
                if (sym)
                    SKM_V4T(0, true);
                else
                    SKM_V4T(0, false);
            } else {
                if (sym)
                    SKM_V4T(1, true);
                else
                    SKM_V4T(1, false);
            }
#undef SKM_V4T
            return skm_check_launch("k_cosine_dense_i8");
        }
    } else if (v2) {
        if (sym && forced0 != 2) {  // (SKM_DENSE_VARIANT=2: the rectangular launch)
            const int64_t T = skm_ceil_div(n, BM);
            const dim3 gsym((unsigned)(T * (T + 1) / 2));
            if (mode == 0)
                k_cosine_dense_i8_v2<0, true><<<gsym, 256, 0, ctx->stream>>>(n, m, kdim, d_x, d_y, d_xrnorm, d_yrnorm, d_out, ld);
            else
                k_cosine_dense_i8_v2<1, true><<<gsym, 256, 0, ctx->stream>>>(n, m, kdim, d_x, d_y, d_xrnorm, d_yrnorm, d_out, ld);
        } else if (mode == 0)
            k_cosine_dense_i8_v2<0><<<grid, 256, 0, ctx->stream>>>(n, m, kdim, d_x, d_y, d_xrnorm, d_yrnorm, d_out, ld);
        else
            k_cosine_dense_i8_v2<1><<<grid, 256, 0, ctx->stream>>>(n, m, kdim, d_x, d_y, d_xrnorm, d_yrnorm, d_out, ld);
    } else {
        if (mode == 0)
            k_cosine_dense_i8<0><<<grid, 256, 0, ctx->stream>>>(n, m, kdim, d_x, d_y, d_xrnorm, d_yrnorm, d_out, ld);
        else
            k_cosine_dense_i8<1><<<grid, 256, 0, ctx->stream>>>(n, m, kdim, d_x, d_y, d_xrnorm, d_yrnorm, d_out, ld);
    }
    return skm_check_launch("k_cosine_dense_i8");
}

extern "C" int skm_dense_to_csr(skm_ctx *ctx, int64_t n, int64_t ncols, int dtype, const void *d_in, int64_t ld,
                                int64_t cap_entries, int64_t *d_rowptr, uint32_t *d_col, uint32_t *d_val, int64_t *h_nnz)
{
    SKM_REQUIRE(ctx && n >= 0 && ncols >= 0 && ld >= ncols && cap_entries >= 0 && d_rowptr && h_nnz, SKM_E_BADARG,
                "skm_dense_to_csr: bad argument");
    SKM_REQUIRE(dtype >= 0 && dtype <= 2, SKM_E_BADARG, "skm_dense_to_csr: dtype must be 0 (uint16), 1 (uint32) or 2 (int8)");
    SKM_REQUIRE(ncols < ((int64_t)1 << 32) - 1, SKM_E_OVERFLOW, "skm_dense_to_csr: ncols >= 2^32");
    SKM_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    *h_nnz = 0;
    if (n == 0 || ncols == 0) {
        SKM_HIP(hipMemsetAsync(d_rowptr, 0, sizeof(int64_t) * (size_t)(n + 1), st));
        return SKM_OK;
    }
    SKM_REQUIRE(d_in, SKM_E_BADARG, "skm_dense_to_csr: null matrix");
    const size_t cell = dtype == 0 ? 2 : (dtype == 1 ? 4 : 1);
    const int vec_ok = (((uintptr_t)d_in & 15) == 0 && ((size_t)ld * cell) % 16 == 0) ? 1 : 0;
    void *p;
    SKM_TRY(skm_ws(ctx, WS_H, sizeof(int64_t) * (size_t)(4 * n + 1), &p));
    int64_t *seg = (int64_t *)p;
    SKM_TRY(skm_ws(ctx, WS_I, sizeof(int64_t) * (size_t)(4 * n + 1), &p));
    int64_t *seg_pos = (int64_t *)p;
    SKM_HIP(hipMemsetAsync(seg + 4 * n, 0, sizeof(int64_t), st));
    const int grid = skm_grid_cap(ctx, n, 16);
#define SKM_D2C(WRITE, SEG)                                                                                              \
    do {                                                                                                                 \
        if (dtype == 0)                                                                                                  \
            k_dense_to_csr<uint16_t, WRITE><<<grid, 256, 0, st>>>(n, ncols, (const uint16_t *)d_in, ld, vec_ok, SEG, d_col, d_val); \
        else if (dtype == 1)                                                                                             \
            k_dense_to_csr<uint32_t, WRITE><<<grid, 256, 0, st>>>(n, ncols, (const uint32_t *)d_in, ld, vec_ok, SEG, d_col, d_val); \
        else                                                                                                             \
            k_dense_to_csr<uint8_t, WRITE><<<grid, 256, 0, st>>>(n, ncols, (const uint8_t *)d_in, ld, vec_ok, SEG, d_col, d_val);   \
    } while (0)
    {
        SKM_PROF(ctx, "k_dense_to_csr_count");
        SKM_D2C(false, seg);
    }
    SKM_TRY(skm_check_launch("k_dense_to_csr_count"));
    {
        size_t tmp = 0;
        SKM_HIP(rocprim::exclusive_scan(nullptr, tmp, seg, seg_pos, (int64_t)0, (size_t)(4 * n + 1), rocprim::plus<int64_t>(), st));
        void *t;
        SKM_TRY(skm_ws(ctx, WS_ROCPRIM, tmp, &t));
        SKM_HIP(rocprim::exclusive_scan(t, tmp, seg, seg_pos, (int64_t)0, (size_t)(4 * n + 1), rocprim::plus<int64_t>(), st));
    }
    k_rowptr_from_segments<<<(unsigned)skm_ceil_div(n + 1, 256), 256, 0, st>>>(n, seg_pos, d_rowptr);
    SKM_TRY(skm_check_launch("k_rowptr_from_segments"));
    int64_t *h = (int64_t *)ctx->h_pinned;
    SKM_HIP(hipMemcpyAsync(h, seg_pos + 4 * n, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    SKM_HIP(hipStreamSynchronize(st));
    *h_nnz = *h;
    SKM_REQUIRE(*h <= cap_entries, SKM_E_OVERFLOW, "skm_dense_to_csr: %lld non-zeros exceed cap_entries %lld", (long long)*h,
                (long long)cap_entries);
    if (*h == 0)
        return SKM_OK;
    SKM_REQUIRE(d_col && d_val, SKM_E_BADARG, "skm_dense_to_csr: null output arrays");
    {
        SKM_PROF(ctx, "k_dense_to_csr_write");
        SKM_D2C(true, seg_pos);
    }
#undef SKM_D2C
    return skm_check_launch("k_dense_to_csr_write");
}

extern "C" int skm_row_norms_i8(skm_ctx *ctx, int64_t n, int64_t kdim, const int8_t *d_in, float *d_rnorm, uint64_t *d_normsq)
{
    SKM_REQUIRE(ctx && n >= 0 && kdim >= 0 && kdim % 64 == 0, SKM_E_BADARG, "skm_row_norms_i8: bad argument (kdim must be a multiple of 64)");
    if (n == 0)
        return SKM_OK;
    SKM_REQUIRE(d_in && ((uintptr_t)d_in & 15) == 0, SKM_E_BADARG, "skm_row_norms_i8: matrix must be 16-byte aligned");
    SKM_HIP(hipSetDevice(ctx->device));
    SKM_PROF(ctx, "k_row_norms_i8");
    k_row_norms_i8<<<skm_grid_cap(ctx, skm_ceil_div(n, 4), 16), 256, 0, ctx->stream>>>(n, kdim, d_in, d_rnorm, d_normsq);
    return skm_check_launch("k_row_norms_i8");
}

// ------------------------------------------------------------------------------- dense route of engine.Pipeline
// Small full bases (|S|^k <= 2^17: the reference's own CI configuration, solvacc k=8 = 6561 columns) are dense enough
// for the cosine to be a true GEMM.  The count stage's CSR (column id == k-mer code) becomes the int8 operand of the MFMA
// kernel; a count above 127 saturates there, its row is listed as IRREGULAR, and k_cosine_fixup_rows recomputes that
// row's cells (and, by symmetry, its column) from the CSR itself with float64 accumulators - so the route is exact for
// any input and nothing is decided on the host.  Regular rows need no range test: 127^2 * kdim < 2^31 for kdim <= 133 143.
namespace {

__global__ __launch_bounds__(256) void k_csr_to_dense_i8_flag(int64_t n, const int64_t *__restrict__ rowptr,
                                                              const uint32_t *__restrict__ codes,
                                                              const uint32_t *__restrict__ counts, int64_t kdim,
                                                              int8_t *__restrict__ out, uint32_t *__restrict__ irr_list,
                                                              uint32_t *__restrict__ irr_count)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += nwaves) {
        const int64_t b = rowptr[i], e = rowptr[i + 1];
        // the row is zeroed by the wave that fills it (kdim is a multiple of 64; 1 KiB per store instruction): no separate
        // fill of the whole matrix in front of the kernel.  The wave waits for its zeros before it stores a count over them.
        int4 *row16 = reinterpret_cast<int4 *>(out + i * kdim);
        for (int64_t z = lane; z < kdim / 16; z += 64)
            row16[z] = make_int4(0, 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bool big = false;
        for (int64_t t = b + lane; t < e; t += 64) {
            const uint32_t v = counts[t];
            big |= v > 127u;
            out[i * kdim + codes[t]] = (int8_t)(v > 127u ? 127u : v);
        }
        if (__any(big) && lane == 0)
            irr_list[atomicAdd(irr_count, 1u)] = (uint32_t)i;
    }
}

// Exact cells of the irregular rows.  Work item = (irregular row i, 256 consecutive rows j); row i's (code, count) list
// goes through LDS in sorted tiles of FX_TILE entries; thread j walks its own row's entries once (they are sorted too)
// and looks each one up in the tile with a binary search.  out[i][j] for i in [row0, row1), out[j][i] for j in [row0, row1).
constexpr int FX_TILE = 2048;

template <int MODE>
__global__ __launch_bounds__(256) void k_cosine_fixup_rows(int64_t n, const int64_t *__restrict__ rowptr,
                                                           const uint32_t *__restrict__ codes,
                                                           const uint32_t *__restrict__ counts,
                                                           const float *__restrict__ rnorm, int64_t row0, int64_t row1,
                                                           const uint32_t *__restrict__ irr_list,
                                                           const uint32_t *__restrict__ irr_count, float *__restrict__ out,
                                                           int64_t ld)
{
    __shared__ uint32_t s_code[FX_TILE], s_val[FX_TILE];
    const int tid = threadIdx.x;
    const int64_t nchunks = (n + 255) / 256;
    const int64_t items = (int64_t)*irr_count * nchunks;
    for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
        const int64_t i = irr_list[item / nchunks];
        const int64_t j = (item % nchunks) * 256 + tid;
        const int64_t ib = rowptr[i], ie = rowptr[i + 1];
        int64_t pj = 0, ej = 0;
        uint32_t cj = 0xFFFFFFFFu;
        if (j < n) {
            pj = rowptr[j];
            ej = rowptr[j + 1];
            if (pj < ej)
                cj = codes[pj];
        }
        double acc = 0.0;
        for (int64_t t0 = ib; t0 < ie; t0 += FX_TILE) {
            const int cnt = (int)min((int64_t)FX_TILE, ie - t0);
            __syncthreads();
            for (int z = tid; z < cnt; z += 256) {
                s_code[z] = codes[t0 + z];
                s_val[z] = counts[t0 + z];
            }
            __syncthreads();
            const uint32_t tile_max = s_code[cnt - 1];
            while (pj < ej && cj <= tile_max) {
                int lo = 0, hi = cnt - 1;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (s_code[mid] < cj)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                if (s_code[lo] == cj)
                    acc += (double)s_val[lo] * (double)counts[pj];
                ++pj;
                cj = pj < ej ? codes[pj] : 0xFFFFFFFFu;
            }
        }
        if (j < n) {
            // a cell whose row AND column are irregular is written by two work items ((i, j) and (j, i)); the integer sum is
            // the same in both (exact in float64 below 2^53) and the scale is formed symmetrically, so both store one value
            const double scale = (double)rnorm[i < j ? i : j] * (double)rnorm[i < j ? j : i];
            float o = (float)(acc * scale);
            if (MODE == 1) {
                o = fminf(fmaxf(1.0f - o, 0.0f), 2.0f);
                if (i == j)
                    o = 0.0f;
            }
            if (i >= row0 && i < row1)
                out[(i - row0) * ld + j] = o;
            if (j >= row0 && j < row1)
                out[(j - row0) * ld + i] = o;
        }
    }
}

}  // namespace

extern "C" int skm_csr_to_dense_i8(skm_ctx *ctx, int64_t n, const int64_t *d_rowptr, const uint32_t *d_codes,
                                   const uint32_t *d_counts, int64_t kdim, int8_t *d_out, uint32_t *d_irr_list,
                                   uint32_t *d_irr_count)
{
    SKM_REQUIRE(ctx && n >= 0 && kdim >= 0 && kdim % 64 == 0, SKM_E_BADARG, "skm_csr_to_dense_i8: bad argument (kdim must be a multiple of 64)");
    SKM_REQUIRE(d_irr_count, SKM_E_BADARG, "skm_csr_to_dense_i8: null counter");
    SKM_HIP(hipSetDevice(ctx->device));
    SKM_HIP(hipMemsetAsync(d_irr_count, 0, sizeof(uint32_t), ctx->stream));
    if (n == 0 || kdim == 0)
        return SKM_OK;
    SKM_REQUIRE(d_rowptr && d_codes && d_counts && d_out && d_irr_list, SKM_E_BADARG, "skm_csr_to_dense_i8: null array");
    SKM_PROF(ctx, "k_csr_to_dense_i8");  // (zeroes every row it fills: round 5; a fill of the whole matrix in front before)
    k_csr_to_dense_i8_flag<<<skm_grid_cap(ctx, skm_ceil_div(n, 4), 16), 256, 0, ctx->stream>>>(n, d_rowptr, d_codes, d_counts, kdim,
                                                                                             d_out, d_irr_list, d_irr_count);
    return skm_check_launch("k_csr_to_dense_i8");
}

extern "C" int skm_cosine_fixup_rows(skm_ctx *ctx, int64_t n, const int64_t *d_rowptr, const uint32_t *d_codes,
                                     const uint32_t *d_counts, const float *d_rnorm, int64_t row0, int64_t row1,
                                     const uint32_t *d_irr_list, const uint32_t *d_irr_count, int mode, float *d_out, int64_t ld)
{
    SKM_REQUIRE(ctx && n >= 0 && row0 >= 0 && row0 <= row1 && row1 <= n && ld >= n && (mode == 0 || mode == 1), SKM_E_BADARG,
                "skm_cosine_fixup_rows: bad argument");
    if (n == 0 || row0 == row1)
        return SKM_OK;
    SKM_REQUIRE(d_rowptr && d_codes && d_counts && d_rnorm && d_irr_list && d_irr_count && d_out, SKM_E_BADARG,
                "skm_cosine_fixup_rows: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    SKM_PROF(ctx, "k_cosine_fixup_rows");
    const int grid = skm_grid_cap(ctx, n, 4);  // strides over (irregular rows) x (chunks of 256 rows); empty list: exits at once
    if (mode == 0)
        k_cosine_fixup_rows<0><<<grid, 256, 0, ctx->stream>>>(n, d_rowptr, d_codes, d_counts, d_rnorm, row0, row1, d_irr_list,
                                                               d_irr_count, d_out, ld);
    else
        k_cosine_fixup_rows<1><<<grid, 256, 0, ctx->stream>>>(n, d_rowptr, d_codes, d_counts, d_rnorm, row0, row1, d_irr_list,
                                                               d_irr_count, d_out, ld);
    return skm_check_launch("k_cosine_fixup_rows");
}
