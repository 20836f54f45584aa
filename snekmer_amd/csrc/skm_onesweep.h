// Stable LSD radix sort of (key, entry index) pairs for the basis stage: one-sweep passes with decoupled look-back.
//
// Why not the vendor building block here: rocPRIM's Onesweep (skm_sort.h) is a good sort for 3e7 pairs, but a call is
// ~17 stream operations (a histogram and a scan kernel, and per digit a memset of the look-back state plus the pass), so
// a FASTA file of a few hundred to a few thousand records - the reference's real job size, one file per Snakemake job,
// snekmer/rules/kmerize.smk:57-65 - pays 0.12-0.16 ms of launch latency for a sort whose data fits L2.  It also sorts
// the capacity (one key per residue, sentinel-filled tail) because its size is a host argument.  This sort
//   * reads the entry count on the device (nothing behind it is touched: no sentinel fill, no tail),
//   * is 2 + P launches for P digits: one memset of its small state, one histogram kernel for all digits, P passes,
//   * generates the payload (the entry's index) in the first pass instead of reading a counting array.
// A pass: workgroups take tiles in arrival order (atomic ticket, so every tile a look-back waits for is already
// running); the tile's keys sit in registers in wave-blocked order (element = wave * 64 * IPT + r * 64 + lane), ranks
// within a wave come from 8 ballots per item (the lanes holding the same digit) plus a per-wave digit counter in LDS;
// per-digit tile counts are published as (status, value) words, the exclusive prefix over earlier tiles is collected by
// look-back; keys and payloads are reordered through LDS so that every digit's run leaves as consecutive addresses.
#pragma once
#include <cstdint>

#include "skm_common.h"

namespace skm_onesweep {

constexpr int RADIX_BITS = 8, RADIX = 1 << RADIX_BITS;
constexpr uint32_t ST_SHIFT = 30, ST_MASK = (1u << ST_SHIFT) - 1u;  // tile word: status << 30 | value (n < 2^30)
constexpr uint32_t ST_AGG = 1u, ST_PREFIX = 2u;
constexpr int MAX_PASSES = 8;

struct state_header {
    uint32_t hist[MAX_PASSES][RADIX];  // digit counts of the whole input, per pass
    uint32_t ticket[MAX_PASSES];       // next tile of every pass
    uint32_t pad[RADIX - MAX_PASSES];
};

// counts of every digit of every pass; one read of the keys
template <typename K, int TB>
__global__ __launch_bounds__(TB) void k_histogram(const int64_t *__restrict__ d_n, const K *__restrict__ keys, int passes,
                                                  int key_bits, state_header *st)
{
    __shared__ uint32_t s_h[MAX_PASSES][RADIX];
    for (int z = threadIdx.x; z < MAX_PASSES * RADIX; z += TB)
        (&s_h[0][0])[z] = 0u;
    __syncthreads();
    const int64_t n = *d_n;
    const int64_t stride = (int64_t)gridDim.x * TB;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += stride) {
        const K key = keys[i];
        for (int p = 0; p < passes; ++p) {
            const int shift = p * RADIX_BITS, bits = min(RADIX_BITS, key_bits - shift);
            atomicAdd(&s_h[p][(uint32_t)(key >> shift) & ((1u << bits) - 1u)], 1u);
        }
    }
    __syncthreads();
    for (int z = threadIdx.x; z < passes * RADIX; z += TB) {
        const uint32_t c = (&s_h[0][0])[z];
        if (c)
            atomicAdd(&(&st->hist[0][0])[z], c);
    }
}

template <typename K, int TB, int IPT, bool FIRST>
__global__ __launch_bounds__(TB) void k_pass(const int64_t *__restrict__ d_n, const K *__restrict__ kin, K *__restrict__ kout,
                                             const uint32_t *__restrict__ vin, uint32_t *__restrict__ vout, state_header *st,
                                             uint32_t *__restrict__ tile_state, int pass, int key_bits)
{
    constexpr int NW = TB / 64, TILE = TB * IPT, WCHUNK = 64 * IPT;
    __shared__ uint32_t s_whist[NW][RADIX];   // per-wave digit counters, then exclusive prefixes over the waves
    __shared__ uint32_t s_tile_excl[RADIX];   // first slot of every digit inside the sorted tile
    __shared__ uint32_t s_gbase[RADIX];       // global position of the tile's first element of every digit
    __shared__ K s_key[TILE];
    __shared__ uint32_t s_val[TILE];
    __shared__ uint32_t s_tile;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int shift = pass * RADIX_BITS;
    const uint32_t mask = (1u << min(RADIX_BITS, key_bits - shift)) - 1u;
    const int64_t n = *d_n;
    if (tid == 0)
        s_tile = atomicAdd(&st->ticket[pass], 1u);
    for (int z = tid; z < NW * RADIX; z += TB)
        (&s_whist[0][0])[z] = 0u;
    __syncthreads();
    const int64_t tile = s_tile;
    const int64_t tile_base = tile * TILE;
    if (tile_base >= n)  // uniform; no earlier tile ever waits for this one
        return;
    const int count = (int)min((int64_t)TILE, n - tile_base);

    K key[IPT];
    uint32_t val[IPT], dig[IPT], rank[IPT];
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const int e = wid * WCHUNK + r * 64 + lane;
        const bool valid = e < count;
        key[r] = valid ? kin[tile_base + e] : K(0);
        val[r] = FIRST ? (uint32_t)(tile_base + e) : (valid ? vin[tile_base + e] : 0u);
        dig[r] = valid ? ((uint32_t)(key[r] >> shift) & mask) : 0xFFFFFFFFu;
    }
    // ranks within the wave, in element order: items r = 0.. in turn, lanes of equal digit found with ballots
    const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const bool valid = dig[r] != 0xFFFFFFFFu;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
            const bool bit = (dig[r] >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        uint32_t base = 0;
        if (valid) {
            const int leader = __ffsll((long long)peers) - 1;
            if (lane == leader)
                base = atomicAdd(&s_whist[wid][dig[r]], (uint32_t)__popcll(peers));
            base = __shfl(base, leader);
            rank[r] = base + (uint32_t)__popcll(peers & lt);
        } else {
            rank[r] = 0;
        }
    }
    __syncthreads();
    // per digit: exclusive prefix over the waves, tile count, publication, look-back
    uint32_t tile_cnt = 0;  // of digit d = tid (threads 0..RADIX-1)
    if (tid < RADIX) {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const uint32_t c = s_whist[w][tid];
            s_whist[w][tid] = run;
            run += c;
        }
        tile_cnt = run;
        uint32_t *mine = tile_state + (size_t)tile * RADIX + tid;
        if (tile == 0) {
            __hip_atomic_store(mine, (ST_PREFIX << ST_SHIFT) | tile_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(mine, (ST_AGG << ST_SHIFT) | tile_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // exclusive scan of the tile counts over the digits -> s_tile_excl (RADIX = 256 = 4 waves of the first 256 threads)
    {
        __shared__ uint32_t s_wsum[RADIX / 64];
        uint32_t incl = tile_cnt;
        if (tid < RADIX) {
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(incl, o);
                if (lane >= o)
                    incl += up;
            }
            if (lane == 63)
                s_wsum[wid] = incl;
        }
        __syncthreads();
        if (tid < RADIX) {
            uint32_t before = 0;
#pragma unroll
            for (int w = 0; w < RADIX / 64; ++w)
                before += w < wid ? s_wsum[w] : 0u;
            s_tile_excl[tid] = before + incl - tile_cnt;
        }
    }
    if (tid < RADIX) {
        // global digit base: exclusive scan of the whole-input histogram, recomputed per tile from 256 words (L2)
        uint32_t g = 0;
        {
            const uint32_t h = st->hist[pass][tid];
            uint32_t incl = h;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(incl, o);
                if (lane >= o)
                    incl += up;
            }
            // totals of the earlier waves of the 256 digits: read the histogram words directly (at most 3 x 64 adds
            // would need another LDS round; 3 partial sums are cheaper through a second shuffle tree per wave)
            uint32_t before = 0;
            for (int w = 0; w < wid; ++w) {
                uint32_t part = st->hist[pass][w * 64 + lane];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1)
                    part += __shfl_xor(part, o);
                before += part;
            }
            g = before + incl - h;
        }
        uint32_t excl = 0;
        if (tile > 0) {
            int64_t t = tile - 1;
            while (true) {
                const uint32_t w = __hip_atomic_load(tile_state + (size_t)t * RADIX + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t status = w >> ST_SHIFT;
                if (status == 0u) {
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                excl += w & ST_MASK;
                if (status == ST_PREFIX)
                    break;
                --t;
            }
            __hip_atomic_store(tile_state + (size_t)tile * RADIX + tid, (ST_PREFIX << ST_SHIFT) | (excl + tile_cnt), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        s_gbase[tid] = g + excl;
    }
    __syncthreads();
    // reorder through LDS: position inside the tile's sorted image
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        if (dig[r] != 0xFFFFFFFFu) {
            const uint32_t pos = s_tile_excl[dig[r]] + s_whist[wid][dig[r]] + rank[r];
            s_key[pos] = key[r];
            s_val[pos] = val[r];
        }
    }
    __syncthreads();
    for (int j = tid; j < count; j += TB) {
        const K k2 = s_key[j];
        const uint32_t d = (uint32_t)(k2 >> shift) & mask;
        const uint32_t dst = s_gbase[d] + ((uint32_t)j - s_tile_excl[d]);
        kout[dst] = k2;
        vout[dst] = s_val[j];
    }
}

// bytes of device scratch the sort needs for `cap` pairs with tiles of `tile` keys
static inline size_t state_bytes(int64_t cap, int tile, int passes)
{
    const int64_t ntiles = (cap + tile - 1) / tile + 1;
    return sizeof(state_header) + sizeof(uint32_t) * (size_t)ntiles * RADIX * (size_t)passes;
}

// Stable sort of the first *d_n (<= cap < 2^30) keys of `kin` with payload = index; result in kout / vout.  ktmp / vtmp:
// scratch of cap elements each; d_state: state_bytes().  Nothing waits for the device.
template <typename K>
static int sort_pairs_dev(skm_ctx *ctx, const int64_t *d_n, int64_t cap, const K *kin, K *kout, uint32_t *vout, K *ktmp,
                          uint32_t *vtmp, void *d_state, int key_bits, const char *label)
{
    hipStream_t s = ctx->stream;
    const int passes = (key_bits + RADIX_BITS - 1) / RADIX_BITS;
    SKM_REQUIRE(passes >= 1 && passes <= MAX_PASSES && cap < ((int64_t)1 << 30), SKM_E_BADARG, "onesweep: bad size");
    // small inputs: 256-thread tiles of 2048 keys (more tiles in flight: the passes are latency-bound there);
    // large ones: 1024 x 8 (the shape rocPRIM's tuning also prefers on this chip)
    const bool small = cap <= ((int64_t)1 << 22);
    const int tile = small ? 256 * 8 : 1024 * 8;
    const int64_t ntiles = (cap + tile - 1) / tile + 1;
    state_header *st = (state_header *)d_state;
    uint32_t *tile_state = (uint32_t *)((uint8_t *)d_state + sizeof(state_header));
    SKM_PROF(ctx, label);
    SKM_HIP(hipMemsetAsync(d_state, 0, state_bytes(cap, tile, passes), s));
    {
        const int grid = skm_grid_cap(ctx, skm_ceil_div(cap, 256 * 16), 4);
        k_histogram<K, 256><<<grid, 256, 0, s>>>(d_n, kin, passes, key_bits, st);
    }
    const K *src_k = kin;
    const uint32_t *src_v = nullptr;
    for (int p = 0; p < passes; ++p) {
        // the last pass lands in (kout, vout); passes alternate between the two buffer pairs
        const bool to_out = ((passes - 1 - p) & 1) == 0;
        K *dst_k = to_out ? kout : ktmp;
        uint32_t *dst_v = to_out ? vout : vtmp;
        uint32_t *ts = tile_state + (size_t)p * (size_t)ntiles * RADIX;
#define SKM_OS_PASS(TB, FIRST)                                                                                        \
    k_pass<K, TB, 8, FIRST><<<(unsigned)(ntiles - 1), TB, 0, s>>>(d_n, src_k, dst_k, src_v, dst_v, st, ts, p, key_bits)
        if (small) {
            if (p == 0)
                SKM_OS_PASS(256, true);
            else
                SKM_OS_PASS(256, false);
        } else {
            if (p == 0)
                SKM_OS_PASS(1024, true);
            else
                SKM_OS_PASS(1024, false);
        }
#undef SKM_OS_PASS
        src_k = dst_k;
        src_v = dst_v;
    }
    return skm_check_launch(label);
}

}  // namespace skm_onesweep
