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

#ifndef SKM_OS_LB
#define SKM_OS_LB 8
#endif
#ifndef SKM_OS_LARGE_TB
#define SKM_OS_LARGE_TB 512  // (29 M pairs, 4 passes: 1024 x 8 0.99 ms, 512 x 8 0.89, 256 x 8 1.13; rocPRIM 0.85)
#endif
#ifndef SKM_OS_SMALL_LOG2
#define SKM_OS_SMALL_LOG2 19
#endif

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
    // tiles in arrival order (atomic ticket), so that every tile a look-back waits for is already running: a workgroup
    // only ever waits for workgroups that hold a ticket, whatever else shares the chip.  (Tiles by blockIdx for a grid that
    // is resident as a whole measured 2 us faster per pass, and is a deadlock with other streams around: workgroups go
    // round-robin to the 8 XCDs, each XCD starts its share in order, and the passes of TWO contexts' sorts - or a sort
    // beside another stream's 128 KiB writer workgroups - can fill XCD a with tiles of sort A that spin on a tile waiting
    // for room on XCD b, which is full of tiles of sort B spinning on one that waits for room on XCD a.  A long fuzz run
    // of round 5 stopped once with that path in the library; the ticket is taken always.)
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
    // ranks within the wave, in element order: items r = 0.. in turn, lanes of equal digit found with ballots.  Three
    // loops instead of one so that the LDS atomics of the IPT items leave back to back (one wave's atomics execute in issue
    // order, which is what keeps the sort stable) and are waited for ONCE: in a single loop every item waited for its own
    // returning atomic before the next item's ballots started - IPT LDS round trips in series, most of a small pass's time.
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned long long peers[IPT];
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const bool valid = dig[r] != 0xFFFFFFFFu;
        unsigned long long pm = __ballot(valid);
#pragma unroll
        for (int b = 0; b < RADIX_BITS; ++b) {
            const bool bit = (dig[r] >> b) & 1u;
            const unsigned long long bal = __ballot(bit);
            pm &= bit ? bal : ~bal;
        }
        peers[r] = valid ? pm : 0ull;
    }
    // every lane issues the atomic, non-leaders (and invalid lanes, on slot 0) with an addend of 0: no branch around it, so
    // the IPT instructions are issued in a row; the leader's return value is the counter before its own add whatever the
    // order in which the lanes of one instruction are applied, because nobody else changes the counter in that instruction
    uint32_t base[IPT];
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const bool lead = peers[r] && lane == __ffsll((long long)peers[r]) - 1;
        base[r] = atomicAdd(&s_whist[wid][peers[r] ? dig[r] : 0u], lead ? (uint32_t)__popcll(peers[r]) : 0u);
    }
    __builtin_amdgcn_sched_barrier(0);  // (the uses below must not be scheduled between the atomics)
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const int leader = peers[r] ? __ffsll((long long)peers[r]) - 1 : lane;
        const uint32_t b0 = __shfl(base[r], leader);
        rank[r] = peers[r] ? b0 + (uint32_t)__popcll(peers[r] & lt) : 0u;
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
            // look-back, LB tiles per round trip: the words of the LB nearest predecessors are loaded together and consumed
            // nearest first up to the first inclusive prefix or the first unpublished word.  One word per round trip made
            // a chain of dependent agent-scope loads (each a trip past the XCD's L2) as long as the distance to the nearest
            // finished tile - with every tile of a small input resident at once, most of a pass's time.
            constexpr int LB = SKM_OS_LB;
            int64_t t = tile - 1;
            bool done = false;
            while (!done) {
                uint32_t w[LB];
#pragma unroll
                for (int q = 0; q < LB; ++q)
                    w[q] = t - q >= 0 ? __hip_atomic_load(tile_state + (size_t)(t - q) * RADIX + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                      : (ST_PREFIX << ST_SHIFT);
                int adv = 0;
#pragma unroll
                for (int q = 0; q < LB; ++q) {
                    const uint32_t status = w[q] >> ST_SHIFT;
                    if (done || adv != q || status == 0u)
                        continue;
                    excl += w[q] & ST_MASK;
                    ++adv;
                    done = status == ST_PREFIX;
                }
                t -= adv;
                if (!done && adv == 0)
                    __builtin_amdgcn_s_sleep(1);
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

// Tile shape (threads, keys per thread) for `cap` keys of `key_bytes` bytes.  Up to 0.5 M keys: 256 x 8 (many tiles in flight:
// the passes are latency-bound there; 1.0 M keys: 0.109 ms with it, 0.094 with 1024 x 8).  Above, the passes are bound by how many ROUNDS of workgroups the grid
// takes (a pass of 354 tiles of 1024 x 8 on 256 CUs takes as long as one of 512), so the tile grows with the input to
// keep the whole grid resident at once - 1024 x 8 / x 12 / x 16, one workgroup per CU (LDS: 8 or 12 bytes per key) - up to
// 4 M keys; above that 512 x 8 (29 M pairs, 4 passes: 1024 x 8 0.99 ms, 512 x 8 0.89, 256 x 8 1.13; rocPRIM 0.85).
static inline void tile_shape(int64_t cap, int key_bytes, int *tb, int *ipt)
{
    *tb = SKM_OS_LARGE_TB;
    *ipt = 8;
    if (cap <= ((int64_t)1 << SKM_OS_SMALL_LOG2)) {
        *tb = 256;
    } else if (cap <= ((int64_t)1 << 21)) {
        *tb = 1024;
    } else if (key_bytes <= 4 && cap <= ((int64_t)3 << 20)) {
        *tb = 1024;
        *ipt = 12;
    } else if (key_bytes <= 4 && cap <= ((int64_t)1 << 22)) {
        *tb = 1024;
        *ipt = 16;
    }
}

// what sort_pairs_dev clears (or wants cleared) at d_state for `cap` keys
static inline size_t sort_state_bytes(int64_t cap, int key_bits, int key_bytes)
{
    const int passes = (key_bits + RADIX_BITS - 1) / RADIX_BITS;
    int tb, ipt;
    tile_shape(cap, key_bytes, &tb, &ipt);
    return state_bytes(cap, tb * ipt, passes);
}

// Stable sort of the first *d_n (<= cap < 2^30) keys of `kin` with payload = index; result in kout / vout.  ktmp / vtmp:
// scratch of cap elements each; d_state: state_bytes() of the smallest tile (2048).  Nothing waits for the device.
template <typename K>
static int sort_pairs_dev(skm_ctx *ctx, const int64_t *d_n, int64_t cap, const K *kin, K *kout, uint32_t *vout, K *ktmp,
                          uint32_t *vtmp, void *d_state, int key_bits, const char *label, bool state_is_zero = false,
                          bool hist_ready = false)
{
    hipStream_t s = ctx->stream;
    const int passes = (key_bits + RADIX_BITS - 1) / RADIX_BITS;
    SKM_REQUIRE(passes >= 1 && passes <= MAX_PASSES && cap < ((int64_t)1 << 30), SKM_E_BADARG, "onesweep: bad size");
    int tb, ipt;
    tile_shape(cap, (int)sizeof(K), &tb, &ipt);
    const int tile = tb * ipt;
    const int64_t ntiles = (cap + tile - 1) / tile + 1;
    state_header *st = (state_header *)d_state;
    uint32_t *tile_state = (uint32_t *)((uint8_t *)d_state + sizeof(state_header));
    SKM_PROF(ctx, label);
    if (!state_is_zero)  // (a fused call has the previous stage's last kernel clear it: sort_state_bytes() words from d_state)
        SKM_HIP(hipMemsetAsync(d_state, 0, state_bytes(cap, tile, passes), s));
    if (!hist_ready) {  // (a fused call: the previous stage's last kernel counted the digits while it wrote the keys)
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
#define SKM_OS_PASS(TB, IPT)                                                                                                   \
    do {                                                                                                                       \
        if (p == 0)                                                                                                            \
            k_pass<K, TB, IPT, true><<<(unsigned)(ntiles - 1), TB, 0, s>>>(d_n, src_k, dst_k, src_v, dst_v, st, ts, p, key_bits);  \
        else                                                                                                                   \
            k_pass<K, TB, IPT, false><<<(unsigned)(ntiles - 1), TB, 0, s>>>(d_n, src_k, dst_k, src_v, dst_v, st, ts, p, key_bits); \
    } while (0)
        if (tb == 256)
            SKM_OS_PASS(256, 8);
        else if (tb == 1024 && ipt == 8)
            SKM_OS_PASS(1024, 8);
        else if (sizeof(K) <= 4 && tb == 1024 && ipt == 12)
            SKM_OS_PASS(1024, (sizeof(K) <= 4 ? 12 : 8));
        else if (sizeof(K) <= 4 && tb == 1024 && ipt == 16)
            SKM_OS_PASS(1024, (sizeof(K) <= 4 ? 16 : 8));
        else
            SKM_OS_PASS(SKM_OS_LARGE_TB, 8);
#undef SKM_OS_PASS
        src_k = dst_k;
        src_v = dst_v;
    }
    return skm_check_launch(label);
}

}  // namespace skm_onesweep
