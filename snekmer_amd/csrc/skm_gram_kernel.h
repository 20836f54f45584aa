// k_gram_sparse: exact sparse Gram rows (neighbour lists) of X against the postings of Y.
// Included by skm_cosine_csr.hip inside its anonymous namespace (needs CH from there).
//
// For the rows of one strip every non-zero (row i, column c, count v) is a task whose posting list
// (rows j of Y holding c, with counts v') contributes v*v' to G[i][j].  The strip's tasks are
// binned by list length in LDS; groups of 4, 16 or G lanes then take one list each and walk it
// (every posting is read once; the loads of a group's next list are in flight while it inserts
// the current one; lists of thousands of postings are walked by the whole workgroup); products are
// summed in per-row LDS hash tables keyed by j; pairs whose key is not at its first-probe slot are
// parked in a per-wave queue and inserted 64 at a time.  Finally each row's (j, dot) entries are
// written to a global list (in no particular order: the streaming writer and the top-k kernel do
// not need one).
//
// Measured on the bench workload (PMC, profiles/): the kernel is bound by VALU issue (about 100
// vector instructions per 64 pairs, lanes 45 % busy because list lengths vary) on top of a gather
// of 8-byte postings that alone takes 0.85 ms (4.6 TB/s of 128-byte lines from beyond L2).  An
// earlier version flattened all (task, posting) pairs with a prefix sum and a per-step binary
// search: perfectly balanced lanes, but ~8 LDS operations per pair.
#pragma once
#include <type_traits>

#ifdef SKM_DIAG
// diagnostic build only (GABL == 3): summed shader-clock ticks per phase over all workgroups
__device__ unsigned long long g_gram_phase_ticks[8];
#endif

// Posting words.  64-bit form: row | count << 32.  32-bit form (half the gather bytes; rows < 2^24):
// row | min(count, 255) << 24, where 255 is an escape: the real count is postcnt[position], a side
// array that is written and read for such postings only (a k-mer repeated >= 255 times in one row).
template <typename PW>
struct posting;
template <>
struct posting<uint64_t> {
    static __device__ __forceinline__ uint32_t row(uint64_t w) { return (uint32_t)w; }
    static __device__ __forceinline__ uint32_t count(uint64_t w, const uint32_t *, uint32_t) { return (uint32_t)(w >> 32); }
};
template <>
struct posting<uint32_t> {
    static __device__ __forceinline__ uint32_t row(uint32_t w) { return w & 0x00FFFFFFu; }
    static __device__ __forceinline__ uint32_t count(uint32_t w, const uint32_t *postcnt, uint32_t at)
    {
        uint32_t c = w >> 24;
        if (__builtin_expect(c == 255u, 0))
            c = postcnt[at];
        return c;
    }
};

constexpr uint32_t G_OVERFLOW = 0xFFFFFFFFu;
constexpr uint32_t G_SINGLETON = 0xFFFFFFFFu;  // colidx of a k-mer that occurs in one row only

// The int32 guard.  The sparse kernels sum exact integer products in 32-bit cells, so a dot product must stay
// below 2^31.  By Cauchy-Schwarz <x, y> <= |x||y| = 1 / (rnorm_x * rnorm_y), hence a row of X is safe against EVERY
// row of Y when rnorm_x * min_j rnorm_y > 2^-31 (the margin covers the float32 rounding of the two reciprocal
// norms; an all-zero row has rnorm 1 and dot 0).  Rows that fail the test ("wide" rows: a count in the tens of
// thousands, a homopolymer of 46 342 windows or more, aggregated count matrices) never enter the 32-bit kernels:
// skm_cosine_csr hands their strips to the float64-accumulator form of k_cosine_strip, skm_gram_neighbors (whose
// output format IS a 32-bit dot) reports them as rows it cannot hold.
__device__ __forceinline__ bool skm_row_is_wide(float rnorm_x, float min_rnorm_y)
{
    return (double)rnorm_x * (double)min_rnorm_y <= 0x1p-31 * (1.0 + 1e-6);
}

// GABL  diagnostic ablation (0 = real kernel)
// GR    rows per workgroup           GH  hash slots per row (at most 3/4 GH distinct neighbours)
// GT    threads per workgroup        GQ  non-zeros per thread (GQ*GT non-zeros per strip)
// G     lanes that share one posting list          U  posting loads a lane issues before it inserts
template <int GABL, int GR, int GH, int GT, int GQ, int G, int U, typename PW>
__device__ __forceinline__ void gram_strip(const int64_t i0, const int64_t *__restrict__ xrowptr,
                                                    const uint32_t *__restrict__ xcolidx,
                                                    const uint32_t *__restrict__ xcounts,
                                                    const uint32_t *__restrict__ ycolptr,
                                                    const PW *__restrict__ ypost,
                                                    const uint32_t *__restrict__ ypostcnt, int64_t row0, int64_t row1,
                                                    unsigned long long fixed_stride, int64_t slot0,
                                                    uint64_t *__restrict__ g_ent, unsigned long long cap_ent,
                                                    unsigned long long *__restrict__ g_counter,
                                                    uint64_t *__restrict__ g_start, uint32_t *__restrict__ g_len,
                                                    uint32_t *__restrict__ over_list, uint32_t *__restrict__ over_count)
{
    // fixed_stride != 0: row r of the launch owns g_ent[(slot0 + r) * fixed_stride ...) (no allocation);
    // fixed_stride == 0: lists are allocated back to back from *g_counter, up to cap_ent entries.
    // g_start / g_len / over_list are indexed by the row's position in the launch (i - row0).
    static_assert(GR <= 16 && G <= 64 && 64 % G == 0 && GT % 64 == 0 && G * U >= 16, "shape");
    constexpr int GTCAP = GQ * GT;
    constexpr int GMAXD = GH / 4 * 3;  // load factor at most 0.75
    constexpr int HBITS = __builtin_ctz(GH);
    constexpr int LONG_DF = 4 * G * U;
    __shared__ uint32_t t_start[GTCAP], t_df[GTCAP], t_liv[GTCAP];
    __shared__ uint32_t hkeys[GR][GH];
    __shared__ int hvals[GR][GH];
    __shared__ int64_t s_rp[GR + 1];
    __shared__ uint32_t s_distinct[GR];
    __shared__ uint32_t s_fill[GR];
    __shared__ uint32_t s_self[GR];
    __shared__ unsigned long long s_off[GR];
    constexpr int NBIN = 4, B0 = 4, B1 = 16;  // list-length bins: <= 4, <= 16, <= LONG_DF, longer
    __shared__ uint32_t s_cnt[NBIN], s_fillc[NBIN];
    constexpr int GQCAP = 128;  // queue words per wave
    __shared__ uint64_t s_queue[GT / 64][GQCAP];
    __shared__ int s_over;
    __shared__ uint32_t s_pairs;
    constexpr int HEAVY_PAIRS = 24576;  // 16 visits per slot of a 2048-slot table's capacity
    const int tid = threadIdx.x, lane = tid & 63;
    const int rows = (int)min((int64_t)GR, row1 - i0);
    unsigned long long stamp = 0;
    auto phase = [&](int idx) {
#ifdef SKM_DIAG
        if (GABL == 3 && tid == 0) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (idx >= 0)
                atomicAdd(&g_gram_phase_ticks[idx], now - stamp);
            stamp = now;
        }
#else
        (void)idx;
        (void)stamp;
#endif
    };
    phase(-1);
    // rows this kernel cannot hold are flagged (and listed for the large-table pass)
    auto flag_rows = [&]() {
        if (tid < rows) {
            g_len[i0 - row0 + tid] = G_OVERFLOW;
            if (over_list)
                over_list[atomicAdd(over_count, 1u)] = (uint32_t)(i0 - row0 + tid);
        }
    };
    // G[li][j] += prod.  Most pairs hit a key that is already present, so look with a plain LDS
    // read first and pay for the compare-and-swap only when the slot still looks empty.
    auto insert = [&](int li, uint32_t j, int prod) {
        const uint32_t key = j + 1u;
        uint32_t h = (j * 2654435761u) >> (32 - HBITS);
        for (int probe = 0; probe < GH; ++probe) {
            uint32_t seen = __atomic_load_n(&hkeys[li][h], __ATOMIC_RELAXED);
            if (seen == 0u) {
                if (s_over)  // the row is already lost: no new keys, so the table never fills up
                    return;  // (a full table would make every later probe walk all GH slots)
                seen = atomicCAS(&hkeys[li][h], 0u, key);
                if (seen == 0u) {
                    seen = key;
                    if (atomicAdd(&s_distinct[li], 1u) >= (uint32_t)GMAXD)
                        s_over = 1;
                }
            }
            if (seen == key) {
                atomicAdd(&hvals[li][h], prod);
                break;
            }
            h = (h + 1) & (GH - 1);
        }
    };

    if (tid <= GR)
        s_rp[tid] = xrowptr[i0 + (tid <= rows ? tid : rows)];
    if (tid < GR) {
        s_distinct[tid] = 0;
        s_fill[tid] = 0;
        s_self[tid] = 0;
    }
    if (tid == 0) {
        s_over = 0;
        s_pairs = 0;
    }
    if (tid < NBIN) {
        s_cnt[tid] = 0;
        s_fillc[tid] = 0;
    }
    for (int z = tid; z < GR * GH; z += GT) {
        (&hkeys[0][0])[z] = 0u;
        (&hvals[0][0])[z] = 0;
    }
    __syncthreads();
    phase(0);  // table zeroing + row pointers
    const int64_t e0 = s_rp[0];
    const int64_t nnz64 = s_rp[GR] - e0;
    if (nnz64 > GTCAP) {
        flag_rows();
        return;
    }
    const int nnz = (int)nnz64;

    // Tasks: one per non-zero (row li, column c, count v) whose k-mer also occurs in another row.
    // A k-mer of this row only (skm_basis_build, ELIDE_SINGLETONS) pairs with nothing but its own
    // row: its v*v goes straight to the row's self product.
    uint32_t col[GQ], val[GQ], pb[GQ], pe[GQ];
    uint32_t self = 0;  // GR == 1: summed over the wave, then one LDS add
#pragma unroll
    for (int q = 0; q < GQ; ++q) {
        const int t = q * GT + tid;
        col[q] = G_SINGLETON;
        val[q] = 0;
        if (t < nnz) {
            col[q] = xcolidx[e0 + t];
            val[q] = xcounts[e0 + t];
        }
    }
#pragma unroll
    for (int q = 0; q < GQ; ++q) {
        pb[q] = pe[q] = 0;
        if (col[q] != G_SINGLETON && GABL != 9) {  // GABL 9 (diagnostic): no column-start gathers, no pair loop
            pb[q] = ycolptr[col[q]];
            pe[q] = ycolptr[col[q] + 1];
            if (GABL == 5)  // diagnostic: half of every list (the visit count of a symmetric half-Gram)
                pb[q] += (pe[q] - pb[q] + 1) / 2;
        }
    }
    // Lists are binned by length so that every bin gets lane groups of a fitting width (a list
    // of 2 postings on a 32-lane group would leave 30 lanes idle: at 100 k sequences half of the
    // lists have fewer than 5 postings).  Bin c holds t_*[s_off[c] .. s_off[c + 1]).
    // First pass only (rows with a fixed list slot): a row whose posting lists hold more than HEAVY_PAIRS postings
    // in total is handed on at once.  It would almost certainly outgrow the table, and finding that out by walking
    // costs most of the walk: on a batch with families of thousands
    // the aborted walks were 13 ms of a 39 ms step.  The pass behind this one is exact for any row, so the rule only
    // moves work; it never changes a result.
    if (fixed_stride != 0ull) {
        uint32_t pairs = 0;
#pragma unroll
        for (int q = 0; q < GQ; ++q)
            pairs += pe[q] - pb[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            pairs += __shfl_xor(pairs, o);
        if (lane == 0 && pairs)
            atomicAdd(&s_pairs, pairs);
    }
    int cls[GQ];
#pragma unroll
    for (int q = 0; q < GQ; ++q) {
        const uint32_t df = pe[q] - pb[q];
        cls[q] = pe[q] > pb[q] ? (df <= (uint32_t)B0 ? 0 : df <= (uint32_t)B1 ? 1 : df <= (uint32_t)LONG_DF ? 2 : 3) : -1;
#pragma unroll
        for (int c = 0; c < NBIN; ++c) {
            const unsigned long long bal = __ballot(cls[q] == c);
            if (bal && lane == 0)
                atomicAdd(&s_cnt[c], (uint32_t)__popcll(bal));
        }
    }
    __syncthreads();
    if (fixed_stride != 0ull && s_pairs > (uint32_t)HEAVY_PAIRS) {  // uniform for the workgroup
        flag_rows();
        return;
    }
    uint32_t boff[NBIN + 1];
    boff[0] = 0;
#pragma unroll
    for (int c = 0; c < NBIN; ++c)
        boff[c + 1] = boff[c] + s_cnt[c];
#pragma unroll
    for (int q = 0; q < GQ; ++q) {
        const int t = q * GT + tid;
        int li = 0;
#pragma unroll
        for (int r = 1; r < GR; ++r)
            li += (e0 + t >= s_rp[r]) ? 1 : 0;
        if (col[q] == G_SINGLETON && t < nnz) {
            const uint32_t vv = (val[q] & 0x0FFFFFFFu) * (val[q] & 0x0FFFFFFFu);
            if (GR == 1)
                self += vv;
            else
                atomicAdd(&s_self[li], vv);
        }
#pragma unroll
        for (int c = 0; c < NBIN; ++c) {
            const unsigned long long bal = __ballot(cls[q] == c);
            if (bal) {
                uint32_t base = 0;
                if (lane == 0)
                    base = atomicAdd(&s_fillc[c], (uint32_t)__popcll(bal));
                base = __shfl(base, 0);
                if (cls[q] == c) {
                    const uint32_t pos = boff[c] + base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                    t_start[pos] = pb[q];
                    t_df[pos] = pe[q] - pb[q];
                    t_liv[pos] = ((uint32_t)li << 28) | (val[q] & 0x0FFFFFFFu);
                }
            }
        }
    }
    if (GR == 1) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            self += __shfl_xor(self, o);
        if (lane == 0 && self)
            atomicAdd(&s_self[0], self);
    }
    __syncthreads();
    phase(1);  // task loads + compaction
    if (tid < rows && s_self[tid] && GABL != 1 && GABL != 4)
        insert(tid, (uint32_t)(i0 + tid), (int)s_self[tid]);

    // A batch = U postings of one list per lane (lane u-th posting at first + u*stride).
    // Lanes past the end of the list load posting 0 and discard it: loads under a branch would make
    // the compiler wait for ALL outstanding loads (vmcnt(0)) before the next LDS phase.  Only called
    // when the strip has a task, so posting 0 exists.
    auto load_batch = [&](auto &pw, uint32_t start, uint32_t df, uint32_t first, uint32_t stride) {
        constexpr int UU = (int)(sizeof(pw) / sizeof(pw[0]));
#pragma unroll
        for (int u = 0; u < UU; ++u)
            pw[u] = ypost[first + u * stride < df ? start + first + u * stride : 0u];
    };
    // Pairs whose key is not where the first probe looks (new neighbours, collisions: about a
    // quarter of all pairs) are parked in a per-wave queue and inserted QDRAIN at a time with every
    // lane busy.  Taking the general insert on the spot would run its branchy probe loop for a
    // handful of lanes in nearly every batch (measured: two thirds of the kernel's instructions).
    uint32_t qn = 0;  // wave-uniform
    uint64_t *const queue = s_queue[tid >> 6];
    auto drain = [&]() {
        for (uint32_t q = (uint32_t)lane; q < qn; q += 64) {
            const uint64_t it = queue[q];
            insert((int)(it >> 60), (uint32_t)(it >> 28), (int)(uint32_t)(it & 0x0FFFFFFFull));
        }
        qn = 0;
    };
    // all first-probe key reads, then the adds
    // `uniform`: every lane of the wave makes this call together (the queue counter is wave-uniform);
    // elsewhere misses take the general insert directly.
    auto insert_batch = [&](const auto &pw, uint32_t start, uint32_t df, int v, int li, uint32_t first, uint32_t stride,
                            auto uniform) {
        constexpr int UU = (int)(sizeof(pw) / sizeof(pw[0]));
        if (GABL == 4) {  // diagnostic: loads only, no hash insert
#pragma unroll
            for (int u = 0; u < UU; ++u)
                asm volatile("" ::"v"((uint32_t)pw[u]), "v"((uint32_t)((uint64_t)pw[u] >> 16)));
            return;
        }
        uint32_t hh[UU], seen[UU];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            hh[u] = (posting<PW>::row(pw[u]) * 2654435761u) >> (32 - HBITS);
            seen[u] = __atomic_load_n(&hkeys[li][hh[u]], __ATOMIC_RELAXED);
        }
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            const bool valid = first + u * stride < df;
            const uint32_t j = posting<PW>::row(pw[u]);
            const uint32_t prod = (uint32_t)v * posting<PW>::count(pw[u], ypostcnt, valid ? start + first + u * stride : 0u);
            const bool hit = valid && seen[u] == j + 1u;
            if (hit)
                atomicAdd(&hvals[li][hh[u]], (int)prod);
            const bool miss = valid && !hit;
            const bool parked = decltype(uniform)::value && miss && prod <= 0x0FFFFFFFu;
            if (miss && !parked)
                insert(li, j, (int)prod);  // does not fit the queue word (a count above 2^14)
            const unsigned long long bal = decltype(uniform)::value ? __ballot(parked) : 0ull;
            if (bal) {
                if (parked)
                    queue[qn + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] =
                        ((uint64_t)li << 60) | ((uint64_t)j << 28) | prod;
                qn += (uint32_t)__popcll(bal);
                if (qn > (uint32_t)(GQCAP - 64))
                    drain();
            }
        }
    };
    // workgroup walk of one list: every wave runs the same number of steps with all lanes
    auto walk_all = [&](uint32_t start, uint32_t df, int v, int li) {
        for (uint32_t p0 = 0; p0 < df && !s_over; p0 += GT * U) {
            PW pw[U];
            load_batch(pw, start, df, p0 + (uint32_t)tid, GT);
            insert_batch(pw, start, df, v, li, p0 + (uint32_t)tid, GT, std::true_type{});
        }
    };
    // Bins 0-2: GC consecutive lanes walk one list, UC postings per lane and step, software
    // pipelined so that the loads of a group's next list are in flight while it inserts the
    // current one.  Bin 3 (a few low-complexity k-mers occur in thousands of rows) is walked by the
    // whole workgroup, list after list.
    struct hdr {
        uint32_t start, df, lv;
    };
    auto run_bin = [&](auto gc_, auto uc_, int tb, int te) {
        constexpr int GC = decltype(gc_)::value, UC = decltype(uc_)::value, NGC = GT / GC;
        const uint32_t gl = (uint32_t)(tid % GC);
        auto fetch = [&](PW (&pw)[UC], int t) -> hdr {
            hdr h = {0u, 0u, 0u};
            if (t < te)
                h = {t_start[t], t_df[t], t_liv[t]};
            load_batch(pw, h.start, h.df, gl, GC);
            return h;
        };
        auto consume = [&](const PW (&pw)[UC], const hdr &h) {
            const int v = (int)(h.lv & 0x0FFFFFFFu), li = (int)(h.lv >> 28);
            insert_batch(pw, h.start, h.df, v, li, gl, GC, std::true_type{});
            if (h.df > (uint32_t)(GC * UC)) {  // the rest of a list longer than one batch (bin 2 only)
                for (uint32_t p = gl + GC * UC; p < h.df && !s_over; p += GC * UC) {
                    PW more[UC];  // lanes of one wave may be in different lists here:
                    load_batch(more, h.start, h.df, p, GC);  // misses are inserted directly
                    insert_batch(more, h.start, h.df, v, li, p, GC, std::false_type{});
                }
            }
        };
        // D register sets used in turn (no copies: a copy would wait for the load)
        constexpr int D = 2;
        PW buf[D][UC];
        hdr hd[D];
        int t = tb + tid / GC;
        int tw = tb + (tid >> 6) * (64 / GC);  // the wave's first list: every lane of a wave runs the
#pragma unroll                                 // same number of steps (the queue counter needs that)
        for (int d = 0; d < D; ++d)
            hd[d] = fetch(buf[d], t + d * NGC);
        while (tw < te && !s_over) {
#pragma unroll
            for (int d = 0; d < D; ++d) {  // lists past the end have df == 0: nothing is inserted
                consume(buf[d], hd[d]);
                hd[d] = fetch(buf[d], t + (d + D) * NGC);
            }
            t += D * NGC;
            tw += D * NGC;
        }
    };
    if (GABL != 1 && boff[NBIN] > 0) {
        // longest lists first: they bring most of a row's distinct neighbours, so a row that will
        // overflow the table does so early and wastes little work before the larger-table pass
        for (int tl = (int)boff[3]; tl < (int)boff[4] && !s_over; ++tl) {
            const uint32_t lv = t_liv[tl];
            walk_all(t_start[tl], t_df[tl], (int)(lv & 0x0FFFFFFFu), (int)(lv >> 28));
        }
        // GABL 6 / 7 / 8 (diagnostic): without the long / middle / short bin
        if (boff[3] > boff[2] && GABL != 6)
            run_bin(std::integral_constant<int, G>{}, std::integral_constant<int, U>{}, (int)boff[2], (int)boff[3]);
        if (boff[2] > boff[1] && GABL != 7)
            run_bin(std::integral_constant<int, B1>{}, std::integral_constant<int, 1>{}, (int)boff[1], (int)boff[2]);
        if (boff[1] > boff[0] && GABL != 8)
            run_bin(std::integral_constant<int, B0>{}, std::integral_constant<int, 1>{}, (int)boff[0], (int)boff[1]);
    }
    if (GABL != 1 && GABL != 4)
        drain();
    __syncthreads();
    phase(2);  // pair loop
    if (s_over) {
        flag_rows();
        return;
    }
    if (GABL == 2) {
        if (tid < rows)
            g_len[i0 - row0 + tid] = 0;
        return;
    }

    // emit each row's (j, dot) entries; one LDS counter bump per wave and step
    if (tid < GR) {
        unsigned long long off = 0;
        if (tid < rows)
            off = fixed_stride ? (unsigned long long)(slot0 + i0 - row0 + tid) * fixed_stride
                               : atomicAdd(g_counter, (unsigned long long)s_distinct[tid]);
        s_off[tid] = off;
    }
    __syncthreads();
    bool fits[GR];
#pragma unroll
    for (int r = 0; r < GR; ++r)
        fits[r] = fixed_stride ? (unsigned long long)s_distinct[r] <= fixed_stride
                               : s_off[r] + (unsigned long long)s_distinct[r] <= cap_ent;
    static_assert((GR * GH) % GT == 0 && GH % 64 == 0, "emit loop must be wave-uniform");
    for (int z = tid; z < GR * GH; z += GT) {
        const uint32_t key = (&hkeys[0][0])[z];
        const int r = z / GH;  // the same for all lanes of a wave
        const bool has = key != 0u && fits[r];
        const unsigned long long bal = __ballot(has);
        if (bal) {
            uint32_t base = 0;
            if (lane == 0)
                base = atomicAdd(&s_fill[r], (uint32_t)__popcll(bal));
            base = __shfl(base, 0);
            if (has)
                g_ent[s_off[r] + base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] =
                    ((uint64_t)(key - 1u) << 32) | (uint32_t)(&hvals[0][0])[z];
        }
    }
    if (tid < rows) {
        g_start[i0 - row0 + tid] = s_off[tid];
        g_len[i0 - row0 + tid] = fits[tid] ? s_distinct[tid] : G_OVERFLOW;
    }
    phase(3);  // emit
}

// One strip of GR consecutive rows per workgroup.
template <int GABL, int GR, int GH, int GT, int GQ, int G, int U, typename PW>
__global__ __launch_bounds__(GT) void k_gram_sparse(const int64_t *__restrict__ xrowptr,
                                                    const uint32_t *__restrict__ xcolidx,
                                                    const uint32_t *__restrict__ xcounts,
                                                    const uint32_t *__restrict__ ycolptr,
                                                    const PW *__restrict__ ypost,
                                                    const uint32_t *__restrict__ ypostcnt, int64_t row0, int64_t row1,
                                                    unsigned long long fixed_stride, int64_t slot0,
                                                    uint64_t *__restrict__ g_ent, unsigned long long cap_ent,
                                                    unsigned long long *__restrict__ g_counter,
                                                    uint64_t *__restrict__ g_start, uint32_t *__restrict__ g_len,
                                                    uint32_t *__restrict__ over_list, uint32_t *__restrict__ over_count,
                                                    const float *__restrict__ xrnorm,
                                                    const float *__restrict__ min_yrnorm, uint32_t wide_mark)
{
    static_assert(GR == 1, "the wide-row test below is per workgroup");
    // a row whose dot products may reach 2^31 (skm_row_is_wide) is marked and left to the caller's wide path
    if (skm_row_is_wide(xrnorm[row0 + blockIdx.x], *min_yrnorm)) {
        if (threadIdx.x == 0)
            g_len[blockIdx.x] = wide_mark;
        return;
    }
    gram_strip<GABL, GR, GH, GT, GQ, G, U, PW>(row0 + (int64_t)blockIdx.x * GR, xrowptr, xcolidx, xcounts, ycolptr, ypost,
                                             ypostcnt, row0, row1, fixed_stride, slot0, g_ent, cap_ent, g_counter, g_start,
                                             g_len, over_list, over_count);
}

// Second pass with a large table (one row per workgroup) over the rows the first pass listed.
// The grid is fixed; workgroups stride over the list, whose length is only known on the device.
template <int GH, int GT, int GQ, int G, int U, typename PW>
__global__ __launch_bounds__(GT) void k_gram_sparse_big(const int64_t *__restrict__ xrowptr,
                                                        const uint32_t *__restrict__ xcolidx,
                                                        const uint32_t *__restrict__ xcounts,
                                                        const uint32_t *__restrict__ ycolptr,
                                                        const PW *__restrict__ ypost,
                                                        const uint32_t *__restrict__ ypostcnt, int64_t row0, int64_t row1,
                                                        uint64_t *__restrict__ g_ent,
                                                        unsigned long long cap_ent,
                                                        unsigned long long *__restrict__ g_counter,
                                                        uint64_t *__restrict__ g_start, uint32_t *__restrict__ g_len,
                                                        const uint32_t *__restrict__ row_list,
                                                        const uint32_t *__restrict__ row_count,
                                                        uint32_t *__restrict__ over_list,
                                                        uint32_t *__restrict__ over_count)
{
    const uint32_t cnt = *row_count;
    for (uint32_t idx = blockIdx.x; idx < cnt; idx += gridDim.x) {
        const int64_t i0 = row0 + row_list[idx];
        // a one-row strip: clamp row1 so that the strip never spills into the next row
        gram_strip<0, 1, GH, GT, GQ, G, U, PW>(i0, xrowptr, xcolidx, xcounts, ycolptr, ypost, ypostcnt, row0, i0 + 1, 0ull, 0,
                                             g_ent, cap_ent, g_counter, g_start, g_len, over_list, over_count);
        __syncthreads();
    }
}
