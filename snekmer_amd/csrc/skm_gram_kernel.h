// k_gram_sparse: exact sparse Gram rows (neighbour lists) of X against the postings of Y.
// Included by skm_cosine_csr.hip inside its anonymous namespace (needs CH from there).
//
// For the rows of one strip every non-zero (row i, column c, count v) is a task whose posting list
// (rows j of Y holding c, with counts v') contributes v*v' to G[i][j].  All (task, posting) pairs
// of the strip are flattened with an LDS prefix sum over the posting-list lengths, so each
// posting is read exactly once, by independent and mostly coalesced loads; products are summed in
// per-row LDS hash tables keyed by j.  Finally each row's (j, dot) entries are written to a
// global list (in no particular order: the streaming writer and the top-k kernel do not need one).
#pragma once

// diagnostic only (GABL == 3): summed shader-clock ticks per phase over all workgroups
__device__ unsigned long long g_gram_phase_ticks[8];

constexpr uint32_t G_OVERFLOW = 0xFFFFFFFFu;
constexpr uint32_t G_SINGLETON = 0xFFFFFFFFu;  // colidx of a k-mer that occurs in one row only

// GABL  diagnostic ablation (0 = real kernel)
// GR    rows per workgroup           GH  hash slots per row (at most GH/2 distinct neighbours)
// GT    threads per workgroup        GQ  tasks per thread (GQ*GT non-zeros per strip)
// SL    consecutive pairs a thread handles per step
template <int GABL, int GR, int GH, int GT, int GQ, int SL>
__device__ __forceinline__ void gram_strip(const int64_t i0, const int64_t *__restrict__ xrowptr,
                                                    const uint32_t *__restrict__ xcolidx,
                                                    const uint32_t *__restrict__ xcounts,
                                                    const uint32_t *__restrict__ ycolptr,
                                                    const uint64_t *__restrict__ ypost, int64_t row0, int64_t row1,
                                                    unsigned long long fixed_stride, int64_t slot0,
                                                    uint64_t *__restrict__ g_ent, unsigned long long cap_ent,
                                                    unsigned long long *__restrict__ g_counter,
                                                    uint64_t *__restrict__ g_start, uint32_t *__restrict__ g_len,
                                                    uint32_t *__restrict__ over_list, uint32_t *__restrict__ over_count)
{
    // fixed_stride != 0: row r of the launch owns g_ent[(slot0 + r) * fixed_stride ...) (no allocation);
    // fixed_stride == 0: lists are allocated back to back from *g_counter, up to cap_ent entries.
    // g_start / g_len / over_list are indexed by the row's position in the launch (i - row0).
    constexpr int GTCAP = GQ * GT;
    constexpr int GMAXD = GH / 2;
    constexpr int HBITS = __builtin_ctz(GH);
    __shared__ uint32_t s_u[3 * GTCAP + 1];
    __shared__ uint32_t hkeys[GR][GH];
    __shared__ int hvals[GR][GH];
    __shared__ int64_t s_rp[GR + 1];
    __shared__ uint32_t s_wsum[GT / 64];
    __shared__ uint32_t s_distinct[GR];
    __shared__ uint32_t s_fill[GR];
    __shared__ unsigned long long s_off[GR];
    __shared__ int s_over;
    uint32_t *t_start = s_u, *t_scan = s_u + GTCAP, *t_liv = s_u + 2 * GTCAP + 1;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int rows = (int)min((int64_t)GR, row1 - i0);
    unsigned long long stamp = 0;
    auto phase = [&](int idx) {
        if (GABL == 3 && tid == 0) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (idx >= 0)
                atomicAdd(&g_gram_phase_ticks[idx], now - stamp);
            stamp = now;
        }
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

    if (tid <= GR)
        s_rp[tid] = xrowptr[i0 + (tid <= rows ? tid : rows)];
    if (tid < GR) {
        s_distinct[tid] = 0;
        s_fill[tid] = 0;
    }
    if (tid == 0)
        s_over = 0;
    for (int z = tid; z < GR * GH; z += GT) {
        (&hkeys[0][0])[z] = 0u;
        (&hvals[0][0])[z] = 0;
    }
    __syncthreads();
    phase(0);  // table zeroing + row pointers
    const int64_t e0 = s_rp[0];
    const int64_t ntasks64 = s_rp[GR] - e0;
    if (ntasks64 > GTCAP) {
        flag_rows();
        return;
    }
    const int ntasks = (int)ntasks64;

    // posting-list length of every task + exclusive prefix sum over the strip
    uint32_t mydf[GQ];
    uint32_t mysum = 0;
#pragma unroll
    for (int q = 0; q < GQ; ++q) {
        const int t = tid * GQ + q;
        mydf[q] = 0;
        if (t < ntasks) {
            const int64_t e = e0 + t;
            int li = 0;
#pragma unroll
            for (int r = 1; r < GR; ++r)
                li += (e >= s_rp[r]) ? 1 : 0;
            const uint32_t c = xcolidx[e];
            t_liv[t] = ((uint32_t)li << 28) | (xcounts[e] & 0x0FFFFFFFu);
            if (c == G_SINGLETON) {  // k-mer of this row only (skm_basis_build, ELIDE_SINGLETONS)
                t_start[t] = G_SINGLETON;
                mydf[q] = 1;
            } else {
                const uint32_t pb = ycolptr[c], pe = ycolptr[c + 1];
                t_start[t] = pb;
                mydf[q] = pe - pb;
            }
        }
        mysum += mydf[q];
    }
    uint32_t incl = mysum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t up = __shfl_up(incl, o);
        if (lane >= o)
            incl += up;
    }
    if (lane == 63)
        s_wsum[wid] = incl;
    __syncthreads();
    uint32_t wbase = 0, total = 0;
#pragma unroll
    for (int q = 0; q < GT / 64; ++q) {
        wbase += q < wid ? s_wsum[q] : 0;
        total += s_wsum[q];
    }
    uint32_t run = wbase + incl - mysum;
#pragma unroll
    for (int q = 0; q < GQ; ++q) {
        const int t = tid * GQ + q;
        if (t < ntasks)
            t_scan[t] = run;
        run += mydf[q];
    }
    if (tid == 0)
        t_scan[ntasks] = total;
    __syncthreads();
    phase(1);  // task loads + prefix sum

    // every (task, posting) pair once.  A thread takes SL consecutive pairs per step: one
    // branch-free binary search locates the first pair's task, the rest walk forward; all posting
    // loads of the step are issued before the first hash insert.
    for (uint32_t g0 = (uint32_t)tid * SL; GABL != 1 && g0 < total; g0 += GT * SL) {
        if (s_over)
            break;
        int t = 0;
#pragma unroll
        for (int w = GTCAP / 2; w > 0; w >>= 1) {
            const int cand = t + w;
            if (cand < ntasks && t_scan[cand] <= g0)
                t = cand;
        }
        uint32_t jj[SL], ww[SL], lv[SL];
#pragma unroll
        for (int u = 0; u < SL; ++u) {
            const uint32_t g = g0 + u;
            jj[u] = ww[u] = lv[u] = 0;
            if (g < total) {
                while (t_scan[t + 1] <= g)
                    ++t;
                lv[u] = t_liv[t];
                if (t_start[t] == G_SINGLETON) {  // pairs with its own row only
                    jj[u] = (uint32_t)(i0 + (lv[u] >> 28));
                    ww[u] = lv[u] & 0x0FFFFFFFu;
                } else {
                    const uint64_t pw = ypost[t_start[t] + (g - t_scan[t])];
                    jj[u] = (uint32_t)pw;
                    ww[u] = (uint32_t)(pw >> 32);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < SL; ++u) {
            if (GABL == 4) {  // diagnostic: loads only, no hash insert
                asm volatile("" ::"v"(jj[u]), "v"(ww[u]), "v"(lv[u]));
            } else if (g0 + u < total) {
                const uint32_t j = jj[u];
                const int prod = (int)(lv[u] & 0x0FFFFFFFu) * (int)ww[u];
                const int li = (int)(lv[u] >> 28);
                const uint32_t key = j + 1u;
                uint32_t h = (j * 2654435761u) >> (32 - HBITS);
                // Most pairs hit a key that is already present, so look with a plain LDS read first
                // and pay for the compare-and-swap only when the slot still looks empty.
                for (int probe = 0; probe < GH; ++probe) {
                    uint32_t seen = __atomic_load_n(&hkeys[li][h], __ATOMIC_RELAXED);
                    if (seen == 0u) {
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
            }
        }
    }
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
template <int GABL, int GR, int GH, int GT, int GQ, int SL>
__global__ __launch_bounds__(GT) void k_gram_sparse(const int64_t *__restrict__ xrowptr,
                                                    const uint32_t *__restrict__ xcolidx,
                                                    const uint32_t *__restrict__ xcounts,
                                                    const uint32_t *__restrict__ ycolptr,
                                                    const uint64_t *__restrict__ ypost, int64_t row0, int64_t row1,
                                                    unsigned long long fixed_stride, int64_t slot0,
                                                    uint64_t *__restrict__ g_ent, unsigned long long cap_ent,
                                                    unsigned long long *__restrict__ g_counter,
                                                    uint64_t *__restrict__ g_start, uint32_t *__restrict__ g_len,
                                                    uint32_t *__restrict__ over_list, uint32_t *__restrict__ over_count)
{
    gram_strip<GABL, GR, GH, GT, GQ, SL>(row0 + (int64_t)blockIdx.x * GR, xrowptr, xcolidx, xcounts, ycolptr, ypost, row0,
                                         row1, fixed_stride, slot0, g_ent, cap_ent, g_counter, g_start, g_len, over_list, over_count);
}

// Second pass with a large table (one row per workgroup) over the rows the first pass listed.
// The grid is fixed; workgroups stride over the list, whose length is only known on the device.
template <int GH, int GT, int GQ, int SL>
__global__ __launch_bounds__(GT) void k_gram_sparse_big(const int64_t *__restrict__ xrowptr,
                                                        const uint32_t *__restrict__ xcolidx,
                                                        const uint32_t *__restrict__ xcounts,
                                                        const uint32_t *__restrict__ ycolptr,
                                                        const uint64_t *__restrict__ ypost, int64_t row0, int64_t row1,
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
        gram_strip<0, 1, GH, GT, GQ, SL>(i0, xrowptr, xcolidx, xcounts, ycolptr, ypost, row0, i0 + 1, 0ull, 0, g_ent, cap_ent,
                                         g_counter, g_start, g_len, over_list, over_count);
        __syncthreads();
    }
}
