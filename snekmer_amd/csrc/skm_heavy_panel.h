// Heavy rows on the matrix cores (included by skm_cosine_csr.hip inside its anonymous namespace).
//
// k_cosine_heavy walks, for every heavy row, the posting list of every one of its k-mers: on a batch with families of
// thousands each list is re-read once per member (measured on synth_skewed at 100 k rows: 99 GB of postings walked for
// 0.26 GB of distinct postings, 22 ms).  But the members of a family share their LONG lists: rows x long-list columns
// is a dense block, and the dot products over those columns are a small int8 GEMM (BASELINE north_star: "MFMA on the ...
// count matrix since N x N similarity is a true dense GEMM").  So, for the heavy rows of a call:
//
//   k_panel_key      per heavy row: its smallest long-list column (a min-hash: members of a family agree on it often)
//   one-sweep sort   heavy rows ordered by that key; consecutive PB_ROWS rows form a BLOCK (no group detection: a block
//                    that mixes families is merely less dense)
//   k_panel_dict     per block: the distinct long-list columns of its rows (LDS hash set) -> slots 0..K-1 (at most K_MAX;
//                    what does not fit stays with the walk), columns holding a count above 127 marked bad
//   k_panel_rows     per block: the union J of the posting lists of its slots as a bitmap over Y's rows (LDS), prefix
//                    popcounts -> rank of every row of J, the list J itself; counts above 127 on the Y side mark a slot bad
//   k_panel_zero / _fill   int8 panels A [PB_ROWS x K] (the block's rows) and B [|J| x K] (Y's rows of J), bad slots left zero
//   k_panel_gemm     G = A B^T (int32, v_mfma_i32_32x32x32_i8), |dot| <= 127^2 K_MAX: no overflow
//   k_cosine_heavy   (PANEL form) walks only what the panel does not cover - short lists, lists longer than DF_MAX,
//                    slots that did not fit or are bad - and adds the row of G through J
//
// Every posting of a long list is then read twice per BLOCK (bitmap, fill) instead of once per ROW.  Exactness does
// not depend on how good the grouping is: a (row, column) pair is either in the panel (slot found and not bad) or
// walked, never both, and everything is integer arithmetic.  Blocks whose J outgrows J_MAX are disabled (K = 0: all
// walked).  All sizes live on the device; every launch covers the worst case and exits on the device-side counts.
#pragma once

constexpr int PB_ROWS = 256;       // heavy rows per block
constexpr int PB_KMAX = 1024;      // panel columns per block
constexpr int PB_DICT = 8192;      // hash slots of a block's column dictionary (round 4: 2048, and a tenth of the long entries of
                                   // synth_skewed's heavy rows found no room: mixed blocks hold thousands of distinct long columns)
constexpr int PB_JMAX = 16384;     // rows of Y a block's panel may touch
constexpr uint32_t PB_DF_LONG = 64, PB_DF_MAX = 8192;  // a column is panel material when DF_LONG < df <= DF_MAX
constexpr uint32_t PB_NOSLOT = 0xFFFFFFFFu;
#ifndef SKM_PB_MIN_ROWS
#define SKM_PB_MIN_ROWS 24  // (bench.py skewed_workload, ms per step: 2 30.2, 4 29.0, 8 23.5, 16 23.2, 24 22.7, 32 23.0, 48 23.8)
#endif
constexpr uint32_t PB_MIN_ROWS = SKM_PB_MIN_ROWS;  // rows of a block that must share a column for it to get a slot
constexpr int PB_STEP_COLS = 32768; // columns per step of the PANEL form of k_cosine_heavy
constexpr int PB_STEPS = 32;       // such steps over at most 2^20 rows of Y
constexpr int PB_MAXBLOCKS = 512;  // at most 131072 heavy rows get panels; the rest are walked

struct panel_bufs {
    uint32_t *key;        // [hcap] clustering key per heavy row (over_list order)
    uint32_t *skey;       // [hcap] sorted keys
    uint32_t *perm;       // [hcap] sorted position -> index into over_list
    int64_t *count64;     // the heavy-row count as the sort wants it
    uint2 *dict;          // [nb][PB_DICT]: (column, slot); the slot is PB_NOSLOT for a column without one and, once k_panel_rows
                          // has run, for a bad one: a lookup is ONE 8-byte load per probe (keys, slots and bad flags in three
                          // arrays were three dependent loads at the end of the heavy kernel's per-row set-up chain)
    uint32_t *cols;       // [nb][PB_KMAX]
    uint8_t *bad;         // [nb][PB_KMAX]
    uint32_t *meta;       // [nb][4]: K, |J|, rows in the block, 0
    uint32_t *jlist;      // [nb][PB_JMAX]
    uint32_t *jbound;     // [nb][PB_STEPS + 1]: first entry of jlist at or after row t * PB_STEP_COLS (the heavy kernel's column steps)
    int8_t *A;            // [nb][PB_ROWS][PB_KMAX]
    int *G;               // [nb][PB_ROWS][PB_JMAX]
    uint32_t mwords;      // words of a bitmap over Y's rows
    int nb;               // blocks the buffers hold
};

__device__ __forceinline__ uint32_t panel_hash(uint32_t c) { return (c * 2654435761u) >> (32 - 13); }  // PB_DICT = 2^13
static_assert(PB_DICT == 1 << 13, "panel_hash");

// slot of column c in block b's dictionary, PB_NOSLOT when absent / not given a slot / bad
__device__ __forceinline__ uint32_t panel_lookup(const panel_bufs &pb, uint32_t b, uint32_t c)
{
    const uint2 *dict = pb.dict + (size_t)b * PB_DICT;
    uint32_t h = panel_hash(c);
    for (int probe = 0; probe < 32; ++probe) {
        const uint2 e = dict[h];
        if (e.x == c)
            return e.y;
        if (e.x == NONE)
            return PB_NOSLOT;
        h = (h + 1) & (PB_DICT - 1);
    }
    return PB_NOSLOT;
}

// Wave per heavy row: the clustering key = the smallest row of Y that shares a long-list column with this row (the
// first posting of every such column is its smallest row).  Members of a family nearly all share some long k-mer with
// the family's first member, so they get the same key and sort next to each other - a single min-hash over the
// columns (smallest shared column id) was tried first: a member holds only ~28 % of the family's k-mers, so a family
// split into groups of 28 %, 20 %, 15 % ... of its size scattered over the order, and a quarter of the pairs stayed
// outside the panels.  Also leaves the row count as int64 for the sort.
template <typename PW>
__global__ __launch_bounds__(256) void k_panel_key(const int64_t *__restrict__ xrowptr, const uint32_t *__restrict__ xcolidx,
                                                   const uint32_t *__restrict__ ycolptr, const PW *__restrict__ ypost,
                                                   int64_t row0, int64_t rbase, const uint32_t *__restrict__ row_list,
                                                   const uint32_t *__restrict__ row_count, panel_bufs pb)
{
    const uint32_t cnt = *row_count;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    if (wave == 0 && lane == 0)
        *pb.count64 = (int64_t)cnt;
    for (int64_t idx = wave; idx < (int64_t)cnt; idx += nwaves) {
        const int64_t i = row0 + rbase + row_list[idx];
        uint32_t best = 0xFFFFFFFFu;
        constexpr int UN = 4;  // 256 entries per round trip: column id -> column start / end -> first posting are dependent gathers
        const int64_t e1 = xrowptr[i + 1];
        for (int64_t t0 = xrowptr[i]; t0 < e1; t0 += 64 * UN) {
            uint32_t c[UN], p0[UN], df[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int64_t t = t0 + u * 64 + lane;
                c[u] = t < e1 ? xcolidx[t] : NONE;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                p0[u] = df[u] = 0;
                if (c[u] != NONE) {
                    p0[u] = ycolptr[c[u]];
                    df[u] = ycolptr[c[u] + 1] - p0[u];
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (df[u] > PB_DF_LONG && df[u] <= PB_DF_MAX)
                    best = min(best, posting<PW>::row(ypost[p0[u]]));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
            best = min(best, (uint32_t)__shfl_xor(best, o));
        if (lane == 0)
            pb.key[idx] = best;
    }
}

// One workgroup per block: dictionary of the long-list columns its rows share, and the A panel.
__global__ __launch_bounds__(1024) void k_panel_dict(const int64_t *__restrict__ xrowptr, const uint32_t *__restrict__ xcolidx,
                                                     const uint32_t *__restrict__ xcounts,
                                                     const uint32_t *__restrict__ ycolptr, int64_t row0, int64_t rbase,
                                                     const uint32_t *__restrict__ row_list,
                                                     const uint32_t *__restrict__ row_count, panel_bufs pb)
{
    // (two arrays = 64 KiB: two blocks' workgroups per CU, and the ~310 blocks of a 100 k-row skewed batch in one round of the grid)
    __shared__ uint32_t s_key[PB_DICT], s_slot[PB_DICT];  // s_slot: the number of rows holding the column until step 2 turns it into the slot
    __shared__ uint32_t s_k;
    const uint32_t cnt = min(*row_count, (uint32_t)(pb.nb * PB_ROWS));
    const uint32_t b = blockIdx.x, first = b * PB_ROWS;
    if (first >= cnt)
        return;
    const uint32_t rows = min((uint32_t)PB_ROWS, cnt - first);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    for (int z = tid; z < PB_DICT; z += 1024) {
        s_key[z] = NONE;
        s_slot[z] = 0;
    }
    if (tid < PB_KMAX)
        pb.bad[(size_t)b * PB_KMAX + tid] = 0;
    int8_t *A = pb.A + (size_t)b * PB_ROWS * PB_KMAX;
    for (int z = tid; z < PB_ROWS * PB_KMAX / 16; z += 1024)
        reinterpret_cast<int4 *>(A)[z] = make_int4(0, 0, 0, 0);
    if (tid == 0)
        s_k = 0;
    __syncthreads();
    // the wave's rows (r = wid, wid + 16, ...: at most 16 of them): row id and entry range, one lane per row, loaded together
    // (perm -> row list -> row pointers was a chain of three dependent loads in front of every row)
    int64_t my_e0 = 0, my_e1 = 0;
    {
        const uint32_t r = (uint32_t)wid + 16u * (uint32_t)lane;
        if (lane < 16 && r < rows) {
            const int64_t i = row0 + rbase + row_list[pb.perm[first + r]];
            my_e0 = xrowptr[i];
            my_e1 = xrowptr[i + 1];
        }
    }
    auto for_entries = [&](auto need_df, auto &&fn) {
        constexpr int UN = 4;  // 64 x UN entries of a row per round trip (one at a time: each a chain of two or three dependent
                               // gathers at HBM latency, 16 rows x 5 rounds x 2 passes of them in series per wave)
        for (uint32_t r = wid, k = 0; r < rows; r += 16, ++k) {
            const int64_t e0 = __shfl(my_e0, (int)k), e1 = __shfl(my_e1, (int)k);
            for (int64_t t0 = e0; t0 < e1; t0 += 64 * UN) {
                uint32_t c[UN], v[UN], lo[UN], hi[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int64_t t = t0 + u * 64 + lane;
                    c[u] = t < e1 ? xcolidx[t] : NONE;
                    v[u] = t < e1 ? xcounts[t] : 0u;
                }
                if (decltype(need_df)::value) {  // (two random gathers per entry: only the pass that builds the table pays)
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        lo[u] = hi[u] = 0;
                        if (c[u] != NONE) {
                            lo[u] = ycolptr[c[u]];
                            hi[u] = ycolptr[c[u] + 1];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    if (c[u] == NONE)
                        continue;
                    if (decltype(need_df)::value) {
                        const uint32_t df = hi[u] - lo[u];
                        if (!(df > PB_DF_LONG && df <= PB_DF_MAX))
                            continue;
                    }
                    fn(r, c[u], v[u]);
                }
            }
        }
    };
    // 1. the columns, with the number of the block's rows that hold each
    for_entries(std::true_type{}, [&](uint32_t, uint32_t c, uint32_t) {
        uint32_t h = panel_hash(c);
        for (int probe = 0; probe < 32; ++probe) {
            const uint32_t old = atomicCAS(&s_key[h], NONE, c);
            if (old == NONE || old == c) {
                atomicAdd(&s_slot[h], 1u);
                break;
            }
            h = (h + 1) & (PB_DICT - 1);
        }
    });
    __syncthreads();
    // 2. slots for the first PB_KMAX of those that at least PB_MIN_ROWS rows of the block share: a column of one or two
    // rows gains nothing from a panel, and its posting list would drag unrelated rows into J
    for (int z = tid; z < PB_DICT; z += 1024) {
        const uint32_t holders = s_slot[z];
        s_slot[z] = PB_NOSLOT;
        if (s_key[z] != NONE && holders >= PB_MIN_ROWS) {
            const uint32_t s = atomicAdd(&s_k, 1u);
            if (s < (uint32_t)PB_KMAX) {
                s_slot[z] = s;
                pb.cols[(size_t)b * PB_KMAX + s] = s_key[z];
            }
        }
    }
    __syncthreads();
    // 3. the A panel (a column that is not in the table was not long, or found no room); a count above 127 (X side)
    // makes the whole column bad: it stays with the walk
    for_entries(std::false_type{}, [&](uint32_t r, uint32_t c, uint32_t v) {
        uint32_t h = panel_hash(c);
        for (int probe = 0; probe < 32; ++probe) {
            const uint32_t k = s_key[h];
            if (k == c) {
                const uint32_t s = s_slot[h];
                if (s != PB_NOSLOT) {
                    if (v > 127u)
                        pb.bad[(size_t)b * PB_KMAX + s] = 1;
                    else
                        A[(size_t)r * PB_KMAX + s] = (int8_t)v;
                }
                break;
            }
            if (k == NONE)
                break;
            h = (h + 1) & (PB_DICT - 1);
        }
    });
    for (int z = tid; z < PB_DICT; z += 1024) {
        pb.dict[(size_t)b * PB_DICT + z] = make_uint2(s_key[z], s_slot[z]);
    }
    if (tid == 0) {
        pb.meta[b * 4 + 0] = min(s_k, (uint32_t)PB_KMAX);
        pb.meta[b * 4 + 1] = 0;
        pb.meta[b * 4 + 2] = rows;
        pb.meta[b * 4 + 3] = 0;
    }
}

// one workgroup per block: J = union of the posting lists of its slots (bitmap over Y's rows in LDS), as a sorted list
template <typename PW>
__global__ __launch_bounds__(1024) void k_panel_rows(const uint32_t *__restrict__ ycolptr, const PW *__restrict__ ypost,
                                                     const uint32_t *__restrict__ ypostcnt,
                                                     const uint32_t *__restrict__ row_count, panel_bufs pb)
{
    extern __shared__ uint32_t s_bm[];  // mwords
    __shared__ uint32_t s_wsum[16];
    const uint32_t cnt = min(*row_count, (uint32_t)(pb.nb * PB_ROWS));
    const uint32_t b = blockIdx.x;
    if (b * PB_ROWS >= cnt)
        return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const uint32_t K = pb.meta[b * 4 + 0], mw = pb.mwords;
    for (uint32_t z = tid; z < mw; z += 1024)
        s_bm[z] = 0u;
    __syncthreads();
    for (uint32_t s = wid; s < K; s += 16) {
        const uint32_t c = pb.cols[(size_t)b * PB_KMAX + s];
        bool big = false;
        const uint32_t p0 = ycolptr[c], pe = ycolptr[c + 1];
        for (uint32_t pp = p0 + lane; pp < pe; pp += 64 * 4) {  // four loads in flight per lane
            PW pw[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                pw[u] = ypost[min(pp + 64u * u, pe - 1u)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t p = pp + 64u * u;
                if (p < pe) {
                    const uint32_t j = posting<PW>::row(pw[u]);
                    atomicOr(&s_bm[j >> 5], 1u << (j & 31));
                    big |= posting<PW>::count(pw[u], ypostcnt, p) > 127u;
                }
            }
        }
        if (__any(big) && lane == 0)
            pb.bad[(size_t)b * PB_KMAX + s] = 1;  // a count above 127 on the Y side
    }
    __syncthreads();
    // exclusive prefix popcount over the words: thread t owns words [t * per, (t + 1) * per)
    const uint32_t per = (mw + 1023) / 1024;
    const uint32_t w0 = min(mw, tid * per), w1 = min(mw, w0 + per);
    uint32_t mine = 0;
    for (uint32_t w = w0; w < w1; ++w)
        mine += __popc(s_bm[w]);
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if (lane >= o)
            incl += up;
    }
    if (lane == 63)
        s_wsum[wid] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        before += w < wid ? s_wsum[w] : 0u;
        total += s_wsum[w];
    }
    const bool fits = total <= (uint32_t)PB_JMAX;
    uint32_t at = before + incl - mine;
    if (fits) {
        for (uint32_t w = w0; w < w1; ++w) {
            uint32_t bits = s_bm[w];
            while (bits) {
                const int bit = __ffs((int)bits) - 1;
                pb.jlist[(size_t)b * PB_JMAX + at++] = w * 32u + (uint32_t)bit;
                bits &= bits - 1u;
            }
        }
    }
    // jbound[t] = rows of J below t * PB_STEP_COLS = popcount of the bitmap words below word t * PB_STEP_COLS / 32: the
    // owner of that word adds its leading words to its own exclusive prefix
    for (int t = 0; t <= PB_STEPS; ++t) {
        const uint32_t W = (uint32_t)t * (PB_STEP_COLS / 32);
        if (W >= mw) {
            if (tid == 0)
                pb.jbound[(size_t)b * (PB_STEPS + 1) + t] = total;
        } else if (W / per == (uint32_t)tid) {
            uint32_t acc = before + incl - mine;
            for (uint32_t w = w0; w < W; ++w)
                acc += __popc(s_bm[w]);
            pb.jbound[(size_t)b * (PB_STEPS + 1) + t] = acc;
        }
    }
    if (tid == 0) {
        pb.meta[b * 4 + 1] = fits ? total : 0u;
        if (!fits)
            pb.meta[b * 4 + 0] = 0u;  // block disabled: its rows are walked in full
    }
    // bad slots (a count above 127 on either side; the X side was marked by k_panel_dict, the Y side above, in front of a
    // barrier) leave the dictionary: panel_lookup then needs no flag array
    for (int z = tid; z < PB_DICT; z += 1024) {
        const uint32_t s = pb.dict[(size_t)b * PB_DICT + z].y;
        if (s != PB_NOSLOT && pb.bad[(size_t)b * PB_KMAX + s])
            pb.dict[(size_t)b * PB_DICT + z].y = PB_NOSLOT;
    }
}

// G[b] = A[b] B[b]^T with B never stored: a workgroup owns a run of 128-row tiles of J; per tile it builds the B tile
// [128 x K] in LDS straight from the posting lists (a cursor per slot in registers: postings are sorted by row, so the
// tile's share of a list follows the previous tile's; rank within the tile = position in the tile's slice of J, found
// by binary search in LDS) and multiplies both halves of A against it (v_mfma_i32_32x32x32_i8).
typedef int pi32x16 __attribute__((ext_vector_type(16)));
typedef int pi32x4 __attribute__((ext_vector_type(4)));
constexpr int PG_CHUNKS = 8;               // workgroups per block
constexpr int PG_KP = PB_KMAX + 16;        // LDS row stride of the B tile (bank spread)
#ifndef SKM_PG_U
#define SKM_PG_U 8
#endif
constexpr int PG_U = SKM_PG_U;             // postings per slot and round trip of the B-tile fill
constexpr int PG_TB = 512;                 // 8 waves: all fill the B tile, four multiply each half of A against it

template <typename PW>
__global__ __launch_bounds__(PG_TB) void k_panel_gemm(const uint32_t *__restrict__ ycolptr, const PW *__restrict__ ypost,
                                                      const uint32_t *__restrict__ ypostcnt,
                                                      const uint32_t *__restrict__ row_count, panel_bufs pb)
{
    constexpr int TN = 128, TK = 64, LROW = TK + 16;
    extern __shared__ __attribute__((aligned(16))) int8_t s_dynb[];  // B tile [TN][PG_KP]
    __shared__ __attribute__((aligned(16))) int8_t s_a[PB_ROWS * LROW];
    __shared__ uint32_t s_j[TN];
    // rank of a row of Y inside the tile's slice of J: a hash table of (row << 7 | rank) words, 512 slots for 128 rows - one
    // LDS read per posting where the binary search in s_j took seven dependent ones (and, one thread per list, was the
    // fill's time: 45 postings per thread and tile)
    constexpr int PG_TAB = 512;
    __shared__ uint32_t s_tab[PG_TAB];
    static_assert(TN == 128, "s_tab packs the rank in 7 bits (and rows of Y in 25: panels need m <= 2^20)");
    int8_t *s_b = s_dynb;
    const uint32_t cnt = min(*row_count, (uint32_t)(pb.nb * PB_ROWS));
    const uint32_t b = blockIdx.y;
    if (b * PB_ROWS >= cnt)
        return;
    const uint32_t K = pb.meta[b * 4 + 0], J = pb.meta[b * 4 + 1], rows = pb.meta[b * 4 + 2];
    if (K == 0 || J == 0)
        return;
    const uint32_t ntiles = (J + TN - 1) / TN, per = (ntiles + PG_CHUNKS - 1) / PG_CHUNKS;
    const uint32_t t0 = blockIdx.x * per, t1 = min(ntiles, t0 + per);
    if (t0 >= t1)
        return;
    const uint32_t kpad = (K + 63) / 64 * 64, kq = kpad / 16;
    const int8_t *A = pb.A + (size_t)b * PB_ROWS * PB_KMAX;
    const uint32_t *jl = pb.jlist + (size_t)b * PB_JMAX;
    int *G = pb.G + (size_t)b * PB_ROWS * PB_JMAX;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int half = wid >> 2, wq = wid & 3, wr = wq >> 1, wc = wq & 1;  // `half`: which 128 rows of A this wave multiplies
    const bool active = (uint32_t)half * 128u < rows;                   // (wave-uniform)
    // this thread's slots: tid, tid + PG_TB; cursor = first posting at or after the chunk's first row of J
    constexpr int SPT = PB_KMAX / PG_TB;
    uint32_t cur[SPT], end[SPT];
    const uint32_t jfirst = jl[t0 * TN];
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        const uint32_t s = tid + q * PG_TB;
        cur[q] = end[q] = 0;
        if (s < K && !pb.bad[(size_t)b * PB_KMAX + s]) {
            const uint32_t c = pb.cols[(size_t)b * PB_KMAX + s];
            uint32_t lo = ycolptr[c], hi = ycolptr[c + 1];
            end[q] = hi;
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (posting<PW>::row(ypost[mid]) < jfirst)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            cur[q] = lo;
        }
    }
    for (uint32_t t = t0; t < t1; ++t) {
        const uint32_t col0 = t * TN, ncol = min((uint32_t)TN, J - col0);
        __syncthreads();  // the previous tile's fragment reads are done
        uint32_t myj = 0xFFFFFFFFu;
        if (tid < TN)
            s_j[tid] = myj = tid < (int)ncol ? jl[col0 + tid] : 0xFFFFFFFFu;
        s_tab[tid] = 0xFFFFFFFFu;
        static_assert(PG_TB == PG_TAB, "one table slot per thread");
        for (uint32_t z = tid; z < (uint32_t)TN * kq; z += PG_TB)
            *reinterpret_cast<int4 *>(s_b + (z / kq) * PG_KP + (z % kq) * 16) = make_int4(0, 0, 0, 0);
        __syncthreads();
        if (myj != 0xFFFFFFFFu) {
            uint32_t h = (myj * 2654435761u) >> (32 - 9);
            while (atomicCAS(&s_tab[h], 0xFFFFFFFFu, (myj << 7) | (uint32_t)tid) != 0xFFFFFFFFu)
                h = (h + 1u) & (PG_TAB - 1);
        }
        __syncthreads();
        const uint32_t jhi = s_j[ncol - 1];
        // both of the thread's slots in the same round trip, PG_U postings each (round 4: one slot after the other, four
        // postings per round trip - the fill, a chain of such round trips per tile, was most of the kernel's time)
        bool more[SPT];
        bool any_more = false;
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            more[q] = cur[q] < end[q];
            any_more |= more[q];
        }
        while (any_more) {
            PW pw[SPT][PG_U];
#pragma unroll
            for (int q = 0; q < SPT; ++q)
#pragma unroll
                for (int u = 0; u < PG_U; ++u)
                    pw[q][u] = ypost[more[q] ? min(cur[q] + (uint32_t)u, end[q] - 1u) : 0u];  // (posting 0 exists: K > 0)
            any_more = false;
#pragma unroll
            for (int q = 0; q < SPT; ++q) {
                const uint32_t s = tid + q * PG_TB;
                int took = 0;
#pragma unroll
                for (int u = 0; u < PG_U; ++u) {
                    const uint32_t j = posting<PW>::row(pw[q][u]);
                    if (!more[q] || took != u || cur[q] + (uint32_t)u >= end[q] || j > jhi)
                        continue;
                    // every row of a slot's list is in J, and j lies in this tile's slice of it (the cursor starts at the
                    // chunk's first row and j <= jhi): the probe finds it
                    uint32_t h = (j * 2654435761u) >> (32 - 9), e = s_tab[h];
                    for (int probe = 0; probe < PG_TAB && (e >> 7) != j; ++probe) {
                        h = (h + 1u) & (PG_TAB - 1);
                        e = s_tab[h];
                    }
                    s_b[(e & 127u) * PG_KP + s] = (int8_t)posting<PW>::count(pw[q][u], ypostcnt, cur[q] + (uint32_t)u);
                    took = u + 1;
                }
                cur[q] += (uint32_t)took;
                more[q] = more[q] && took == PG_U && cur[q] < end[q];
                any_more |= more[q];
            }
        }
        pi32x16 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[a][c][r] = 0;
        const int st_row = tid >> 2, st_chunk = tid & 3;  // 128 rows x 64 B per pass, two passes: all 256 rows of A
        for (uint32_t k0 = 0; k0 < kpad; k0 += TK) {
            pi32x4 va[2];
#pragma unroll
            for (int pss = 0; pss < 2; ++pss)
                va[pss] = *reinterpret_cast<const pi32x4 *>(A + (size_t)(st_row + pss * 128) * PB_KMAX + k0 + st_chunk * 16);
            __syncthreads();  // (first pass: also closes the fill of the B tile)
#pragma unroll
            for (int pss = 0; pss < 2; ++pss)
                *reinterpret_cast<pi32x4 *>(s_a + (st_row + pss * 128) * LROW + st_chunk * 16) = va[pss];
            __syncthreads();
            if (active) {
                const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
                for (int ks = 0; ks < TK / 32; ++ks) {
                    pi32x4 fa[2], fb[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        fa[u] = *reinterpret_cast<const pi32x4 *>(s_a + (half * 128 + wr * 64 + u * 32 + fr) * LROW + ks * 32 + fh * 16);
                        fb[u] = *reinterpret_cast<const pi32x4 *>(s_b + (wc * 64 + u * 32 + fr) * PG_KP + k0 + ks * 32 + fh * 16);
                    }
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            acc[a][c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[a], fb[c], acc[a][c], 0, 0, 0);
                }
            }
        }
        if (active) {
            const int ccol = lane & 31, chalf = lane >> 5;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const uint32_t j = col0 + wc * 64 + c * 32 + ccol;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t i = half * 128 + wr * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * chalf;
                        if (j < J)
                            G[(size_t)i * PB_JMAX + j] = acc[a][c][r];
                    }
                }
        }
    }
}
