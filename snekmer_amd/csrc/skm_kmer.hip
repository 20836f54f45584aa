// Stage 1 of the hot path: recode, k-mer window codes and per-sequence k-mer counts (CSR).
//
// Reference behaviour being replaced (per record, in Python):
//   reduce()                 snekmer/vectorize.py:173-195
//   KmerVec._kmer_gen        snekmer/vectorize.py:239-249
//   KmerVec.reduce_vectorize snekmer/vectorize.py:292-328
//   count loops              snekmer/rules/learn.smk:359-383, snekmer/rules/apply.smk:188-206
//
// Device design (gfx950): one 64-lane wavefront owns one sequence.  The wave stages the
// sequence's class ranks in LDS once (1 B/residue read from HBM), every lane then forms the codes
// of windows lane, lane+64, ... straight from LDS, the wave sorts its <= 512 codes with a bitonic
// network held in registers (cross-lane steps are wave shuffles), run-length encodes them with
// ballots and writes (code, count) pairs.  Sequences with more than 512 windows take a
// workgroup-per-sequence variant with the keys in LDS (<= 8192 windows) or in global scratch.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "skm_common.h"
#include "skm_onesweep.h"
#include "skm_wave_sort.h"

namespace {

constexpr int SHORT_MAX = 512;    // windows handled by one wave's register sort (8 per lane)
constexpr int LONGSEQ_MAX = 8192;    // windows handled by one workgroup: 16 waves' register sorts merged through LDS
constexpr int MIDSEQ_MAX = 4096;     // ... by a workgroup of 8 waves (round 5: nearly all of a proteome's long sequences; 4 such workgroups fit a CU)
constexpr int NBUCKET = 7;        // 0: no windows, 1: short, 2: 513..4096, 3: 4097..8192, 6: keys in global scratch; 4-5 unused
constexpr int BLK = 256;

template <typename K>
__device__ __forceinline__ K sentinel()
{
    return ~K(0);
}

// ------------------------------------------------------------------------------- recode
__global__ void k_recode_bytes(skm_lut256 lut, const uint8_t *__restrict__ seq, uint8_t *__restrict__ out,
                               int64_t total)
{
    __shared__ uint8_t s_lut[256];
    if (threadIdx.x < 64)
        reinterpret_cast<uint32_t *>(s_lut)[threadIdx.x] = reinterpret_cast<const uint32_t *>(lut.b)[threadIdx.x];
    __syncthreads();
    const int64_t nvec = total >> 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const uint4 *in4 = reinterpret_cast<const uint4 *>(seq);
    uint4 *out4 = reinterpret_cast<uint4 *>(out);
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        uint4 w = in4[v];
        uint32_t *p = reinterpret_cast<uint32_t *>(&w);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t x = p[q];
            p[q] = (uint32_t)s_lut[x & 255] | ((uint32_t)s_lut[(x >> 8) & 255] << 8) |
                   ((uint32_t)s_lut[(x >> 16) & 255] << 16) | ((uint32_t)s_lut[x >> 24] << 24);
        }
        out4[v] = w;
    }
    for (int64_t b = (nvec << 4) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < total; b += stride)
        out[b] = s_lut[seq[b]];
}

// Stripped length (trailing '*' removed, snekmer/vectorize.py:193), window count and size class.
__global__ void k_classify(const uint8_t *__restrict__ seq, const int64_t *__restrict__ off, int64_t n, int k,
                           int32_t *__restrict__ slen, int32_t *__restrict__ nwin, uint32_t *__restrict__ lists,
                           uint32_t *__restrict__ bucket_fill, int32_t *__restrict__ row_nnz, int64_t max_win = -1,
                           uint32_t *err = nullptr)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    int64_t b = off[i], e = off[i + 1];
    while (e > b && seq[e - 1] == '*')
        --e;
    int64_t len = e - b;
    int64_t w = len - k + 1;
    if (w < 0)
        w = 0;
    if (max_win >= 0 && w > max_win) {
        // longer than the bound the caller gave (which sized the launches and the scratch behind this kernel): the row
        // stays empty and the context's sticky error word says so at the next host wait (skm_check_device_error)
        w = 0;
        if (err)  // one bit is defined: a plain system-scope store (a read-modify-write on host memory needs PCIe atomics)
            __hip_atomic_store(err, SKM_DEVERR_SEQ_TOO_LONG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (slen)
        slen[i] = (int32_t)len;
    if (nwin)
        nwin[i] = (int32_t)w;
    if (lists) {
        int bucket;
        if (w == 0) {
            bucket = 0;
            row_nnz[i] = 0;
        } else if (w <= SHORT_MAX)
            bucket = 1;
        else if (w <= MIDSEQ_MAX)
            bucket = 2;
        else if (w <= LONGSEQ_MAX)
            bucket = 3;
        else
            bucket = 6;
        if (i == 0)
            row_nnz[n] = 0;  // the closing element of the scan over row_nnz
        // one atomic per (wave, bucket) instead of one per sequence
        const int lane = threadIdx.x & 63;
        uint32_t slot = 0;
        for (int bk = 0; bk < NBUCKET; ++bk) {
            unsigned long long mask = __ballot(bucket == bk);
            if (!mask)
                continue;
            uint32_t base = 0;
            const int leader = __ffsll((long long)mask) - 1;
            if (lane == leader)
                base = atomicAdd(&bucket_fill[bk], (uint32_t)__popcll(mask));
            base = __shfl(base, leader);
            if (bucket == bk)
                slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        }
        lists[(int64_t)bucket * n + slot] = (uint32_t)i;
    }
}

// ------------------------------------------------------------------------------- window codes
template <typename K>
__device__ __forceinline__ K window_code(const uint8_t *ranks, int p, int k, int nsym)
{
    K c = 0;
    uint32_t bad = 0;
    for (int j = 0; j < k; ++j) {
        uint32_t r = ranks[p + j];
        bad |= (r == 0xFFu);
        c = c * (K)nsym + (K)r;
    }
    return bad ? sentinel<K>() : c;
}

// Wave per sequence, windows in window order (a5/a6).  Tiles of 1024 windows with a k-1 halo.
template <typename K>
__global__ __launch_bounds__(64) void k_kmer_codes(skm_lut256 lut, int nsym, int k,
                                                   const uint8_t *__restrict__ seq,
                                                   const int64_t *__restrict__ off, int64_t n,
                                                   const int32_t *__restrict__ slen, K *__restrict__ codes)
{
    constexpr int TILE = 1024;
    __shared__ uint8_t s_lut[256];
    __shared__ uint8_t s_rank[TILE + 64];
    const int lane = threadIdx.x;
    reinterpret_cast<uint32_t *>(s_lut)[lane] = reinterpret_cast<const uint32_t *>(lut.b)[lane];
    for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
        const int64_t b = off[i];
        const int len = slen[i];
        const int w = len - k + 1;
        for (int t0 = 0; t0 < w; t0 += TILE) {
            __syncthreads();
            int span = min(TILE + k - 1, len - t0);
            for (int p = lane; p < span; p += 64)
                s_rank[p] = s_lut[seq[b + t0 + p]];
            __syncthreads();
            int wt = min(TILE, w - t0);
            for (int p = lane; p < wt; p += 64)
                codes[b + t0 + p] = window_code<K>(s_rank, p, k, nsym);
        }
    }
}

// Wave-per-sequence count kernel for sequences with 1..512 windows.  Lane l owns the 8
// consecutive windows 8l..8l+7 (one code from k ranks, the next seven by a rolling update), the
// wave sorts the 512 codes in registers (skm_wave_sort.h), run heads are ranked with a wave scan,
// and (code, run start) go through LDS so that the global stores are contiguous.
template <typename K, bool WITH_POS>
__global__ __launch_bounds__(64) void k_count_short(skm_lut256 lut, int nsym, int k,
                                                    const uint8_t *__restrict__ seq,
                                                    const int64_t *__restrict__ off,
                                                    const int32_t *__restrict__ slen,
                                                    const uint32_t *__restrict__ list,
                                                    const uint32_t *__restrict__ nlist_ptr,
                                                    K *__restrict__ tmp_codes, uint32_t *__restrict__ tmp_counts,
                                                    uint32_t *__restrict__ tmp_first, int32_t *__restrict__ row_nnz,
                                                    uint32_t *__restrict__ zero, int64_t zero_words)
{
    if (zero) {  // state of the NEXT stage (skm_count_extras): cleared here, by the widest launch of this one
        uint4 *z4 = reinterpret_cast<uint4 *>(zero);
        const int64_t n4 = zero_words >> 2, gthreads = (int64_t)gridDim.x * 64;
        for (int64_t z = (int64_t)blockIdx.x * 64 + threadIdx.x; z < n4; z += gthreads)
            z4[z] = make_uint4(0u, 0u, 0u, 0u);
        for (int64_t z = (n4 << 2) + (int64_t)blockIdx.x * 64 + threadIdx.x; z < zero_words; z += gthreads)
            zero[z] = 0u;
    }
    // the list length is read on the device: the launch does not wait for the host to learn it
    const uint32_t nlist = *nlist_ptr;
    __shared__ uint8_t s_lut[256];
    __shared__ uint8_t s_rank[SHORT_MAX + 64 + 8];
    __shared__ K s_uniq[SHORT_MAX];
    __shared__ uint32_t s_start[SHORT_MAX + 1];
    __shared__ uint32_t s_pos[WITH_POS ? SHORT_MAX : 1];
    const int lane = threadIdx.x;
    const K SENT = sentinel<K>();
    reinterpret_cast<uint32_t *>(s_lut)[lane] = reinterpret_cast<const uint32_t *>(lut.b)[lane];
    K msd = 1;  // nsym^(k-1): weight of the rank that leaves a window
    for (int j = 1; j < k; ++j)
        msd *= (K)nsym;

    for (uint32_t it = blockIdx.x; it < nlist; it += gridDim.x) {
        const uint32_t i = list[it];
        const int64_t b = off[i];
        const int len = slen[i];
        const int w = len - k + 1;
        __syncthreads();
        for (int p = lane; p < len; p += 64)
            s_rank[p] = s_lut[seq[b + p]];
        __syncthreads();

        K v[8];
        K orig[WITH_POS ? 8 : 1];
        const int p0 = lane * 8;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            v[r] = SENT;
        if (p0 < w) {
            K c = 0;
            int bad = 0;
            for (int j = 0; j < k; ++j) {
                const uint32_t x = s_rank[p0 + j];
                bad += x == 0xFFu;
                c = c * (K)nsym + (K)(x == 0xFFu ? 0u : x);
            }
            v[0] = bad ? SENT : c;
#pragma unroll
            for (int r = 1; r < 8; ++r) {
                // ranks past the sequence end may be stale; they only reach windows >= w, which stay SENT
                const uint32_t out = s_rank[p0 + r - 1], in = s_rank[p0 + r - 1 + k];
                bad += (int)(in == 0xFFu) - (int)(out == 0xFFu);
                c = (c - (K)(out == 0xFFu ? 0u : out) * msd) * (K)nsym + (K)(in == 0xFFu ? 0u : in);
                if (p0 + r < w && bad == 0)
                    v[r] = c;
            }
        }
        if constexpr (WITH_POS) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                orig[r] = v[r];
        }
        wave_bitonic_512_blocked<K>(v, lane);

        // run heads: element e = lane*8 + r starts a run if it differs from element e-1
        K prev = shfl_idx_k<K>(v[7], (lane + 63) & 63);
        uint32_t hm = 0, nval = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const bool valid = v[r] != SENT;
            const bool head = valid && ((r == 0 && lane == 0) || v[r] != prev);
            hm |= head ? (1u << r) : 0u;
            nval += valid ? 1u : 0u;
            prev = v[r];
        }
        // exclusive wave scan of the per-lane head counts (low half) and valid counts (high half)
        const uint32_t mine = (uint32_t)__popc(hm) | (nval << 16);
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o)
                incl += up;
        }
        const uint32_t tot = __shfl(incl, 63);
        const int nruns = (int)(tot & 0xFFFFu), nvalid = (int)(tot >> 16);
        const uint32_t base = (incl - mine) & 0xFFFFu;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if ((hm >> r) & 1u) {
                const uint32_t idx = base + (uint32_t)__popc(hm & ((1u << r) - 1u));
                s_uniq[idx] = v[r];
                s_start[idx] = (uint32_t)(p0 + r);
            }
        }
        if (lane == 0) {
            s_start[nruns] = (uint32_t)nvalid;
            row_nnz[i] = nruns;
        }
        if constexpr (WITH_POS) {
            for (int t = lane; t < nruns; t += 64)
                s_pos[t] = 0xFFFFFFFFu;
        }
        __syncthreads();
        for (int t = lane; t < nruns; t += 64) {
            tmp_codes[b + t] = s_uniq[t];
            tmp_counts[b + t] = s_start[t + 1] - s_start[t];
        }
        if constexpr (WITH_POS) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                K c = orig[r];
                if (c != SENT) {
                    int lo = 0, hi = nruns - 1;
                    while (lo < hi) {
                        int mid = (lo + hi) >> 1;
                        if (s_uniq[mid] < c)
                            lo = mid + 1;
                        else
                            hi = mid;
                    }
                    atomicMin(&s_pos[lo], (uint32_t)(p0 + r));
                }
            }
            __syncthreads();
            for (int t = lane; t < nruns; t += 64)
                tmp_first[b + t] = s_pos[t];
        }
    }
}

// ------------------------------------------------------------------------------- long sequences (513..8192 windows)
// One workgroup of 16 waves per sequence.  The keys stay in registers in the GLOBAL blocked layout (element e =
// wave*512 + lane*8 + r): every wave sorts its 512 with the register network of the short kernel, then the sorted runs
// are merged pairwise (bitonic merges in flip form); only the stages whose partner sits in another wave go through LDS
// (10 exchanges for 8192 keys), the rest are the same DPP / shuffle stages.  A real proteome has 15-25 % of its
// sequences here; the kernel this replaces kept the keys in LDS and ran all 55-91 stages of a bitonic sort there with
// 256 threads: 20 / 40 / 80 us for ONE sequence of 1024 / 2048 / 4096 windows, a launch per size class (the host had
// to learn the class sizes first), 0.16 ms for the reference's CI proteome.  This one reads its list length on the device.
template <typename K>
__device__ __forceinline__ void lds_exchange(K (&v)[8], K *s_x, int e0, int partner_base, bool flip, bool lower)
{
#pragma unroll
    for (int r = 0; r < 8; ++r)
        s_x[e0 + r] = v[r];
    __syncthreads();
    K o[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
        o[r] = s_x[partner_base + (flip ? 7 - r : r)];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const K mn = v[r] < o[r] ? v[r] : o[r], mx = v[r] < o[r] ? o[r] : v[r];
        v[r] = lower ? mn : mx;
    }
}

// LONG_TB threads, 8 windows each: 1024 threads for up to 8192 windows, 512 for up to 4096 (the 37 KiB of LDS of that form
// let four sequences share a CU where the 16-wave form holds two, most of its waves sorting sentinels)
template <typename K, bool WITH_POS, int MAXW>
constexpr size_t count_long_lds()
{
    return (size_t)MAXW * sizeof(K) + sizeof(uint32_t) * (MAXW + 1) + (WITH_POS ? sizeof(uint32_t) * MAXW : 0) +
           (size_t)MAXW + 64 + 16;
}

template <typename K, bool WITH_POS, int LONG_TB>
__global__ __launch_bounds__(LONG_TB) void k_count_long(skm_lut256 lut, int nsym, int k, const uint8_t *__restrict__ seq,
                                                        const int64_t *__restrict__ off, const int32_t *__restrict__ slen,
                                                        const uint32_t *__restrict__ list,
                                                        const uint32_t *__restrict__ nlist_ptr, K *__restrict__ tmp_codes,
                                                        uint32_t *__restrict__ tmp_counts, uint32_t *__restrict__ tmp_first,
                                                        int32_t *__restrict__ row_nnz)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    __shared__ uint8_t s_lut[256];
    __shared__ K s_last[LONG_TB / 64];
    __shared__ uint32_t s_wtot[LONG_TB / 64];
    constexpr int MAXW = LONG_TB * 8;
    K *s_x = reinterpret_cast<K *>(s_dyn);  // exchange buffer of the merges, then the distinct keys
    uint32_t *s_start = reinterpret_cast<uint32_t *>(s_x + MAXW);
    uint32_t *s_pos = s_start + MAXW + 1;
    uint8_t *s_rank = reinterpret_cast<uint8_t *>(s_pos + (WITH_POS ? MAXW : 0));
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const K SENT = sentinel<K>();
    const uint32_t nlist = *nlist_ptr;
    if (tid < 64)
        reinterpret_cast<uint32_t *>(s_lut)[tid] = reinterpret_cast<const uint32_t *>(lut.b)[tid];
    K msd = 1;
    for (int j = 1; j < k; ++j)
        msd *= (K)nsym;
    const int e0 = tid * 8;  // first of this thread's 8 consecutive elements (= windows, before the sort)

    for (uint32_t it = blockIdx.x; it < nlist; it += gridDim.x) {
        const uint32_t i = list[it];
        const int64_t b = off[i];
        const int len = slen[i];
        const int w = len - k + 1;
        __syncthreads();
        for (int p = tid; p < len; p += LONG_TB)
            s_rank[p] = s_lut[seq[b + p]];
        __syncthreads();

        K v[8];
        K orig[WITH_POS ? 8 : 1];
#pragma unroll
        for (int r = 0; r < 8; ++r)
            v[r] = SENT;
        if (e0 < w) {
            K c = 0;
            int bad = 0;
            for (int j = 0; j < k; ++j) {
                const uint32_t x = s_rank[e0 + j];
                bad += x == 0xFFu;
                c = c * (K)nsym + (K)(x == 0xFFu ? 0u : x);
            }
            v[0] = bad ? SENT : c;
#pragma unroll
            for (int r = 1; r < 8; ++r) {
                if (e0 + r < w) {  // (the short kernel reads stale ranks past the end; here the buffer is exactly sized)
                    const uint32_t out = s_rank[e0 + r - 1], in = s_rank[e0 + r - 1 + k];
                    bad += (int)(in == 0xFFu) - (int)(out == 0xFFu);
                    c = (c - (K)(out == 0xFFu ? 0u : out) * msd) * (K)nsym + (K)(in == 0xFFu ? 0u : in);
                    if (bad == 0)
                        v[r] = c;
                }
            }
        }
        if constexpr (WITH_POS) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
                orig[r] = v[r];
        }
        wave_bitonic_512_blocked<K>(v, lane);
        // merges of sorted runs: sizes 1024, 2048, ... up to the power of two that holds the sequence (uniform)
        for (int S = 2 * SHORT_MAX; (S >> 1) < w; S <<= 1) {
            lds_exchange<K>(v, s_x, e0, (e0 ^ (S - 1)) & ~7, true, (e0 & (S >> 1)) == 0);
            for (int stride = S >> 2; stride >= SHORT_MAX; stride >>= 1)
                lds_exchange<K>(v, s_x, e0, e0 ^ stride, false, (e0 & stride) == 0);
            wave_stage<32, false>(v, lane);
            wave_stage<16, false>(v, lane);
            wave_stage<8, false>(v, lane);
            wave_stage<4, false>(v, lane);
            wave_stage<2, false>(v, lane);
            wave_stage<1, false>(v, lane);
            reg_tail(v);
        }
        // run heads over the whole workgroup
        if (lane == 63)
            s_last[wid] = v[7];
        __syncthreads();
        K prev = shfl_idx_k<K>(v[7], (lane + 63) & 63);
        if (lane == 0 && wid > 0)
            prev = s_last[wid - 1];
        uint32_t hm = 0, nval = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const bool valid = v[r] != SENT;
            const bool head = valid && ((r == 0 && tid == 0) || v[r] != prev);
            hm |= head ? (1u << r) : 0u;
            nval += valid ? 1u : 0u;
            prev = v[r];
        }
        const uint32_t mine = (uint32_t)__popc(hm) | (nval << 16);  // at most 8192 each: 16 bits hold them
        uint32_t incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= o)
                incl += up;
        }
        if (lane == 63)
            s_wtot[wid] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int q = 0; q < LONG_TB / 64; ++q) {
            const uint32_t t = s_wtot[q];
            before += q < wid ? t : 0u;
            total += t;
        }
        const int nruns = (int)(total & 0xFFFFu), nvalid = (int)(total >> 16);
        const uint32_t base = (before + incl - mine) & 0xFFFFu;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if ((hm >> r) & 1u) {
                const uint32_t idx = base + (uint32_t)__popc(hm & ((1u << r) - 1u));
                s_x[idx] = v[r];
                s_start[idx] = (uint32_t)(e0 + r);
            }
        }
        if (tid == 0) {
            s_start[nruns] = (uint32_t)nvalid;
            row_nnz[i] = nruns;
        }
        if constexpr (WITH_POS) {
            for (int t = tid; t < nruns; t += LONG_TB)
                s_pos[t] = 0xFFFFFFFFu;
        }
        __syncthreads();
        for (int t = tid; t < nruns; t += LONG_TB) {
            tmp_codes[b + t] = s_x[t];
            tmp_counts[b + t] = s_start[t + 1] - s_start[t];
        }
        if constexpr (WITH_POS) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const K c = orig[r];
                if (c != SENT) {
                    int lo = 0, hi = nruns - 1;
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (s_x[mid] < c)
                            lo = mid + 1;
                        else
                            hi = mid;
                    }
                    atomicMin(&s_pos[lo], (uint32_t)(e0 + r));
                }
            }
            __syncthreads();
            for (int t = tid; t < nruns; t += LONG_TB)
                tmp_first[b + t] = s_pos[t];
        }
    }
}

// ------------------------------------------------------------------------------- block variant
// Workgroup per sequence; keys / run heads / first positions live in LDS (GLOBAL=false, cap <=
// 8192) or in a per-workgroup slice of global scratch (GLOBAL=true, any length).
template <typename K, bool WITH_POS, bool GLOBAL>
__global__ __launch_bounds__(BLK) void k_count_block(skm_lut256 lut, int nsym, int k,
                                                     const uint8_t *__restrict__ seq,
                                                     const int64_t *__restrict__ off,
                                                     const int32_t *__restrict__ slen,
                                                     const uint32_t *__restrict__ list, uint32_t nlist_host,
                                                     const uint32_t *__restrict__ nlist_ptr, uint32_t cap_lds,
                                                     const int64_t *__restrict__ g_base, int64_t slice_bytes,
                                                     uint8_t *__restrict__ g_scratch,
                                                     K *__restrict__ tmp_codes, uint32_t *__restrict__ tmp_counts,
                                                     uint32_t *__restrict__ tmp_first, int32_t *__restrict__ row_nnz)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    __shared__ uint8_t s_lut[256];
    __shared__ uint32_t s_wsum[BLK / 64];
    __shared__ uint32_t s_nvalid;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const K SENT = sentinel<K>();
    if (tid < 64)
        reinterpret_cast<uint32_t *>(s_lut)[tid] = reinterpret_cast<const uint32_t *>(lut.b)[tid];
    // nlist_ptr: the list length lives on the device (no host wait); every workgroup then owns ONE scratch slice of
    // slice_bytes (sized by the caller's bound on the longest sequence) instead of a slice per sequence
    const uint32_t nlist = nlist_ptr ? *nlist_ptr : nlist_host;

    for (uint32_t it = blockIdx.x; it < nlist; it += gridDim.x) {
        const uint32_t i = list[it];
        const int64_t b = off[i];
        const int len = slen[i];
        const uint32_t w = (uint32_t)(len - k + 1);
        uint32_t cap;
        K *keys;
        uint32_t *heads, *pos;
        uint8_t *ranks = nullptr;
        if constexpr (GLOBAL) {
            cap = 1;
            while (cap < w)
                cap <<= 1;
            uint8_t *base = g_scratch + (slice_bytes ? (int64_t)blockIdx.x * slice_bytes : g_base[it]);
            keys = reinterpret_cast<K *>(base);
            heads = reinterpret_cast<uint32_t *>(base + (size_t)cap * sizeof(K));
            pos = heads + cap + 1;
        } else {
            cap = cap_lds;
            keys = reinterpret_cast<K *>(s_dyn);
            heads = reinterpret_cast<uint32_t *>(s_dyn + (size_t)cap * sizeof(K));
            pos = heads + cap + 1;
            ranks = reinterpret_cast<uint8_t *>(pos + (WITH_POS ? cap : 0));
        }
        __syncthreads();
        if (tid == 0)
            s_nvalid = 0;
        if constexpr (!GLOBAL) {
            for (int p = tid; p < len; p += BLK)
                ranks[p] = s_lut[seq[b + p]];
        }
        __syncthreads();
        uint32_t myvalid = 0;
        for (uint32_t p = tid; p < cap; p += BLK) {
            K c = SENT;
            if (p < w) {
                if constexpr (GLOBAL) {
                    K acc = 0;
                    uint32_t bad = 0;
                    for (int j = 0; j < k; ++j) {
                        uint32_t r = s_lut[seq[b + p + j]];
                        bad |= (r == 0xFFu);
                        acc = acc * (K)nsym + (K)r;
                    }
                    c = bad ? SENT : acc;
                } else {
                    c = window_code<K>(ranks, (int)p, k, nsym);
                }
            }
            myvalid += (c != SENT);
            keys[p] = c;
        }
        atomicAdd(&s_nvalid, myvalid);
        __syncthreads();
        // bitonic sort of `cap` keys
        for (uint32_t size = 2; size <= cap; size <<= 1) {
            for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
                for (uint32_t t = tid; t < (cap >> 1); t += BLK) {
                    uint32_t lo = 2 * t - (t & (stride - 1));
                    uint32_t hi = lo + stride;
                    bool asc = (lo & size) == 0;
                    K a = keys[lo], c2 = keys[hi];
                    if ((a > c2) == asc) {
                        keys[lo] = c2;
                        keys[hi] = a;
                    }
                }
                __syncthreads();
            }
        }
        const uint32_t nvalid = s_nvalid;
        // heads[idx] = position of the idx-th distinct key
        uint32_t running = 0;
        for (uint32_t base = 0; base < nvalid; base += BLK) {
            uint32_t e = base + tid;
            bool head = e < nvalid && (e == 0 || keys[e] != keys[e - 1]);
            unsigned long long bal = __ballot(head);
            if (lane == 0)
                s_wsum[wid] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t before = 0, total = 0;
#pragma unroll
            for (int q = 0; q < BLK / 64; ++q) {
                uint32_t s = s_wsum[q];
                before += q < wid ? s : 0;
                total += s;
            }
            if (head) {
                uint32_t idx = running + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
                heads[idx] = e;
                tmp_codes[b + idx] = keys[e];
            }
            running += total;
            __syncthreads();
        }
        const uint32_t nruns = running;
        if (tid == 0) {
            heads[nruns] = nvalid;
            row_nnz[i] = (int32_t)nruns;
        }
        if constexpr (WITH_POS) {
            for (uint32_t t = tid; t < nruns; t += BLK)
                pos[t] = 0xFFFFFFFFu;
        }
        __syncthreads();
        for (uint32_t t = tid; t < nruns; t += BLK)
            tmp_counts[b + t] = heads[t + 1] - heads[t];
        if constexpr (WITH_POS) {
            for (uint32_t p = tid; p < w; p += BLK) {
                K c;
                if constexpr (GLOBAL) {
                    K acc = 0;
                    uint32_t bad = 0;
                    for (int j = 0; j < k; ++j) {
                        uint32_t r = s_lut[seq[b + p + j]];
                        bad |= (r == 0xFFu);
                        acc = acc * (K)nsym + (K)r;
                    }
                    c = bad ? SENT : acc;
                } else {
                    c = window_code<K>(ranks, (int)p, k, nsym);
                }
                if (c != SENT) {
                    uint32_t lo = 0, hi = nruns - 1;
                    while (lo < hi) {
                        uint32_t mid = (lo + hi) >> 1;
                        if (keys[heads[mid]] < c)
                            lo = mid + 1;
                        else
                            hi = mid;
                    }
                    atomicMin(&pos[lo], p);
                }
            }
            __syncthreads();
            for (uint32_t t = tid; t < nruns; t += BLK)
                tmp_first[b + t] = pos[t];
        }
    }
}

// Copy each row's entries from its padded slot (at off[i]) to the tight CSR position.
// Optional by-products of the same pass (the fused vectorize entry point asks for them, so that the basis and
// norm stages need no pass of their own over the entries): rowcount[e] = row | count << 32, the posting word of
// entry e; rnorm[i] = 1/||row i|| (1 for an all-zero row) and normsq[i] = its exact squared norm.
template <typename K>
__global__ __launch_bounds__(BLK) void k_compact_rows(const int64_t *__restrict__ off,
                                                      const int64_t *__restrict__ rowptr, int64_t n,
                                                      const K *__restrict__ tmp_codes,
                                                      const uint32_t *__restrict__ tmp_counts,
                                                      const uint32_t *__restrict__ tmp_first,
                                                      K *__restrict__ codes, uint32_t *__restrict__ counts,
                                                      uint32_t *__restrict__ first, uint64_t *__restrict__ rowcount,
                                                      float *__restrict__ rnorm, uint64_t *__restrict__ normsq,
                                                      uint32_t *__restrict__ bucket_fill, uint32_t *__restrict__ colidx_ff,
                                                      uint32_t *__restrict__ hist, int hist_passes, int hist_key_bits)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    // the stage's last kernel: every count kernel has read its list length, so the size-class counters go back to zero
    // for the next call (no fill operation in front of k_classify); and state the next stage wants cleared
    if (bucket_fill && blockIdx.x == 0 && threadIdx.x < NBUCKET)
        bucket_fill[threadIdx.x] = 0u;
    // digit histograms of the next stage's sort, counted while the codes pass through (skm_count_extras::hist)
    __shared__ uint32_t s_h[skm_onesweep::MAX_PASSES][skm_onesweep::RADIX];
    if (hist) {
        for (int z = threadIdx.x; z < hist_passes * skm_onesweep::RADIX; z += BLK)
            (&s_h[0][0])[z] = 0u;
        __syncthreads();
    }
    for (int64_t i = wave; i < n; i += nwaves) {
        const int64_t src = off[i], dst = rowptr[i];
        const int64_t cnt = rowptr[i + 1] - dst;
        unsigned long long sq = 0;
        for (int64_t t = lane; t < cnt; t += 64) {
            const uint32_t c = tmp_counts[src + t];
            codes[dst + t] = tmp_codes[src + t];
            counts[dst + t] = c;
            if (first)
                first[dst + t] = tmp_first[src + t];
            if (rowcount)
                rowcount[dst + t] = (uint64_t)(uint32_t)i | ((uint64_t)c << 32);
            if (colidx_ff)
                colidx_ff[dst + t] = 0xFFFFFFFFu;
            if (hist) {
                const K code = tmp_codes[src + t];
                for (int p = 0; p < hist_passes; ++p) {
                    const int shift = p * skm_onesweep::RADIX_BITS, bits = min(skm_onesweep::RADIX_BITS, hist_key_bits - shift);
                    atomicAdd(&s_h[p][(uint32_t)(code >> shift) & ((1u << bits) - 1u)], 1u);
                }
            }
            sq += (unsigned long long)c * c;
        }
        if (rnorm || normsq) {
            for (int o = 32; o > 0; o >>= 1)
                sq += __shfl_down(sq, o);
            if (lane == 0) {
                if (normsq)
                    normsq[i] = sq;
                if (rnorm)
                    rnorm[i] = sq ? (float)(1.0 / sqrt((double)sq)) : 1.0f;
            }
        }
    }
    if (hist) {
        __syncthreads();
        for (int z = threadIdx.x; z < hist_passes * skm_onesweep::RADIX; z += BLK) {
            const uint32_t c = (&s_h[0][0])[z];
            if (c)
                atomicAdd(hist + z, c);
        }
    }
}

struct to_i64 {
    __device__ int64_t operator()(int32_t x) const { return (int64_t)x; }
};

int make_lut(const uint8_t *h, skm_lut256 *out)
{
    if (!h) {
        skm_set_error("null lookup table");
        return SKM_E_BADARG;
    }
    memcpy(out->b, h, 256);
    return SKM_OK;
}

int check_code_space(int nsym, int k, int code_bits)
{
    if (nsym < 1 || nsym > 254 || k < 1 || k > 64 || (code_bits != 32 && code_bits != 64)) {
        skm_set_error("bad alphabet size / k / code width (nsym=%d k=%d bits=%d)", nsym, k, code_bits);
        return SKM_E_BADARG;
    }
    // nsym^k must be < 2^code_bits so that the all-ones word stays free as the sentinel
    unsigned __int128 space = 1, lim = (unsigned __int128)1 << code_bits;
    for (int j = 0; j < k; ++j) {
        space *= (unsigned)nsym;
        if (space >= lim) {
            skm_set_error("k-mer space %d^%d does not fit %d-bit codes", nsym, k, code_bits);
            return SKM_E_UNSUPPORTED;
        }
    }
    return SKM_OK;
}

// `h_nnz` == nullptr: nothing waits for the device (the caller sizes later stages by `total_residues` and reads the
// entry count from d_rowptr[n] on the device); d_rowcount / d_rnorm / d_normsq: by-products of the compaction pass.
template <typename K, bool WITH_POS>
int count_csr_impl(skm_ctx *ctx, const skm_lut256 &lut, int nsym, int k, const uint8_t *d_seq,
                   const int64_t *d_off, int64_t n, int64_t total_residues, int64_t max_seq_len, int64_t *d_rowptr, K *d_codes,
                   uint32_t *d_counts, uint32_t *d_firstpos, int64_t *h_nnz, uint64_t *d_rowcount = nullptr,
                   float *d_rnorm = nullptr, uint64_t *d_normsq = nullptr, const skm_count_extras &extras = skm_count_extras())
{
    hipStream_t st = ctx->stream;
    void *p;
    SKM_TRY(skm_ws(ctx, WS_A, sizeof(int32_t) * (size_t)(n + 1), &p));
    int32_t *slen = (int32_t *)p;
    SKM_TRY(skm_ws(ctx, WS_B, sizeof(int32_t) * (size_t)(n + 1), &p));
    int32_t *row_nnz = (int32_t *)p;
    SKM_TRY(skm_ws(ctx, WS_C, sizeof(uint32_t) * (size_t)n * NBUCKET + 64, &p));
    uint32_t *lists = (uint32_t *)p;
    SKM_TRY(skm_ws(ctx, WS_ZERO, 256, &p));
    uint32_t *fill = (uint32_t *)p;
    SKM_TRY(skm_ws(ctx, WS_D, sizeof(K) * (size_t)(total_residues + 1), &p));
    K *tmp_codes = (K *)p;
    SKM_TRY(skm_ws(ctx, WS_E, sizeof(uint32_t) * (size_t)(total_residues + 1), &p));
    uint32_t *tmp_counts = (uint32_t *)p;
    uint32_t *tmp_first = nullptr;
    if (WITH_POS) {
        SKM_TRY(skm_ws(ctx, WS_F, sizeof(uint32_t) * (size_t)(total_residues + 1), &p));
        tmp_first = (uint32_t *)p;
    }

    // zero when allocated and put back to zero by every call's last kernel (k_compact_rows); a call that failed between
    // its first and last launch left the flag set
    if (ctx->count_fill_dirty)
        SKM_HIP(hipMemsetAsync(fill, 0, sizeof(uint32_t) * 8, st));
    ctx->count_fill_dirty = true;
    {
        SKM_PROF(ctx, "k_classify");
        const int64_t max_win = max_seq_len > 0 ? (max_seq_len - k + 1 > 0 ? max_seq_len - k + 1 : 0) : -1;
        k_classify<<<(unsigned)skm_ceil_div(n, 256), 256, 0, st>>>(d_seq, d_off, n, k, slen, nullptr, lists, fill,
                                                                     row_nnz, max_win, ctx->d_err);
    }
    SKM_TRY(skm_check_launch("k_classify"));
    // Every count kernel reads its list length on the device and is launched with a grid that covers the worst case,
    // so nothing here waits for the size-class histogram - provided the caller bounds the longest sequence
    // (max_seq_len > 0: the packed batch's offsets are host data at every call site of the Python layer).  Without the
    // bound the histogram is fetched (asynchronously, awaited behind the two common-case kernels) to learn whether, and
    // for which sequences, the global-scratch kernel is needed.
    const bool bounded = max_seq_len > 0;
    uint32_t *h_fill = (uint32_t *)ctx->h_pinned;
    if (!bounded) {
        SKM_HIP(hipMemcpyAsync(h_fill, fill, sizeof(uint32_t) * NBUCKET, hipMemcpyDeviceToHost, st));
        if (!ctx->ev_host)
            SKM_REQUIRE((ctx->ev_host = skm_event_acquire(ctx->device, false)) != nullptr, SKM_E_HIP, "no event");
        SKM_HIP(hipEventRecord(ctx->ev_host, st));
    }
    {
        SKM_PROF(ctx, "k_count_short");
        int grid = skm_grid_cap(ctx, n, 64);
        k_count_short<K, WITH_POS><<<grid, 64, 0, st>>>(lut, nsym, k, d_seq, d_off, slen, lists + 1 * n, fill + 1,
                                                         tmp_codes, tmp_counts, tmp_first, row_nnz, extras.zero, extras.zero_words);
        SKM_TRY(skm_check_launch("k_count_short"));
    }
    if (!bounded || max_seq_len - k + 1 > SHORT_MAX) {
        // 513..4096 windows: one workgroup of 8 waves per sequence, persistent grid
        SKM_PROF(ctx, "k_count_long");
        auto kern = k_count_long<K, WITH_POS, MIDSEQ_MAX / 8>;
        constexpr size_t lds = count_long_lds<K, WITH_POS, MIDSEQ_MAX>();
        if (lds > 64 * 1024)  // (8-byte codes with first positions: 70 KiB)
            SKM_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kern<<<skm_grid_cap(ctx, n, 4), MIDSEQ_MAX / 8, lds, st>>>(lut, nsym, k, d_seq, d_off, slen, lists + (int64_t)2 * n, fill + 2,
                                                                    tmp_codes, tmp_counts, tmp_first, row_nnz);
        SKM_TRY(skm_check_launch("k_count_long"));
    }
    if (!bounded || max_seq_len - k + 1 > MIDSEQ_MAX) {
        // 4097..8192 windows: one workgroup of 16 waves per sequence, persistent grid (one workgroup per CU: ~100-137 KB of LDS)
        SKM_PROF(ctx, "k_count_long");
        auto kern = k_count_long<K, WITH_POS, LONGSEQ_MAX / 8>;
        constexpr size_t lds = count_long_lds<K, WITH_POS, LONGSEQ_MAX>();
        SKM_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kern<<<skm_grid_cap(ctx, n, 1), LONGSEQ_MAX / 8, lds, st>>>(lut, nsym, k, d_seq, d_off, slen, lists + (int64_t)3 * n, fill + 3,
                                                                     tmp_codes, tmp_counts, tmp_first, row_nnz);
        SKM_TRY(skm_check_launch("k_count_long"));
    }
    if (bounded && max_seq_len - k + 1 > LONGSEQ_MAX) {
        // more than 8192 windows (rare): keys in global scratch, one slice per workgroup sized by the caller's bound
        uint64_t w = (uint64_t)(max_seq_len - k + 1), cap = 1;
        while (cap < w)
            cap <<= 1;
        size_t slice = cap * sizeof(K) + sizeof(uint32_t) * (2 * cap + 2);
        slice = (slice + 255) & ~(size_t)255;
        const int grid = skm_grid_cap(ctx, n, 1);
        SKM_TRY(skm_ws(ctx, WS_G, slice * (size_t)grid, &p));
        SKM_PROF(ctx, "k_count_block_global");
        k_count_block<K, WITH_POS, true><<<grid, BLK, 16, st>>>(lut, nsym, k, d_seq, d_off, slen, lists + 6 * n, 0, fill + 6, 0,
                                                                 nullptr, (int64_t)slice, (uint8_t *)p, tmp_codes, tmp_counts,
                                                                 tmp_first, row_nnz);
        SKM_TRY(skm_check_launch("k_count_block_global"));
    }
    if (!bounded) {
        SKM_HIP(hipEventSynchronize(ctx->ev_host));
        if (h_fill[6]) {
            // Sequences with > 8192 windows and no bound from the caller: size a scratch slice per sequence on the host.
            uint32_t nl = h_fill[6];
            std::vector<uint32_t> ids(nl);
            SKM_HIP(hipMemcpyAsync(ids.data(), lists + 6 * n, sizeof(uint32_t) * nl, hipMemcpyDeviceToHost, st));
            SKM_HIP(hipStreamSynchronize(st));
            std::vector<int32_t> lens(nl);
            for (uint32_t q = 0; q < nl; ++q)
                SKM_HIP(hipMemcpyAsync(&lens[q], slen + ids[q], sizeof(int32_t), hipMemcpyDeviceToHost, st));
            SKM_HIP(hipStreamSynchronize(st));
            std::vector<int64_t> base(nl);
            size_t tot = 0;
            for (uint32_t q = 0; q < nl; ++q) {
                uint64_t w = (uint64_t)(lens[q] - k + 1), cap = 1;
                while (cap < w)
                    cap <<= 1;
                base[q] = (int64_t)tot;
                tot += cap * sizeof(K) + sizeof(uint32_t) * (2 * cap + 2);
                tot = (tot + 255) & ~(size_t)255;
            }
            SKM_TRY(skm_ws(ctx, WS_G, tot, &p));
            uint8_t *scratch = (uint8_t *)p;
            SKM_TRY(skm_ws(ctx, WS_H, sizeof(int64_t) * nl, &p));
            int64_t *d_base = (int64_t *)p;
            SKM_HIP(hipMemcpyAsync(d_base, base.data(), sizeof(int64_t) * nl, hipMemcpyHostToDevice, st));
            SKM_HIP(hipStreamSynchronize(st));
            SKM_PROF(ctx, "k_count_block_global");
            k_count_block<K, WITH_POS, true><<<nl, BLK, 16, st>>>(lut, nsym, k, d_seq, d_off, slen, lists + 6 * n, nl, nullptr, 0,
                                                                   d_base, 0, scratch, tmp_codes, tmp_counts, tmp_first, row_nnz);
            SKM_TRY(skm_check_launch("k_count_block_global"));
        }
    }

    // rowptr = exclusive scan of row_nnz (n+1 entries, the last one zero)
    {
        size_t tmp_bytes = 0;
        auto in = rocprim::make_transform_iterator(row_nnz, to_i64());
        SKM_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, in, d_rowptr, (int64_t)0, (size_t)(n + 1),
                                        rocprim::plus<int64_t>(), st));
        SKM_TRY(skm_ws(ctx, WS_SCAN, tmp_bytes, &p));
        SKM_PROF(ctx, "rocprim_scan_rowptr");
        SKM_HIP(rocprim::exclusive_scan(p, tmp_bytes, in, d_rowptr, (int64_t)0, (size_t)(n + 1),
                                        rocprim::plus<int64_t>(), st));
    }
    {
        SKM_PROF(ctx, "k_compact_rows");
        int grid = skm_grid_cap(ctx, skm_ceil_div(n, BLK / 64), 16);
        k_compact_rows<K><<<grid, BLK, 0, st>>>(d_off, d_rowptr, n, tmp_codes, tmp_counts, tmp_first, d_codes, d_counts,
                                                 WITH_POS ? d_firstpos : nullptr, d_rowcount, d_rnorm, d_normsq, fill,
                                                 extras.colidx_ff, extras.hist ? extras.zero : nullptr, extras.hist_passes,
                                                 extras.hist_key_bits);
    }
    SKM_TRY(skm_check_launch("k_compact_rows"));
    ctx->count_fill_dirty = false;
    if (h_nnz) {
        int64_t *h_n = (int64_t *)ctx->h_pinned;
        SKM_HIP(hipMemcpyAsync(h_n, d_rowptr + n, sizeof(int64_t), hipMemcpyDeviceToHost, st));
        SKM_HIP(hipStreamSynchronize(st));
        *h_nnz = *h_n;
        SKM_TRY(skm_check_device_error(ctx, "skm_count_csr"));
    }
    return SKM_OK;
}

}  // namespace

// =============================================================================== C ABI
extern "C" int skm_recode(skm_ctx *ctx, const uint8_t *h_translate, const uint8_t *d_seq, const int64_t *d_off,
                          int64_t n, uint8_t *d_out, int32_t *d_outlen)
{
    SKM_REQUIRE(ctx && d_off && d_outlen && n >= 0, SKM_E_BADARG, "skm_recode: bad argument");
    skm_lut256 lut;
    SKM_TRY(make_lut(h_translate, &lut));
    if (n == 0)
        return SKM_OK;
    SKM_HIP(hipSetDevice(ctx->device));
    int64_t *h_tot = (int64_t *)ctx->h_pinned;
    SKM_HIP(hipMemcpyAsync(h_tot, d_off + n, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    SKM_HIP(hipStreamSynchronize(ctx->stream));
    int64_t total = *h_tot;
    SKM_REQUIRE(total == 0 || (d_seq && d_out), SKM_E_BADARG, "skm_recode: null sequence buffer");
    SKM_REQUIRE(((uintptr_t)d_seq & 15) == 0 && ((uintptr_t)d_out & 15) == 0, SKM_E_BADARG,
                "skm_recode: buffers must be 16-byte aligned");
    if (total) {
        SKM_PROF(ctx, "k_recode_bytes");
        int grid = skm_grid_cap(ctx, skm_ceil_div(total, 256 * 16), 8);
        k_recode_bytes<<<grid, 256, 0, ctx->stream>>>(lut, d_seq, d_out, total);
        SKM_TRY(skm_check_launch("k_recode_bytes"));
    }
    {
        SKM_PROF(ctx, "k_classify");
        k_classify<<<(unsigned)skm_ceil_div(n, 256), 256, 0, ctx->stream>>>(d_seq, d_off, n, 1, d_outlen, nullptr,
                                                                             nullptr, nullptr, nullptr);
    }
    return skm_check_launch("k_classify");
}

extern "C" int skm_kmer_codes(skm_ctx *ctx, const uint8_t *h_rank, int nsym, int k, int code_bits,
                              const uint8_t *d_seq, const int64_t *d_off, int64_t n, void *d_codes,
                              int32_t *d_nwin)
{
    SKM_REQUIRE(ctx && d_off && d_nwin && d_codes && n >= 0, SKM_E_BADARG, "skm_kmer_codes: bad argument");
    skm_lut256 lut;
    SKM_TRY(make_lut(h_rank, &lut));
    SKM_TRY(check_code_space(nsym, k, code_bits));
    if (n == 0)
        return SKM_OK;
    SKM_HIP(hipSetDevice(ctx->device));
    void *p;
    SKM_TRY(skm_ws(ctx, WS_A, sizeof(int32_t) * (size_t)(n + 1), &p));
    int32_t *slen = (int32_t *)p;
    {
        SKM_PROF(ctx, "k_classify");
        k_classify<<<(unsigned)skm_ceil_div(n, 256), 256, 0, ctx->stream>>>(d_seq, d_off, n, k, slen, d_nwin, nullptr,
                                                                             nullptr, nullptr);
    }
    SKM_TRY(skm_check_launch("k_classify"));
    int grid = skm_grid_cap(ctx, n, 64);
    SKM_PROF(ctx, "k_kmer_codes");
    if (code_bits == 32)
        k_kmer_codes<uint32_t><<<grid, 64, 0, ctx->stream>>>(lut, nsym, k, d_seq, d_off, n, slen, (uint32_t *)d_codes);
    else
        k_kmer_codes<uint64_t><<<grid, 64, 0, ctx->stream>>>(lut, nsym, k, d_seq, d_off, n, slen, (uint64_t *)d_codes);
    return skm_check_launch("k_kmer_codes");
}

extern "C" int skm_count_csr(skm_ctx *ctx, const uint8_t *h_rank, int nsym, int k, int code_bits,
                             const uint8_t *d_seq, const int64_t *d_off, int64_t n, int64_t total_residues,
                             int64_t max_seq_len, int64_t cap_entries, int64_t *d_rowptr, void *d_codes, uint32_t *d_counts,
                             uint32_t *d_firstpos, int64_t *h_nnz)
{
    SKM_REQUIRE(ctx && d_off && d_rowptr && d_codes && d_counts && h_nnz && n >= 0, SKM_E_BADARG,
                "skm_count_csr: bad argument");
    SKM_REQUIRE(cap_entries >= total_residues, SKM_E_BADARG,
                "skm_count_csr: cap_entries (%lld) must be >= total residues (%lld)", (long long)cap_entries,
                (long long)total_residues);
    SKM_REQUIRE(total_residues < ((int64_t)1 << 32), SKM_E_OVERFLOW,
                "skm_count_csr: more than 2^32 residues in one batch; split the batch");
    skm_lut256 lut;
    SKM_TRY(make_lut(h_rank, &lut));
    SKM_TRY(check_code_space(nsym, k, code_bits));
    SKM_HIP(hipSetDevice(ctx->device));
    *h_nnz = 0;
    if (n == 0) {
        SKM_HIP(hipMemsetAsync(d_rowptr, 0, sizeof(int64_t), ctx->stream));
        return SKM_OK;
    }
    if (code_bits == 32) {
        if (d_firstpos)
            return count_csr_impl<uint32_t, true>(ctx, lut, nsym, k, d_seq, d_off, n, total_residues, max_seq_len, d_rowptr,
                                                  (uint32_t *)d_codes, d_counts, d_firstpos, h_nnz);
        return count_csr_impl<uint32_t, false>(ctx, lut, nsym, k, d_seq, d_off, n, total_residues, max_seq_len, d_rowptr,
                                               (uint32_t *)d_codes, d_counts, nullptr, h_nnz);
    }
    if (d_firstpos)
        return count_csr_impl<uint64_t, true>(ctx, lut, nsym, k, d_seq, d_off, n, total_residues, max_seq_len, d_rowptr,
                                              (uint64_t *)d_codes, d_counts, d_firstpos, h_nnz);
    return count_csr_impl<uint64_t, false>(ctx, lut, nsym, k, d_seq, d_off, n, total_residues, max_seq_len, d_rowptr,
                                           (uint64_t *)d_codes, d_counts, nullptr, h_nnz);
}

// Count stage of skm_vectorize_csr: the same kernels as skm_count_csr, nothing waits for the device, d_codes is
// pre-filled with the sentinel so that everything past the (device-side) entry count sorts last, and the
// compaction pass also emits the posting words and the row norms.
int skm_count_stage_async(skm_ctx *ctx, const uint8_t *h_rank, int nsym, int k, int code_bits, const uint8_t *d_seq,
                          const int64_t *d_off, int64_t n, int64_t total_residues, int64_t max_seq_len, int64_t *d_rowptr, void *d_codes,
                          uint32_t *d_counts, uint64_t *d_rowcount, float *d_rnorm, uint64_t *d_normsq,
                          const skm_count_extras &extras)
{
    skm_lut256 lut;
    SKM_TRY(make_lut(h_rank, &lut));
    SKM_TRY(check_code_space(nsym, k, code_bits));
    if (d_rowcount && !skm_use_onesweep(total_residues))  // the sentinel tail is for rocPRIM's capacity-sized sort
        SKM_HIP(hipMemsetAsync(d_codes, 0xFF, (size_t)(code_bits / 8) * (size_t)(total_residues + 1), ctx->stream));
    if (code_bits == 32)
        return count_csr_impl<uint32_t, false>(ctx, lut, nsym, k, d_seq, d_off, n, total_residues, max_seq_len, d_rowptr, (uint32_t *)d_codes,
                                               d_counts, nullptr, nullptr, d_rowcount, d_rnorm, d_normsq, extras);
    return count_csr_impl<uint64_t, false>(ctx, lut, nsym, k, d_seq, d_off, n, total_residues, max_seq_len, d_rowptr, (uint64_t *)d_codes,
                                           d_counts, nullptr, nullptr, d_rowcount, d_rnorm, d_normsq, extras);
}
