// Wave-level (64-lane) sorting helpers shared by the count and the sparse-Gram kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

template <typename K>
__device__ __forceinline__ K shfl_xor_k(K v, int m)
{
    if constexpr (sizeof(K) == 8) {
        uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
        lo = __shfl_xor(lo, m);
        hi = __shfl_xor(hi, m);
        return ((K)hi << 32) | lo;
    } else {
        return (K)__shfl_xor((uint32_t)v, m);
    }
}

template <typename K>
__device__ __forceinline__ K shfl_idx_k(K v, int src)
{
    if constexpr (sizeof(K) == 8) {
        uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
        lo = __shfl(lo, src);
        hi = __shfl(hi, src);
        return ((K)hi << 32) | lo;
    } else {
        return (K)__shfl((uint32_t)v, src);
    }
}

// x[lane ^ M] for a compile-time M.  Masks a DPP control can express (quad permutes, half-row and
// row mirrors, and pairs of them) stay in the VALU; the rest go through ds_bpermute.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov_u32(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true);
}

template <int M>
__device__ __forceinline__ uint32_t lane_xor_u32(uint32_t x)
{
    constexpr int QUAD_X1 = 0xB1, QUAD_X2 = 0x4E, QUAD_X3 = 0x1B, ROW_MIRROR = 0x140, HALF_MIRROR = 0x141;
    if constexpr (M == 1)
        return dpp_mov_u32<QUAD_X1>(x);
    else if constexpr (M == 2)
        return dpp_mov_u32<QUAD_X2>(x);
    else if constexpr (M == 3)
        return dpp_mov_u32<QUAD_X3>(x);
    else if constexpr (M == 7)
        return dpp_mov_u32<HALF_MIRROR>(x);  // i -> 7 - i within 8 lanes == i ^ 7
    else if constexpr (M == 15)
        return dpp_mov_u32<ROW_MIRROR>(x);  // i -> 15 - i within 16 lanes == i ^ 15
    else if constexpr (M == 4)
        return dpp_mov_u32<QUAD_X3>(dpp_mov_u32<HALF_MIRROR>(x));  // 7 ^ 3
    else if constexpr (M == 8)
        return dpp_mov_u32<HALF_MIRROR>(dpp_mov_u32<ROW_MIRROR>(x));  // 15 ^ 7
    else
        return (uint32_t)__shfl_xor(x, M);
}

template <int M, typename K>
__device__ __forceinline__ K lane_xor_k(K v)
{
    if constexpr (sizeof(K) == 8) {
        const uint32_t lo = lane_xor_u32<M>((uint32_t)v), hi = lane_xor_u32<M>((uint32_t)(v >> 32));
        return ((K)hi << 32) | lo;
    } else {
        return (K)lane_xor_u32<M>((uint32_t)v);
    }
}

// One cross-lane stage of the blocked network: partner lane = lane ^ M; FLIP also mirrors the
// register index (partner element = e ^ (size - 1)).  The lower lane of a pair keeps the minima.
template <int M, bool FLIP, typename K>
__device__ __forceinline__ void wave_stage(K (&v)[8], int lane)
{
    constexpr int TOP = (M + 1) / 2 > 0 && ((M + 1) & M) == 0 ? (M + 1) / 2 : M;  // highest set bit of M
    const bool lower = (lane & TOP) == 0;
    K nv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const K other = lane_xor_k<M, K>(v[FLIP ? 7 - r : r]);
        const K mn = v[r] < other ? v[r] : other;
        const K mx = v[r] < other ? other : v[r];
        nv[r] = lower ? mn : mx;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r)
        v[r] = nv[r];
}

template <typename K>
__device__ __forceinline__ void reg_cas(K &a, K &b)
{
    const K mn = a < b ? a : b, mx = a < b ? b : a;
    a = mn;
    b = mx;
}

// The three in-register stages (strides 4, 2, 1) that end every merge of 8 or more elements.
template <typename K>
__device__ __forceinline__ void reg_tail(K (&v)[8])
{
    reg_cas(v[0], v[4]), reg_cas(v[1], v[5]), reg_cas(v[2], v[6]), reg_cas(v[3], v[7]);
    reg_cas(v[0], v[2]), reg_cas(v[1], v[3]), reg_cas(v[4], v[6]), reg_cas(v[5], v[7]);
    reg_cas(v[0], v[1]), reg_cas(v[2], v[3]), reg_cas(v[4], v[5]), reg_cas(v[6], v[7]);
}

// Sort 512 keys laid out BLOCKED, element e = lane*8 + r, ascending in e.  Bitonic network in
// its "flip" form (the first stage of a merge pairs e with e ^ (size-1), all later stages pair e
// with e ^ stride), so every compare-exchange is ascending.  24 of the 45 stages stay inside a
// lane; 18 of the 21 cross-lane stages are DPP moves, 3 use ds_bpermute.
template <typename K>
__device__ __forceinline__ void wave_bitonic_512_blocked(K (&v)[8], int lane)
{
    // size 2, 4, 8: inside the lane
    reg_cas(v[0], v[1]), reg_cas(v[2], v[3]), reg_cas(v[4], v[5]), reg_cas(v[6], v[7]);
    reg_cas(v[0], v[3]), reg_cas(v[1], v[2]), reg_cas(v[4], v[7]), reg_cas(v[5], v[6]);
    reg_cas(v[0], v[1]), reg_cas(v[2], v[3]), reg_cas(v[4], v[5]), reg_cas(v[6], v[7]);
    reg_cas(v[0], v[7]), reg_cas(v[1], v[6]), reg_cas(v[2], v[5]), reg_cas(v[3], v[4]);
    reg_cas(v[0], v[2]), reg_cas(v[1], v[3]), reg_cas(v[4], v[6]), reg_cas(v[5], v[7]);
    reg_cas(v[0], v[1]), reg_cas(v[2], v[3]), reg_cas(v[4], v[5]), reg_cas(v[6], v[7]);
    // size 16
    wave_stage<1, true>(v, lane);
    reg_tail(v);
    // size 32
    wave_stage<3, true>(v, lane);
    wave_stage<1, false>(v, lane);
    reg_tail(v);
    // size 64
    wave_stage<7, true>(v, lane);
    wave_stage<2, false>(v, lane);
    wave_stage<1, false>(v, lane);
    reg_tail(v);
    // size 128
    wave_stage<15, true>(v, lane);
    wave_stage<4, false>(v, lane);
    wave_stage<2, false>(v, lane);
    wave_stage<1, false>(v, lane);
    reg_tail(v);
    // size 256
    wave_stage<31, true>(v, lane);
    wave_stage<8, false>(v, lane);
    wave_stage<4, false>(v, lane);
    wave_stage<2, false>(v, lane);
    wave_stage<1, false>(v, lane);
    reg_tail(v);
    // size 512
    wave_stage<63, true>(v, lane);
    wave_stage<16, false>(v, lane);
    wave_stage<8, false>(v, lane);
    wave_stage<4, false>(v, lane);
    wave_stage<2, false>(v, lane);
    wave_stage<1, false>(v, lane);
    reg_tail(v);
}
