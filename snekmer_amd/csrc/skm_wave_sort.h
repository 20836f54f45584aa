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

// Sort 512 keys laid out as element e = r*64 + lane, ascending in e.
template <typename K>
__device__ __forceinline__ void wave_bitonic_512(K (&v)[8], int lane)
{
#pragma unroll
    for (int size = 2; size <= 512; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 64) {
                const int rs = stride >> 6;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if ((r & rs) == 0) {
                        const bool asc = ((r << 6) & size) == 0;  // size >= 128 here: depends on r only
                        K a = v[r], b = v[r | rs];
                        bool sw = asc ? (a > b) : (a < b);
                        v[r] = sw ? b : a;
                        v[r | rs] = sw ? a : b;
                    }
                }
            } else {
                const bool lower = (lane & stride) == 0;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const bool asc = size >= 64 ? (((r << 6) & size) == 0) : ((lane & size) == 0);
                    K other = shfl_xor_k<K>(v[r], stride);
                    K mn = v[r] < other ? v[r] : other;
                    K mx = v[r] < other ? other : v[r];
                    v[r] = (lower == asc) ? mn : mx;
                }
            }
        }
    }
}

