// Shared rocPRIM sort wrapper of the basis translation units (included inside no namespace; the
// functions are static).
#pragma once
#include <cstring>  // rocPRIM headers use memset without including it

#include <rocprim/rocprim.hpp>

#include "skm_common.h"

// stable sort of (key, position) pairs: the payload is the entry's index before the sort
// rocPRIM ships no gfx950 tuning for Onesweep.  Measured on MI355X for 2.9e7 pairs with a 4-byte
// payload (tools/sort_tune.hip): 4-byte keys, 1024 threads x 8 items, 8-bit digits: 0.78 ms (generic
// default 0.91); 8-byte keys of 34 bits: 9-bit digits need 4 passes instead of 5: 1.06 ms (1024 x 8)
// against 1.32 ms (8-bit digits, 1024 x 6) and 1.45 ms (default).  9-bit digits are used whenever they
// save a pass; 11-bit digits (3 passes for 32 bits) are slower than 4 passes of 8.
template <typename K, unsigned R>
struct sort_config;
template <unsigned R>
struct sort_config<uint32_t, R> {
    using type = rocprim::radix_sort_config<
        rocprim::default_config, rocprim::default_config,
        rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 8>, R,
                                            rocprim::block_radix_rank_algorithm::match>>;
};
template <>
struct sort_config<uint64_t, 8> {
    using type = rocprim::radix_sort_config<
        rocprim::default_config, rocprim::default_config,
        rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 6>, rocprim::kernel_config<1024, 6>, 8,
                                            rocprim::block_radix_rank_algorithm::match>>;
};
template <>
struct sort_config<uint64_t, 9> {
    using type = rocprim::radix_sort_config<
        rocprim::default_config, rocprim::default_config,
        rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 8>, 9,
                                            rocprim::block_radix_rank_algorithm::match>>;
};

template <typename K, unsigned R>
static int sort_pairs_r(skm_ctx *ctx, const K *kin, K *kout, uint32_t *vout, int64_t nnz, int bits, const char *label)
{
    using config = typename sort_config<K, R>::type;
    const rocprim::counting_iterator<uint32_t> vin(0);
    size_t tmp = 0;
    SKM_HIP(rocprim::radix_sort_pairs<config>(nullptr, tmp, kin, kout, vin, vout, (size_t)nnz, 0u, (unsigned)bits,
                                              ctx->stream));
    void *p;
    SKM_TRY(skm_ws(ctx, WS_ROCPRIM, tmp, &p));
    SKM_PROF(ctx, label);
    SKM_HIP(rocprim::radix_sort_pairs<config>(p, tmp, kin, kout, vin, vout, (size_t)nnz, 0u, (unsigned)bits, ctx->stream));
    return SKM_OK;
}

template <typename K>
static int sort_pairs(skm_ctx *ctx, const K *kin, K *kout, uint32_t *vout, int64_t nnz, int bits, const char *label)
{
    if ((bits + 8) / 9 < (bits + 7) / 8)  // 9-bit digits save a pass
        return sort_pairs_r<K, 9>(ctx, kin, kout, vout, nnz, bits, label);
    return sort_pairs_r<K, 8>(ctx, kin, kout, vout, nnz, bits, label);
}
