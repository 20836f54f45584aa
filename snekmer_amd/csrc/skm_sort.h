// Shared rocPRIM sort wrapper of the basis translation units (included inside no namespace; the
// functions are static).
#pragma once
#include <cstring>  // rocPRIM headers use memset without including it

#include <rocprim/rocprim.hpp>

#include "skm_common.h"

// stable sort of (key, position) pairs: the payload is the entry's index before the sort
// rocPRIM ships no gfx950 tuning for Onesweep; for 4-byte keys with a 4-byte payload 1024 threads x 8
// items beat its generic default on MI355X (0.80 vs 0.92 ms for 2.9e7 pairs, tools/sort_tune.hip).
template <typename K>
struct sort_config {
    using type = rocprim::default_config;
};
template <>
struct sort_config<uint32_t> {
    using type = rocprim::radix_sort_config<
        rocprim::default_config, rocprim::default_config,
        rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 8>, rocprim::kernel_config<1024, 8>, 8,
                                            rocprim::block_radix_rank_algorithm::match>>;
};

template <typename K>
static int sort_pairs(skm_ctx *ctx, const K *kin, K *kout, uint32_t *vout, int64_t nnz, int bits, const char *label)
{
    using config = typename sort_config<K>::type;
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

