// Micro-benchmark: rocPRIM radix_sort_pairs (u32 key, u32 position payload) configurations on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
#include <rocprim/rocprim.hpp>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class Config>
int run(const char *name, const uint32_t *kin, uint32_t *kout, uint32_t *vout, size_t n, unsigned bits)
{
    const rocprim::counting_iterator<uint32_t> vin(0);
    size_t tmp = 0;
    CK((rocprim::radix_sort_pairs<Config>(nullptr, tmp, kin, kout, vin, vout, n, 0u, bits, 0)));
    void *p;
    CK(hipMalloc(&p, tmp));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i)
        CK((rocprim::radix_sort_pairs<Config>(p, tmp, kin, kout, vin, vout, n, 0u, bits, 0)));
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < 10; ++i)
        CK((rocprim::radix_sort_pairs<Config>(p, tmp, kin, kout, vin, vout, n, 0u, bits, 0)));
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-28s %.3f ms\n", name, ms / 10);
    CK(hipFree(p));
    return 0;
}

template <unsigned T, unsigned I, unsigned R = 8>
using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<T, I>, rocprim::kernel_config<T, I>, R,
                                        rocprim::block_radix_rank_algorithm::match>>;

// second table: 8-byte keys of 34 bits (standard k=12): 9-bit digits save one of five passes
template <class Config>
int run64(const char *name, const unsigned long long *kin, unsigned long long *kout, uint32_t *vout, size_t n, unsigned bits)
{
    const rocprim::counting_iterator<uint32_t> vin(0);
    size_t tmp = 0;
    CK((rocprim::radix_sort_pairs<Config>(nullptr, tmp, kin, kout, vin, vout, n, 0u, bits, 0)));
    void *p;
    CK(hipMalloc(&p, tmp));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i)
        CK((rocprim::radix_sort_pairs<Config>(p, tmp, kin, kout, vin, vout, n, 0u, bits, 0)));
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < 10; ++i)
        CK((rocprim::radix_sort_pairs<Config>(p, tmp, kin, kout, vin, vout, n, 0u, bits, 0)));
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("u64/34 bits %-18s %.3f ms\n", name, ms / 10);
    CK(hipFree(p));
    return 0;
}

int main64()
{
    const size_t n = 28888366;
    std::vector<unsigned long long> h(n);
    std::mt19937_64 g(1);
    for (auto &x : h) x = g() % 13841287201ull;  // 7^12
    unsigned long long *kin, *kout; uint32_t *vout;
    CK(hipMalloc(&kin, 8 * n)); CK(hipMalloc(&kout, 8 * n)); CK(hipMalloc(&vout, 4 * n));
    CK(hipMemcpy(kin, h.data(), 8 * n, hipMemcpyHostToDevice));
    run64<rocprim::default_config>("default", kin, kout, vout, n, 34);
    run64<cfg<1024, 6>>("1024x6 r8", kin, kout, vout, n, 34);
    run64<cfg<1024, 8>>("1024x8 r8", kin, kout, vout, n, 34);
    run64<cfg<1024, 8, 9>>("1024x8 r9", kin, kout, vout, n, 34);
    run64<cfg<1024, 6, 9>>("1024x6 r9", kin, kout, vout, n, 34);
    run64<cfg<512, 16, 9>>("512x16 r9", kin, kout, vout, n, 34);
    CK(hipFree(kin)); CK(hipFree(kout)); CK(hipFree(vout));
    return 0;
}

int main()
{
    const size_t n = 28888318;
    std::vector<uint32_t> h(n);
    std::mt19937_64 g(1);
    for (auto &x : h) x = (uint32_t)(g() % 2176782336ull);
    uint32_t *kin, *kout, *vout;
    CK(hipMalloc(&kin, 4 * n)); CK(hipMalloc(&kout, 4 * n)); CK(hipMalloc(&vout, 4 * n));
    CK(hipMemcpy(kin, h.data(), 4 * n, hipMemcpyHostToDevice));
    run<rocprim::default_config>("default", kin, kout, vout, n, 32);
    run<cfg<1024, 8>>("1024x8", kin, kout, vout, n, 32);
    run<cfg<1024, 6>>("1024x6", kin, kout, vout, n, 32);
    run<cfg<1024, 12>>("1024x12", kin, kout, vout, n, 32);
    run<cfg<512, 16>>("512x16", kin, kout, vout, n, 32);
    run<cfg<512, 12>>("512x12", kin, kout, vout, n, 32);
    run<cfg<512, 8>>("512x8", kin, kout, vout, n, 32);
    run<cfg<256, 16>>("256x16", kin, kout, vout, n, 32);
    run<cfg<256, 12>>("256x12", kin, kout, vout, n, 32);
    run<cfg<512, 22>>("512x22", kin, kout, vout, n, 32);
    run<cfg<1024, 8, 7>>("1024x8 r7", kin, kout, vout, n, 32);
    run<cfg<512, 16, 6>>("512x16 r6", kin, kout, vout, n, 32);
    run<cfg<1024, 8, 11>>("1024x8 r11", kin, kout, vout, n, 32);
    run<cfg<1024, 16, 11>>("1024x16 r11", kin, kout, vout, n, 32);
    return main64();
}
