// Microbenchmark: LDS hash-accumulate primitives on gfx950 (lane-ops per clock per CU).
//   mode 0: ds_add_u32 (no return) at random slots     mode 1: read key + ds_add_u32
//   mode 2: ds_add_rtn_u64                              mode 3: plain read-modify-write (not atomic)
//   mode 4: ds_add_u32, all 4 lane groups hit the same 16 slots (4-way same-address conflicts)
// build: hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_bench.hip -o tools/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, uint32_t *out)
{
    __shared__ uint32_t keys[2048];
    __shared__ unsigned long long vals[2048];
    for (int z = threadIdx.x; z < 2048; z += 256) { keys[z] = z; vals[z] = 0; }
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            x = x * 1664525u + 1013904223u;
            uint32_t h = (x >> 11) & 2047u;
            if (MODE == 4) h = ((x >> 11) & 2032u & 0u) + (threadIdx.x & 15) * 37u + (it & 63) * 16u;
            h &= 2047u;
            if (MODE == 0 || MODE == 4) atomicAdd(reinterpret_cast<uint32_t *>(&vals[h]), x & 7u);
            if (MODE == 1) { uint32_t kk = __atomic_load_n(&keys[h], __ATOMIC_RELAXED); if (kk == h) atomicAdd(reinterpret_cast<uint32_t *>(&vals[h]), x & 7u); }
            if (MODE == 2) acc += (uint32_t)(atomicAdd(&vals[h], (unsigned long long)(x & 7u)) >> 32);
            if (MODE == 3) { volatile uint32_t *p = reinterpret_cast<uint32_t *>(&vals[h]); *p = *p + (x & 7u); }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (uint32_t)vals[5] + acc;
}
template <int MODE> void run(const char *name, uint32_t *d)
{
    const int grid = 256 * 7, iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<grid, 256>>>(10, d);
    hipEventRecord(a);
    k<MODE><<<grid, 256>>>(iters, d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = (double)grid * 256 * iters * 4;
    printf("%-34s %8.3f ms  %7.1f G lane-ops/s  %.2f lane-ops/clk/CU (2.4 GHz, 256 CU)\n", name, ms, ops / ms * 1e-6,
           ops / (ms * 1e-3) / 2.4e9 / 256);
}
int main()
{
    uint32_t *d; hipMalloc(&d, 4 * 256 * 7);
    run<0>("ds_add_u32 random", d);
    run<1>("read key + ds_add_u32 random", d);
    run<2>("ds_add_rtn_u64 random", d);
    run<3>("plain rmw random", d);
    run<4>("ds_add_u32 4-way same address", d);
    return 0;
}
