// Stage 3 of the hot path: all-pairs cosine over sparse count rows.
//
// Reference behaviour being replaced:
//   sklearn.metrics.pairwise.cosine_similarity(X, Y)     snekmer/rules/apply.smk:282-284,
//                                                        snekmer/rules/learn.smk:821-823,
//                                                        snekmer/rules/evaluate.smk:434-436
//   pairwise_distances(X, metric="cosine")               snekmer/score.py:169-171
//
// Why not a dense GEMM here: at k = 12 the observed basis has ~200 columns per sequence
// (SURVEY.md D2/H1); the dense N x B operand would be terabytes and >99.9 % of the N x N Gram
// entries are zero.  The output matrix itself (N*M float32) is the irreducible traffic, so the
// kernel is organised as a streaming writer that is bound by HBM write bandwidth:
//
//   - a workgroup owns a strip of R = 8 output rows and walks the M output columns in chunks of
//     CH = 1024; the chunk's exact integer dot products are accumulated in LDS (int32, 32 KiB);
//   - every non-zero (row i, column c, count v) of the strip is a "task" holding a cursor into
//     the posting list of c (rows of Y containing c, ascending).  Because posting lists are
//     sorted, a task contributes to chunk [j0, j1) exactly the entries its cursor passes while
//     the posting's row is < j1: LDS atomic add of v * v', cursor state in registers;
//   - the epilogue converts the chunk to float32, scales by 1/|x_i| * 1/|y_j| and streams it out
//     with 16-byte stores (each wave store instruction writes 1 KiB of one output row).
//
// The dense small-basis case (a true dense GEMM) is served by the i8 MFMA kernel in
// skm_dense.hip instead.
#include "skm_common.h"

namespace {

constexpr int R = 8;
constexpr int CH = 1024;
constexpr int TB = 256;
constexpr int Q = 10;  // register-resident tasks per thread -> Q*TB = 2560 tasks per strip
constexpr uint32_t NONE = 0xFFFFFFFFu;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool VEC>
__global__ __launch_bounds__(TB) void k_cosine_strip(const int64_t *__restrict__ xrowptr,
                                                     const uint32_t *__restrict__ xcolidx,
                                                     const uint32_t *__restrict__ xcounts,
                                                     const float *__restrict__ xrnorm, int64_t m,
                                                     const uint32_t *__restrict__ ycolptr,
                                                     const uint32_t *__restrict__ yprow,
                                                     const uint32_t *__restrict__ ypval,
                                                     const float *__restrict__ yrnorm, int64_t row0, int64_t row1,
                                                     float *__restrict__ out, int64_t ld)
{
    __shared__ __attribute__((aligned(16))) int s_acc[R][CH];
    __shared__ int64_t s_rp[R + 1];
    __shared__ float s_rni[R];
    const int tid = threadIdx.x;
    const int64_t i0 = row0 + (int64_t)blockIdx.x * R;
    const int rows = (int)min((int64_t)R, row1 - i0);

    if (tid <= R) {
        int64_t r = tid <= rows ? tid : rows;
        s_rp[tid] = xrowptr[i0 + r];
    }
    if (tid < R)
        s_rni[tid] = tid < rows ? xrnorm[i0 + tid] : 0.0f;
    for (int z = tid; z < R * CH / 4; z += TB)
        reinterpret_cast<int4 *>(&s_acc[0][0])[z] = make_int4(0, 0, 0, 0);
    __syncthreads();

    const int64_t e0 = s_rp[0];
    const int64_t ntasks = s_rp[R] - e0;

    uint32_t cur[Q], rem[Q], nj[Q], nv[Q], liv[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int64_t t = (int64_t)tid + (int64_t)q * TB;
        nj[q] = NONE;
        cur[q] = rem[q] = nv[q] = liv[q] = 0;
        if (t < ntasks) {
            const int64_t e = e0 + t;
            int li = 0;
#pragma unroll
            for (int r = 1; r < R; ++r)
                li += (e >= s_rp[r]) ? 1 : 0;
            const uint32_t c = xcolidx[e];
            const uint32_t v = xcounts[e];
            const uint32_t pb = ycolptr[c], pe = ycolptr[c + 1];
            cur[q] = pb;
            rem[q] = pe - pb;
            liv[q] = ((uint32_t)li << 28) | (v & 0x0FFFFFFFu);
            if (pe > pb) {
                nj[q] = yprow[pb];
                nv[q] = ypval[pb];
            }
        }
    }

    for (int64_t j0 = 0; j0 < m; j0 += CH) {
        const int64_t j1 = min(j0 + (int64_t)CH, m);
        const uint32_t j1u = (uint32_t)j1, j0u = (uint32_t)j0;
        // ---- accumulate: advance every cursor through [j0, j1)
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            while (nj[q] < j1u) {
                const int li = (int)(liv[q] >> 28);
                const int v = (int)(liv[q] & 0x0FFFFFFFu);
                atomicAdd(&s_acc[li][nj[q] - j0u], v * (int)nv[q]);
                ++cur[q];
                if (--rem[q]) {
                    nj[q] = yprow[cur[q]];
                    nv[q] = ypval[cur[q]];
                } else {
                    nj[q] = NONE;
                }
            }
        }
        // ---- strips with more than Q*TB non-zeros (very long sequences): stateless tasks
        for (int64_t t = (int64_t)Q * TB + tid; t < ntasks; t += TB) {
            const int64_t e = e0 + t;
            int li = 0;
#pragma unroll
            for (int r = 1; r < R; ++r)
                li += (e >= s_rp[r]) ? 1 : 0;
            const uint32_t c = xcolidx[e];
            const int v = (int)xcounts[e];
            uint32_t lo = ycolptr[c], hi = ycolptr[c + 1];
            const uint32_t pe = hi;
            while (lo < hi) {
                uint32_t mid = lo + ((hi - lo) >> 1);
                if (yprow[mid] < j0u)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            for (; lo < pe; ++lo) {
                const uint32_t j = yprow[lo];
                if (j >= j1u)
                    break;
                atomicAdd(&s_acc[li][j - j0u], v * (int)ypval[lo]);
            }
        }
        __syncthreads();
        // ---- epilogue: scale, store, clear
        const int64_t jc = j0 + 4 * tid;
        float rj[4] = {0.f, 0.f, 0.f, 0.f};
        if (VEC && jc + 3 < m) {
            const float4 t4 = *reinterpret_cast<const float4 *>(yrnorm + jc);
            rj[0] = t4.x, rj[1] = t4.y, rj[2] = t4.z, rj[3] = t4.w;
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (jc + u < m)
                    rj[u] = yrnorm[jc + u];
        }
#pragma unroll
        for (int li = 0; li < R; ++li) {
            int4 a = *reinterpret_cast<int4 *>(&s_acc[li][4 * tid]);
            *reinterpret_cast<int4 *>(&s_acc[li][4 * tid]) = make_int4(0, 0, 0, 0);
            if (li < rows) {
                const float ri = s_rni[li];
                float o[4] = {(float)a.x * ri * rj[0], (float)a.y * ri * rj[1], (float)a.z * ri * rj[2],
                              (float)a.w * ri * rj[3]};
                const int64_t i = i0 + li;
                if (MODE == 1) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        float d = 1.0f - o[u];
                        d = fminf(fmaxf(d, 0.0f), 2.0f);
                        o[u] = (jc + u == i) ? 0.0f : d;
                    }
                }
                float *dst = out + (i - row0) * ld + jc;
                if (VEC && jc + 3 < m) {
                    f32x4 pack = {o[0], o[1], o[2], o[3]};
                    __builtin_nontemporal_store(pack, reinterpret_cast<f32x4 *>(dst));
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (jc + u < m)
                            dst[u] = o[u];
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int skm_cosine_csr(skm_ctx *ctx, int64_t n, const int64_t *d_xrowptr, const uint32_t *d_xcolidx,
                              const uint32_t *d_xcounts, const float *d_xrnorm, int64_t m, int64_t ncols,
                              const uint32_t *d_ycolptr, const uint32_t *d_yprow, const uint32_t *d_ypval,
                              const float *d_yrnorm, int64_t row0, int64_t row1, int mode, float *d_out, int64_t ld)
{
    SKM_REQUIRE(ctx && n >= 0 && m >= 0 && ncols >= 0, SKM_E_BADARG, "skm_cosine_csr: bad argument");
    SKM_REQUIRE(row0 >= 0 && row0 <= row1 && row1 <= n, SKM_E_BADARG, "skm_cosine_csr: bad row range [%lld,%lld) of %lld",
                (long long)row0, (long long)row1, (long long)n);
    SKM_REQUIRE(ld >= m, SKM_E_BADARG, "skm_cosine_csr: ld (%lld) < m (%lld)", (long long)ld, (long long)m);
    SKM_REQUIRE(mode == 0 || mode == 1, SKM_E_BADARG, "skm_cosine_csr: mode must be 0 or 1");
    SKM_REQUIRE(m < ((int64_t)1 << 32) - 1, SKM_E_OVERFLOW, "skm_cosine_csr: m >= 2^32");
    if (row1 == row0 || m == 0)
        return SKM_OK;
    SKM_REQUIRE(d_xrowptr && d_xrnorm && d_ycolptr && d_yrnorm && d_out, SKM_E_BADARG, "skm_cosine_csr: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    const int64_t strips = skm_ceil_div(row1 - row0, R);
    SKM_REQUIRE(strips < ((int64_t)1 << 31), SKM_E_OVERFLOW, "skm_cosine_csr: too many rows in one call");
    const bool vec = (ld % 4 == 0) && (((uintptr_t)d_out & 15) == 0) && (((uintptr_t)d_yrnorm & 15) == 0);
    SKM_PROF(ctx, "k_cosine_strip");
#define SKM_LAUNCH(MODE, VEC)                                                                                        \
    k_cosine_strip<MODE, VEC><<<(unsigned)strips, TB, 0, ctx->stream>>>(d_xrowptr, d_xcolidx, d_xcounts, d_xrnorm, m, \
                                                                         d_ycolptr, d_yprow, d_ypval, d_yrnorm, row0, \
                                                                         row1, d_out, ld)
    if (mode == 0) {
        if (vec)
            SKM_LAUNCH(0, true);
        else
            SKM_LAUNCH(0, false);
    } else {
        if (vec)
            SKM_LAUNCH(1, true);
        else
            SKM_LAUNCH(1, false);
    }
#undef SKM_LAUNCH
    return skm_check_launch("k_cosine_strip");
}
