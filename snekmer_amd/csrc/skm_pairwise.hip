// The `else` branch of snekmer/score.py:166-171: connection_matrix_from_features(X, metric=m) =
// sklearn.metrics.pairwise_distances(X, metric=m) for the metrics that are sums (or maxima) over columns of a function
// of (x_ic, y_jc), and for scipy's boolean dissimilarities, which are functions of the four agreement counts.  No
// Snekmer rule passes such a metric (the rules use the default branch and "cosine", which have their own kernels);
// this is the general form of the API hook, for tutorial-scale dense matrices: a float64 LDS-tiled kernel, one 64 x 64
// output tile per workgroup, 4 x 4 outputs per thread.  Not a matrix-core shape: |x - y| and max are not bilinear.
#include "skm_common.h"

namespace {

enum metric_id {
    M_CITYBLOCK = 0,    // sum |x - y|                                  (manhattan, l1)
    M_SQEUCLIDEAN = 1,  // sum (x - y)^2
    M_EUCLIDEAN = 2,    // sqrt of it                                   (l2)
    M_CHEBYSHEV = 3,    // max |x - y|
    M_CANBERRA = 4,     // sum |x - y| / (|x| + |y|), 0/0 terms skipped
    M_BRAYCURTIS = 5,   // sum |x - y| / sum |x + y|
    M_MINKOWSKI = 6,    // (sum |x - y|^p)^(1/p)
    M_NAN_EUCLIDEAN = 7,  // sklearn.metrics.pairwise.nan_euclidean_distances: sqrt(k / present * sum over the columns where
                          // neither value is NaN of (x - y)^2), NaN when no column is present in both rows
    M_HAVERSINE = 8,    // two columns (latitude, longitude in radians): 2 asin(sqrt(sin^2(dlat / 2) + cos cos sin^2(dlon / 2)))
    M_DICE = 10,        // the boolean family: x, y read as x != 0, y != 0 (scipy's definitions, 0/0 -> nan as scipy's C)
    M_ROGERSTANIMOTO = 11,
    M_RUSSELLRAO = 12,
    M_SOKALMICHENER = 13,
    M_SOKALSNEATH = 14,
    M_YULE = 15,
};

constexpr int PT = 64, PK = 16, PROW = PK + 1;

template <int M>
__global__ __launch_bounds__(256) void k_pairwise_f64(int64_t n, int64_t m, int64_t k, const double *__restrict__ X, int64_t ldx,
                                                      const double *__restrict__ Y, int64_t ldy, double p, int square,
                                                      double *__restrict__ out, int64_t ld)
{
    __shared__ double s_x[PT * PROW];
    __shared__ double s_y[PT * PROW];
    constexpr bool BOOLEAN = M >= 10;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t row0 = (int64_t)blockIdx.y * PT, col0 = (int64_t)blockIdx.x * PT;
    double a[4][4], b[4][4], c[4][4];  // a: the metric's sum (or ntt); b: second sum (or ntf); c: nft
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v)
            a[u][v] = b[u][v] = c[u][v] = 0.0;
    const int sr = tid >> 2, sc = (tid & 3) * 4;  // staging: 64 rows x 16 columns, 4 values per thread
    for (int64_t k0 = 0; k0 < k; k0 += PK) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t cc = k0 + sc + u;
            const int64_t gi = row0 + sr, gj = col0 + sr;
            s_x[sr * PROW + sc + u] = (gi < n && cc < k) ? X[gi * ldx + cc] : 0.0;
            s_y[sr * PROW + sc + u] = (gj < m && cc < k) ? Y[gj * ldy + cc] : 0.0;
        }
        __syncthreads();
        const int kk_end = (int)min((int64_t)PK, k - k0);
        for (int kk = 0; kk < kk_end; ++kk) {
            double xv[4], yv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xv[u] = s_x[(ty * 4 + u) * PROW + kk];
                yv[u] = s_y[(tx * 4 + u) * PROW + kk];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const double x = xv[u], y = yv[v];
                    if constexpr (BOOLEAN) {
                        const bool bx = x != 0.0, by = y != 0.0;
                        a[u][v] += (bx && by) ? 1.0 : 0.0;
                        b[u][v] += (bx && !by) ? 1.0 : 0.0;
                        c[u][v] += (!bx && by) ? 1.0 : 0.0;
                    } else {
                        const double d = fabs(x - y);
                        if constexpr (M == M_CITYBLOCK)
                            a[u][v] += d;
                        else if constexpr (M == M_SQEUCLIDEAN || M == M_EUCLIDEAN)
                            a[u][v] += d * d;
                        else if constexpr (M == M_NAN_EUCLIDEAN) {
                            if (x == x && y == y) {  // neither is NaN
                                a[u][v] += d * d;
                                b[u][v] += 1.0;
                            }
                        } else if constexpr (M == M_HAVERSINE) {  // k == 2 (checked by the launcher): k0 == 0, kk = 0, 1
                            const double sh = sin(0.5 * (x - y));
                            if (kk == 0) {
                                a[u][v] = sh * sh;
                                c[u][v] = cos(x) * cos(y);
                            } else
                                b[u][v] = sh * sh;
                        }
                        else if constexpr (M == M_CHEBYSHEV)
                            a[u][v] = fmax(a[u][v], d);
                        else if constexpr (M == M_CANBERRA) {
                            const double den = fabs(x) + fabs(y);
                            if (den > 0.0)
                                a[u][v] += d / den;
                        } else if constexpr (M == M_BRAYCURTIS) {
                            a[u][v] += d;
                            b[u][v] += fabs(x + y);
                        } else if constexpr (M == M_MINKOWSKI)
                            a[u][v] += pow(d, p);
                    }
                }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int64_t i = row0 + ty * 4 + u, j = col0 + tx * 4 + v;
            if (i >= n || j >= m)
                continue;
            double o;
            if constexpr (BOOLEAN) {
                const double ntt = a[u][v], ntf = b[u][v], nft = c[u][v], nff = (double)k - ntt - ntf - nft;
                const double r = ntf + nft;
                if constexpr (M == M_DICE)
                    o = r / (2.0 * ntt + r);
                else if constexpr (M == M_ROGERSTANIMOTO || M == M_SOKALMICHENER)
                    o = (2.0 * r) / (ntt + nff + 2.0 * r);
                else if constexpr (M == M_RUSSELLRAO)
                    o = ((double)k - ntt) / (double)k;
                else if constexpr (M == M_SOKALSNEATH)
                    o = (2.0 * r) / (2.0 * r + ntt);
                else {  // yule
                    const double half_r = ntf * nft;
                    o = half_r == 0.0 ? 0.0 : (2.0 * half_r) / (ntt * nff + half_r);
                }
            } else if constexpr (M == M_EUCLIDEAN)
                o = sqrt(a[u][v]);
            else if constexpr (M == M_NAN_EUCLIDEAN)
                o = b[u][v] > 0.0 ? sqrt(a[u][v] / b[u][v] * (double)k) : nan("");
            else if constexpr (M == M_HAVERSINE)
                o = 2.0 * asin(sqrt(a[u][v] + c[u][v] * b[u][v]));
            else if constexpr (M == M_BRAYCURTIS)
                o = a[u][v] / b[u][v];
            else if constexpr (M == M_MINKOWSKI)
                o = pow(a[u][v], 1.0 / p);
            else
                o = a[u][v];
            if (square && i == j)  // squareform(pdist(X)) / euclidean_distances(X): an exact-zero diagonal
                o = (M == M_NAN_EUCLIDEAN && !(b[u][v] > 0.0)) ? nan("") : 0.0;  // (sklearn: a row without a value stays NaN)
            out[i * ld + j] = o;
        }
}

}  // namespace

extern "C" int skm_pairwise_f64(skm_ctx *ctx, int metric, double p, int64_t n, int64_t m, int64_t k, const double *d_x,
                                int64_t ldx, const double *d_y, int64_t ldy, double *d_out, int64_t ld)
{
    SKM_REQUIRE(ctx && n >= 0 && m >= 0 && k >= 1 && ldx >= k && ldy >= k && ld >= m, SKM_E_BADARG, "skm_pairwise_f64: bad argument");
    SKM_REQUIRE(metric != M_MINKOWSKI || p > 0.0, SKM_E_BADARG, "skm_pairwise_f64: minkowski needs p > 0");
    SKM_REQUIRE(metric != M_HAVERSINE || k == 2, SKM_E_BADARG, "skm_pairwise_f64: haversine needs exactly two columns");
    if (n == 0 || m == 0)
        return SKM_OK;
    SKM_REQUIRE(d_x && d_y && d_out, SKM_E_BADARG, "skm_pairwise_f64: null array");
    SKM_HIP(hipSetDevice(ctx->device));
    const int square = d_x == d_y && n == m && ldx == ldy;
    dim3 grid((unsigned)skm_ceil_div(m, PT), (unsigned)skm_ceil_div(n, PT));
    SKM_PROF(ctx, "k_pairwise_f64");
#define SKM_PW(M)                                                                                                    \
    case M:                                                                                                          \
        k_pairwise_f64<M><<<grid, 256, 0, ctx->stream>>>(n, m, k, d_x, ldx, d_y, ldy, p, square, d_out, ld);          \
        break
    switch (metric) {
        SKM_PW(M_CITYBLOCK);
        SKM_PW(M_SQEUCLIDEAN);
        SKM_PW(M_EUCLIDEAN);
        SKM_PW(M_CHEBYSHEV);
        SKM_PW(M_CANBERRA);
        SKM_PW(M_BRAYCURTIS);
        SKM_PW(M_MINKOWSKI);
        SKM_PW(M_NAN_EUCLIDEAN);
        SKM_PW(M_HAVERSINE);
        SKM_PW(M_DICE);
        SKM_PW(M_ROGERSTANIMOTO);
        SKM_PW(M_RUSSELLRAO);
        SKM_PW(M_SOKALMICHENER);
        SKM_PW(M_SOKALSNEATH);
        SKM_PW(M_YULE);
    default:
        skm_set_error("skm_pairwise_f64: unknown metric id %d", metric);
        return SKM_E_UNSUPPORTED;
    }
#undef SKM_PW
    return skm_check_launch("k_pairwise_f64");
}
