// Bivariate normal / skew-normal helpers of the contour samplers (gfx950).
//   cu_logpdf_grid : BivariateNormal.logpdf / BivariateSkewNormal.logpdf (reference contour_uncertainty/distributions/
//                    bivariatenormal.py:15-36, bivariateskewnormal.py:19-49) for M distributions on P points
//                    (outer product, or pairwise when pairwise = 1 and M == P).  Closed-form 2x2 algebra; the skew
//                    affine uses Sigma^-1/2 = 1/(s t) [[b+s, -c], [-c, a+s]], s = sqrt(det), t = sqrt(tr + 2 s).
//   cu_skew_rvs    : BivariateSkewNormal.rvs_fast (bivariateskewnormal.py:159-191): (x0, x) ~ N(0, [[1, d^T], [d, S]]),
//                    d = S a / sqrt(1 + a^T S a); x <- -x where x0 <= 0; + mu.  Draws via the Cholesky factor of the
//                    3x3 covariance, exactly like MultivariateNormal.sample.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void logpdf_kernel(int M, int P, int pairwise, const float* __restrict__ pts,
                                                     const float* __restrict__ mu, const float* __restrict__ sigma,
                                                     const float* __restrict__ alpha, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = pairwise ? (size_t)P : (size_t)M * P;
    if (i >= total) return;
    const int m = pairwise ? (int)i : (int)(i / P);
    const int pt = pairwise ? (int)i : (int)(i % P);
    const float a = sigma[3 * m], b = sigma[3 * m + 1], c = sigma[3 * m + 2];
    const float d1 = pts[2 * pt] - mu[2 * m], d2 = pts[2 * pt + 1] - mu[2 * m + 1];
    const float det = a * b - c * c;
    const float quad = (b * d1 * d1 - 2.f * c * d1 * d2 + a * d2 * d2) / det;
    float lp = -1.8378770664093453f - 0.5f * logf(det) - 0.5f * quad;     // K/2 log(2 pi) with K = 2
    if (alpha) {
        const float s = sqrtf(det), t = sqrtf(a + b + 2.f * s);
        const float al1 = alpha[2 * m], al2 = alpha[2 * m + 1];
        const float z = (al1 * ((b + s) * d1 - c * d2) + al2 * ((a + s) * d2 - c * d1)) / (s * t);
        const float cdf = 0.5f * (1.f + erff(z * 0.70710678118654752f));
        lp += 0.6931471805599453f + logf(cdf + 1e-7f);
    }
    out[i] = lp;
}

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ void gauss2(unsigned long long key, float& g0, float& g1) {
    const unsigned long long r = mix64(key);
    const float u1 = ((unsigned)(r >> 40) + 1.0f) * (1.0f / 16777217.0f);
    const float u2 = (unsigned)((r >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.f * logf(u1));
    g0 = rad * cosf(6.283185307179586f * u2);
    g1 = rad * sinf(6.283185307179586f * u2);
}

__global__ __launch_bounds__(256) void skew_rvs_kernel(int M, int S, const float* __restrict__ mu,
                                                       const float* __restrict__ sigma, const float* __restrict__ alpha,
                                                       const float* __restrict__ eps, unsigned long long seed,
                                                       float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)M * S) return;
    const int m = (int)(i / S);
    const float a = sigma[3 * m], b = sigma[3 * m + 1], c = sigma[3 * m + 2];
    const float al1 = alpha[2 * m], al2 = alpha[2 * m + 1];
    const float sa1 = a * al1 + c * al2, sa2 = c * al1 + b * al2;          // Sigma alpha
    const float norm = 1.f / sqrtf(1.f + al1 * sa1 + al2 * sa2);
    const float dl1 = sa1 * norm, dl2 = sa2 * norm;                         // delta
    // Cholesky of [[1, dl1, dl2], [dl1, a, c], [dl2, c, b]]
    const float l21 = dl1, l31 = dl2;
    const float l22 = sqrtf(fmaxf(a - l21 * l21, 0.f));
    const float l32 = (c - l31 * l21) / l22;
    const float l33 = sqrtf(fmaxf(b - l31 * l31 - l32 * l32, 0.f));
    float e0, e1, e2;
    if (eps) { e0 = eps[3 * i]; e1 = eps[3 * i + 1]; e2 = eps[3 * i + 2]; }
    else {
        float dummy;
        gauss2(seed ^ (i * 2ull) * 0x9E3779B97F4A7C15ull, e0, e1);
        gauss2(seed ^ (i * 2ull + 1ull) * 0x9E3779B97F4A7C15ull, e2, dummy);
    }
    const float x0 = e0;
    float x1 = l21 * e0 + l22 * e1;
    float x2 = l31 * e0 + l32 * e1 + l33 * e2;
    if (x0 <= 0.f) { x1 = -x1; x2 = -x2; }
    out[2 * i] = x1 + mu[2 * m];
    out[2 * i + 1] = x2 + mu[2 * m + 1];
}

}  // namespace

extern "C" int cu_logpdf_grid(int M, int P, int pairwise, const float* pts, const float* mu, const float* sigma,
                              const float* alpha, float* out, void* stream) {
    CU_CHECK_ARG(M > 0 && P > 0 && pts && mu && sigma && out, "cu_logpdf_grid: bad argument");
    CU_CHECK_ARG(!pairwise || M == P, "cu_logpdf_grid: pairwise mode needs M == P");
    const size_t total = pairwise ? (size_t)P : (size_t)M * P;
    hipLaunchKernelGGL(logpdf_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), M, P, pairwise, pts, mu, sigma, alpha, out);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_skew_rvs(int M, int S, const float* mu, const float* sigma, const float* alpha, const float* eps,
                           uint64_t seed, float* out, void* stream) {
    CU_CHECK_ARG(M > 0 && S > 0 && mu && sigma && alpha && out, "cu_skew_rvs: bad argument");
    const size_t total = (size_t)M * S;
    hipLaunchKernelGGL(skew_rvs_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), M, S, mu, sigma, alpha, eps, (unsigned long long)seed, out);
    CU_LAUNCH_CHECK();
    return 0;
}
