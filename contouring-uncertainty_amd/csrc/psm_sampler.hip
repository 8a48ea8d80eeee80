// Monte-Carlo contour sampler: hierarchical Gaussian posterior-shape-model (PSM) sampling, one workgroup per frame.
//
// Replaces PosteriorShapeModelSampler.__call__ / sample_endo_contour / sample_points / merge_priors
// (reference contour_uncertainty/sampler/posterior_shape_model/psm.py:73-93,199-440) and pca / posterior_shape_model
// (posteriorshapemodel.py:9-81) for the Gaussian tasks.  The reference re-fits a 42x42 PCA with torch.linalg.eig per
// frame and runs two 42x42 inverses per level PER SAMPLE in a Python loop; here (SURVEY.md 3D "key algebraic fact"):
//   * with C = Q Q^T the PSM conditional is the Gaussian conditional with slack,
//         mu_c = m + C[:,g] (C[g,g] + s2 I)^-1 (s_g - m_g),   cov_c = C - C[:,g] (C[g,g] + s2 I)^-1 C[g,:],
//     and C = Cov0 + (xbar - m)(xbar - m)^T (Cov0 = training covariance about its own mean): no eigendecomposition;
//   * the gains and the merged 2x2 covariances depend on the frame and the level only, so they are computed once per
//     frame in LDS (Gauss-Jordan of at most 34x34) and every sample is a handful of mat-vecs + 2x2 algebra + RNG.
// Draws follow MultivariateNormal.rsample: x = mu + chol(Sigma) eps.  eps comes from the caller (exact comparison with
// the oracle) or from a counter-based generator (splitmix64 + Box-Muller).
#include "common.h"

namespace {

constexpr int MAXP = 48;       // flat shape dimension (2K), K <= 24 (the reference contours have K = 21)
constexpr int MAXLV = 6;       // levels incl. the final fill
constexpr int ST = 256;

struct PsmArgs {
    const float* mu_pred; const float* cov_pred; const float* cov0; const float* xbar; const float* smean;
    const float* sscale; const float* eps; float* out;
    const int* tables;         // per level: [ng, nt, g_flat[MAXP], t_pts[32]]  (stride 2 + MAXP + 32 ints)
    int F, S, K, n_init, n_levels;
    int init_pts[8];
    float sigma2[MAXLV];
    int sample_level[MAXLV];   // 1: draw the level's points, 0: fill them with the conditional mean (last level)
    unsigned long long seed;
};

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ void normal2(unsigned long long key, float& a, float& b) {
    const unsigned long long r = splitmix64(key);
    const float u1 = ((unsigned)(r >> 40) + 1.0f) * (1.0f / 16777217.0f);     // (0, 1)
    const float u2 = (unsigned)((r >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);   // [0, 1)
    const float rad = sqrtf(-2.f * logf(u1));
    a = rad * cosf(6.283185307179586f * u2);
    b = rad * sinf(6.283185307179586f * u2);
}

__global__ __launch_bounds__(ST) void psm_gauss_kernel(const PsmArgs p) {
    extern __shared__ __attribute__((aligned(16))) double lds_d[];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int K = p.K, P = 2 * K;
    constexpr int TSTRIDE = 2 + MAXP + 32;
    // LDS carve.  The per-frame linear algebra (C, the Gauss-Jordan inverse of C[g,g] + s2 I, the gains) runs in f64:
    // C[g,g] + I has a condition number of 1e4-1e5 at the deepest level, which costs f32 (the reference) ~0.3 px.
    double* C = lds_d;                        // [P][P]
    double* aug = C + MAXP * MAXP;            // [MAXP][2*MAXP]  Gauss-Jordan workspace
    float* m = reinterpret_cast<float*>(aug + MAXP * 2 * MAXP);   // [P] transformed predicted contour (PCA mean)
    float* G = m + MAXP;                      // per level gains: level l at goff[l], rows = 2*nt, cols = ng
    float* mrg = G + MAXLV * 32 * MAXP;       // per point 11 floats: M1 (4), M2 (4), chol(Sigma_f) (3)
    float* anch = mrg + 32 * 11;              // per point chol of the predicted covariance (3)
    float* cont = anch + 32 * 3;              // [P][ST] per-thread contour in pixel units
    __shared__ int goff[MAXLV];

    const float* mu_f = p.mu_pred + (size_t)f * P;
    const float* cv_f = p.cov_pred + (size_t)f * K * 3;

    // ---- a. C = Cov0 + d d^T with d = xbar - m
    if (tid < P) m[tid] = (mu_f[tid] - p.smean[tid]) / p.sscale[tid];
    if (tid < K) {       // Cholesky of every predicted covariance (anchors use it)
        const float a = cv_f[3 * tid], b = cv_f[3 * tid + 1], c = cv_f[3 * tid + 2];
        const float l11 = sqrtf(a), l21 = c / l11;
        anch[3 * tid] = l11; anch[3 * tid + 1] = l21; anch[3 * tid + 2] = sqrtf(fmaxf(b - l21 * l21, 0.f));
    }
    __syncthreads();
    for (int i = tid; i < P * P; i += ST) {
        const int r = i / P, c = i - r * P;
        C[r * MAXP + c] = (double)p.cov0[i] + ((double)p.xbar[r] - m[r]) * ((double)p.xbar[c] - m[c]);
    }
    if (tid == 0) {
        int off = 0;
        for (int l = 0; l < p.n_levels; ++l) {
            goff[l] = off;
            off += 2 * p.tables[l * TSTRIDE + 1] * p.tables[l * TSTRIDE];
        }
    }
    __syncthreads();

    // ---- b. per level: A^-1 (Gauss-Jordan, A SPD), gains for the target rows, merged 2x2 parameters
    for (int l = 0; l < p.n_levels; ++l) {
        const int* tb = p.tables + l * TSTRIDE;
        const int ng = tb[0], nt = tb[1];
        const int* gi = tb + 2;
        const int* tp = tb + 2 + MAXP;
        const float s2 = p.sigma2[l];
        for (int i = tid; i < ng * 2 * ng; i += ST) {
            const int r = i / (2 * ng), c = i - r * 2 * ng;
            double v;
            if (c < ng) v = C[gi[r] * MAXP + gi[c]] + (r == c ? (double)s2 : 0.0);
            else v = (c - ng == r) ? 1.0 : 0.0;
            aug[r * 2 * MAXP + c] = v;
        }
        __syncthreads();
        for (int k = 0; k < ng; ++k) {
            const double piv = 1.0 / aug[k * 2 * MAXP + k];
            __syncthreads();
            for (int c = tid; c < 2 * ng; c += ST) aug[k * 2 * MAXP + c] *= piv;
            __syncthreads();
            for (int i = tid; i < ng * 2 * ng; i += ST) {
                const int r = i / (2 * ng), c = i - r * 2 * ng;
                if (r != k && c != k) aug[r * 2 * MAXP + c] -= aug[r * 2 * MAXP + k] * aug[k * 2 * MAXP + c];
            }
            __syncthreads();
            for (int r = tid; r < ng; r += ST)
                if (r != k) aug[r * 2 * MAXP + k] = 0.0;
            __syncthreads();
        }
        // gains: G[row][j] = sum_i C[t_row][g_i] * Ainv[i][j]
        float* Gl = G + goff[l];
        for (int i = tid; i < 2 * nt * ng; i += ST) {
            const int row = i / ng, j = i - row * ng;
            const int trow = 2 * tp[row >> 1] + (row & 1);
            double s = 0.0;
            for (int q = 0; q < ng; ++q) s += C[trow * MAXP + gi[q]] * aug[q * 2 * MAXP + ng + j];
            Gl[row * ng + j] = (float)s;
        }
        __syncthreads();
        if (p.sample_level[l] && tid < nt) {
            const int pt = tp[tid];
            float sc[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    double s = C[(2 * pt + a) * MAXP + 2 * pt + b];
                    for (int q = 0; q < ng; ++q) s -= (double)Gl[(2 * tid + a) * ng + q] * C[gi[q] * MAXP + 2 * pt + b];
                    sc[a][b] = (float)s * p.sscale[2 * pt + b];      // psm.py:276 `cov_c *= self.scale` (column broadcast)
                }
            const float a1 = cv_f[3 * pt], b1 = cv_f[3 * pt + 1], c1 = cv_f[3 * pt + 2];
            const float s1[2][2] = {{a1, c1}, {c1, b1}};
            float t[2][2] = {{s1[0][0] + sc[0][0], s1[0][1] + sc[0][1]}, {s1[1][0] + sc[1][0], s1[1][1] + sc[1][1]}};
            const float idet = 1.f / (t[0][0] * t[1][1] - t[0][1] * t[1][0]);
            const float w[2][2] = {{t[1][1] * idet, -t[0][1] * idet}, {-t[1][0] * idet, t[0][0] * idet}};
            float m1[2][2], m2[2][2], sf[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    m1[a][b] = s1[a][0] * w[0][b] + s1[a][1] * w[1][b];
                    m2[a][b] = sc[a][0] * w[0][b] + sc[a][1] * w[1][b];
                }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) sf[a][b] = m1[a][0] * sc[0][b] + m1[a][1] * sc[1][b];   // S1 W S2
            float* o = mrg + pt * 11;
            o[0] = m1[0][0]; o[1] = m1[0][1]; o[2] = m1[1][0]; o[3] = m1[1][1];
            o[4] = m2[0][0]; o[5] = m2[0][1]; o[6] = m2[1][0]; o[7] = m2[1][1];
            const float l11 = sqrtf(sf[0][0]), l21 = sf[1][0] / l11;     // torch.linalg.cholesky reads the lower triangle
            o[8] = l11; o[9] = l21; o[10] = sqrtf(fmaxf(sf[1][1] - l21 * l21, 0.f));
        }
        __syncthreads();
    }

    // ---- c. samples: one thread per sample, contour kept in LDS (thread-minor)
    for (int s = tid; s < p.S; s += ST) {
        float* ct = cont + tid;
        const size_t obase = ((size_t)f * p.S + s) * P;
        auto draw = [&](int pt, float& e0, float& e1) {
            if (p.eps) { e0 = p.eps[obase + 2 * pt]; e1 = p.eps[obase + 2 * pt + 1]; }
            else normal2(p.seed ^ (((unsigned long long)f * p.S + s) * 64ull + pt) * 0x9E3779B97F4A7C15ull, e0, e1);
        };
        for (int i = 0; i < p.n_init; ++i) {
            const int pt = p.init_pts[i];
            float e0, e1;
            draw(pt, e0, e1);
            ct[(2 * pt) * ST] = mu_f[2 * pt] + anch[3 * pt] * e0;
            ct[(2 * pt + 1) * ST] = mu_f[2 * pt + 1] + anch[3 * pt + 1] * e0 + anch[3 * pt + 2] * e1;
        }
        for (int l = 0; l < p.n_levels; ++l) {
            const int* tb = p.tables + l * TSTRIDE;
            const int ng = tb[0], nt = tb[1];
            const int* gi = tb + 2;
            const int* tp = tb + 2 + MAXP;
            const float* Gl = G + goff[l];
            for (int q = 0; q < nt; ++q) {
                const int pt = tp[q];
                float mc[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float acc = m[2 * pt + a];
                    for (int j = 0; j < ng; ++j) {
                        const int gj = gi[j];
                        acc += Gl[(2 * q + a) * ng + j] * ((ct[gj * ST] - p.smean[gj]) / p.sscale[gj] - m[gj]);
                    }
                    mc[a] = acc * p.sscale[2 * pt + a] + p.smean[2 * pt + a];     // inverse_transform
                }
                if (p.sample_level[l]) {
                    const float* o = mrg + pt * 11;
                    const float mu1x = mu_f[2 * pt], mu1y = mu_f[2 * pt + 1];
                    const float mfx = o[0] * mc[0] + o[1] * mc[1] + o[4] * mu1x + o[5] * mu1y;
                    const float mfy = o[2] * mc[0] + o[3] * mc[1] + o[6] * mu1x + o[7] * mu1y;
                    float e0, e1;
                    draw(pt, e0, e1);
                    mc[0] = mfx + o[8] * e0;
                    mc[1] = mfy + o[9] * e0 + o[10] * e1;
                }
                // targets of a level are never part of its own conditioning set: safe to write immediately
                ct[(2 * pt) * ST] = mc[0];
                ct[(2 * pt + 1) * ST] = mc[1];
            }
        }
        for (int i = 0; i < P; ++i) p.out[obase + i] = ct[i * ST];
    }
}

}  // namespace

extern "C" int cu_psm_sample_gauss(int F, int S, int K, const float* mu_pred, const float* cov_pred, const float* cov0,
                                   const float* xbar, const float* smean, const float* sscale, int n_init,
                                   const int* init_pts, int n_levels, const int* tables, const float* sigma2,
                                   const int* sample_level, const float* eps, uint64_t seed, float* out, void* stream) {
    CU_CHECK_ARG(F > 0 && S > 0 && K > 0 && 2 * K <= MAXP, "cu_psm_sample_gauss: bad sizes F=%d S=%d K=%d", F, S, K);
    CU_CHECK_ARG(n_init > 0 && n_init <= 8 && n_levels > 0 && n_levels <= MAXLV, "cu_psm_sample_gauss: bad level counts");
    CU_CHECK_ARG(mu_pred && cov_pred && cov0 && xbar && smean && sscale && init_pts && tables && sigma2 && sample_level && out,
                 "cu_psm_sample_gauss: null pointer");
    PsmArgs a;
    memset(&a, 0, sizeof(a));
    a.mu_pred = mu_pred; a.cov_pred = cov_pred; a.cov0 = cov0; a.xbar = xbar; a.smean = smean; a.sscale = sscale;
    a.eps = eps; a.out = out; a.tables = tables; a.F = F; a.S = S; a.K = K; a.n_init = n_init; a.n_levels = n_levels;
    for (int i = 0; i < n_init; ++i) a.init_pts[i] = init_pts[i];         // host arrays (small)
    for (int l = 0; l < n_levels; ++l) { a.sigma2[l] = sigma2[l]; a.sample_level[l] = sample_level[l]; }
    a.seed = seed;
    const size_t lds = sizeof(double) * ((size_t)MAXP * MAXP + (size_t)MAXP * 2 * MAXP) +
                       sizeof(float) * (MAXP + (size_t)MAXLV * 32 * MAXP + 32 * 11 + 32 * 3 + (size_t)MAXP * ST);
    CU_CHECK_ARG(lds <= 160 * 1024, "cu_psm_sample_gauss: LDS %zu too large", lds);
    auto k = psm_gauss_kernel;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    CU_CHECK_ARG(e == hipSuccess, "cu_psm_sample_gauss: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k, dim3(F), dim3(ST), lds, reinterpret_cast<hipStream_t>(stream), a);
    CU_LAUNCH_CHECK();
    return 0;
}
