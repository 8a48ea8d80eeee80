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

//
// The skew-normal sampler (SkewPosteriorShapeModelSampler, psm_skew.py:45-158,162-503) and the ED/ES sequence samplers
// (sequence_sampler.py:13-160, psm_skew_sequence.py:21-166) share the same per-frame algebra (psm_frame_setup below):
//   cu_psm_setup          gains / conditional covariances of every level into a global per-frame record;
//   cu_psm_sample_skew    one WAVE per (frame, sample), four samples of a frame per workgroup: anchors by rvs_fast, every other point by inverse-CDF
//                         sampling of  skew-pdf(prediction) x N(mu_c, cov_c)  on the 256x256 pixel grid
//                         (`numerical_sampling`), evaluated on the fly - no pdf grid ever touches HBM;
//   cu_psm_condition      conditional mean of a 1-level record given sampled contours (+ product-of-Gaussians merge):
//                         the ED -> ES coupling of the sequence samplers.
namespace {

constexpr int MAXP = 48;       // flat shape dimension (2K), K <= 24 (the reference contours have K = 21)
constexpr int MAXG = 48;       // conditioning-set size per level
constexpr int MAXT = 32;       // target points per level
constexpr int MAXLV = 6;       // levels incl. the final fill
constexpr int ST = 256;
constexpr int TSTRIDE = 2 + MAXG + MAXT;

struct SetupArgs {
    const float* mu_pred;      // [F][P] predicted contour, pixel units (the PCA is re-centred on it, psm.py:91)
    const float* cov0; const float* xbar; const float* smean; const float* sscale;
    const int* tables;         // per level: [ng, nt, g_flat[MAXG], t_pts[MAXT]]
    int P, n_levels;
    float sigma2[MAXLV];
};

// Per-frame record written by cu_psm_setup (floats): m[REC_M] | covc[48][4] | G (level l at goff[l], rows 2*nt, cols ng)
constexpr int REC_M = 96, REC_COVC = REC_M, REC_G = REC_COVC + 48 * 4;

struct PsmArgs {
    SetupArgs su;
    const float* cov_pred; const float* eps; float* out;
    int F, S, K, n_init;
    int init_pts[8];
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
__device__ __forceinline__ float uniform01(unsigned long long key) {
    return (unsigned)(splitmix64(key) >> 40) * (1.0f / 16777216.0f);           // [0, 1)
}

// Frame algebra shared by every sampler.  All ST threads of the block take part.  MP = leading dimension of C.
//   m[P]        transformed predicted contour (= PCA mean)
//   G           gains of every level, level l at goff[l]: G[row][j] = sum_i C[t_row][g_i] (C[g,g] + s2 I)^-1[i][j]
//   covc[pt][4] conditional covariance block of target point pt, scaled like psm.py:276 (`cov_c *= self.scale`)
// The linear algebra (C, Gauss-Jordan inverse, gains) runs in f64: C[g,g] + I has a condition number of 1e4-1e5 at the
// deepest level, which costs f32 (the reference) ~0.3 px.  m / G / covc may live in LDS or in global memory.
template <int MP>
__device__ void psm_frame_setup(const SetupArgs& p, int f, double* C, double* aug, float* m, float* G, float* covc,
                                int* goff) {
    const int tid = threadIdx.x, P = p.P;
    const float* mu_f = p.mu_pred + (size_t)f * P;
    if (tid < P) m[tid] = (mu_f[tid] - p.smean[tid]) / p.sscale[tid];
    if (tid == 0) {
        int off = 0;
        for (int l = 0; l < p.n_levels; ++l) {
            goff[l] = off;
            off += 2 * p.tables[l * TSTRIDE + 1] * p.tables[l * TSTRIDE];
        }
    }
    __syncthreads();
    // C = Cov0 + d d^T with d = xbar - m
    for (int i = tid; i < P * P; i += ST) {
        const int r = i / P, c = i - r * P;
        C[r * MP + c] = (double)p.cov0[i] + ((double)p.xbar[r] - m[r]) * ((double)p.xbar[c] - m[c]);
    }
    __syncthreads();
    for (int l = 0; l < p.n_levels; ++l) {
        const int* tb = p.tables + l * TSTRIDE;
        const int ng = tb[0], nt = tb[1];
        const int* gi = tb + 2;
        const int* tp = tb + 2 + MAXG;
        const float s2 = p.sigma2[l];
        constexpr int LD = 2 * MAXG;
        for (int i = tid; i < ng * 2 * ng; i += ST) {
            const int r = i / (2 * ng), c = i - r * 2 * ng;
            double v;
            if (c < ng) v = C[gi[r] * MP + gi[c]] + (r == c ? (double)s2 : 0.0);
            else v = (c - ng == r) ? 1.0 : 0.0;
            aug[r * LD + c] = v;
        }
        __syncthreads();
        for (int k = 0; k < ng; ++k) {       // Gauss-Jordan, A SPD: no pivoting needed
            const double piv = 1.0 / aug[k * LD + k];
            __syncthreads();
            for (int c = tid; c < 2 * ng; c += ST) aug[k * LD + c] *= piv;
            __syncthreads();
            for (int i = tid; i < ng * 2 * ng; i += ST) {
                const int r = i / (2 * ng), c = i - r * 2 * ng;
                if (r != k && c != k) aug[r * LD + c] -= aug[r * LD + k] * aug[k * LD + c];
            }
            __syncthreads();
            for (int r = tid; r < ng; r += ST)
                if (r != k) aug[r * LD + k] = 0.0;
            __syncthreads();
        }
        float* Gl = G + goff[l];
        for (int i = tid; i < 2 * nt * ng; i += ST) {
            const int row = i / ng, j = i - row * ng;
            const int trow = 2 * tp[row >> 1] + (row & 1);
            double s = 0.0;
            for (int q = 0; q < ng; ++q) s += C[trow * MP + gi[q]] * aug[q * LD + ng + j];
            Gl[row * ng + j] = (float)s;
        }
        __syncthreads();
        for (int i = tid; i < 4 * nt; i += ST) {
            const int q = i >> 2, a = (i >> 1) & 1, b = i & 1, pt = tp[q];
            double s = C[(2 * pt + a) * MP + 2 * pt + b];
            for (int j = 0; j < ng; ++j) s -= (double)Gl[(2 * q + a) * ng + j] * C[gi[j] * MP + 2 * pt + b];
            covc[pt * 4 + a * 2 + b] = (float)s * p.sscale[2 * pt + b];      // column broadcast, like the reference
        }
        __syncthreads();
    }
}

// product of Gaussians (psm.py:424-440): M1 = S1 W, M2 = Sc W, Sf = S1 W Sc with W = (S1 + Sc)^-1;  mu_f = M1 mu_c + M2 mu_1
__device__ __forceinline__ void merge2x2(const float s1[2][2], const float sc[2][2], float m1[2][2], float m2[2][2],
                                         float sf[2][2]) {
    const float t[2][2] = {{s1[0][0] + sc[0][0], s1[0][1] + sc[0][1]}, {s1[1][0] + sc[1][0], s1[1][1] + sc[1][1]}};
    const float idet = 1.f / (t[0][0] * t[1][1] - t[0][1] * t[1][0]);
    const float w[2][2] = {{t[1][1] * idet, -t[0][1] * idet}, {-t[1][0] * idet, t[0][0] * idet}};
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
        for (int b = 0; b < 2; ++b) sf[a][b] = m1[a][0] * sc[0][b] + m1[a][1] * sc[1][b];
}

__global__ __launch_bounds__(ST) void psm_gauss_kernel(const PsmArgs p) {
    extern __shared__ __attribute__((aligned(16))) double lds_d[];
    const int f = blockIdx.x, tid = threadIdx.x;
    const int K = p.K, P = 2 * K;
    double* C = lds_d;                        // [P][P]
    double* aug = C + MAXP * MAXP;            // [MAXG][2*MAXG]  Gauss-Jordan workspace
    float* m = reinterpret_cast<float*>(aug + MAXG * 2 * MAXG);   // [P] transformed predicted contour (PCA mean)
    float* G = m + MAXP;                      // per level gains
    float* covc = G + MAXLV * MAXT * MAXP;    // [48][4]
    float* mrg = covc + 48 * 4;               // per point 11 floats: M1 (4), M2 (4), chol(Sigma_f) (3)
    float* anch = mrg + MAXT * 11;            // per point chol of the predicted covariance (3)
    float* cont = anch + MAXT * 3;            // [P][ST] per-thread contour in pixel units
    __shared__ int goff[MAXLV];

    const float* mu_f = p.su.mu_pred + (size_t)f * P;
    const float* cv_f = p.cov_pred + (size_t)f * K * 3;
    if (tid < K) {       // Cholesky of every predicted covariance (anchors use it)
        const float a = cv_f[3 * tid], b = cv_f[3 * tid + 1], c = cv_f[3 * tid + 2];
        const float l11 = sqrtf(a), l21 = c / l11;
        anch[3 * tid] = l11; anch[3 * tid + 1] = l21; anch[3 * tid + 2] = sqrtf(fmaxf(b - l21 * l21, 0.f));
    }
    psm_frame_setup<MAXP>(p.su, f, C, aug, m, G, covc, goff);
    // merged 2x2 parameters of every sampled point
    for (int l = 0; l < p.su.n_levels; ++l) {
        const int* tb = p.su.tables + l * TSTRIDE;
        const int nt = tb[1];
        const int* tp = tb + 2 + MAXG;
        if (p.sample_level[l] && tid < nt) {
            const int pt = tp[tid];
            const float sc[2][2] = {{covc[pt * 4], covc[pt * 4 + 1]}, {covc[pt * 4 + 2], covc[pt * 4 + 3]}};
            const float a1 = cv_f[3 * pt], b1 = cv_f[3 * pt + 1], c1 = cv_f[3 * pt + 2];
            const float s1[2][2] = {{a1, c1}, {c1, b1}};
            float m1[2][2], m2[2][2], sf[2][2];
            merge2x2(s1, sc, m1, m2, sf);
            float* o = mrg + pt * 11;
            o[0] = m1[0][0]; o[1] = m1[0][1]; o[2] = m1[1][0]; o[3] = m1[1][1];
            o[4] = m2[0][0]; o[5] = m2[0][1]; o[6] = m2[1][0]; o[7] = m2[1][1];
            const float l11 = sqrtf(sf[0][0]), l21 = sf[1][0] / l11;     // torch.linalg.cholesky reads the lower triangle
            o[8] = l11; o[9] = l21; o[10] = sqrtf(fmaxf(sf[1][1] - l21 * l21, 0.f));
        }
    }
    __syncthreads();

    // samples: one thread per sample, contour kept in LDS (thread-minor)
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
        for (int l = 0; l < p.su.n_levels; ++l) {
            const int* tb = p.su.tables + l * TSTRIDE;
            const int ng = tb[0], nt = tb[1];
            const int* gi = tb + 2;
            const int* tp = tb + 2 + MAXG;
            const float* Gl = G + goff[l];
            for (int q = 0; q < nt; ++q) {
                const int pt = tp[q];
                float mc[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    float acc = m[2 * pt + a];
                    for (int j = 0; j < ng; ++j) {
                        const int gj = gi[j];
                        acc += Gl[(2 * q + a) * ng + j] * ((ct[gj * ST] - p.su.smean[gj]) / p.su.sscale[gj] - m[gj]);
                    }
                    mc[a] = acc * p.su.sscale[2 * pt + a] + p.su.smean[2 * pt + a];     // inverse_transform
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

// ---- per-frame record for the grid sampler / the sequence coupling -------------------------------------------------
template <int MP>
__global__ __launch_bounds__(ST) void psm_setup_kernel(const SetupArgs p, float* rec, int rec_stride) {
    extern __shared__ __attribute__((aligned(16))) double lds_d[];
    __shared__ int goff[MAXLV];
    double* C = lds_d;
    double* aug = C + MP * MP;
    float* r = rec + (size_t)blockIdx.x * rec_stride;
    psm_frame_setup<MP>(p, blockIdx.x, C, aug, r, r + REC_G, r + REC_COVC, goff);
}

// ---- skew-normal grid sampler ----------------------------------------------------------------------------------------
struct Gauss2 { float mx, my, qa, qb, qc, lc; };     // log N(x) = lc - 0.5 (qa d1^2 + qb d1 d2 + qc d2^2)
struct Skew2 { Gauss2 g; float z1, z2; };            // x 2 Phi(z1 d1 + z2 d2)

__device__ __forceinline__ Gauss2 make_gauss(float mx, float my, float a, float b, float c) {
    const float det = a * b - c * c, id = 1.f / det;
    return Gauss2{mx, my, b * id, -2.f * c * id, a * id, -1.8378770664093453f - 0.5f * logf(det)};
}
__device__ __forceinline__ float gauss_pdf(const Gauss2& g, float x, float y) {
    const float d1 = x - g.mx, d2 = y - g.my;
    return expf(g.lc - 0.5f * (g.qa * d1 * d1 + g.qb * d1 * d2 + g.qc * d2 * d2));
}
__device__ __forceinline__ float skew_pdf(const Skew2& k, float x, float y) {
    const float d1 = x - k.g.mx, d2 = y - k.g.my;
    const float n = expf(k.g.lc - 0.5f * (k.g.qa * d1 * d1 + k.g.qb * d1 * d2 + k.g.qc * d2 * d2));
    const float cdf = 0.5f * (1.f + erff((k.z1 * d1 + k.z2 * d2) * 0.70710678118654752f));
    return 2.f * n * (cdf + 1e-7f);                  // bivariateskewnormal.py:30-48: log 2 + log N + log(Phi + 1e-7)
}

struct SkewArgs {
    const float* mu_pred; const float* cov_pred; const float* alpha;   // [F][K][2], [F][K][3] {xx,yy,xy}, [F][K][2]
    const float* rec; int rec_stride;
    const float* smean; const float* sscale;
    const int* tables;
    const float* prior_mu; const float* prior_cov;      // optional extra Gaussian factor per point: [F][S][K][2], [F][S][K][3]
    const float* eps; const float* u; float* out;
    int F, S, K, n_init, n_levels, grid, use_initial_pdf;
    int merged_window;                                  // window of the PRODUCT of the Gaussian factors (grid_sample)
    int init_pts[8];
    int sample_level[MAXLV];
    unsigned long long skew_bits;                       // bit k: point k is drawn from the skew grid product
    float alpha_y_sign;                                 // -1: psm_skew.py:232 undoes the predict-time flip of alpha_y
    unsigned long long seed;
};

// ---- one WAVE per sample (round 4) -----------------------------------------------------------------------------------
// The first version gave a sample a whole 256-thread workgroup: thread = grid row, two 256-wide Hillis-Steele scans through LDS
// and ~22 barriers per drawn point, with the window of non-zero cells (30-70 rows for the PSM's tight conditionals) keeping one
// or two of the four waves busy.  Now a workgroup holds FOUR samples of one frame (the frame record is staged once for the
// four), each wave runs its sample on its own: scans and reductions are wave shuffles, the contour lives in a register per
// lane (lane i = flat coordinate i), and after the staging barrier there is no workgroup synchronisation at all.
constexpr float NARROW_HALF = 36.f;     // grid_sample: the narrow box holds the cells within exp(-36) of the product's peak

__device__ __forceinline__ double wave_scan(double v, int lane) {         // inclusive prefix over the 64 lanes
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

// Draw one grid cell from  pA(x,y) * pB(x,y) [* pC(x,y)]  on the grid x grid lattice of pixel coordinates
// linspace(0, 255, grid) (numerical_sampling, psm_skew.py:45-158): inverse CDF in the flat order of torch's
// meshgrid(indexing='ij') (x major).  Cells where the Gaussian factor(s) are exactly 0 in f32 are skipped: they carry
// no mass in the reference's table either.  Returns false when the table has no mass (the reference's multinomial
// raises and it falls back to mu_c, psm_skew.py:135-154).  All arguments are wave-uniform; so are the results.
// Window of nx rows x ny cells: k = 64 / nx (a power of two) lanes share a row when the window is narrow, each summing every
// k-th cell of it; wider windows take ceil(nx / 64) passes of one lane per row (grid <= 256: at most 4).
__device__ bool grid_sample(const Skew2& A, const Gauss2& B, const Gauss2* Cg, float covBxx, float covByy, float covCxx,
                            float covCyy, int grid, float u, int lane, float& sx, float& sy, int merged_window) {
    const float step = 255.f / (float)(grid - 1), istep = (float)(grid - 1) / 255.f;
    // window where exp(lc - q/2) can be non-zero in f32 (smallest denormal = exp(-103.3)); q >= d1^2 / Sigma_xx
    float L = 2.f * (104.f + fmaxf(B.lc, 0.f));
    float rx = sqrtf(L * covBxx), ry = sqrtf(L * covByy);
    float x0 = B.mx - rx, x1 = B.mx + rx, y0 = B.my - ry, y1 = B.my + ry;
    if (Cg) {
        L = 2.f * (104.f + fmaxf(Cg->lc, 0.f));
        rx = sqrtf(L * covCxx); ry = sqrtf(L * covCyy);
        x0 = fmaxf(x0, Cg->mx - rx); x1 = fminf(x1, Cg->mx + rx);
        y0 = fmaxf(y0, Cg->my - ry); y1 = fminf(y1, Cg->my + ry);
    }
    bool narrow_ok = false;
    float n_fx = 0.f, n_fy = 0.f, n_rx = 0.f, n_ry = 0.f, n_lcs = 0.f, n_r0 = 0.f;
    if (merged_window) {
        // The Gaussian factors multiply to c N(x; mu_f, P^-1), P = sum of the inverse covariances, and the skew factor
        // 2 (Phi + 1e-7) is at most 2.0000002: in the reference's f32 table p1 * p2 [* p3] (psm_skew.py:96-107) a cell is exactly
        // 0 -- below the smallest denormal, exp(-103.28) -- where
        //     sum lc_i + log 2 - (r0 + (x - mu_f)^T P (x - mu_f)) / 2 < -103.28,     r0 = sum q_i(mu_f)
        // (one unit of slack in the exponent for the f32 evaluation).  The bounding box of that ellipse is never larger than the
        // window of the conditional Gaussian alone and much smaller when the prediction is the tighter factor.
        float Pa = A.g.qa + B.qa, Pb = A.g.qb + B.qb, Pc = A.g.qc + B.qc;                 // P = [[Pa, Pb/2], [Pb/2, Pc]]
        float hx = A.g.qa * A.g.mx + 0.5f * A.g.qb * A.g.my + B.qa * B.mx + 0.5f * B.qb * B.my;
        float hy = 0.5f * A.g.qb * A.g.mx + A.g.qc * A.g.my + 0.5f * B.qb * B.mx + B.qc * B.my;
        float lcs = A.g.lc + B.lc;
        if (Cg) {
            Pa += Cg->qa; Pb += Cg->qb; Pc += Cg->qc;
            hx += Cg->qa * Cg->mx + 0.5f * Cg->qb * Cg->my;
            hy += 0.5f * Cg->qb * Cg->mx + Cg->qc * Cg->my;
            lcs += Cg->lc;
        }
        const float det = Pa * Pc - 0.25f * Pb * Pb, idet = 1.f / det;
        const float fx = (Pc * hx - 0.5f * Pb * hy) * idet, fy = (Pa * hy - 0.5f * Pb * hx) * idet;
        auto quad = [&](const Gauss2& g) {
            const float d1 = fx - g.mx, d2 = fy - g.my;
            return g.qa * d1 * d1 + g.qb * d1 * d2 + g.qc * d2 * d2;
        };
        float r0 = quad(A.g) + quad(B);
        if (Cg) r0 += quad(*Cg);
        const float Lf = 2.f * (lcs + 0.7f + 104.3f) - r0;
        if (!(det > 0.f) || !(Lf == Lf)) {
            // (degenerate algebra: keep the window of the conditional Gaussian)
        } else if (Lf <= 0.f) {
            x1 = x0 - 1.f;                                        // no cell carries mass
        } else {
            const float rfx = sqrtf(Lf * Pc * idet), rfy = sqrtf(Lf * Pa * idet);
            x0 = fmaxf(x0, fx - rfx); x1 = fminf(x1, fx + rfx);
            y0 = fmaxf(y0, fy - rfy); y1 = fminf(y1, fy + rfy);
            if (merged_window > 1 && Lf > 2.f * NARROW_HALF) {
                narrow_ok = true;
                n_fx = fx; n_fy = fy; n_lcs = lcs; n_r0 = r0;
                n_rx = sqrtf(2.f * NARROW_HALF * Pc * idet) + step;      // (+ one cell: the f32 centre and radii)
                n_ry = sqrtf(2.f * NARROW_HALF * Pa * idet) + step;
            }
        }
    }
    const int wxlo = max(0, (int)ceilf(x0 * istep)), wxhi = min(grid - 1, (int)floorf(x1 * istep));
    const int wylo = max(0, (int)ceilf(y0 * istep)), wyhi = min(grid - 1, (int)floorf(y1 * istep));
    if (wxlo > wxhi || wylo > wyhi) return false;
    auto cell = [&](int ix, int iy) -> double {
        const float x = ix * step, y = iy * step;
        double v = (double)skew_pdf(A, x, y) * (double)gauss_pdf(B, x, y);
        if (Cg) v *= (double)gauss_pdf(*Cg, x, y);
        return v;
    };
    // Inverse CDF over the window [xlo, xhi] x [ylo, yhi].  delta > 0: the window is a NARROW box and delta bounds the mass of
    // the cells left out; the draw is accepted (certain = true) only if no prefix it was compared with lies within delta of the
    // threshold -- then the full window selects the same cell (see below).
    auto draw = [&](int xlo, int xhi, int ylo, int yhi, double delta, int& r_out, int& j_out, bool& certain) -> bool {
        const int nx = xhi - xlo + 1, ny = yhi - ylo + 1;
        int k = 1;                                   // lanes per row
        while (2 * k * nx <= 64) k <<= 1;
        const int rpp = 64 / k;                      // rows per pass
        const int sub = lane & (k - 1), rl = lane / k;
        double rs[4], cum[4];                        // row sum / inclusive prefix of the lane's row in pass p (lanes with sub == 0)
        double carry = 0.0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            rs[p] = 0.0; cum[p] = 0.0;
            if (p * rpp >= nx) continue;             // (uniform)
            const int ri = p * rpp + rl;
            double acc = 0.0;
            if (ri < nx)
                for (int iy = ylo + sub; iy <= yhi; iy += k) acc += cell(xlo + ri, iy);
            for (int off = 1; off < k; off <<= 1) acc += __shfl_xor(acc, off, 64);
            rs[p] = (sub == 0 && ri < nx) ? acc : 0.0;
            cum[p] = carry + wave_scan(rs[p], lane);
            carry = __shfl(cum[p], 63, 64);
        }
        const double total = carry;
        if (!(total > 0.0) || !(total < 1e300)) return false;
        const double T = (double)u * total;
        const double band = delta > 0.0 ? delta + 1e-12 * total : -1.0;     // (+ the rounding of the double sums)
        bool near = T <= band;                       // the rows before the window have prefix <= delta
        int r = 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (p * rpp >= nx) continue;
            const bool mine = sub == 0 && p * rpp + rl < nx;
            r += __popcll(__ballot(mine && cum[p] <= T));
            near = near || (mine && fabs(cum[p] - T) <= band);
        }
        r = min(r, nx - 1);
        // prefix before row r, from the lane that owns it
        const int pr = r / rpp, lr = (r - pr * rpp) * k;
        double mb = 0.0;
#pragma unroll
        for (int p = 0; p < 4; ++p) mb = p == pr ? cum[p] - rs[p] : mb;
        const double base = __shfl(mb, lr, 64);
        int j = 0;
        double carry2 = 0.0;
        for (int p = 0; p * 64 < ny; ++p) {
            const int iy = ylo + p * 64 + lane;
            const double v = iy <= yhi ? cell(xlo + r, iy) : 0.0;
            const double c2 = carry2 + wave_scan(v, lane);
            carry2 = __shfl(c2, 63, 64);
            j += __popcll(__ballot(iy <= yhi && base + c2 <= T));
            near = near || (iy <= yhi && fabs(base + c2 - T) <= band);
        }
        j = min(j, ny - 1);
        r_out = xlo + r; j_out = ylo + j;
        certain = __ballot(near) == 0;
        return true;
    };
    int r = 0, j = 0;
    bool certain = false;
    if (narrow_ok) {
        // Narrow box first.  N = bounding box of the ellipse (x - mu_f)^T P (x - mu_f) <= 2 NARROW_HALF: a cell of the full window W
        // outside N is outside the ellipse, so it is at most vmax = 2.0000002 exp(sum lc_i - r0 / 2 - NARROW_HALF), and the mass
        // left out is at most delta = |W| vmax.  With total_W = total_N + tail, 0 <= tail <= delta: the threshold u total moves up
        // by at most delta, and so does every prefix (the mass before a cell in the flat order gains the part of the tail that
        // precedes it).  A comparison `prefix <= threshold` can therefore only change if prefix_N is within delta of threshold_N;
        // when none is -- rows, the cells of the selected row, and the empty prefix of the rows before N -- the full window draws
        // the same cell.  Otherwise (u within ~1e-11 of a cell boundary) the full window is evaluated.
        const double vmax = 2.0000002 * exp((double)n_lcs - 0.5 * (double)n_r0 - (double)NARROW_HALF);
        const double delta = (double)(wxhi - wxlo + 1) * (double)(wyhi - wylo + 1) * vmax;
        const int nxlo = max(wxlo, (int)ceilf((n_fx - n_rx) * istep)), nxhi = min(wxhi, (int)floorf((n_fx + n_rx) * istep));
        const int nylo = max(wylo, (int)ceilf((n_fy - n_ry) * istep)), nyhi = min(wyhi, (int)floorf((n_fy + n_ry) * istep));
        if (nxlo <= nxhi && nylo <= nyhi && delta > 0.0)
            if (!draw(nxlo, nxhi, nylo, nyhi, delta, r, j, certain)) certain = false;
    }
    if (!certain && !draw(wxlo, wxhi, wylo, wyhi, 0.0, r, j, certain)) return false;
    sx = r * step; sy = j * step;
    return true;
}

constexpr int SPW = ST / 64;        // samples per workgroup

__global__ __launch_bounds__(ST) void psm_skew_kernel(const SkewArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    const int f = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x * SPW + wave;
    const int K = p.K, P = 2 * K;
    float* rec = lds_f;                                        // frame record: m | covc | G
    __shared__ int goff[MAXLV];
    __shared__ Skew2 skA[MAXP / 2];
    __shared__ Gauss2 gsC[SPW][MAXP / 2];
    __shared__ float cvC[SPW][MAXP / 2][2];

    const float* grec = p.rec + (size_t)f * p.rec_stride;
    for (int i = tid; i < p.rec_stride; i += ST) rec[i] = grec[i];
    const float* m = rec;
    const float* covc = rec + REC_COVC;
    const float* G = rec + REC_G;
    const float* mu_f = p.mu_pred + (size_t)f * P;
    const float* cv_f = p.cov_pred + (size_t)f * K * 3;
    const float* al_f = p.alpha + (size_t)f * P;
    const size_t sidx = (size_t)f * p.S + min(s, p.S - 1);
    if (tid < K) {
        const float a = cv_f[3 * tid], b = cv_f[3 * tid + 1], c = cv_f[3 * tid + 2];
        Skew2 k;
        k.g = make_gauss(mu_f[2 * tid], mu_f[2 * tid + 1], a, b, c);
        const float det = a * b - c * c, sd = sqrtf(det), t = sqrtf(a + b + 2.f * sd), ist = 1.f / (sd * t);
        const float al1 = al_f[2 * tid], al2 = al_f[2 * tid + 1] * p.alpha_y_sign;
        k.z1 = (al1 * (b + sd) - al2 * c) * ist;               // alpha^T Sigma^-1/2, Sigma^-1/2 = [[b+s,-c],[-c,a+s]]/(s t)
        k.z2 = (al2 * (a + sd) - al1 * c) * ist;
        skA[tid] = k;
    }
    if (p.prior_mu && lane < K) {                              // the extra Gaussian factor is per SAMPLE: one table per wave
        const float* pm = p.prior_mu + (sidx * K + lane) * 2;
        const float* pc = p.prior_cov + (sidx * K + lane) * 3;
        gsC[wave][lane] = make_gauss(pm[0], pm[1], pc[0], pc[1], pc[2]);
        cvC[wave][lane][0] = pc[0]; cvC[wave][lane][1] = pc[1];
    }
    if (tid == 0) {
        int off = 0;
        for (int l = 0; l < p.n_levels; ++l) {
            goff[l] = off;
            off += 2 * p.tables[l * TSTRIDE + 1] * p.tables[l * TSTRIDE];
        }
    }
    __syncthreads();
    if (s >= p.S) return;                                      // (no workgroup barrier below)
    auto normals = [&](int pt, int j, float& e0, float& e1) {
        if (p.eps) { e0 = p.eps[(sidx * K + pt) * 3 + 2 * j]; e1 = j ? 0.f : p.eps[(sidx * K + pt) * 3 + 1]; }
        else normal2(p.seed ^ ((sidx * 64ull + pt) * 4ull + j) * 0x9E3779B97F4A7C15ull, e0, e1);
    };
    auto uniform = [&](int pt) -> float {
        return p.u ? p.u[sidx * K + pt] : uniform01(p.seed ^ ((sidx * 64ull + pt) * 4ull + 3ull) * 0x9E3779B97F4A7C15ull);
    };
    float ct = 0.f;                                            // lane i < P: flat coordinate i of the contour, pixel units
    auto set_point = [&](int pt, float x, float y) {           // (x, y wave-uniform)
        ct = lane == 2 * pt ? x : (lane == 2 * pt + 1 ? y : ct);
    };
    const Gauss2* gC = gsC[wave];

    // ---- anchors
    if (!p.use_initial_pdf) {
        float ax = 0.f, ay = 0.f;
        if (lane < p.n_init) {      // BivariateSkewNormal.rvs_fast (bivariateskewnormal.py:159-191)
            const int pt = p.init_pts[lane];
            const float a = cv_f[3 * pt], b = cv_f[3 * pt + 1], c = cv_f[3 * pt + 2];
            const float al1 = al_f[2 * pt], al2 = al_f[2 * pt + 1] * p.alpha_y_sign;
            const float sa1 = a * al1 + c * al2, sa2 = c * al1 + b * al2;
            const float nrm = 1.f / sqrtf(1.f + al1 * sa1 + al2 * sa2);
            const float l21 = sa1 * nrm, l31 = sa2 * nrm;
            const float l22 = sqrtf(fmaxf(a - l21 * l21, 0.f));
            const float l32 = (c - l31 * l21) / l22;
            const float l33 = sqrtf(fmaxf(b - l31 * l31 - l32 * l32, 0.f));
            float e0, e1, e2, dummy;
            normals(pt, 0, e0, e1);
            normals(pt, 1, e2, dummy);
            float x1 = l21 * e0 + l22 * e1, x2 = l31 * e0 + l32 * e1 + l33 * e2;
            if (e0 <= 0.f) { x1 = -x1; x2 = -x2; }
            ax = x1 + mu_f[2 * pt];
            ay = x2 + mu_f[2 * pt + 1];
        }
        for (int i = 0; i < p.n_init; ++i) set_point(p.init_pts[i], __shfl(ax, i, 64), __shfl(ay, i, 64));
    } else {                       // numerical_sample of the supplied pdfs (psm_skew.py:450-466): skew x prior
        for (int i = 0; i < p.n_init; ++i) {
            const int pt = p.init_pts[i];
            float sx, sy;
            const bool ok = grid_sample(skA[pt], gC[pt], nullptr, cvC[wave][pt][0], cvC[wave][pt][1], 0.f, 0.f, p.grid,
                                        uniform(pt), lane, sx, sy, p.merged_window);
            set_point(pt, ok ? sx : gC[pt].mx, ok ? sy : gC[pt].my);
        }
    }

    // ---- levels
    for (int l = 0; l < p.n_levels; ++l) {
        const int* tb = p.tables + l * TSTRIDE;
        const int ng = tb[0], nt = tb[1];
        const int* gi = tb + 2;
        const int* tp = tb + 2 + MAXG;
        const float* Gl = G + goff[l];
        // conditional means of the level's points: lane < 2 nt owns one coordinate (2 nt <= 2 MAXT = 64)
        const bool own = lane < 2 * nt;
        const int row = own ? 2 * tp[lane >> 1] + (lane & 1) : 0;
        float acc = m[row];
        for (int j = 0; j < ng; ++j) {
            const int gj = gi[j];
            const float cg = __shfl(ct, gj, 64);
            if (own) acc += Gl[lane * ng + j] * ((cg - p.smean[gj]) / p.sscale[gj] - m[gj]);
        }
        const float mcl = acc * p.sscale[row] + p.smean[row];
        if (!p.sample_level[l]) {
            for (int q = 0; q < nt; ++q) set_point(tp[q], __shfl(mcl, 2 * q, 64), __shfl(mcl, 2 * q + 1, 64));
            continue;
        }
        for (int q = 0; q < nt; ++q) {
            const int pt = tp[q];
            const float mcx = __shfl(mcl, 2 * q, 64), mcy = __shfl(mcl, 2 * q + 1, 64);
            const float* cc = covc + pt * 4;
            if ((p.skew_bits >> pt) & 1ull) {
                // p2 = exp(MultivariateNormal(mu_c, cov_c).log_prob): the Cholesky factor reads the lower triangle
                const Gauss2 B = make_gauss(mcx, mcy, cc[0], cc[3], cc[2]);
                float sx, sy;
                const bool ok = grid_sample(skA[pt], B, p.prior_mu ? &gC[pt] : nullptr, cc[0], cc[3], cvC[wave][pt][0],
                                            cvC[wave][pt][1], p.grid, uniform(pt), lane, sx, sy, p.merged_window);
                set_point(pt, ok ? sx : mcx, ok ? sy : mcy);
            } else {                 // Gaussian point: product-of-Gaussians merge + draw (psm.py:424-440, 387-421)
                const float a1 = cv_f[3 * pt], b1 = cv_f[3 * pt + 1], c1 = cv_f[3 * pt + 2];
                const float s1[2][2] = {{a1, c1}, {c1, b1}};
                const float sc[2][2] = {{cc[0], cc[1]}, {cc[2], cc[3]}};
                float m1[2][2], m2[2][2], sf[2][2];
                merge2x2(s1, sc, m1, m2, sf);
                const float mu1x = mu_f[2 * pt], mu1y = mu_f[2 * pt + 1];
                const float mfx = m1[0][0] * mcx + m1[0][1] * mcy + m2[0][0] * mu1x + m2[0][1] * mu1y;
                const float mfy = m1[1][0] * mcx + m1[1][1] * mcy + m2[1][0] * mu1x + m2[1][1] * mu1y;
                const float l11 = sqrtf(sf[0][0]), l21 = sf[1][0] / l11, l22 = sqrtf(fmaxf(sf[1][1] - l21 * l21, 0.f));
                float e0, e1;
                normals(pt, 0, e0, e1);
                set_point(pt, mfx + l11 * e0, mfy + l21 * e0 + l22 * e1);
            }
        }
    }
    if (lane < P) p.out[sidx * P + lane] = ct;
}

// ---- conditional mean of a 1-level record given sampled contours (+ merge with the predictions) ----------------------
struct CondArgs {
    const float* rec; int rec_stride; int per_rec;       // sample i uses record i / per_rec
    const int* table;                                    // one level row
    const float* known;                                  // [N][P] pixel units (only the conditioning entries are read)
    const float* smean; const float* sscale;
    const float* mu_p; const float* cov_p;               // optional predictions [R][P], [R][P/2][3]
    float* mu_c; float* cov_c; float* mu_f; float* cov_f;   // [N][nt][2], [R][nt][4], [N][nt][2], [R][nt][4]
    int N, P;
};

__global__ __launch_bounds__(ST) void psm_condition_kernel(const CondArgs p) {
    const int ng = p.table[0], nt = p.table[1];
    const int* gi = p.table + 2;
    const int* tp = p.table + 2 + MAXG;
    const size_t i = (size_t)blockIdx.x * ST + threadIdx.x;
    if (i >= (size_t)p.N * nt) return;
    const int n = (int)(i / nt), q = (int)(i - (size_t)n * nt), pt = tp[q], r = n / p.per_rec;
    const float* rec = p.rec + (size_t)r * p.rec_stride;
    const float* m = rec;
    const float* cc = rec + REC_COVC + pt * 4;
    const float* Gl = rec + REC_G;
    const float* kn = p.known + (size_t)n * p.P;
    float mc[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        float acc = m[2 * pt + a];
        for (int j = 0; j < ng; ++j) {
            const int gj = gi[j];
            acc += Gl[(2 * q + a) * ng + j] * ((kn[gj] - p.smean[gj]) / p.sscale[gj] - m[gj]);
        }
        mc[a] = acc * p.sscale[2 * pt + a] + p.smean[2 * pt + a];
    }
    p.mu_c[i * 2] = mc[0]; p.mu_c[i * 2 + 1] = mc[1];
    const bool first = (n % p.per_rec) == 0;
    if (first)
        for (int e = 0; e < 4; ++e) p.cov_c[((size_t)r * nt + q) * 4 + e] = cc[e];
    if (p.mu_p) {
        const float* cp = p.cov_p + ((size_t)r * (p.P / 2) + pt) * 3;
        const float s1[2][2] = {{cp[0], cp[2]}, {cp[2], cp[1]}};
        const float sc[2][2] = {{cc[0], cc[1]}, {cc[2], cc[3]}};
        float m1[2][2], m2[2][2], sf[2][2];
        merge2x2(s1, sc, m1, m2, sf);
        const float mu1x = p.mu_p[(size_t)r * p.P + 2 * pt], mu1y = p.mu_p[(size_t)r * p.P + 2 * pt + 1];
        p.mu_f[i * 2] = m1[0][0] * mc[0] + m1[0][1] * mc[1] + m2[0][0] * mu1x + m2[0][1] * mu1y;
        p.mu_f[i * 2 + 1] = m1[1][0] * mc[0] + m1[1][1] * mc[1] + m2[1][0] * mu1x + m2[1][1] * mu1y;
        if (first) {
            float* o = p.cov_f + ((size_t)r * nt + q) * 4;
            o[0] = sf[0][0]; o[1] = sf[0][1]; o[2] = sf[1][0]; o[3] = sf[1][1];
        }
    }
}

}  // namespace

static void fill_setup(SetupArgs& su, const float* mu_pred, const float* cov0, const float* xbar, const float* smean,
                       const float* sscale, const int* tables, int P, int n_levels, const float* sigma2) {
    su.mu_pred = mu_pred; su.cov0 = cov0; su.xbar = xbar; su.smean = smean; su.sscale = sscale; su.tables = tables;
    su.P = P; su.n_levels = n_levels;
    for (int l = 0; l < n_levels; ++l) su.sigma2[l] = sigma2[l];
}

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
    fill_setup(a.su, mu_pred, cov0, xbar, smean, sscale, tables, 2 * K, n_levels, sigma2);
    a.cov_pred = cov_pred; a.eps = eps; a.out = out; a.F = F; a.S = S; a.K = K; a.n_init = n_init;
    for (int i = 0; i < n_init; ++i) a.init_pts[i] = init_pts[i];         // host arrays (small)
    for (int l = 0; l < n_levels; ++l) a.sample_level[l] = sample_level[l];
    a.seed = seed;
    const size_t lds = sizeof(double) * ((size_t)MAXP * MAXP + (size_t)MAXG * 2 * MAXG) +
                       sizeof(float) * (MAXP + (size_t)MAXLV * MAXT * MAXP + 48 * 4 + MAXT * 11 + MAXT * 3 + (size_t)MAXP * ST);
    CU_CHECK_ARG(lds <= 160 * 1024, "cu_psm_sample_gauss: LDS %zu too large", lds);
    auto k = psm_gauss_kernel;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    CU_CHECK_ARG(e == hipSuccess, "cu_psm_sample_gauss: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k, dim3(F), dim3(ST), lds, reinterpret_cast<hipStream_t>(stream), a);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_psm_record_floats(int n_levels, const int* ng, const int* nt) {
    int g = 0;
    for (int l = 0; l < n_levels; ++l) g += 2 * nt[l] * ng[l];
    return REC_G + g;
}

extern "C" int cu_psm_setup(int F, int P, const float* mu_pred, const float* cov0, const float* xbar, const float* smean,
                            const float* sscale, int n_levels, const int* tables, const float* sigma2, float* rec,
                            int rec_stride, void* stream) {
    CU_CHECK_ARG(F > 0 && P > 0 && P <= 96 && (P & 1) == 0, "cu_psm_setup: bad sizes F=%d P=%d", F, P);
    CU_CHECK_ARG(n_levels > 0 && n_levels <= MAXLV && rec_stride >= REC_G, "cu_psm_setup: bad level count / record stride");
    CU_CHECK_ARG(mu_pred && cov0 && xbar && smean && sscale && tables && sigma2 && rec, "cu_psm_setup: null pointer");
    SetupArgs a;
    memset(&a, 0, sizeof(a));
    fill_setup(a, mu_pred, cov0, xbar, smean, sscale, tables, P, n_levels, sigma2);
    const int MP = P <= 48 ? 48 : 96;
    const size_t lds = sizeof(double) * ((size_t)MP * MP + (size_t)MAXG * 2 * MAXG);
    auto k = P <= 48 ? psm_setup_kernel<48> : psm_setup_kernel<96>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    CU_CHECK_ARG(e == hipSuccess, "cu_psm_setup: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k, dim3(F), dim3(ST), lds, reinterpret_cast<hipStream_t>(stream), a, rec, rec_stride);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_psm_sample_skew(int F, int S, int K, const float* mu_pred, const float* cov_pred, const float* alpha,
                                  float alpha_y_sign, uint64_t skew_bits, const float* rec, int rec_stride,
                                  const float* smean, const float* sscale, int n_init, const int* init_pts, int n_levels,
                                  const int* tables, const int* sample_level, const float* prior_mu,
                                  const float* prior_cov, int use_initial_pdf, int grid, const float* eps, const float* u,
                                  uint64_t seed, float* out, void* stream) {
    CU_CHECK_ARG(F > 0 && S > 0 && K > 0 && 2 * K <= MAXP && F <= 65535, "cu_psm_sample_skew: bad sizes F=%d S=%d K=%d", F, S, K);
    CU_CHECK_ARG(n_init > 0 && n_init <= 8 && n_levels > 0 && n_levels <= MAXLV, "cu_psm_sample_skew: bad level counts");
    CU_CHECK_ARG(grid >= 2 && grid <= ST, "cu_psm_sample_skew: grid size %d not in [2, %d]", grid, ST);
    CU_CHECK_ARG(rec_stride >= REC_G && rec_stride <= 16384, "cu_psm_sample_skew: bad record stride %d", rec_stride);
    CU_CHECK_ARG(mu_pred && cov_pred && alpha && rec && smean && sscale && init_pts && tables && sample_level && out,
                 "cu_psm_sample_skew: null pointer");
    CU_CHECK_ARG((prior_mu == nullptr) == (prior_cov == nullptr) && (!use_initial_pdf || prior_mu),
                 "cu_psm_sample_skew: use_initial_pdf needs the prior factor");
    SkewArgs a;
    memset(&a, 0, sizeof(a));
    a.mu_pred = mu_pred; a.cov_pred = cov_pred; a.alpha = alpha; a.rec = rec; a.rec_stride = rec_stride;
    a.smean = smean; a.sscale = sscale; a.tables = tables; a.prior_mu = prior_mu; a.prior_cov = prior_cov;
    a.eps = eps; a.u = u; a.out = out; a.F = F; a.S = S; a.K = K; a.n_init = n_init; a.n_levels = n_levels;
    a.grid = grid; a.use_initial_pdf = use_initial_pdf; a.skew_bits = skew_bits; a.alpha_y_sign = alpha_y_sign;
    a.seed = seed;
    a.merged_window = cu_env_int("CU_PSM_MERGED_WINDOW", 2);     // 0: conditional Gaussian's window, 1: + product, 2: + narrow box first
    for (int i = 0; i < n_init; ++i) a.init_pts[i] = init_pts[i];
    for (int l = 0; l < n_levels; ++l) a.sample_level[l] = sample_level[l];
    const size_t lds = sizeof(float) * (size_t)rec_stride;
    auto k = psm_skew_kernel;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    CU_CHECK_ARG(e == hipSuccess, "cu_psm_sample_skew: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k, dim3((S + SPW - 1) / SPW, F), dim3(ST), lds, reinterpret_cast<hipStream_t>(stream), a);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_psm_condition(int N, int P, int per_rec, const float* rec, int rec_stride, const int* table,
                                int nt, const float* known, const float* smean, const float* sscale, const float* mu_p,
                                const float* cov_p, float* mu_c, float* cov_c, float* mu_f, float* cov_f, void* stream) {
    CU_CHECK_ARG(N > 0 && P > 0 && P <= 96 && per_rec > 0 && nt > 0 && nt <= MAXT, "cu_psm_condition: bad sizes");
    CU_CHECK_ARG(rec && table && known && smean && sscale && mu_c && cov_c, "cu_psm_condition: null pointer");
    CU_CHECK_ARG((mu_p == nullptr) == (cov_p == nullptr) && (!mu_p || (mu_f && cov_f)), "cu_psm_condition: merge needs mu_p, cov_p, mu_f, cov_f");
    CondArgs a;
    a.rec = rec; a.rec_stride = rec_stride; a.per_rec = per_rec; a.table = table; a.known = known; a.smean = smean;
    a.sscale = sscale; a.mu_p = mu_p; a.cov_p = cov_p; a.mu_c = mu_c; a.cov_c = cov_c; a.mu_f = mu_f; a.cov_f = cov_f;
    a.N = N; a.P = P;
    const size_t total = (size_t)N * nt;
    hipLaunchKernelGGL(psm_condition_kernel, dim3((unsigned)((total + ST - 1) / ST)), dim3(ST), 0,
                       reinterpret_cast<hipStream_t>(stream), a);
    CU_LAUNCH_CHECK();
    return 0;
}
