// InstanceNorm2d(affine) + LeakyReLU in fused form (reference models/nnUnet/layers.py:193-194,203-204), NHWC.
//
// Forward: only the per-(image,channel) statistics are computed here (one streaming read of the raw conv output);
// the normalise + affine + LeakyReLU is applied by the consumer while it loads its operand (igemm_conv.hip /
// igemm_wgrad.hip), so the activated tensor never exists in HBM.
// Backward: one reduction pass (A1 = sum g*l', A2 = sum g*l'*xhat) and one in-place pass that turns dL/d(activated)
// into dL/dz; d(gamma), d(beta) and d(conv bias) fall out of the same passes.
// All kernels are HBM-bound streaming reductions: 16-byte loads, one channel piece per thread, LDS tree, few atomics.
#include "common.h"

namespace {

constexpr int NT = 256;

// thread -> (channel piece, pixel row) mapping shared by the streaming kernels
struct RowMap {
    int tpp;     // threads (pieces) per pixel
    int rows;    // pixel rows handled per iteration
};
static inline RowMap row_map(int C, int piece) {
    RowMap m;
    m.tpp = C / piece;
    m.rows = NT / m.tpp;
    if (m.rows < 1) m.rows = 1;
    return m;
}

// Block-level reduction over pixel rows of per-thread partials v[NV][PIECE]; result lands in thread (prow == 0).
// When tpp (threads per pixel) divides the wave, lanes l, l+tpp, l+2tpp ... hold the same channel piece: butterfly
// shuffles reduce the wave, then 4 LDS rows combine the waves.  (The first version let `tpp` threads sum up to 64 LDS
// rows serially, which dominated these HBM-bound kernels.)
template <int NV, int PIECE>
__device__ __forceinline__ void reduce_rows(float (&v)[NV][PIECE], float* lds, int piece, int prow, int rows, int tpp,
                                            bool active) {
    const int stride = NV * PIECE;
    const bool pow2 = (tpp & (tpp - 1)) == 0 && tpp <= 32;
    if (pow2) {       // uniform; every thread of the block is active when tpp divides 256
        for (int off = tpp; off < 64; off <<= 1) {
#pragma unroll
            for (int a = 0; a < NV; ++a)
#pragma unroll
                for (int e = 0; e < PIECE; ++e) v[a][e] += __shfl_xor(v[a][e], off, 64);
        }
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane < tpp) {
            float* d = lds + ((size_t)wave * tpp + lane) * stride;
#pragma unroll
            for (int a = 0; a < NV; ++a)
#pragma unroll
                for (int e = 0; e < PIECE; ++e) d[a * PIECE + e] = v[a][e];
        }
        __syncthreads();
        if (prow == 0) {      // threads 0 .. tpp-1 (wave 0, lane == piece)
#pragma unroll
            for (int a = 0; a < NV; ++a)
#pragma unroll
                for (int e = 0; e < PIECE; ++e) v[a][e] = 0.f;
            for (int w = 0; w < NT / 64; ++w) {
                const float* s = lds + ((size_t)w * tpp + piece) * stride;
#pragma unroll
                for (int a = 0; a < NV; ++a)
#pragma unroll
                    for (int e = 0; e < PIECE; ++e) v[a][e] += s[a * PIECE + e];
            }
        }
        return;
    }
    // general case (C = 480: 60 or 120 pieces, 4 or 2 rows): lds[rows][tpp][NV*PIECE]
    if (active) {
        float* d = lds + ((size_t)prow * tpp + piece) * stride;
#pragma unroll
        for (int a = 0; a < NV; ++a)
#pragma unroll
            for (int e = 0; e < PIECE; ++e) d[a * PIECE + e] = v[a][e];
    }
    __syncthreads();
    if (active && prow == 0) {
        for (int rr = 1; rr < rows; ++rr) {
            const float* s = lds + ((size_t)rr * tpp + piece) * stride;
#pragma unroll
            for (int a = 0; a < NV; ++a)
#pragma unroll
                for (int e = 0; e < PIECE; ++e) v[a][e] += s[a * PIECE + e];
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward statistics
template <typename T>
__global__ __launch_bounds__(NT) void stats_partial_kernel(const T* __restrict__ z, float* __restrict__ ws, int HW, int C,
                                                           int tpp, int rows, int chunk) {
    constexpr int PIECE = Elem<T>::PIECE;
    extern __shared__ float lds[];
    const int n = blockIdx.x;
    const int p0 = blockIdx.y * chunk;
    const int p1 = min(HW, p0 + chunk);
    const int piece = threadIdx.x % tpp, prow = threadIdx.x / tpp;
    const bool active = prow < rows && piece < tpp;
    float acc[2][PIECE];
#pragma unroll
    for (int e = 0; e < PIECE; ++e) acc[0][e] = acc[1][e] = 0.f;
    if (active) {
        const T* base = z + (size_t)n * HW * C + piece * PIECE;
        float k[PIECE];
        load_piece<T>(base, k);   // shift by the image's first pixel: avoids E[x^2]-E[x]^2 cancellation
        // 4 independent 16-byte loads in flight per thread (the loop is latency-bound otherwise)
        int p = p0 + prow;
        for (; p + 3 * rows < p1; p += 4 * rows) {
            float v[4][PIECE];
#pragma unroll
            for (int u = 0; u < 4; ++u) load_piece<T>(base + (size_t)(p + u * rows) * C, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < PIECE; ++e) {
                    const float d = v[u][e] - k[e];
                    acc[0][e] += d;
                    acc[1][e] += d * d;
                }
        }
        for (; p < p1; p += rows) {
            float v[PIECE];
            load_piece<T>(base + (size_t)p * C, v);
#pragma unroll
            for (int e = 0; e < PIECE; ++e) {
                const float d = v[e] - k[e];
                acc[0][e] += d;
                acc[1][e] += d * d;
            }
        }
    }
    reduce_rows<2, PIECE>(acc, lds, piece, prow, rows, tpp, active);
    if (active && prow == 0) {
        float* o = ws + ((size_t)n * C + piece * PIECE) * 2;
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            unsafeAtomicAdd(o + 2 * e, acc[0][e]);
            unsafeAtomicAdd(o + 2 * e + 1, acc[1][e]);
        }
    }
}

template <typename T>
__global__ void stats_finalize_kernel(const T* __restrict__ z, const float* __restrict__ ws,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                      float* __restrict__ stats, int N, int HW, int C, int nimg) {
    // N = images of the whole tensor (plane stride of stats), nimg = images of this launch
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nimg * C) return;
    const int n = i / C, c = i - n * C;
    const float k = Elem<T>::ld(z + (size_t)n * HW * C + c);
    const float inv = 1.f / (float)HW;
    const float m1 = ws[2 * i] * inv, m2 = ws[2 * i + 1] * inv;
    const float mean = k + m1;
    const float var = fmaxf(m2 - m1 * m1, 0.f);
    const float rstd = 1.f / sqrtf(var + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const size_t NC = (size_t)N * C;
    stats[i] = mean;
    stats[NC + i] = rstd;
    stats[2 * NC + i] = g * rstd;
    stats[3 * NC + i] = b - mean * g * rstd;
}

// the same from sums a convolution epilogue gathered (tconv.hip): sums[n][c] = {sum, sum of squares} of (z - shift[c]) in
// f32 BEFORE the rounding of z to its storage type (shift = the conv bias, or NULL = 0)
__global__ void stats_finalize_given_kernel(const float* __restrict__ sums, const float* __restrict__ shift,
                                            const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                            float* __restrict__ stats, int N, int HW, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int c = i % C;
    const float inv = 1.f / (float)HW;
    const float m1 = sums[2 * i] * inv, m2 = sums[2 * i + 1] * inv;
    const float mean = (shift ? shift[c] : 0.f) + m1;
    const float var = fmaxf(m2 - m1 * m1, 0.f);
    const float rstd = 1.f / sqrtf(var + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const size_t NC = (size_t)N * C;
    stats[i] = mean;
    stats[NC + i] = rstd;
    stats[2 * NC + i] = g * rstd;
    stats[3 * NC + i] = b - mean * g * rstd;
}

// materialise a = LeakyReLU(z*scale + shift) (one streaming pass): the MFMA kernels then stage plain operands.
// Measured on MI355X: re-doing this affine + activation inside every consumer's operand load made the thin-channel
// convolutions VALU-bound (40-50 % of their time); one extra 2-byte write + read per element is far cheaper.
// GIVEN (round 4): the statistics are finalised HERE from sums a convolution epilogue gathered -- the arithmetic of
// stats_finalize_given_kernel, expression for expression, so the values are bit-identical to the two-launch form -- and the
// first pixel chunk of every image stores them for the backward: one launch per layer less (18 per step).
struct GivenSums {
    const float* sums;     // [N][C][2] = {sum, sum of squares} of (z - shift[c])
    const float* shift;    // conv bias or NULL
    const float* gamma;
    const float* beta;
    float eps;
    float* stats_out;      // [4][N][C]
};

template <typename T, bool GIVEN = false>
__global__ __launch_bounds__(NT) void apply_kernel(const T* __restrict__ z, const float* __restrict__ stats, float slope,
                                                   T* __restrict__ out, int N, int HW, int C, int tpp, int rows,
                                                   int chunk, GivenSums gs = GivenSums()) {
    constexpr int PIECE = Elem<T>::PIECE;
    const int n = blockIdx.x;
    const int p0 = blockIdx.y * chunk, p1 = min(HW, p0 + chunk);
    const int piece = threadIdx.x % tpp, prow = threadIdx.x / tpp;
    if (prow >= rows) return;
    const size_t NC = (size_t)N * C;
    const size_t sidx = (size_t)n * C + piece * PIECE;
    float sc[PIECE], sh[PIECE];
    if constexpr (GIVEN) {
        const float inv = 1.f / (float)HW;
        const bool keep = blockIdx.y == 0 && prow == 0;
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            const size_t i = sidx + e;
            const int c = piece * PIECE + e;
            const float m1 = gs.sums[2 * i] * inv, m2 = gs.sums[2 * i + 1] * inv;
            const float mean = (gs.shift ? gs.shift[c] : 0.f) + m1;
            const float var = fmaxf(m2 - m1 * m1, 0.f);
            const float rstd = 1.f / sqrtf(var + gs.eps);
            const float g = gs.gamma ? gs.gamma[c] : 1.f, b = gs.beta ? gs.beta[c] : 0.f;
            sc[e] = g * rstd;
            sh[e] = b - mean * g * rstd;
            if (keep) {
                gs.stats_out[i] = mean;
                gs.stats_out[NC + i] = rstd;
                gs.stats_out[2 * NC + i] = sc[e];
                gs.stats_out[3 * NC + i] = sh[e];
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < PIECE; ++e) { sc[e] = stats[2 * NC + sidx + e]; sh[e] = stats[3 * NC + sidx + e]; }
    }
    const size_t base = (size_t)n * HW * C + piece * PIECE;
    int p = p0 + prow;
    for (; p + 3 * rows < p1; p += 4 * rows) {
        float v[4][PIECE];
#pragma unroll
        for (int u = 0; u < 4; ++u) load_piece<T>(z + base + (size_t)(p + u * rows) * C, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int e = 0; e < PIECE; ++e) {
                const float y = v[u][e] * sc[e] + sh[e];
                v[u][e] = y > 0.f ? y : y * slope;
            }
            store_piece<T>(out + base + (size_t)(p + u * rows) * C, v[u]);
        }
    }
    for (; p < p1; p += rows) {
        float v[PIECE];
        load_piece<T>(z + base + (size_t)p * C, v);
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            const float y = v[e] * sc[e] + sh[e];
            v[e] = y > 0.f ? y : y * slope;
        }
        store_piece<T>(out + base + (size_t)p * C, v);
    }
}

// ------------------------------------------------------------------------------------------------ backward
template <typename T>
__global__ __launch_bounds__(NT) void bwd_reduce_kernel(const T* __restrict__ g, const T* __restrict__ z,
                                                        const float* __restrict__ stats, float slope,
                                                        float* __restrict__ ws, int N, int HW, int C, int tpp, int rows,
                                                        int chunk) {
    constexpr int PIECE = Elem<T>::PIECE;
    extern __shared__ float lds[];
    const int n = blockIdx.x;
    const int p0 = blockIdx.y * chunk, p1 = min(HW, p0 + chunk);
    const int piece = threadIdx.x % tpp, prow = threadIdx.x / tpp;
    const bool active = prow < rows;
    const size_t NC = (size_t)N * C;
    float acc[2][PIECE];
#pragma unroll
    for (int e = 0; e < PIECE; ++e) acc[0][e] = acc[1][e] = 0.f;
    if (active) {
        const size_t sidx = (size_t)n * C + piece * PIECE;
        float mean[PIECE], rstd[PIECE], sc[PIECE], sh[PIECE];
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            mean[e] = stats[sidx + e]; rstd[e] = stats[NC + sidx + e];
            sc[e] = stats[2 * NC + sidx + e]; sh[e] = stats[3 * NC + sidx + e];
        }
        const size_t base = (size_t)n * HW * C + piece * PIECE;
        int p = p0 + prow;
        for (; p + rows < p1; p += 2 * rows) {
            float zv[2][PIECE], gv[2][PIECE];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                load_piece<T>(z + base + (size_t)(p + u * rows) * C, zv[u]);
                load_piece<T>(g + base + (size_t)(p + u * rows) * C, gv[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int e = 0; e < PIECE; ++e) {
                    const float y = zv[u][e] * sc[e] + sh[e];
                    const float gl = y > 0.f ? gv[u][e] : gv[u][e] * slope;
                    acc[0][e] += gl;
                    acc[1][e] += gl * (zv[u][e] - mean[e]) * rstd[e];
                }
        }
        for (; p < p1; p += rows) {
            float zv[PIECE], gv[PIECE];
            load_piece<T>(z + base + (size_t)p * C, zv);
            load_piece<T>(g + base + (size_t)p * C, gv);
#pragma unroll
            for (int e = 0; e < PIECE; ++e) {
                const float y = zv[e] * sc[e] + sh[e];
                const float gl = y > 0.f ? gv[e] : gv[e] * slope;
                acc[0][e] += gl;
                acc[1][e] += gl * (zv[e] - mean[e]) * rstd[e];
            }
        }
    }
    reduce_rows<2, PIECE>(acc, lds, piece, prow, rows, tpp, active);
    if (active && prow == 0) {
        float* o = ws + ((size_t)n * C + piece * PIECE) * 2;
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            unsafeAtomicAdd(o + 2 * e, acc[0][e]);
            unsafeAtomicAdd(o + 2 * e + 1, acc[1][e]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void bwd_apply_kernel(T* __restrict__ g, const T* __restrict__ z,
                                                       const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       float slope, const float* __restrict__ ws,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                       float* __restrict__ dbias, int N, int HW, int C, int tpp, int rows,
                                                       int chunk) {
    constexpr int PIECE = Elem<T>::PIECE;
    extern __shared__ float lds[];
    const int n = blockIdx.x;
    const int p0 = blockIdx.y * chunk, p1 = min(HW, p0 + chunk);
    const int piece = threadIdx.x % tpp, prow = threadIdx.x / tpp;
    const bool active = prow < rows;
    const size_t NC = (size_t)N * C;
    float acc[1][PIECE];
#pragma unroll
    for (int e = 0; e < PIECE; ++e) acc[0][e] = 0.f;
    if (active) {
        const size_t sidx = (size_t)n * C + piece * PIECE;
        float mean[PIECE], rstd[PIECE], sc[PIECE], sh[PIECE], a1[PIECE], a2[PIECE], gr[PIECE];
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            mean[e] = stats[sidx + e]; rstd[e] = stats[NC + sidx + e];
            sc[e] = stats[2 * NC + sidx + e]; sh[e] = stats[3 * NC + sidx + e];
            a1[e] = ws[(sidx + e) * 2] * inv; a2[e] = ws[(sidx + e) * 2 + 1] * inv;
            gr[e] = (gamma ? gamma[piece * PIECE + e] : 1.f) * rstd[e];
        }
        if (blockIdx.y == 0 && prow == 0) {   // one contribution per (image, channel)
#pragma unroll
            for (int e = 0; e < PIECE; ++e) {
                if (dbeta) unsafeAtomicAdd(dbeta + piece * PIECE + e, ws[(sidx + e) * 2]);
                if (dgamma) unsafeAtomicAdd(dgamma + piece * PIECE + e, ws[(sidx + e) * 2 + 1]);
            }
        }
        const size_t base = (size_t)n * HW * C + piece * PIECE;
        int p = p0 + prow;
        for (; p + rows < p1; p += 2 * rows) {
            float zv[2][PIECE], gv[2][PIECE];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                load_piece<T>(z + base + (size_t)(p + u * rows) * C, zv[u]);
                load_piece<T>(g + base + (size_t)(p + u * rows) * C, gv[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int e = 0; e < PIECE; ++e) {
                    const float y = zv[u][e] * sc[e] + sh[e];
                    const float gl = y > 0.f ? gv[u][e] : gv[u][e] * slope;
                    const float xh = (zv[u][e] - mean[e]) * rstd[e];
                    gv[u][e] = gr[e] * (gl - a1[e] - xh * a2[e]);
                    acc[0][e] += gv[u][e];
                }
                store_piece<T>(g + base + (size_t)(p + u * rows) * C, gv[u]);
            }
        }
        for (; p < p1; p += rows) {
            float zv[PIECE], gv[PIECE];
            load_piece<T>(z + base + (size_t)p * C, zv);
            load_piece<T>(g + base + (size_t)p * C, gv);
#pragma unroll
            for (int e = 0; e < PIECE; ++e) {
                const float y = zv[e] * sc[e] + sh[e];
                const float gl = y > 0.f ? gv[e] : gv[e] * slope;
                const float xh = (zv[e] - mean[e]) * rstd[e];
                gv[e] = gr[e] * (gl - a1[e] - xh * a2[e]);
                acc[0][e] += gv[e];
            }
            store_piece<T>(g + base + (size_t)p * C, gv);
        }
    }
    if (dbias) {
        reduce_rows<1, PIECE>(acc, lds, piece, prow, rows, tpp, active);
        if (active && prow == 0) {
#pragma unroll
            for (int e = 0; e < PIECE; ++e) unsafeAtomicAdd(dbias + piece * PIECE + e, acc[0][e]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ small feature maps
// HW <= 1024 (32x32 and below): one workgroup owns ALL pixels of one image for a slice of 8 pieces (32 pixel rows x 8
// pieces), so the statistics / the backward sums never leave the workgroup: one launch instead of memset + partial +
// finalize (forward) or memset + reduce + apply (backward), and no atomics on the per-(image, channel) sums.
constexpr int SG = 8;             // pieces per workgroup
constexpr int SR = NT / SG;       // pixel rows per workgroup (32)

template <typename T>
__global__ __launch_bounds__(NT) void stats_small_kernel(const T* __restrict__ z, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float eps,
                                                         float* __restrict__ stats, int N, int HW, int C) {
    constexpr int PIECE = Elem<T>::PIECE;
    __shared__ float lds[(NT / 64) * SG * 2 * PIECE];
    const int n = blockIdx.x;
    const int pl = threadIdx.x & (SG - 1), prow = threadIdx.x / SG;
    const int piece = blockIdx.y * SG + pl;
    const bool active = piece * PIECE < C;
    float acc[2][PIECE], k[PIECE];
#pragma unroll
    for (int e = 0; e < PIECE; ++e) acc[0][e] = acc[1][e] = k[e] = 0.f;
    if (active) {
        const T* base = z + (size_t)n * HW * C + piece * PIECE;
        load_piece<T>(base, k);   // shift by the image's first pixel: avoids E[x^2]-E[x]^2 cancellation
        // 4 independent 16-byte loads in flight per thread: one workgroup per (image, 8 pieces) leaves nothing else to
        // hide the memory latency behind (32x32: 32 dependent round trips, 19 us for a 7 us read)
        int p = prow;
        for (; p + 3 * SR < HW; p += 4 * SR) {
            float v[4][PIECE];
#pragma unroll
            for (int u = 0; u < 4; ++u) load_piece<T>(base + (size_t)(p + u * SR) * C, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < PIECE; ++e) {
                    const float d = v[u][e] - k[e];
                    acc[0][e] += d;
                    acc[1][e] += d * d;
                }
        }
        for (; p < HW; p += SR) {
            float v[PIECE];
            load_piece<T>(base + (size_t)p * C, v);
#pragma unroll
            for (int e = 0; e < PIECE; ++e) {
                const float d = v[e] - k[e];
                acc[0][e] += d;
                acc[1][e] += d * d;
            }
        }
    }
    reduce_rows<2, PIECE>(acc, lds, pl, prow, SR, SG, true);
    if (active && prow == 0) {
        const size_t NC = (size_t)N * C;
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            const int c = piece * PIECE + e;
            const size_t i = (size_t)n * C + c;
            const float m1 = acc[0][e] * inv, m2 = acc[1][e] * inv;
            const float mean = k[e] + m1;
            const float rstd = 1.f / sqrtf(fmaxf(m2 - m1 * m1, 0.f) + eps);
            const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
            stats[i] = mean;
            stats[NC + i] = rstd;
            stats[2 * NC + i] = g * rstd;
            stats[3 * NC + i] = b - mean * g * rstd;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void bwd_small_kernel(T* __restrict__ g, const T* __restrict__ z,
                                                       const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       float slope, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                       float* __restrict__ dbias, int N, int HW, int C, int pstride = 0) {
    // pstride != 0 (CU_NORM_PARAM_PARTS): dgamma / dbeta are per-image planes [N][pstride]; this workgroup is the only writer
    // of its (image, channel) entries: plain stores, no atomics (the caller sums over the images)
    constexpr int PIECE = Elem<T>::PIECE;
    __shared__ float lds[(NT / 64) * SG * 2 * PIECE];
    __shared__ float sums[SG][2 * PIECE];
    const int n = blockIdx.x;
    const int pl = threadIdx.x & (SG - 1), prow = threadIdx.x / SG;
    const int piece = blockIdx.y * SG + pl;
    const bool active = piece * PIECE < C;
    const size_t NC = (size_t)N * C;
    const size_t sidx = (size_t)n * C + (active ? piece : 0) * PIECE;
    const size_t base = (size_t)n * HW * C + (active ? piece : 0) * PIECE;
    float mean[PIECE], rstd[PIECE], sc[PIECE], sh[PIECE];
#pragma unroll
    for (int e = 0; e < PIECE; ++e) {
        mean[e] = stats[sidx + e]; rstd[e] = stats[NC + sidx + e];
        sc[e] = stats[2 * NC + sidx + e]; sh[e] = stats[3 * NC + sidx + e];
    }
    float acc[2][PIECE];
#pragma unroll
    for (int e = 0; e < PIECE; ++e) acc[0][e] = acc[1][e] = 0.f;
    if (active) {
        auto sum1 = [&](const float (&zv)[PIECE], const float (&gv)[PIECE]) {
#pragma unroll
            for (int e = 0; e < PIECE; ++e) {
                const float y = zv[e] * sc[e] + sh[e];
                const float gl = y > 0.f ? gv[e] : gv[e] * slope;
                acc[0][e] += gl;
                acc[1][e] += gl * (zv[e] - mean[e]) * rstd[e];
            }
        };
        int p = prow;
        for (; p + 3 * SR < HW; p += 4 * SR) {      // 8 independent loads in flight (see stats_small_kernel)
            float zv[4][PIECE], gv[4][PIECE];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                load_piece<T>(z + base + (size_t)(p + u * SR) * C, zv[u]);
                load_piece<T>(g + base + (size_t)(p + u * SR) * C, gv[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) sum1(zv[u], gv[u]);
        }
        for (; p < HW; p += SR) {
            float zv[PIECE], gv[PIECE];
            load_piece<T>(z + base + (size_t)p * C, zv);
            load_piece<T>(g + base + (size_t)p * C, gv);
            sum1(zv, gv);
        }
    }
    reduce_rows<2, PIECE>(acc, lds, pl, prow, SR, SG, true);
    if (prow == 0) {
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            sums[pl][2 * e] = acc[0][e];
            sums[pl][2 * e + 1] = acc[1][e];
            if (active && pstride) {
                if (dbeta) dbeta[(size_t)n * pstride + piece * PIECE + e] = acc[0][e];
                if (dgamma) dgamma[(size_t)n * pstride + piece * PIECE + e] = acc[1][e];
            } else if (active) {
                if (dbeta) unsafeAtomicAdd(dbeta + piece * PIECE + e, acc[0][e]);
                if (dgamma) unsafeAtomicAdd(dgamma + piece * PIECE + e, acc[1][e]);
            }
        }
    }
    __syncthreads();
    float db[1][PIECE];
#pragma unroll
    for (int e = 0; e < PIECE; ++e) db[0][e] = 0.f;
    if (active) {
        const float inv = 1.f / (float)HW;
        float a1[PIECE], a2[PIECE], gr[PIECE];
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            a1[e] = sums[pl][2 * e] * inv; a2[e] = sums[pl][2 * e + 1] * inv;
            gr[e] = (gamma ? gamma[piece * PIECE + e] : 1.f) * rstd[e];
        }
        auto out2 = [&](const float (&zv)[PIECE], float (&gv)[PIECE]) {
#pragma unroll
            for (int e = 0; e < PIECE; ++e) {
                const float y = zv[e] * sc[e] + sh[e];
                const float gl = y > 0.f ? gv[e] : gv[e] * slope;
                const float xh = (zv[e] - mean[e]) * rstd[e];
                gv[e] = gr[e] * (gl - a1[e] - xh * a2[e]);
                db[0][e] += gv[e];
            }
        };
        int p = prow;
        for (; p + 3 * SR < HW; p += 4 * SR) {
            float zv[4][PIECE], gv[4][PIECE];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                load_piece<T>(z + base + (size_t)(p + u * SR) * C, zv[u]);
                load_piece<T>(g + base + (size_t)(p + u * SR) * C, gv[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                out2(zv[u], gv[u]);
                store_piece<T>(g + base + (size_t)(p + u * SR) * C, gv[u]);
            }
        }
        for (; p < HW; p += SR) {
            float zv[PIECE], gv[PIECE];
            load_piece<T>(z + base + (size_t)p * C, zv);
            load_piece<T>(g + base + (size_t)p * C, gv);
            out2(zv, gv);
            store_piece<T>(g + base + (size_t)p * C, gv);
        }
    }
    if (dbias) {
        __syncthreads();
        reduce_rows<1, PIECE>(db, lds, pl, prow, SR, SG, true);
        if (active && prow == 0) {
#pragma unroll
            for (int e = 0; e < PIECE; ++e) unsafeAtomicAdd(dbias + piece * PIECE + e, db[0][e]);
        }
    }
}

// Register-resident form of bwd_small_kernel (round 4) for maps of NR * (THREADS / 8) pixels: a workgroup of THREADS threads
// owns all pixels of one image for 8 pieces; every thread keeps its NR (z, g) pieces in registers between the reduction and the
// application, so both tensors are read ONCE (the two-pass form above re-reads them and, with 256 threads per (image, 64
// channels), leaves a CU 4 waves to hide the latency behind: 30 us at 16^2 x 480 and 48 us at 32^2 x 256 for 10 / 20 us of
// traffic).  16 x 16: 512 threads x 4 rows; 32 x 32: 1024 threads x 8 rows.  bf16 only (16-byte pieces of 8 channels).
template <int NR, int THREADS, int SGT = SG>
__global__ __launch_bounds__(THREADS) void bwd_small_res_kernel(bf16_t* __restrict__ g, const bf16_t* __restrict__ z,
                                                                const float* __restrict__ stats, const float* __restrict__ gamma,
                                                                float slope, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                int N, int HW, int C, int pstride) {
    // SGT = 16-byte pieces (8 channels each) per workgroup: 8 at 16 x 16; 4 at 32 x 32, where 1024 threads x 4 rows keep the
    // register count of 512 threads x 8 rows under the 256 an 8-wave workgroup may use (1024 threads x 8 rows x 8 pieces spilled: 616 bytes
    // of scratch, 97 us instead of 48; 1024 threads x 4 rows x 4 pieces still 116 bytes)
    constexpr int PIECE = 8, ROWS = THREADS / SGT, NW = THREADS / 64;
    __shared__ float part[NW][SGT][2 * PIECE];
    const int n = blockIdx.x;
    const int pl = threadIdx.x & (SGT - 1), prow = threadIdx.x / SGT;
    const int piece = blockIdx.y * SGT + pl;
    const bool active = piece * PIECE < C;
    const size_t NC = (size_t)N * C;
    const size_t sidx = (size_t)n * C + (active ? piece : 0) * PIECE;
    const size_t base = (size_t)n * HW * C + (active ? piece : 0) * PIECE;
    u32x4 zr[NR], gr_[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {          // all 2 NR loads of the thread in flight at once
        const size_t off = base + (size_t)(prow + i * ROWS) * C;
        zr[i] = active ? *reinterpret_cast<const u32x4*>(z + off) : u32x4{0u, 0u, 0u, 0u};
        gr_[i] = active ? *reinterpret_cast<const u32x4*>(g + off) : u32x4{0u, 0u, 0u, 0u};
    }
    float mean[PIECE], rstd[PIECE], sc[PIECE], sh[PIECE];
#pragma unroll
    for (int e = 0; e < PIECE; ++e) {
        mean[e] = stats[sidx + e]; rstd[e] = stats[NC + sidx + e];
        sc[e] = stats[2 * NC + sidx + e]; sh[e] = stats[3 * NC + sidx + e];
    }
    auto unpack = [](const u32x4& r, float (&v)[PIECE]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(r[i] << 16);
            v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
        }
    };
    float acc[2][PIECE];
#pragma unroll
    for (int e = 0; e < PIECE; ++e) acc[0][e] = acc[1][e] = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        float zv[PIECE], gv[PIECE];
        unpack(zr[i], zv);
        unpack(gr_[i], gv);
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            const float y = zv[e] * sc[e] + sh[e];
            const float gl = y > 0.f ? gv[e] : gv[e] * slope;
            acc[0][e] += gl;
            acc[1][e] += gl * (zv[e] - mean[e]) * rstd[e];
        }
    }
    // lanes l, l + 8, ..., l + 56 of a wave hold the same piece: butterfly over the pixel rows, then the waves through LDS
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            float v = acc[a][e];
            if constexpr (SGT <= 4) v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            acc[a][e] = v;
        }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < SGT) {
#pragma unroll
        for (int e = 0; e < PIECE; ++e) { part[wave][lane][2 * e] = acc[0][e]; part[wave][lane][2 * e + 1] = acc[1][e]; }
    }
    __syncthreads();
    float a1[PIECE], a2[PIECE];
#pragma unroll
    for (int e = 0; e < PIECE; ++e) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { s1 += part[w][pl][2 * e]; s2 += part[w][pl][2 * e + 1]; }      // fixed order: every thread the same sums
        a1[e] = s1; a2[e] = s2;
    }
    if (active && prow == 0) {
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            if (pstride) {       // per-image planes, sole writer (see bwd_small_kernel): measured 30.8 -> 13.7 us at 16^2 x 480, batch 64
                if (dbeta) dbeta[(size_t)n * pstride + piece * PIECE + e] = a1[e];
                if (dgamma) dgamma[(size_t)n * pstride + piece * PIECE + e] = a2[e];
            } else {
                if (dbeta) unsafeAtomicAdd(dbeta + piece * PIECE + e, a1[e]);
                if (dgamma) unsafeAtomicAdd(dgamma + piece * PIECE + e, a2[e]);
            }
        }
    }
    if (!active) return;
    const float inv = 1.f / (float)HW;
    float grs[PIECE];
#pragma unroll
    for (int e = 0; e < PIECE; ++e) {
        a1[e] *= inv; a2[e] *= inv;
        grs[e] = (gamma ? gamma[piece * PIECE + e] : 1.f) * rstd[e];
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        float zv[PIECE], gv[PIECE];
        unpack(zr[i], zv);
        unpack(gr_[i], gv);
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            const float y = zv[e] * sc[e] + sh[e];
            const float gl = y > 0.f ? gv[e] : gv[e] * slope;
            const float xh = (zv[e] - mean[e]) * rstd[e];
            gv[e] = grs[e] * (gl - a1[e] - xh * a2[e]);
        }
        store_piece<bf16_t>(g + base + (size_t)(prow + i * ROWS) * C, gv);
    }
}

template <typename T>
__global__ __launch_bounds__(NT) void act_bwd_kernel(T* __restrict__ g, const T* __restrict__ z, float slope,
                                                     float* __restrict__ dbias, int HW, int C, int tpp, int rows,
                                                     int chunk) {
    constexpr int PIECE = Elem<T>::PIECE;
    extern __shared__ float lds[];
    const int n = blockIdx.x;
    const int p0 = blockIdx.y * chunk, p1 = min(HW, p0 + chunk);
    const int piece = threadIdx.x % tpp, prow = threadIdx.x / tpp;
    const bool active = prow < rows;
    float acc[1][PIECE];
#pragma unroll
    for (int e = 0; e < PIECE; ++e) acc[0][e] = 0.f;
    if (active) {
        const size_t base = (size_t)n * HW * C + piece * PIECE;
        for (int p = p0 + prow; p < p1; p += rows) {
            float zv[PIECE], gv[PIECE];
            load_piece<T>(z + base + (size_t)p * C, zv);
            load_piece<T>(g + base + (size_t)p * C, gv);
#pragma unroll
            for (int e = 0; e < PIECE; ++e) {
                gv[e] = zv[e] > 0.f ? gv[e] : gv[e] * slope;
                acc[0][e] += gv[e];
            }
            store_piece<T>(g + base + (size_t)p * C, gv);
        }
    }
    if (dbias) {
        reduce_rows<1, PIECE>(acc, lds, piece, prow, rows, tpp, active);
        if (active && prow == 0) {
#pragma unroll
            for (int e = 0; e < PIECE; ++e) unsafeAtomicAdd(dbias + piece * PIECE + e, acc[0][e]);
        }
    }
}

// x[n][p][c] *= mask[n][c]: nn.Dropout2d(p=0.5) between conv and norm (reference layers.py:154-164,199-202), whole
// channels zeroed or doubled.  Only the three deepest encoder blocks use it (tiny tensors).
template <typename T>
__global__ void channel_scale_kernel(T* __restrict__ x, const float* __restrict__ mask, int N, int HW, int C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)N * HW * C) return;
    const int c = i % C;
    const int n = (i / C) / HW;
    Elem<T>::st(x + i, Elem<T>::ld(x + i) * mask[(size_t)n * C + c]);
}

// ------------------------------------------------------------------------------------------------ layout helpers (tiny tensors)
template <typename T>
__global__ void act_to_nchw_kernel(const T* __restrict__ z, const float* __restrict__ stats, float slope,
                                   float* __restrict__ out, int N, int HW, int C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)N * HW * C;
    if (i >= total) return;
    const int c = i % C;
    const size_t np = i / C;
    const int p = np % HW, n = np / HW;
    float v = Elem<T>::ld(z + i);
    if (stats) {
        const size_t NC = (size_t)N * C;
        v = v * stats[2 * NC + (size_t)n * C + c] + stats[3 * NC + (size_t)n * C + c];
    }
    v = v > 0.f ? v : v * slope;
    out[((size_t)n * C + c) * HW + p] = v;
}
// NCHW f32 [N][C][HW] -> NHWC [N][HW][CP] (channels >= C written as 0).  thread = (pixel, 8-channel group): a wave reads
// 64-byte runs of CP/8 planes and writes whole 16-byte pieces.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, int HW,
                                                           int C, int CP) {
    const int groups = CP / 8;
    const int n = blockIdx.y;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int grp = i % groups;
    const size_t p = i / groups;
    if (p >= (size_t)HW) return;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = grp * 8 + e;
        v[e] = c < C ? in[((size_t)n * C + c) * HW + p] : 0.f;
    }
    T* o = out + ((size_t)n * HW + p) * CP + grp * 8;
    if constexpr (sizeof(T) == 2) {
        store_piece<bf16_t>(reinterpret_cast<bf16_t*>(o), v);
    } else {
        float lo[4] = {v[0], v[1], v[2], v[3]}, hi[4] = {v[4], v[5], v[6], v[7]};
        store_piece<float>(reinterpret_cast<float*>(o), lo);
        store_piece<float>(reinterpret_cast<float*>(o) + 4, hi);
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, int N, int HW, int C, int accum) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)N * HW * C;
    if (i >= total) return;
    const int c = i % C;
    const size_t np = i / C;
    const int p = np % HW, n = np / HW;
    float* o = out + ((size_t)n * C + c) * HW + p;
    const float v = Elem<T>::ld(in + i);
    *o = accum ? *o + v : v;
}

// ------------------------------------------------------------------------------------------------ resident-chunk kernels
// One launch per direction, each tensor read ONCE:
//   forward : a = LeakyReLU((z - mean) * rstd * gamma + beta)        read z, write a            (two-pass: stats + apply)
//   backward: dz = gamma*rstd*(gl - mean(gl) - xhat*mean(gl*xhat))   read g, z, write dz        (two-pass: reduce + apply)
// A workgroup takes one (image, pixel chunk) task, keeps the chunk in REGISTERS (NP 16-byte pieces per thread and tensor),
// adds its partial sums to the image's accumulator row with f32 atomics, then waits until every chunk of that image has
// arrived before it normalises what it holds.  Nothing but the per-(image, channel) sums crosses workgroups.
//
// Progress without any assumption on dispatch order or co-residency of the whole grid: tasks are handed out by a TICKET
// counter (atomicAdd at workgroup start), so the chunks of one image go to nchunks consecutive starters.  A workgroup only
// ever waits for tickets of its own image; if all resident workgroups are waiting, fewer than nchunks tickets of the
// lowest unfinished image are out, i.e. fewer than nchunks workgroups are resident: with nchunks <= 256 (one workgroup
// per CU always fits) a slot is free and the next starter takes the missing ticket.
// Visibility (MI355X_MICROARCH.md, inter-workgroup visibility): the accumulator rows and the arrival counters are only
// touched by agent-scope atomics (executed at the memory side) and read by agent-scope (sc1) loads; the arrival add is
// issued after this workgroup's sum atomics have been acknowledged (s_waitcnt vmcnt(0) + barrier).
//
// Measured on MI355X (tools/norm_bench.py, 64 images, bf16; profiles/r02_norm_bench.txt): the streaming itself runs at
// 4.2 TB/s (256^2 x 32: forward 127 us, backward 180 us without any exchange), but every workgroup holds its registers
// through a chain of dependent memory round trips (adds acknowledged -> arrival add -> poll -> totals), and all chunks of
// an image finish together, so loads and stores alternate in bursts: with the exchange 262 / 335 us against 176 / 291 us
// for the two-pass kernels.  The one-launch form wins where an image is a few chunks (forward at <= 16x16: 17-24 us
// against 25-27) and where the two-pass kernels are latency-bound (backward at 64x64: 66 against 95 us); the host entry
// points below pick per shape.  History of the exchange at 256^2 x 32, forward: every piece holder adding its own 16
// floats (16 four-lane wave instructions per workgroup on one 256-byte row) 965 us; one coalesced add per 64 floats
// 278 us; totals fetched once per workgroup through LDS instead of per thread 262 us; 64-byte segments carrying their
// own arrival count (no acknowledgement wait, no counter, every thread polling) 332 us.
constexpr int RC_MAX_CHUNKS = 256;
#ifdef CU_TUNING
#define RC_DBG(dbg, bits) ((dbg) & (bits))      // CU_NORM_DBG: 1 no sum atomics, 2 no arrive/wait, 4 no totals fetch, 8 no ticket
#else
#define RC_DBG(dbg, bits) 0
#endif
// workspace (32-bit words): a 4-KiB header ([0] = ticket, [1] = spin-limit flag), then one 4-KiB-aligned block per image:
// [0, 2C) the sums [C][2], then on a 256-byte line of its own the image's arrival counter
constexpr int RC_HDR = 1024;
constexpr int RC_LINE = 64;
__host__ __device__ inline int rc_ctr_off(int C) { return (2 * C + RC_LINE - 1) / RC_LINE * RC_LINE; }
__host__ __device__ inline int rc_block_words(int C) { return (rc_ctr_off(C) + RC_LINE + 1023) / 1024 * 1024; }
constexpr unsigned RC_SPIN_LIMIT = 1u << 20;    // ~1 s: the kernel drains and raises ws[1] instead of hanging the GPU

struct RcTask { int n, chunk; };

__device__ __forceinline__ RcTask rc_take_ticket(unsigned* ctr, int nchunks, unsigned* s_slot) {
    if (threadIdx.x == 0) *s_slot = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned t = *s_slot;
    RcTask k;
    k.n = (int)(t / (unsigned)nchunks);
    k.chunk = (int)(t - (unsigned)k.n * (unsigned)nchunks);
    return k;
}

__device__ __forceinline__ float rc_read(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// acc (held by the threads with prow == 0, one channel piece each) -> lds[c * 2 + k] = the image's totals.
// The partial sums go through LDS so that consecutive lanes add consecutive floats (one 256-byte wave instruction per 64
// floats), and the totals come back through LDS from one coalesced agent-scope load per workgroup.
template <int PIECE>
__device__ __forceinline__ void rc_exchange(const float (&acc)[2][PIECE], float* lds, float* row, unsigned* flag, int C,
                                            int piece, bool holder, int nchunks, int dbg, unsigned* s_flag) {
    __syncthreads();                      // reduce_rows is done with the LDS scratch
    if (holder) {
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            lds[(piece * PIECE + e) * 2] = acc[0][e];
            lds[(piece * PIECE + e) * 2 + 1] = acc[1][e];
        }
    }
    __syncthreads();
    if (!RC_DBG(dbg, 1))
        for (int i = threadIdx.x; i < 2 * C; i += NT) unsafeAtomicAdd(row + i, lds[i]);
    if (!RC_DBG(dbg, 2)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's adds are acknowledged ...
        __syncthreads();                                       // ... everybody's are
        if (threadIdx.x == 0) {
            unsigned* arrived = reinterpret_cast<unsigned*>(row) + rc_ctr_off(C);
            __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0, gave_up = 0;
            while (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nchunks) {
                __builtin_amdgcn_s_sleep(32);
                if (++spins > RC_SPIN_LIMIT) {
                    __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gave_up = 1;
                    break;
                }
            }
            *s_flag = gave_up;        // (the ticket in this word was read by every thread before the first barrier above)
        }
    } else if (threadIdx.x == 0) {
        *s_flag = 0;
    }
    __syncthreads();
    // A wait that gave up must not pass incomplete sums on as if they were totals (the flag word lives in the
    // per-pass arena and nobody on the training path reads it): the workgroup POISONS its totals, so the statistics, the
    // activations / gradients and with them the step's loss turn NaN -- loud, and sticky through the optimiser step.
    const bool gave_up = *s_flag != 0;
    if (!RC_DBG(dbg, 4))
        for (int i = threadIdx.x; i < 2 * C; i += NT) lds[i] = gave_up ? __builtin_nanf("") : rc_read(row + i);
    __syncthreads();
}

template <typename T> __device__ __forceinline__ void unpack_piece(const u32x4& r, float (&v)[Elem<T>::PIECE]);
template <> __device__ __forceinline__ void unpack_piece<float>(const u32x4& r, float (&v)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __uint_as_float(r[e]);
}
template <> __device__ __forceinline__ void unpack_piece<bf16_t>(const u32x4& r, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(r[i] << 16);
        v[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
    }
}

template <typename T, int NP>
__global__ __launch_bounds__(NT, 4) void fwd_resident_kernel(const T* __restrict__ z, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps, float slope,
                                                              float* __restrict__ stats, T* __restrict__ out,
                                                              float* ws, int N, int HW, int C, int tpp, int rows,
                                                              int nchunks, int dbg) {
    constexpr int PIECE = Elem<T>::PIECE;
    extern __shared__ float lds[];
    __shared__ unsigned s_slot;
    unsigned* ctr = reinterpret_cast<unsigned*>(ws);
    RcTask task = rc_take_ticket(ctr, nchunks, &s_slot);
    if (RC_DBG(dbg, 8)) { task.n = blockIdx.x / nchunks; task.chunk = blockIdx.x % nchunks; }
    const int n = task.n;
    if (n >= N) return;                                   // uniform
    const int piece = threadIdx.x % tpp, prow = threadIdx.x / tpp;
    const bool active = prow < rows;
    const int p0 = task.chunk * rows * NP;
    const size_t base = (size_t)n * HW * C + (active ? piece : 0) * PIECE;
    u32x4 zr[NP];
    unsigned okm = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int p = p0 + prow + j * rows;
        const bool ok = active && p < HW;
        zr[j] = *reinterpret_cast<const u32x4*>(z + base + (size_t)(ok ? p : 0) * C);
        okm |= (ok ? 1u : 0u) << j;
    }
    float k[PIECE];
    load_piece<T>(z + base, k);              // shift by the image's first pixel: avoids E[x^2]-E[x]^2 cancellation
    float acc[2][PIECE];
#pragma unroll
    for (int e = 0; e < PIECE; ++e) acc[0][e] = acc[1][e] = 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        float v[PIECE];
        unpack_piece<T>(zr[j], v);
        const float m = ((okm >> j) & 1u) ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            const float d = (v[e] - k[e]) * m;
            acc[0][e] += d;
            acc[1][e] += d * d;
        }
    }
    reduce_rows<2, PIECE>(acc, lds, piece, prow, rows, tpp, active);
    rc_exchange<PIECE>(acc, lds, ws + RC_HDR + (size_t)n * rc_block_words(C), ctr + 1, C, piece, active && prow == 0,
                       nchunks, dbg, &s_slot);
    if (!active) return;
    // only the PACKED pieces stay live across the wait (the compiler would otherwise keep their unpacked floats too)
#pragma unroll
    for (int j = 0; j < NP; ++j) asm volatile("" : "+v"(zr[j]));
    float sc[PIECE], sh[PIECE];
    const float inv = 1.f / (float)HW;
    const size_t NC = (size_t)N * C;
    const size_t sidx = (size_t)n * C + piece * PIECE;
#pragma unroll
    for (int e = 0; e < PIECE; ++e) {
        const float m1 = lds[(piece * PIECE + e) * 2] * inv, m2 = lds[(piece * PIECE + e) * 2 + 1] * inv;
        const float mean = k[e] + m1;
        const float rstd = 1.f / sqrtf(fmaxf(m2 - m1 * m1, 0.f) + eps);
        const int c = piece * PIECE + e;
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        sc[e] = g * rstd;
        sh[e] = b - mean * g * rstd;
        if (task.chunk == 0 && prow == 0) {
            stats[sidx + e] = mean;
            stats[NC + sidx + e] = rstd;
            stats[2 * NC + sidx + e] = sc[e];
            stats[3 * NC + sidx + e] = sh[e];
        }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if ((okm >> j) & 1u) {
            float v[PIECE];
            unpack_piece<T>(zr[j], v);
#pragma unroll
            for (int e = 0; e < PIECE; ++e) {
                const float y = v[e] * sc[e] + sh[e];
                v[e] = y > 0.f ? y : y * slope;
            }
            store_piece<T>(out + base + (size_t)(p0 + prow + j * rows) * C, v);
        }
    }
}

template <typename T, int NP>
__global__ __launch_bounds__(NT, 2) void bwd_resident_kernel(T* __restrict__ g, const T* __restrict__ z,
                                                              const float* __restrict__ stats, const float* __restrict__ gamma,
                                                              float slope, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* ws, int N, int HW, int C, int tpp, int rows, int nchunks,
                                                              int dbg) {
    constexpr int PIECE = Elem<T>::PIECE;
    extern __shared__ float lds[];
    __shared__ unsigned s_slot;
    unsigned* ctr = reinterpret_cast<unsigned*>(ws);
    RcTask task = rc_take_ticket(ctr, nchunks, &s_slot);
    if (RC_DBG(dbg, 8)) { task.n = blockIdx.x / nchunks; task.chunk = blockIdx.x % nchunks; }
    const int n = task.n;
    if (n >= N) return;                                   // uniform
    const int piece = threadIdx.x % tpp, prow = threadIdx.x / tpp;
    const bool active = prow < rows;
    const int p0 = task.chunk * rows * NP;
    const size_t base = (size_t)n * HW * C + (active ? piece : 0) * PIECE;
    const size_t NC = (size_t)N * C;
    const size_t sidx = (size_t)n * C + (active ? piece : 0) * PIECE;
    u32x4 zr[NP], gr_[NP];
    unsigned okm = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int p = p0 + prow + j * rows;
        const bool ok = active && p < HW;
        const size_t off = base + (size_t)(ok ? p : 0) * C;
        zr[j] = *reinterpret_cast<const u32x4*>(z + off);
        gr_[j] = *reinterpret_cast<const u32x4*>(g + off);
        okm |= (ok ? 1u : 0u) << j;
    }
    float mean[PIECE], rstd[PIECE], sc[PIECE], sh[PIECE];
#pragma unroll
    for (int e = 0; e < PIECE; ++e) {
        mean[e] = stats[sidx + e]; rstd[e] = stats[NC + sidx + e];
        sc[e] = stats[2 * NC + sidx + e]; sh[e] = stats[3 * NC + sidx + e];
    }
    float acc[2][PIECE];
#pragma unroll
    for (int e = 0; e < PIECE; ++e) acc[0][e] = acc[1][e] = 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        float zv[PIECE], gv[PIECE];
        unpack_piece<T>(zr[j], zv);
        unpack_piece<T>(gr_[j], gv);
        const float m = ((okm >> j) & 1u) ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            const float y = zv[e] * sc[e] + sh[e];
            const float gl = (y > 0.f ? gv[e] : gv[e] * slope) * m;
            acc[0][e] += gl;
            acc[1][e] += gl * (zv[e] - mean[e]) * rstd[e];
        }
    }
    reduce_rows<2, PIECE>(acc, lds, piece, prow, rows, tpp, active);
    rc_exchange<PIECE>(acc, lds, ws + RC_HDR + (size_t)n * rc_block_words(C), ctr + 1, C, piece, active && prow == 0,
                       nchunks, dbg, &s_slot);
    if (!active) return;
#pragma unroll
    for (int j = 0; j < NP; ++j) asm volatile("" : "+v"(zr[j]), "+v"(gr_[j]));
    float a1[PIECE], a2[PIECE], gr[PIECE];
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < PIECE; ++e) {
        const float t1 = lds[(piece * PIECE + e) * 2], t2 = lds[(piece * PIECE + e) * 2 + 1];
        a1[e] = t1 * inv; a2[e] = t2 * inv;
        gr[e] = (gamma ? gamma[piece * PIECE + e] : 1.f) * rstd[e];
        if (task.chunk == 0 && prow == 0) {     // one contribution per (image, channel)
            if (dbeta) unsafeAtomicAdd(dbeta + piece * PIECE + e, t1);
            if (dgamma) unsafeAtomicAdd(dgamma + piece * PIECE + e, t2);
        }
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if ((okm >> j) & 1u) {
            float zv[PIECE], gv[PIECE];
            unpack_piece<T>(zr[j], zv);
            unpack_piece<T>(gr_[j], gv);
#pragma unroll
            for (int e = 0; e < PIECE; ++e) {
                const float y = zv[e] * sc[e] + sh[e];
                const float gl = y > 0.f ? gv[e] : gv[e] * slope;
                const float xh = (zv[e] - mean[e]) * rstd[e];
                gv[e] = gr[e] * (gl - a1[e] - xh * a2[e]);
            }
            store_piece<T>(g + base + (size_t)(p0 + prow + j * rows) * C, gv);
        }
    }
}

// dgamma[c] += sum_n ws[n][c][1], dbeta[c] += sum_n ws[n][c][0] in image order (deterministic mode: the backward kernels
// leave the parameter gradients alone and this pass is their single writer)
__global__ __launch_bounds__(256) void norm_param_grads_kernel(const float* __restrict__ ws, int N, int C,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s0 = 0.f, s1 = 0.f;
    for (int n = 0; n < N; ++n) {
        s0 += ws[((size_t)n * C + c) * 2];
        s1 += ws[((size_t)n * C + c) * 2 + 1];
    }
    if (dbeta) dbeta[c] += s0;
    if (dgamma) dgamma[c] += s1;
}

static int pick_chunk(int N, int HW, int rows, int* nchunks, bool det = false) {
    if (det) {      // one workgroup per image: its LDS reduction has a fixed order and it is the only adder of its sums
        *nchunks = 1;
        return cdiv(HW, rows) * rows;
    }
    // aim for ~2048 workgroups in total; every chunk a multiple of the rows handled per iteration
    int want = cdiv(cu_env_int("CU_NORM_WGS", 2048), N);      // (tuning knob; 1024 / 4096 inside the step: profiles/r04_norm_wgs.txt)
    if (want < 1) want = 1;
    int chunk = cdiv(HW, want);
    // at least 16 pixels per thread: every workgroup ends in one atomic per channel and partial sum, and on the small
    // feature maps (16x16 x 480 channels: 2048 workgroups of 8 pixels) those atomics were the whole run time
    if (chunk < 16 * rows) chunk = 16 * rows;
    chunk = cdiv(chunk, rows) * rows;
    if (chunk < rows) chunk = rows;
    *nchunks = cdiv(HW, chunk);
    return chunk;
}

}  // namespace

#define NORM_COMMON_CHECKS(name)                                                                              \
    CU_CHECK_ARG(dtype == CU_F32 || dtype == CU_BF16, name ": bad dtype %d", dtype);                          \
    const int PIECE = dtype == CU_BF16 ? 8 : 4;                                                               \
    CU_CHECK_ARG(N > 0 && HW > 0 && C > 0 && C % PIECE == 0, name ": bad shape N=%d HW=%d C=%d", N, HW, C);   \
    CU_CHECK_ARG(C / PIECE <= NT, name ": C=%d too wide", C);                                                 \
    const RowMap rm = row_map(C, PIECE);                                                                      \
    int nchunks = 1;                                                                                          \
    const int chunk = pick_chunk(N, HW, rm.rows, &nchunks);                                                   \
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);                                                   \
    (void)chunk; (void)nchunks; (void)st;

// ---- launch helpers: `nimg` images starting at the pointers given, inside a tensor of N images (plane stride of stats)
template <typename T>
static int launch_stats(int nimg, int N, int HW, int C, const void* z, const float* gamma, const float* beta, float eps,
                        float* stats, float* ws, hipStream_t st, bool clean = false, bool det = false) {
    constexpr int PIECE = Elem<T>::PIECE;
    const RowMap rm = row_map(C, PIECE);
    if (HW <= 1024) {       // small feature maps: one fused launch (no atomics: deterministic as it is)
        dim3 sgrid(nimg, cdiv(C / PIECE, SG));
        hipLaunchKernelGGL(stats_small_kernel<T>, sgrid, dim3(NT), 0, st, (const T*)z, gamma, beta, eps, stats, N, HW, C);
        CU_LAUNCH_CHECK();
        return 0;
    }
    int nchunks = 1;
    const int chunk = pick_chunk(nimg, HW, rm.rows, &nchunks, det);
    if (!clean) {
        hipError_t e = hipMemsetAsync(ws, 0, sizeof(float) * 2 * (size_t)nimg * C, st);
        CU_CHECK_ARG(e == hipSuccess, "cu_instnorm_stats: memset failed: %s", hipGetErrorString(e));
    }
    const size_t lds = sizeof(float) * (size_t)rm.rows * rm.tpp * 2 * PIECE;
    hipLaunchKernelGGL(stats_partial_kernel<T>, dim3(nimg, nchunks), dim3(NT), lds, st, (const T*)z, ws, HW, C, rm.tpp,
                       rm.rows, chunk);
    hipLaunchKernelGGL(stats_finalize_kernel<T>, dim3(cdiv(nimg * C, 256)), dim3(256), 0, st, (const T*)z, ws, gamma, beta,
                       eps, stats, N, HW, C, nimg);
    CU_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int launch_apply(int nimg, int N, int HW, int C, const void* z, const float* stats, float slope, void* out,
                        hipStream_t st) {
    const RowMap rm = row_map(C, Elem<T>::PIECE);
    int nchunks = 1;
    const int chunk = pick_chunk(nimg, HW, rm.rows, &nchunks);
    hipLaunchKernelGGL((apply_kernel<T, false>), dim3(nimg, nchunks), dim3(NT), 0, st, (const T*)z, stats, slope, (T*)out, N, HW,
                       C, rm.tpp, rm.rows, chunk, GivenSums());
    CU_LAUNCH_CHECK();
    return 0;
}

template <typename T>
static int launch_bwd(int nimg, int N, int HW, int C, void* g, const void* z, const float* stats, const float* gamma,
                      float slope, float* dgamma, float* dbeta, float* dbias, float* ws, hipStream_t st, bool clean = false,
                      bool det = false, int pstride = 0, bool small_res = false) {
    constexpr int PIECE = Elem<T>::PIECE;
    const RowMap rm = row_map(C, PIECE);
    CU_CHECK_ARG(!pstride || (HW <= 1024 && !det), "CU_NORM_PARAM_PARTS is served on maps of <= 1024 pixels (got %d)", HW);
    if (HW <= 1024 && !det) {       // small feature maps: one fused launch
        dim3 sgrid(nimg, cdiv(C / PIECE, SG));
        if constexpr (sizeof(T) == 2) {
            const dim3 sgrid4(nimg, cdiv(C / PIECE, 4));
            // 16 x 16 and 32 x 32 maps in bf16: the register-resident form (both tensors read once): CU_NORM_SMALL_RES.  Alone it
            // takes 32-35 us instead of 48 (32 x 32) and 20 instead of 30 (16 x 16); inside the training step, beside the weight-
            // gradient stream, its 512-thread x 214-VGPR workgroups (one per CU) cost more than they save: 12.45 vs 12.40 ms per
            // step over six alternating pairs (profiles/r04_norm_small_res_in_step.txt) -- the engine does not ask for it.
            // (tuning build: CU_NORM_SMALL_RES_MASK = 1: 16 x 16, 2: 32 x 32, 3: both, whatever the caller asked)
            const int res_mask = cu_env_int("CU_NORM_SMALL_RES_MASK", small_res ? 3 : 0);
            if (!dbias && ((HW == 256 && (res_mask & 1)) || (HW == 1024 && (res_mask & 2)))) {
                if (HW == 256)
                    hipLaunchKernelGGL((bwd_small_res_kernel<4, 512>), sgrid, dim3(512), 0, st, (bf16_t*)g, (const bf16_t*)z, stats,
                                       gamma, slope, dgamma, dbeta, N, HW, C, pstride);
                else
                    hipLaunchKernelGGL((bwd_small_res_kernel<8, 512, 4>), sgrid4, dim3(512), 0, st, (bf16_t*)g, (const bf16_t*)z, stats,
                                       gamma, slope, dgamma, dbeta, N, HW, C, pstride);
                CU_LAUNCH_CHECK();
                return 0;
            }
        }
        hipLaunchKernelGGL(bwd_small_kernel<T>, sgrid, dim3(NT), 0, st, (T*)g, (const T*)z, stats, gamma, slope, dgamma, dbeta,
                           dbias, N, HW, C, pstride);
        CU_LAUNCH_CHECK();
        return 0;
    }
    int nchunks = 1;
    const int chunk = pick_chunk(nimg, HW, rm.rows, &nchunks, det);
    if (det) dgamma = dbeta = nullptr;      // the caller's norm_param_grads_kernel pass writes them
    if (!clean) {
        hipError_t e = hipMemsetAsync(ws, 0, sizeof(float) * 2 * (size_t)nimg * C, st);
        CU_CHECK_ARG(e == hipSuccess, "cu_instnorm_lrelu_bwd: memset failed: %s", hipGetErrorString(e));
    }
    const size_t lds = sizeof(float) * (size_t)rm.rows * rm.tpp * 2 * PIECE;
    dim3 grid(nimg, nchunks);
    hipLaunchKernelGGL(bwd_reduce_kernel<T>, grid, dim3(NT), lds, st, (const T*)g, (const T*)z, stats, slope, ws, N, HW, C,
                       rm.tpp, rm.rows, chunk);
    hipLaunchKernelGGL(bwd_apply_kernel<T>, grid, dim3(NT), lds, st, (T*)g, (const T*)z, stats, gamma, slope, ws, dgamma, dbeta,
                       dbias, N, HW, C, rm.tpp, rm.rows, chunk);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_instnorm_stats(int dtype, int N, int HW, int C, const void* z, const float* gamma, const float* beta,
                                 float eps, float* stats, float* ws, void* stream) {
    NORM_COMMON_CHECKS("cu_instnorm_stats");
    CU_CHECK_ARG(z && stats && ws, "cu_instnorm_stats: null pointer");
    return dtype == CU_BF16 ? launch_stats<bf16_t>(N, N, HW, C, z, gamma, beta, eps, stats, ws, st)
                            : launch_stats<float>(N, N, HW, C, z, gamma, beta, eps, stats, ws, st);
}

extern "C" int cu_instnorm_apply(int dtype, int N, int HW, int C, const void* z, const float* stats, float slope,
                                 void* out, void* stream) {
    NORM_COMMON_CHECKS("cu_instnorm_apply");
    CU_CHECK_ARG(z && stats && out, "cu_instnorm_apply: null pointer");
    return dtype == CU_BF16 ? launch_apply<bf16_t>(N, N, HW, C, z, stats, slope, out, st)
                            : launch_apply<float>(N, N, HW, C, z, stats, slope, out, st);
}

extern "C" int cu_instnorm_lrelu_bwd(int dtype, int N, int HW, int C, void* g, const void* z, const float* stats,
                                     const float* gamma, float slope, float* dgamma, float* dbeta, float* dbias,
                                     float* ws, void* stream) {
    NORM_COMMON_CHECKS("cu_instnorm_lrelu_bwd");
    CU_CHECK_ARG(g && z && stats && ws, "cu_instnorm_lrelu_bwd: null pointer");
    return dtype == CU_BF16 ? launch_bwd<bf16_t>(N, N, HW, C, g, z, stats, gamma, slope, dgamma, dbeta, dbias, ws, st)
                            : launch_bwd<float>(N, N, HW, C, g, z, stats, gamma, slope, dgamma, dbeta, dbias, ws, st);
}

// ---- production entry points: per shape, the one-launch resident kernels or the two-pass kernels.
// The two-pass form can run on GROUPS of images so that the second pass of a group re-reads what the first pass has just
// pulled through the 256 MiB Infinity Cache.  Measured (profiles/r02_norm_bench.txt): with 32 MiB groups the 4 launches
// per group cost more than the cache saves (256^2 x 32 x 64 images: forward 400 us against 177 us ungrouped, backward
// 544 against 305), so one group = the whole batch is the default; the mechanism stays for larger batches.
constexpr size_t GROUP_BYTES = (size_t)1 << 40;  // one tensor of one group

static int rc_pick_np(int HW, int rows) {
    int np = 8;
    while (np > 1 && rows * (np / 2) >= HW) np >>= 1;      // a smaller chunk still covers the whole image
    return np;
}
static int group_images(int N, int HW, int C, int esz) {
    const size_t per_img = (size_t)HW * C * esz;
    int g = (int)(GROUP_BYTES / per_img);
    if (g < 1) g = 1;
    return g > N ? N : g;
}

extern "C" size_t cu_instnorm_resident_ws_floats(int N, int C) {
    const size_t a = (size_t)RC_HDR + (size_t)N * rc_block_words(C), b = (size_t)N * C * 2;
    return a > b ? a : b;
}

template <typename T, int NP>
static void launch_fwd_resident(int N, int HW, int C, const RowMap& rm, int nch, const void* z, const float* gamma,
                                const float* beta, float eps, float slope, float* stats, void* out, float* ws, hipStream_t st) {
    const size_t lds = sizeof(float) * (size_t)(rm.rows > 4 ? rm.rows : 4) * rm.tpp * 2 * Elem<T>::PIECE;   // >= 2 C floats
    hipLaunchKernelGGL((fwd_resident_kernel<T, NP>), dim3(N * nch), dim3(NT), lds, st, (const T*)z, gamma, beta, eps, slope,
                       stats, (T*)out, ws, N, HW, C, rm.tpp, rm.rows, nch, cu_env_int("CU_NORM_DBG", 0));
}
template <typename T, int NP>
static void launch_bwd_resident(int N, int HW, int C, const RowMap& rm, int nch, void* g, const void* z, const float* stats,
                                const float* gamma, float slope, float* dgamma, float* dbeta, float* ws, hipStream_t st) {
    const size_t lds = sizeof(float) * (size_t)(rm.rows > 4 ? rm.rows : 4) * rm.tpp * 2 * Elem<T>::PIECE;
    hipLaunchKernelGGL((bwd_resident_kernel<T, NP>), dim3(N * nch), dim3(NT), lds, st, (T*)g, (const T*)z, stats, gamma, slope,
                       dgamma, dbeta, ws, N, HW, C, rm.tpp, rm.rows, nch, cu_env_int("CU_NORM_DBG", 0));
}
#define CU_RC_NP(fn, T, np, ...)                                \
    do {                                                        \
        if (np == 8) fn<T, 8>(__VA_ARGS__);                     \
        else if (np == 4) fn<T, 4>(__VA_ARGS__);                \
        else if (np == 2) fn<T, 2>(__VA_ARGS__);                \
        else fn<T, 1>(__VA_ARGS__);                             \
    } while (0)

extern "C" int cu_instnorm_fwd_fused(int dtype, int N, int HW, int C, const void* z, const float* gamma, const float* beta,
                                     float eps, float slope, float* stats, void* out, float* ws, int mode, void* stream) {
    NORM_COMMON_CHECKS("cu_instnorm_fwd_fused");
    CU_CHECK_ARG(z && stats && ws, "cu_instnorm_fwd_fused: null pointer");      // out == NULL: statistics only (two-pass form)
    const bool clean = (mode & CU_NORM_WS_CLEAN) != 0;       // the caller hands over a zeroed workspace: no memset launch
    const bool det = (mode & CU_NORM_DETERMINISTIC) != 0;    // two-pass kernels, one workgroup per image: fixed summation order
    mode &= ~(CU_NORM_WS_CLEAN | CU_NORM_DETERMINISTIC);
    CU_CHECK_ARG(mode >= 0 && mode <= 2, "cu_instnorm_fwd_fused: mode %d", mode);
    if (det) mode = 2;
    const int np = rc_pick_np(HW, rm.rows);
    const int nch = cdiv(HW, rm.rows * np);
    if (mode == 0) mode = (HW <= 256 && nch <= RC_MAX_CHUNKS && out && !cu_env_set("CU_NORM_NO_FWD_RESIDENT")) ? 1 : 2;      // measured crossover (see the kernels' header)
    if (mode == 1) {
        CU_CHECK_ARG(out != nullptr, "cu_instnorm_fwd_fused: the resident kernel always writes the activated tensor");
        CU_CHECK_ARG(nch <= RC_MAX_CHUNKS, "cu_instnorm_fwd_fused: %d chunks per image exceed %d", nch, RC_MAX_CHUNKS);
        if (!clean) {
            hipError_t e = hipMemsetAsync(ws, 0, sizeof(float) * ((size_t)RC_HDR + (size_t)N * rc_block_words(C)), st);
            CU_CHECK_ARG(e == hipSuccess, "cu_instnorm_fwd_fused: memset failed: %s", hipGetErrorString(e));
        }
        if (dtype == CU_BF16) CU_RC_NP(launch_fwd_resident, bf16_t, np, N, HW, C, rm, nch, z, gamma, beta, eps, slope, stats, out, ws, st);
        else CU_RC_NP(launch_fwd_resident, float, np, N, HW, C, rm, nch, z, gamma, beta, eps, slope, stats, out, ws, st);
        CU_LAUNCH_CHECK();
        return 0;
    }
    const int esz = dtype == CU_BF16 ? 2 : 4;
    const int gi = group_images(N, HW, C, esz);
    for (int n0 = 0; n0 < N; n0 += gi) {
        const int ni = N - n0 < gi ? N - n0 : gi;
        const char* zp = (const char*)z + (size_t)n0 * HW * C * esz;
        char* op = (char*)out + (size_t)n0 * HW * C * esz;
        float* sp = stats + (size_t)n0 * C;
        int rc = dtype == CU_BF16 ? launch_stats<bf16_t>(ni, N, HW, C, zp, gamma, beta, eps, sp, ws + (size_t)n0 * C * 2, st, clean, det)
                                  : launch_stats<float>(ni, N, HW, C, zp, gamma, beta, eps, sp, ws + (size_t)n0 * C * 2, st, clean, det);
        if (rc) return rc;
        if (!out) continue;       // the consumers normalise + activate on load (cu_conv_gemm / cu_conv_wgrad with scale / shift)
        rc = dtype == CU_BF16 ? launch_apply<bf16_t>(ni, N, HW, C, zp, sp, slope, op, st)
                              : launch_apply<float>(ni, N, HW, C, zp, sp, slope, op, st);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int cu_instnorm_fwd_given(int dtype, int N, int HW, int C, const void* z, const float* gamma, const float* beta,
                                     float eps, float slope, const float* sums, const float* shift, float* stats, void* out,
                                     void* stream) {
    NORM_COMMON_CHECKS("cu_instnorm_fwd_given");
    CU_CHECK_ARG(z && stats && sums, "cu_instnorm_fwd_given: null pointer");      // out == NULL: statistics only
    if (!out) {
        hipLaunchKernelGGL(stats_finalize_given_kernel, dim3(cdiv(N * C, 256)), dim3(256), 0, st, sums, shift, gamma, beta, eps,
                           stats, N, HW, C);
        CU_LAUNCH_CHECK();
        return 0;
    }
    // statistics finalised inside the apply pass (every workgroup from the sums; chunk 0 of an image stores them)
    GivenSums gs;
    gs.sums = sums; gs.shift = shift; gs.gamma = gamma; gs.beta = beta; gs.eps = eps; gs.stats_out = stats;
    dim3 grid(N, nchunks);
    if (dtype == CU_BF16)
        hipLaunchKernelGGL((apply_kernel<bf16_t, true>), grid, dim3(NT), 0, st, (const bf16_t*)z, (const float*)nullptr, slope,
                           (bf16_t*)out, N, HW, C, rm.tpp, rm.rows, chunk, gs);
    else
        hipLaunchKernelGGL((apply_kernel<float, true>), grid, dim3(NT), 0, st, (const float*)z, (const float*)nullptr, slope,
                           (float*)out, N, HW, C, rm.tpp, rm.rows, chunk, gs);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_instnorm_bwd_given(int dtype, int N, int HW, int C, void* g, const void* z, const float* stats,
                                     const float* gamma, float slope, float* dgamma, float* dbeta, const float* sums,
                                     void* stream) {
    NORM_COMMON_CHECKS("cu_instnorm_bwd_given");
    CU_CHECK_ARG(g && z && stats && sums, "cu_instnorm_bwd_given: null pointer");
    const size_t lds = sizeof(float) * (size_t)rm.rows * rm.tpp * 2 * PIECE;
    dim3 grid(N, nchunks);
    if (dtype == CU_BF16)
        hipLaunchKernelGGL(bwd_apply_kernel<bf16_t>, grid, dim3(NT), lds, st, (bf16_t*)g, (const bf16_t*)z, stats, gamma, slope,
                           sums, dgamma, dbeta, (float*)nullptr, N, HW, C, rm.tpp, rm.rows, chunk);
    else
        hipLaunchKernelGGL(bwd_apply_kernel<float>, grid, dim3(NT), lds, st, (float*)g, (const float*)z, stats, gamma, slope,
                           sums, dgamma, dbeta, (float*)nullptr, N, HW, C, rm.tpp, rm.rows, chunk);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_instnorm_bwd_fused(int dtype, int N, int HW, int C, void* g, const void* z, const float* stats,
                                     const float* gamma, float slope, float* dgamma, float* dbeta, float* ws, int mode,
                                     void* stream) {
    NORM_COMMON_CHECKS("cu_instnorm_bwd_fused");
    CU_CHECK_ARG(g && z && stats && ws, "cu_instnorm_bwd_fused: null pointer");
    const bool clean = (mode & CU_NORM_WS_CLEAN) != 0;       // the caller hands over a zeroed workspace: no memset launch
    const bool det = (mode & CU_NORM_DETERMINISTIC) != 0;
    const bool parts = (mode & CU_NORM_PARAM_PARTS) != 0;    // dgamma / dbeta are per-image planes [N][C] (maps of <= 1024 pixels)
    const bool small_res = (mode & CU_NORM_SMALL_RES) != 0;  // register-resident kernels at 16 x 16 / 32 x 32 (bf16)
    mode &= ~(CU_NORM_WS_CLEAN | CU_NORM_DETERMINISTIC | CU_NORM_PARAM_PARTS | CU_NORM_SMALL_RES);
    CU_CHECK_ARG(mode >= 0 && mode <= 2, "cu_instnorm_bwd_fused: mode %d", mode);
    CU_CHECK_ARG(!parts || (HW <= 1024 && !det && mode != 1), "cu_instnorm_bwd_fused: CU_NORM_PARAM_PARTS needs a map of <= 1024 pixels, "
                 "not the resident or the deterministic form");
    if (det) mode = 2;
    const int np = rc_pick_np(HW, rm.rows);
    const int nch = cdiv(HW, rm.rows * np);
    if (mode == 0) mode = (HW > 1024 && HW <= 4096 && nch <= RC_MAX_CHUNKS && !cu_env_set("CU_NORM_NO_BWD_RESIDENT")) ? 1 : 2;
    if (mode == 1) {
        CU_CHECK_ARG(nch <= RC_MAX_CHUNKS, "cu_instnorm_bwd_fused: %d chunks per image exceed %d", nch, RC_MAX_CHUNKS);
        if (!clean) {
            hipError_t e = hipMemsetAsync(ws, 0, sizeof(float) * ((size_t)RC_HDR + (size_t)N * rc_block_words(C)), st);
            CU_CHECK_ARG(e == hipSuccess, "cu_instnorm_bwd_fused: memset failed: %s", hipGetErrorString(e));
        }
        if (dtype == CU_BF16) CU_RC_NP(launch_bwd_resident, bf16_t, np, N, HW, C, rm, nch, g, z, stats, gamma, slope, dgamma, dbeta, ws, st);
        else CU_RC_NP(launch_bwd_resident, float, np, N, HW, C, rm, nch, g, z, stats, gamma, slope, dgamma, dbeta, ws, st);
        CU_LAUNCH_CHECK();
        return 0;
    }
    const int esz = dtype == CU_BF16 ? 2 : 4;
    const int gi = group_images(N, HW, C, esz);
    for (int n0 = 0; n0 < N; n0 += gi) {
        const int ni = N - n0 < gi ? N - n0 : gi;
        char* gp = (char*)g + (size_t)n0 * HW * C * esz;
        const char* zp = (const char*)z + (size_t)n0 * HW * C * esz;
        const float* sp = stats + (size_t)n0 * C;
        float* dgp = parts && dgamma ? dgamma + (size_t)n0 * C : dgamma;          // per-image planes follow the image group
        float* dbp = parts && dbeta ? dbeta + (size_t)n0 * C : dbeta;
        const int rc = dtype == CU_BF16
            ? launch_bwd<bf16_t>(ni, N, HW, C, gp, zp, sp, gamma, slope, dgp, dbp, nullptr, ws + (size_t)n0 * C * 2, st, clean, det, parts ? C : 0, small_res)
            : launch_bwd<float>(ni, N, HW, C, gp, zp, sp, gamma, slope, dgp, dbp, nullptr, ws + (size_t)n0 * C * 2, st, clean, det, parts ? C : 0, small_res);
        if (rc) return rc;
    }
    if (det && (dgamma || dbeta)) {
        hipLaunchKernelGGL(norm_param_grads_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, ws, N, C, dgamma, dbeta);
        CU_LAUNCH_CHECK();
    }
    return 0;
}
#undef CU_RC_NP

// ---- per-image InstanceNorm parameter gradients -> dgamma / dbeta, every layer of a backward pass in ONE launch (round 4).
// The small-map backward kernels (CU_NORM_PARAM_PARTS; cu_conv_epilogue mode 5) leave sum gl / sum gl zhat of every (image,
// channel) in planes [N][C] instead of adding them into dgamma[c] / dbeta[c] with 64 same-address atomics per channel (17 of
// 31 us of the 16^2 x 480 launch); this kernel adds the images in order: deterministic, one launch per step.
namespace {
__global__ __launch_bounds__(256) void norm_param_grads_batch_kernel(const cu_pgrad_item* __restrict__ items, char* base) {
    // workgroup = (layer, 32 channels); thread = (channel, one of 8 image groups): the groups' loads are independent (a
    // single thread walking the 64 images serially took 60 us for the step's 28 layers), the groups are added in order
    __shared__ float red[2][8][32];
    const cu_pgrad_item it = items[blockIdx.x];
    const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int c = blockIdx.y * 32 + cl;
    if (blockIdx.y * 32 >= it.C) return;
    float sg = 0.f, sb = 0.f;
    if (c < it.C) {
        const int per = (it.N + 7) / 8, n0 = grp * per, n1 = min(it.N, n0 + per);
        for (int n = n0; n < n1; ++n) {
            if (it.dgamma_parts) sg += it.dgamma_parts[(size_t)n * it.C + c];
            if (it.dbeta_parts) sb += it.dbeta_parts[(size_t)n * it.C + c];
        }
    }
    red[0][grp][cl] = sg;
    red[1][grp][cl] = sb;
    __syncthreads();
    if (grp == 0 && c < it.C) {
        float tg = 0.f, tb = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { tg += red[0][k][cl]; tb += red[1][k][cl]; }
        if (it.dgamma_parts) reinterpret_cast<float*>(base + it.dgamma_off)[c] += tg;
        if (it.dbeta_parts) reinterpret_cast<float*>(base + it.dbeta_off)[c] += tb;
    }
}
}  // namespace

extern "C" int cu_norm_param_grads_batch(const cu_pgrad_item* items, int n_items, int max_c, float* grad_base, void* stream) {
    CU_CHECK_ARG(items && n_items > 0 && max_c > 0, "cu_norm_param_grads_batch: bad argument");
    hipLaunchKernelGGL(norm_param_grads_batch_kernel, dim3(n_items, cdiv(max_c, 32)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), items, reinterpret_cast<char*>(grad_base));
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_act_bwd(int dtype, int N, int HW, int C, void* g, const void* z, float slope, float* dbias,
                          void* stream) {
    NORM_COMMON_CHECKS("cu_act_bwd");
    CU_CHECK_ARG(g && z, "cu_act_bwd: null pointer");
    const size_t lds = sizeof(float) * (size_t)rm.rows * rm.tpp * PIECE;
    dim3 grid(N, nchunks);
    if (dtype == CU_BF16)
        hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, grid, dim3(NT), lds, st, (bf16_t*)g, (const bf16_t*)z, slope, dbias, HW,
                           C, rm.tpp, rm.rows, chunk);
    else
        hipLaunchKernelGGL(act_bwd_kernel<float>, grid, dim3(NT), lds, st, (float*)g, (const float*)z, slope, dbias, HW, C,
                           rm.tpp, rm.rows, chunk);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_act_bwd_det(int dtype, int N, int HW, int C, void* g, const void* z, float slope, float* dbias,
                              void* stream) {
    // elementwise in g; dbias sums over every pixel of every image: ONE workgroup walks the batch as one image of N * HW
    // pixels, so the sum has a fixed order and a single adder (the heads this serves are a few hundred pixels)
    CU_CHECK_ARG(dtype == CU_F32 || dtype == CU_BF16, "cu_act_bwd_det: bad dtype %d", dtype);
    const int PIECE = dtype == CU_BF16 ? 8 : 4;
    CU_CHECK_ARG(N > 0 && HW > 0 && C > 0 && C % PIECE == 0 && C / PIECE <= NT && g && z, "cu_act_bwd_det: bad argument");
    CU_CHECK_ARG((long long)N * HW <= (1 << 20), "cu_act_bwd_det: %d x %d pixels for one workgroup", N, HW);
    const RowMap rm = row_map(C, PIECE);
    const int px = N * HW;
    const int chunk = cdiv(px, rm.rows) * rm.rows;
    const size_t lds = sizeof(float) * (size_t)rm.rows * rm.tpp * PIECE;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == CU_BF16)
        hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, dim3(1, 1), dim3(NT), lds, st, (bf16_t*)g, (const bf16_t*)z, slope, dbias, px,
                           C, rm.tpp, rm.rows, chunk);
    else
        hipLaunchKernelGGL(act_bwd_kernel<float>, dim3(1, 1), dim3(NT), lds, st, (float*)g, (const float*)z, slope, dbias, px, C,
                           rm.tpp, rm.rows, chunk);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_act_to_nchw_f32(int dtype, int N, int HW, int C, const void* z, const float* stats, float slope,
                                  float* out, void* stream) {
    CU_CHECK_ARG(dtype == CU_F32 || dtype == CU_BF16, "cu_act_to_nchw_f32: bad dtype");
    CU_CHECK_ARG(z && out && N > 0 && HW > 0 && C > 0, "cu_act_to_nchw_f32: bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t total = (size_t)N * HW * C;
    const int blocks = (int)((total + 255) / 256);
    if (dtype == CU_BF16)
        hipLaunchKernelGGL(act_to_nchw_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)z, stats, slope, out,
                           N, HW, C);
    else
        hipLaunchKernelGGL(act_to_nchw_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)z, stats, slope, out, N,
                           HW, C);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_nchw_f32_to_nhwc(int dtype, int N, int HW, int C, int CP, const float* in, void* out, void* stream) {
    CU_CHECK_ARG(dtype == CU_F32 || dtype == CU_BF16, "cu_nchw_f32_to_nhwc: bad dtype");
    CU_CHECK_ARG(in && out && N > 0 && HW > 0 && C > 0 && CP >= C && CP % 8 == 0, "cu_nchw_f32_to_nhwc: bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t total = (size_t)HW * (CP / 8);
    dim3 grid((unsigned)((total + 255) / 256), N);
    if (dtype == CU_BF16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid, dim3(256), 0, st, in, (bf16_t*)out, HW, C, CP);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, st, in, (float*)out, HW, C, CP);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_nhwc_to_nchw_f32(int dtype, int N, int HW, int C, const void* in, float* out, int accumulate,
                                   void* stream) {
    CU_CHECK_ARG(dtype == CU_F32 || dtype == CU_BF16, "cu_nhwc_to_nchw_f32: bad dtype");
    CU_CHECK_ARG(in && out && N > 0 && HW > 0 && C > 0, "cu_nhwc_to_nchw_f32: bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t total = (size_t)N * HW * C;
    const int blocks = (int)((total + 255) / 256);
    if (dtype == CU_BF16)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)in, out, N, HW, C,
                           accumulate);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)in, out, N, HW, C,
                           accumulate);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_channel_scale(int dtype, int N, int HW, int C, void* x, const float* mask, void* stream) {
    CU_CHECK_ARG(dtype == CU_F32 || dtype == CU_BF16, "cu_channel_scale: bad dtype");
    CU_CHECK_ARG(x && mask && N > 0 && HW > 0 && C > 0, "cu_channel_scale: bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const size_t total = (size_t)N * HW * C;
    const int blocks = (int)((total + 255) / 256);
    if (dtype == CU_BF16)
        hipLaunchKernelGGL(channel_scale_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (bf16_t*)x, mask, N, HW, C);
    else
        hipLaunchKernelGGL(channel_scale_kernel<float>, dim3(blocks), dim3(256), 0, st, (float*)x, mask, N, HW, C);
    CU_LAUNCH_CHECK();
    return 0;
}
