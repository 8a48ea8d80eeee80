// Small kernels of libcontour_hip.so: first-layer direct conv (Cin = 1), operand-copy preparation, fused Adam.
#include <stdlib.h>

#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------- first conv, Cin = 1
// reference: input_block.conv1.conv = nn.Conv2d(1, 32, 3, 1, 1) (models/nnUnet/unet2.py:113-119, layers.py:192).
// 0.13 % of the network's MACs and HBM-bound on its 64-byte/pixel store, so plain VALU FMAs.
// One thread = one 16-byte piece of channels for C1_PX consecutive pixels of a row: the 9 x PIECE weights are loaded once
// per thread and the 3 x (C1_PX + 2) image window slides (the first version reloaded 72 weights per output piece).
constexpr int C1_PX = 8;
// MODE 0: z = conv + bias -> dst.
// MODE 2 (round 3): z -> dst AND LeakyReLU(z * scale + shift) -> dst2 from finished statistics (stats planes 2 and 3, see
//         c1_moments_kernel): the layer's conv and apply passes in one, each tensor written once and none re-read.
// MODE 3: only the activation (dst2); z is never stored -- the backward (c1_bwd_kernel) recomputes it from the image with
//         c1_z(), the SAME expressions, so that both sides decide the LeakyReLU branch on the same value.
template <typename T, int PIECE>
__device__ __forceinline__ void c1_z(const float (&v)[9], const float (&wr)[9][PIECE], const float (&b)[PIECE], float (&z)[PIECE]) {
    // channel pairs as 2-vectors: v_pk_fma_f32 (two IEEE fmas per lane and instruction; the same values as fmaf per channel)
    f32x2 zz[PIECE / 2];
#pragma unroll
    for (int e = 0; e < PIECE / 2; ++e) zz[e] = f32x2{b[2 * e], b[2 * e + 1]};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const f32x2 vv = {v[t], v[t]};
#pragma unroll
        for (int e = 0; e < PIECE / 2; ++e)
            zz[e] = __builtin_elementwise_fma(vv, f32x2{wr[t][2 * e], wr[t][2 * e + 1]}, zz[e]);
    }
#pragma unroll
    for (int e = 0; e < PIECE / 2; ++e) { z[2 * e] = zz[e][0]; z[2 * e + 1] = zz[e][1]; }
}
// the value the layer's consumers see: z as stored (rounded to the storage type)
template <typename T> __device__ __forceinline__ float c1_stored(float z) {
    if constexpr (sizeof(T) == 2) return bf16_to_f32(f32_to_bf16(z));
    return z;
}
template <typename T, int MODE>
__global__ __launch_bounds__(256) void conv_c1_fwd_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                          const float* __restrict__ bias, T* __restrict__ dst, int N,
                                                          int H, int W, int CO, const float* __restrict__ stats,
                                                          float slope, T* __restrict__ dst2) {
    constexpr int PIECE = Elem<T>::PIECE;
    const int ppp = CO / PIECE;
    const int wgroups = (W + C1_PX - 1) / C1_PX;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)N * H * wgroups * ppp;
    if (i >= total) return;
    const int piece = (int)(i % ppp);
    size_t r = i / ppp;
    const int x0 = (int)(r % wgroups) * C1_PX; r /= wgroups;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    float wr[9][PIECE], b[PIECE];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < PIECE; ++e) wr[t][e] = w[t * CO + piece * PIECE + e];
#pragma unroll
    for (int e = 0; e < PIECE; ++e) b[e] = bias ? bias[piece * PIECE + e] : 0.f;
    float sc[MODE >= 2 ? PIECE : 1], sh[MODE >= 2 ? PIECE : 1];
    if constexpr (MODE >= 2) {
        const size_t NC = (size_t)N * CO, si = (size_t)n * CO + piece * PIECE;
#pragma unroll
        for (int e = 0; e < PIECE; ++e) { sc[e] = stats[2 * NC + si + e]; sh[e] = stats[3 * NC + si + e]; }
    }
    float win[3][C1_PX + 2];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int sy = y + dy - 1;
#pragma unroll
        for (int k = 0; k < C1_PX + 2; ++k) {
            const int sx = x0 + k - 1;
            win[dy][k] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? img[((size_t)n * H + sy) * W + sx] : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < C1_PX; ++k) {
        if (x0 + k >= W) break;
        float acc[PIECE], v9[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) v9[t] = win[t / 3][k + t % 3];
        c1_z<T, PIECE>(v9, wr, b, acc);
        const size_t o = (((size_t)n * H + y) * W + x0 + k) * CO + piece * PIECE;
        if constexpr (MODE != 3) store_piece<T>(dst + o, acc);
        if constexpr (MODE >= 2) {
            float a[PIECE];
#pragma unroll
            for (int e = 0; e < PIECE; ++e) {
                const float yv = fmaf(c1_stored<T>(acc[e]), sc[e], sh[e]);
                a[e] = yv > 0.f ? yv : yv * slope;
            }
            store_piece<T>(dst2 + o, a);
        }
    }
}

// InstanceNorm statistics of the first layer's z = conv3x3(img) + bias WITHOUT computing z: z is linear in the image, so
//   sum_p (z - bias)[c]   = sum_t w[t][c] S[t],           S[t]     = sum_p x[p + t]
//   sum_p (z - bias)[c]^2 = sum_{t,u} w[t][c] w[u][c] R[t][u],  R[t][u] = sum_p x[p + t] x[p + u]
// with x zero outside the image (the conv's padding): 9 + 45 moments per image gathered in one pass over the IMAGE
// (17 MB at batch 64, against 268 MB for a statistics pass over z).  One thread = C1M_PX pixels of a row; per-workgroup
// partial moments go to part[n][wg][54] and are summed in workgroup order by c1_stats_kernel: deterministic.  The
// statistics are those of z BEFORE its rounding to the storage type, as in the streaming kernel's epilogue (tconv.hip).
constexpr int C1M_PX = 16;
constexpr int C1M_N = 54;
constexpr int C1M_LD = 68;           // LDS row stride (floats): 16-byte aligned rows
__global__ __launch_bounds__(256) void c1_moments_kernel(const float* __restrict__ img, float* __restrict__ part, int H, int W) {
    __shared__ __attribute__((aligned(16))) float red[4 * C1M_N * C1M_LD];
    __shared__ float tot[4][C1M_N];
    const int n = blockIdx.y;
    const int segs = (W + C1M_PX - 1) / C1M_PX;
    const int s = blockIdx.x * 256 + threadIdx.x;
    float m[C1M_N];
#pragma unroll
    for (int j = 0; j < C1M_N; ++j) m[j] = 0.f;
    if (s < H * segs) {
        const int y = s / segs, x0 = (s % segs) * C1M_PX;
        float win[3][C1M_PX + 2];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int sy = y + dy - 1;
#pragma unroll
            for (int k = 0; k < C1M_PX + 2; ++k) {
                const int sx = x0 + k - 1;
                win[dy][k] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? img[((size_t)n * H + sy) * W + sx] : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < C1M_PX; ++k) {
            const float live = x0 + k < W ? 1.f : 0.f;
            float v[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) v[t] = win[t / 3][k + t % 3] * live;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                m[t] += v[t];
#pragma unroll
                for (int u = t; u < 9; ++u) {
                    const int j = 9 + t * 9 - t * (t - 1) / 2 + (u - t);        // row t of the upper triangle
                    m[j] = fmaf(v[t], v[u], m[j]);
                }
            }
        }
    }
    // reduce over the workgroup through LDS (a shuffle tree costs 6 cross-lane steps per moment: 5x the pixel arithmetic):
    // every lane stores its 54 moments as a column, lane j < 54 then sums row j of its wave, wave 0 the four waves
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* mine = red + wave * (C1M_N * C1M_LD);
#pragma unroll
    for (int j = 0; j < C1M_N; ++j) mine[j * C1M_LD + lane] = m[j];
    __syncthreads();
    if (lane < C1M_N) {
        const f32x4* row = reinterpret_cast<const f32x4*>(mine + lane * C1M_LD);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const f32x4 v = row[k]; a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3]; }
        tot[wave][lane] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    if (threadIdx.x < C1M_N)
        part[((size_t)n * gridDim.x + blockIdx.x) * C1M_N + threadIdx.x] =
            (tot[0][threadIdx.x] + tot[1][threadIdx.x]) + (tot[2][threadIdx.x] + tot[3][threadIdx.x]);
}

// moments -> the four statistics planes (mean, rstd, scale, shift) of cu_instnorm_stats.  One 64-thread workgroup per image:
// thread j < 54 sums moment j over the workgroups' partials (fixed order), then thread c (strided over CO) evaluates the
// 81-term quadratic form in double (it cancels: mean^2 against the second moment).
__global__ __launch_bounds__(64) void c1_stats_kernel(const float* __restrict__ part, int nwg, const float* __restrict__ w,
                                                      const float* __restrict__ bias, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps,
                                                      float* __restrict__ stats, int N, int HW, int CO) {
    __shared__ double m[C1M_N];
    const int n = blockIdx.x;
    if (threadIdx.x < C1M_N) {
        double a = 0.0;
        for (int g = 0; g < nwg; ++g) a += (double)part[((size_t)n * nwg + g) * C1M_N + threadIdx.x];
        m[threadIdx.x] = a;
    }
    __syncthreads();
    const double inv = 1.0 / (double)HW;
    const size_t NC = (size_t)N * CO;
    for (int c = threadIdx.x; c < CO; c += 64) {
        double wt[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t] = (double)w[t * CO + c];
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            s1 += wt[t] * m[t];
#pragma unroll
            for (int u = t; u < 9; ++u)
                s2 += (u == t ? 1.0 : 2.0) * wt[t] * wt[u] * m[9 + t * 9 - t * (t - 1) / 2 + (u - t)];
        }
        const double m1 = s1 * inv, m2 = s2 * inv;
        const double var = fmax(m2 - m1 * m1, 0.0);
        const float mean = (float)((bias ? (double)bias[c] : 0.0) + m1);
        const float rstd = 1.f / sqrtf((float)var + eps);
        const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
        const size_t i = (size_t)n * CO + c;
        stats[i] = mean;
        stats[NC + i] = rstd;
        stats[2 * NC + i] = g * rstd;
        stats[3 * NC + i] = b - mean * g * rstd;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void conv_c1_wgrad_kernel(const float* __restrict__ img, const T* __restrict__ dz,
                                                            float* __restrict__ dw, int N, int H, int W, int CO,
                                                            int chunk) {
    constexpr int PIECE = Elem<T>::PIECE;
    extern __shared__ float lds[];
    const int ppp = CO / PIECE;
    const int rows = 256 / ppp;
    const int piece = threadIdx.x % ppp, prow = threadIdx.x / ppp;
    const int n = blockIdx.y;
    const int HWn = H * W;
    const int p0 = blockIdx.x * chunk, p1 = min(HWn, p0 + chunk);
    float acc[9][PIECE];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < PIECE; ++e) acc[t][e] = 0.f;
    if (prow < rows) {
        for (int p = p0 + prow; p < p1; p += rows) {
            const int y = p / W, x = p - y * W;
            float g[PIECE];
            load_piece<T>(dz + ((size_t)n * HWn + p) * CO + piece * PIECE, g);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int sy = y + t / 3 - 1, sx = x + t % 3 - 1;
                const float v = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? img[((size_t)n * H + sy) * W + sx] : 0.f;
#pragma unroll
                for (int e = 0; e < PIECE; ++e) acc[t][e] += v * g[e];
            }
        }
        float* d = lds + ((size_t)prow * ppp + piece) * 9 * PIECE;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < PIECE; ++e) d[t * PIECE + e] = acc[t][e];
    }
    __syncthreads();
    // 9*CO outputs per block, summed over the rows by the first 9*CO threads (strided if more outputs than threads)
    for (int o = threadIdx.x; o < 9 * CO; o += 256) {
        const int t = o / CO, c = o - t * CO;
        const int pc = c / PIECE, e = c - pc * PIECE;
        float s = 0.f;
        for (int rr = 0; rr < rows; ++rr) s += lds[((size_t)rr * ppp + pc) * 9 * PIECE + t * PIECE + e];
        unsafeAtomicAdd(dw + o, s);
    }
}

// The same sums with the image rows of the chunk staged in LDS (zero-padded border) and four dz pieces in flight per thread:
// the form above issues nine 4-byte global loads per 16-byte piece of dz and ran at a third of HBM speed (164 us for the
// 268 MB of dz at 256^2 x 32 channels x 64 images).  Chunks are whole rows; CO / PIECE a power of two <= 16.
template <typename T>
__global__ __launch_bounds__(256) void conv_c1_wgrad_rows_kernel(const float* __restrict__ img, const T* __restrict__ dz,
                                                                 float* __restrict__ dw, int N, int H, int W, int CO, int R,
                                                                 float* __restrict__ part) {
    constexpr int PIECE = Elem<T>::PIECE;
    extern __shared__ float lds[];                 // (R + 2) x (W + 2) image rows, afterwards the reduction scratch
    const int ppp = CO / PIECE, rows = 256 / ppp;
    const int piece = threadIdx.x % ppp, prow = threadIdx.x / ppp;
    const int n = blockIdx.y, y0 = blockIdx.x * R;
    const int WP = W + 2;
    const int Rv = min(R, H - y0);
    for (int i = threadIdx.x; i < (R + 2) * WP; i += 256) {
        const int ry = i / WP, rx = i - ry * WP;
        const int sy = y0 + ry - 1, sx = rx - 1;
        lds[i] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? img[((size_t)n * H + sy) * W + sx] : 0.f;
    }
    __syncthreads();
    float acc[9][PIECE];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < PIECE; ++e) acc[t][e] = 0.f;
    const int npx = Rv * W;
    const T* dzb = dz + ((size_t)n * H + y0) * W * CO + piece * PIECE;
    auto add = [&](int q, const float (&g)[PIECE]) {
        const int y = q / W, x = q - y * W;
        const float* s = lds + y * WP + x;         // top-left of the 3x3 window in padded coordinates
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float v = s[(t / 3) * WP + t % 3];
#pragma unroll
            for (int e = 0; e < PIECE; ++e) acc[t][e] += v * g[e];
        }
    };
    int p = prow;
    for (; p + 3 * rows < npx; p += 4 * rows) {
        float g[4][PIECE];
#pragma unroll
        for (int u = 0; u < 4; ++u) load_piece<T>(dzb + (size_t)(p + u * rows) * CO, g[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) add(p + u * rows, g[u]);
    }
    for (; p < npx; p += rows) {
        float g[PIECE];
        load_piece<T>(dzb + (size_t)p * CO, g);
        add(p, g);
    }
    // lane = (prow % (64 / ppp)) * ppp + piece: lanes 16 and 32 apart hold the same piece of other pixel rows
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            acc[t][e] += __shfl_xor(acc[t][e], 32);
            acc[t][e] += __shfl_xor(acc[t][e], 16);
        }
    __syncthreads();                               // every wave is done with the image rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rpw = 16 / ppp;                      // partial rows per wave that are left
    if (lane < 16) {
        float* d = lds + ((size_t)(wave * rpw + lane / ppp) * ppp + piece) * 9 * PIECE;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < PIECE; ++e) d[t * PIECE + e] = acc[t][e];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 9 * CO; o += 256) {
        const int t = o / CO, c = o - t * CO;
        const int pc = c / PIECE, e = c - pc * PIECE;
        float sum = 0.f;
        for (int rr = 0; rr < 4 * rpw; ++rr) sum += lds[((size_t)rr * ppp + pc) * 9 * PIECE + t * PIECE + e];
        if (part) part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 9 * CO + o] = sum;      // deterministic mode
        else unsafeAtomicAdd(dw + o, sum);
    }
}

// deterministic mode: dw[o] += the workgroups' partial sums in workgroup order
__global__ __launch_bounds__(256) void conv_c1_wgrad_finish_kernel(const float* __restrict__ part, int nwg, int n_out,
                                                                   float* __restrict__ dw) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out) return;
    float s = 0.f;
    for (int g = 0; g < nwg; ++g) s += part[(size_t)g * n_out + o];
    dw[o] += s;
}

// First layer, whole backward without z and without dz (round 3).  The layer's dz feeds nothing but its own 9 x CO weight
// gradient (there is no input gradient), and its z is 9 FMAs per channel away from the image: so neither tensor needs to
// exist.  Two passes over g = dL/da (each: 268 MB at batch 64 instead of read g + z, write dz, read dz = 1.07 GB):
//   PASS 1: z recomputed (c1_z), gl = g LeakyReLU'(y), per-(image, channel) sums of gl and gl zhat -> sums [N][CO][2] (+=);
//   PASS 2: dz = gamma rstd (gl - S1/HW - zhat S2/HW) (bwd_apply_kernel's expression, norm.hip) formed in registers and
//           accumulated into dw[t][c] += x[p + t] dz[p][c]; dgamma / dbeta += the image's sums.
// Geometry of conv_c1_wgrad_rows_kernel: a workgroup owns R image rows (staged in LDS with a zero border), thread =
// (piece of channels, pixel row slot), four g pieces in flight.
template <typename T, int PASS>
__global__ __launch_bounds__(256, 2) void c1_bwd_kernel(const float* __restrict__ img, const T* __restrict__ g,
                                                     const float* __restrict__ w, const float* __restrict__ bias,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     float slope, float* __restrict__ sums, float* __restrict__ dw,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int H, int W,
                                                     int CO, int R) {
    constexpr int PIECE = Elem<T>::PIECE;
    constexpr int NV = PASS == 1 ? 2 : 9;          // accumulator rows per thread
    extern __shared__ float lds[];                 // (R + 2) x (W + 2) image rows, afterwards the reduction scratch
    const int ppp = CO / PIECE, rows = 256 / ppp;
    const int piece = threadIdx.x % ppp, prow = threadIdx.x / ppp;
    const int n = blockIdx.y, y0 = blockIdx.x * R;
    const int WP = W + 2;
    const int Rv = min(R, H - y0);
    for (int i = threadIdx.x; i < (R + 2) * WP; i += 256) {
        const int ry = i / WP, rx = i - ry * WP;
        const int sy = y0 + ry - 1, sx = rx - 1;
        lds[i] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? img[((size_t)n * H + sy) * W + sx] : 0.f;
    }
    // PASS 1 accumulates sum gl and sum gl z (zhat = (z - mean) rstd is linear in z: fixed up once at the end);
    // PASS 2 uses dz = gr gl + B z + C with B = -gr a2 rstd, C = -gr a1 + gr a2 rstd mean (bwd_apply_kernel's expression, expanded)
    float wr[9][PIECE], b[PIECE], sc[PIECE], sh[PIECE], gr[PIECE], cB[PIECE], cC[PIECE];
    const size_t NC = (size_t)N * CO, sidx = (size_t)n * CO + piece * PIECE;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < PIECE; ++e) wr[t][e] = w[t * CO + piece * PIECE + e];
    const float inv = 1.f / (float)(H * W);
#pragma unroll
    for (int e = 0; e < PIECE; ++e) {
        b[e] = bias ? bias[piece * PIECE + e] : 0.f;
        sc[e] = stats[2 * NC + sidx + e]; sh[e] = stats[3 * NC + sidx + e];
        if constexpr (PASS == 2) {
            const float mean = stats[sidx + e], rstd = stats[NC + sidx + e];
            const float a1 = sums[(sidx + e) * 2] * inv, a2 = sums[(sidx + e) * 2 + 1] * inv;
            gr[e] = (gamma ? gamma[piece * PIECE + e] : 1.f) * rstd;
            cB[e] = -gr[e] * a2 * rstd;
            cC[e] = -gr[e] * a1 - cB[e] * mean;
        }
    }
    if constexpr (PASS == 2) {
        if (blockIdx.x == 0 && prow == 0) {          // one contribution per (image, channel)
#pragma unroll
            for (int e = 0; e < PIECE; ++e) {
                if (dbeta) unsafeAtomicAdd(dbeta + piece * PIECE + e, sums[(sidx + e) * 2]);
                if (dgamma) unsafeAtomicAdd(dgamma + piece * PIECE + e, sums[(sidx + e) * 2 + 1]);
            }
        }
    }
    __syncthreads();
    float acc[NV][PIECE];
#pragma unroll
    for (int t = 0; t < NV; ++t)
#pragma unroll
        for (int e = 0; e < PIECE; ++e) acc[t][e] = 0.f;
    const int npx = Rv * W;
    const T* gb = g + ((size_t)n * H + y0) * W * CO + piece * PIECE;
    auto add = [&](int q, const float (&gv)[PIECE]) {
        const int y = q / W, x = q - y * W;
        const float* s = lds + y * WP + x;         // top-left of the 3x3 window in padded coordinates
        float v[9], z[PIECE];
#pragma unroll
        for (int t = 0; t < 9; ++t) v[t] = s[(t / 3) * WP + t % 3];
        c1_z<T, PIECE>(v, wr, b, z);
        float dzv[PIECE];
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            const float zr = c1_stored<T>(z[e]);
            const float yv = fmaf(zr, sc[e], sh[e]);
            const float gl = yv > 0.f ? gv[e] : gv[e] * slope;
            dzv[e] = 0.f;
            if constexpr (PASS == 1) {
                acc[0][e] += gl;
                acc[1][e] = fmaf(gl, zr, acc[1][e]);
            } else {
                dzv[e] = fmaf(zr, cB[e], fmaf(gr[e], gl, cC[e]));
            }
        }
        if constexpr (PASS == 2) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const f32x2 vv = {v[t], v[t]};
#pragma unroll
                for (int e = 0; e < PIECE / 2; ++e) {
                    const f32x2 r = __builtin_elementwise_fma(vv, f32x2{dzv[2 * e], dzv[2 * e + 1]}, f32x2{acc[t][2 * e], acc[t][2 * e + 1]});
                    acc[t][2 * e] = r[0]; acc[t][2 * e + 1] = r[1];
                }
            }
        }
    };
    constexpr int UF = PASS == 1 ? 4 : 2;          // g pieces in flight per thread (PASS 2 holds 2 x 72 registers of w and dw)
    int p = prow;
    for (; p + (UF - 1) * rows < npx; p += UF * rows) {
        float gv[UF][PIECE];
#pragma unroll
        for (int u = 0; u < UF; ++u) load_piece<T>(gb + (size_t)(p + u * rows) * CO, gv[u]);
#pragma unroll
        for (int u = 0; u < UF; ++u) add(p + u * rows, gv[u]);
    }
    for (; p < npx; p += rows) {
        float gv[PIECE];
        load_piece<T>(gb + (size_t)p * CO, gv);
        add(p, gv);
    }
    // lane = (prow % (64 / ppp)) * ppp + piece: lanes 16 and 32 apart hold the same piece of other pixel rows
#pragma unroll
    for (int t = 0; t < NV; ++t)
#pragma unroll
        for (int e = 0; e < PIECE; ++e) {
            acc[t][e] += __shfl_xor(acc[t][e], 32);
            acc[t][e] += __shfl_xor(acc[t][e], 16);
        }
    __syncthreads();                               // every wave is done with the image rows
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rpw = 16 / ppp;                      // partial rows per wave that are left
    if (lane < 16) {
        float* d = lds + ((size_t)(wave * rpw + lane / ppp) * ppp + piece) * NV * PIECE;
#pragma unroll
        for (int t = 0; t < NV; ++t)
#pragma unroll
            for (int e = 0; e < PIECE; ++e) d[t * PIECE + e] = acc[t][e];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < NV * CO; o += 256) {
        const int t = o / CO, c = o - t * CO;
        const int pc = c / PIECE, e = c - pc * PIECE;
        float sum = 0.f;
        for (int rr = 0; rr < 4 * rpw; ++rr) sum += lds[((size_t)rr * ppp + pc) * NV * PIECE + t * PIECE + e];
        if constexpr (PASS == 1) {
            // row 0: sum gl; row 1: sum gl z -> sum gl zhat = rstd (sum gl z - mean sum gl), per workgroup (linear)
            if (t == 1) {
                float s0 = 0.f;
                for (int rr = 0; rr < 4 * rpw; ++rr) s0 += lds[((size_t)rr * ppp + pc) * NV * PIECE + e];
                sum = stats[NC + (size_t)n * CO + c] * (sum - stats[(size_t)n * CO + c] * s0);
            }
            unsafeAtomicAdd(sums + ((size_t)n * CO + c) * 2 + t, sum);
        } else {
            unsafeAtomicAdd(dw + o, sum);
        }
    }
}

// ---------------------------------------------------------------------------------------- operand copies
// The nn.Parameters keep the reference's logical layouts (Conv2d: [CO][CI][kh][kw]; ConvTranspose2d: [CI][CO][kh][kw]) so
// that reference checkpoints load with strict=True.  The MFMA kernels want tap-major operands:
//   fwd operand   [T][CO][CI]   (rows = GEMM columns, K contiguous)
//   dgrad operand [T][CI][CO]
// element (t, co, ci) of the master sits at  co*s_co + ci*s_ci + t  (s_co, s_ci in elements).  32x32 LDS-tiled per tap.
template <typename T>
__global__ __launch_bounds__(256) void weight_prep_kernel(const float* __restrict__ m, T* __restrict__ wf,
                                                          T* __restrict__ wd, int CO, int CI, int COP, long s_co,
                                                          long s_ci) {
    __shared__ float tile[32][33];
    const int t = blockIdx.z;
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int co = co0 + ty + 8 * k, ci = ci0 + tx;
        float v = 0.f;
        if (co < CO && ci < CI) v = m[(size_t)co * s_co + (size_t)ci * s_ci + t];
        if (co < COP && ci < CI && wf) Elem<T>::st(wf + ((size_t)t * COP + co) * CI + ci, v);
        tile[ty + 8 * k][tx] = v;
    }
    __syncthreads();
    if (wd) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ci = ci0 + ty + 8 * k, co = co0 + tx;
            if (co < COP && ci < CI) Elem<T>::st(wd + ((size_t)t * CI + ci) * COP + co, tile[tx][ty + 8 * k]);
        }
    }
}

// kernel-layout f32 gradient dWk[T][COP][CI] -> logical gradient (same strides as above), rows >= CO dropped
__global__ __launch_bounds__(256) void grad_unprep_kernel(float* __restrict__ dwk, float* __restrict__ g, int CO,
                                                          int CI, int COP, long s_co, long s_ci, int accumulate) {
    const int t = blockIdx.z;
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int co = co0 + ty + 8 * k, ci = ci0 + tx;
        if (co < CO && ci < CI) {
            float* src = dwk + ((size_t)t * COP + co) * CI + ci;
            const float v = *src;
            if (accumulate & 2) *src = 0.f;
            float* o = g + (size_t)co * s_co + (size_t)ci * s_ci + t;
            *o = (accumulate & 1) ? *o + v : v;
        }
    }
}

// The same for logical layouts with the taps innermost (every conv / transposed-conv weight): one workgroup gathers an
// 8 x 32 block of (slow channel, fast channel) with ALL its taps in LDS and writes the logical gradient in contiguous
// rows of 32*T floats (the per-tap kernel above touches every logical line T times, 4 useful bytes per 36).
constexpr int UNPREP_MAXT = 9;
// NT = compile-time tap count (all NT loads of a thread are in flight before the first use; the run-time form waited for
// every load in turn: 10 us per launch on average, 41 launches per step) or 0 = run-time count NTr.
template <int NT>
__global__ __launch_bounds__(256) void grad_unprep_rows_kernel(float* __restrict__ dwk, float* __restrict__ g, int NTr,
                                                               int CO, int CI, int COP, long s_co, long s_ci, int accumulate) {
    __shared__ float tile[8 * (32 * UNPREP_MAXT + 1)];
    const int nt = NT ? NT : NTr;
    const bool co_rows = s_co > s_ci;                  // conv: rows = co, columns = ci; transposed conv: the other way
    const int r0 = blockIdx.y * 8, c0 = blockIdx.x * 32;
    const int RN = co_rows ? CO : CI, CN = co_rows ? CI : CO;
    const int pitch = 32 * nt + 1;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // column tx of row ty
    {
        const int r = r0 + ty, c = c0 + tx;
        const int co = co_rows ? r : c, ci = co_rows ? c : r;
        const bool ok = r < RN && c < CN;
        float* src = dwk + ((size_t)co) * CI + ci;
        const size_t tstride = (size_t)COP * CI;
        if constexpr (NT > 0) {
            float v[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) v[t] = ok ? src[t * tstride] : 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                tile[ty * pitch + tx * NT + t] = v[t];
                if (ok && (accumulate & 2)) src[t * tstride] = 0.f;      // read-and-clear: ready for the next layer
            }
        } else {
            for (int t = 0; t < nt; ++t) {
                tile[ty * pitch + tx * nt + t] = ok ? src[t * tstride] : 0.f;
                if (ok && (accumulate & 2)) src[t * tstride] = 0.f;
            }
        }
    }
    __syncthreads();
    const long s_r = co_rows ? s_co : s_ci;
    const int cvalid = min(32, CN - c0) * nt;
    if (r0 + ty < RN) {
        float* dst = g + (size_t)(r0 + ty) * s_r + (size_t)c0 * nt;
        if constexpr (NT > 0) {
            if (accumulate & 1) {
                float o[NT];
#pragma unroll
                for (int i = 0; i < NT; ++i) o[i] = (tx + 32 * i < cvalid) ? dst[tx + 32 * i] : 0.f;
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    if (tx + 32 * i < cvalid) dst[tx + 32 * i] = o[i] + tile[ty * pitch + tx + 32 * i];
            } else {
#pragma unroll
                for (int i = 0; i < NT; ++i)
                    if (tx + 32 * i < cvalid) dst[tx + 32 * i] = tile[ty * pitch + tx + 32 * i];
            }
        } else {
            for (int k = tx; k < cvalid; k += 32) {
                const float v = tile[ty * pitch + k];
                dst[k] = (accumulate & 1) ? dst[k] + v : v;
            }
        }
    }
}

// Partial tiles of the weight gradient (cu_conv_wgrad_parts): S slabs of E4 16-byte pieces in the accumulators' own
// ("native") layout [block][wave block][weight tap][q][lane][4].  Two element-wise kernels sum them in slab order (a fixed
// summation order, no atomics):
//   parts_reduce_kernel  group j (grid.y) adds the G consecutive slabs j*G .. j*G+G-1 INTO slab j*G -- first level for the
//                        thin layers, whose up to 256 slabs of a small tile would otherwise be walked by a few workgroups;
//   parts_finish_kernel  adds `S` slabs `stride4` pieces apart and writes the plain [T][CO][CI] tile: piece (block, wave
//                        block, t, q, lane) = rows n0 .. n0+3 (n0 = nt*TN + nblk*32 + 8q + 4h), column ct*TC + cblk*32 + r.
__global__ __launch_bounds__(256) void parts_reduce_kernel(float* __restrict__ parts, size_t E4, int S, int G) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= E4) return;
    const int s0 = blockIdx.y * G, n = min(G, S - s0);
    f32x4* base = reinterpret_cast<f32x4*>(parts) + (size_t)s0 * E4 + i;
    f32x4 acc = base[0];
#pragma unroll 8
    for (int s = 1; s < n; ++s) {
        const f32x4 v = base[(size_t)s * E4];
        acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    base[0] = acc;
}

// slab 0 += slabs 1 .. S-1, `stride4` pieces apart, in slab order (plain-layout slabs)
__global__ __launch_bounds__(256) void parts_reduce_strided_kernel(float* __restrict__ parts, size_t E4, size_t stride4, int S) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= E4) return;
    f32x4* base = reinterpret_cast<f32x4*>(parts) + i;
    f32x4 acc = base[0];
#pragma unroll 8
    for (int s = 1; s < S; ++s) {
        const f32x4 v = base[(size_t)s * stride4];
        acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    base[0] = acc;
}

__global__ __launch_bounds__(256) void parts_finish_kernel(const float* __restrict__ parts, size_t E4, size_t stride4, int S,
                                                           float* __restrict__ out, int T, int CO, int CI, int NBLK,
                                                           int CBLK, int ctiles) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= E4) return;
    const int lane = (int)(i & 63), q = (int)((i >> 6) & 3);
    size_t rest = i >> 8;
    const int t = (int)(rest % T); rest /= T;
    const int nwb = NBLK * CBLK;
    const int blk = (int)(rest % nwb);
    const int block = (int)(rest / nwb);
    const int nt = block / ctiles, ct = block - nt * ctiles;
    const int nblk = blk / CBLK, cblk = blk - nblk * CBLK;
    const int n0 = (nt * NBLK + nblk) * 32 + 8 * q + 4 * (lane >> 5);
    const int c = (ct * CBLK + cblk) * 32 + (lane & 31);
    if (n0 >= CO || c >= CI) return;              // regions of inactive waves are never written: never read either
    const f32x4* base = reinterpret_cast<const f32x4*>(parts) + i;
    f32x4 acc = base[0];
#pragma unroll 8
    for (int s = 1; s < S; ++s) {
        const f32x4 v = base[(size_t)s * stride4];
        acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    float* o = out + ((size_t)t * CO + n0) * CI + c;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (n0 + e < CO) o[(size_t)e * CI] = acc[e];
}

// One launch instead of parts_finish_kernel + grad_unprep_rows_kernel (round 4: 73 launches per step less): a workgroup owns
// an 8 x 32 block of (slow channel, fast channel) of the LOGICAL gradient with all its NT taps, adds the S slabs of exactly
// those elements in slab order (the same fixed summation order as before: bit-identical sums) and writes the logical rows
// through LDS like grad_unprep_rows_kernel.
//   NATIVE (conv layouts: rows = co): in the accumulators' layout an 8 (co) x 32 (ci) block of one tap is ONE contiguous
//     1-KiB run -- pieces (q, lane = 32 h + r) = rows 8q + 4h .. + 3 of the 32-row wave block, column r -- so a wave reads a
//     slab's share of a tap with one 16-byte load per lane;
//   plain slabs [T][COP][CI] (gemm_tn.hip, the fused head): element loads, 128-byte rows (conv) or 32-byte runs (transposed).
template <int NT, bool NATIVE>
__global__ __launch_bounds__(256) void parts_sum_unprep_kernel(const float* __restrict__ parts, size_t stride, int S,
                                                               float* __restrict__ g, int CO, int CI, int COP, long s_co,
                                                               long s_ci, int NBLK, int CBLK, int ctiles, int accumulate) {
    __shared__ float tile[8 * (32 * UNPREP_MAXT + 1)];
    constexpr int pitch = 32 * NT + 1;
    const bool co_rows = s_co > s_ci;
    const int r0 = blockIdx.y * 8, c0 = blockIdx.x * 32;
    const int RN = co_rows ? CO : CI, CN = co_rows ? CI : CO;
    if constexpr (NATIVE) {          // co_rows only (checked by the host): rows = co, columns = ci
        const int nt_ = r0 / (32 * NBLK), nblk = (r0 >> 5) % NBLK, q = (r0 & 31) >> 3;
        const int ct = c0 / (32 * CBLK), cblk = (c0 >> 5) % CBLK;
        const size_t blk = (size_t)(nt_ * ctiles + ct) * (NBLK * CBLK) + (nblk * CBLK + cblk);
        for (int pidx = threadIdx.x; pidx < NT * 64; pidx += 256) {
            const int t = pidx >> 6, lane = pidx & 63;
            const int n0 = r0 + 4 * (lane >> 5), c = c0 + (lane & 31);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (n0 < COP && c < CI) {            // regions of inactive waves are never written: never read either
                const f32x4* base = reinterpret_cast<const f32x4*>(parts) + (((blk * NT + t) * 4 + q) * 64 + lane);
                acc = base[0];
#pragma unroll 8
                for (int s = 1; s < S; ++s) {
                    const f32x4 v = base[(size_t)s * (stride / 4)];
                    acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[(4 * (lane >> 5) + e) * pitch + (lane & 31) * NT + t] = acc[e];
        }
    } else {
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        const int r = r0 + ty, c = c0 + tx;
        const int co = co_rows ? r : c, ci = co_rows ? c : r;
        const bool ok = r < RN && c < CN;
        const float* src = parts + (size_t)co * CI + ci;
        const size_t tstride = (size_t)COP * CI;
        float v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) v[t] = ok ? src[t * tstride] : 0.f;
        for (int s = 1; s < S; ++s) {
            float w[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) w[t] = ok ? src[(size_t)s * stride + t * tstride] : 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) v[t] += w[t];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) tile[ty * pitch + tx * NT + t] = v[t];
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long s_r = co_rows ? s_co : s_ci;
    const int cvalid = min(32, CN - c0) * NT;
    if (r0 + ty < RN) {
        float* dst = g + (size_t)(r0 + ty) * s_r + (size_t)c0 * NT;
        if (accumulate & 1) {
            float o[NT];
#pragma unroll
            for (int i = 0; i < NT; ++i) o[i] = (tx + 32 * i < cvalid) ? dst[tx + 32 * i] : 0.f;
#pragma unroll
            for (int i = 0; i < NT; ++i)
                if (tx + 32 * i < cvalid) dst[tx + 32 * i] = o[i] + tile[ty * pitch + tx + 32 * i];
        } else {
#pragma unroll
            for (int i = 0; i < NT; ++i)
                if (tx + 32 * i < cvalid) dst[tx + 32 * i] = tile[ty * pitch + tx + 32 * i];
        }
    }
}

template <bool NATIVE>
static bool launch_sum_unprep(int T, const float* parts, size_t stride, int S, float* grad, int CO, int CI, int COP, long s_co,
                              long s_ci, int NBLK, int CBLK, int ctiles, int accumulate, hipStream_t st) {
    const bool co_rows = s_co > s_ci;
    dim3 grid(cdiv(co_rows ? CI : CO, 32), cdiv(co_rows ? CO : CI, 8));
#define CU_SUM_UNPREP(NT_) hipLaunchKernelGGL((parts_sum_unprep_kernel<NT_, NATIVE>), grid, dim3(256), 0, st, parts, stride, S, \
                                              grad, CO, CI, COP, s_co, s_ci, NBLK, CBLK, ctiles, accumulate)
    if (T == 9) CU_SUM_UNPREP(9);
    else if (T == 4) CU_SUM_UNPREP(4);
    else if (T == 1) CU_SUM_UNPREP(1);
    else return false;
#undef CU_SUM_UNPREP
    return true;
}

// ---- batched form: the operand copies of every conv layer of the network in ONE launch ---------------------------------
__device__ __forceinline__ const cu_prep_item* find_item(const cu_prep_item* items, int n, int blk) {
    int i = 0;
    while (i + 1 < n && items[i + 1].blk0 <= blk) ++i;       // n <= a few dozen, blk0 ascending
    return items + i;
}

// One workgroup = a 32 x 32 block of (co, ci) with ALL its taps.  The logical layouts keep the taps innermost, so along
// the slower channel dimension a block is 32 rows of 32*T contiguous floats: coalesced on the logical side, and the
// kernel layouts are written / read in 64- or 128-byte rows.  (A per-tap version read every logical line T times with
// 4 useful bytes per 36: 0.5 ms per pass over the 25 M weights; it survives as the non-batched entry point.)
constexpr int PREP_MAXT = 9;

template <typename T> __device__ __forceinline__ void store_pair(T* p, float a, float b);
template <> __device__ __forceinline__ void store_pair<float>(float* p, float a, float b) {
    f32x2 t = {a, b};
    *reinterpret_cast<f32x2*>(p) = t;
}
template <> __device__ __forceinline__ void store_pair<bf16_t>(bf16_t* p, float a, float b) {
    *reinterpret_cast<unsigned*>(p) = (unsigned)f32_to_bf16(a) | ((unsigned)f32_to_bf16(b) << 16);
}

// NT = compile-time tap count: the 4 NT loads of a thread are all in flight before the first LDS store (the run-time loop
// waited for each in turn and the pass ran at a quarter of HBM speed), and the copies leave as pairs (4-byte stores for
// bf16).  NT = 0: run-time count, scalar stores (odd channel counts).
template <typename T, int NT>
__device__ __forceinline__ void prep_tile(const cu_prep_item* it, int b, float* tile) {
    const int tci = b % it->tiles_ci, tco = b / it->tiles_ci;
    const int nt = NT ? NT : it->T;
    const int CO = it->CO, CI = it->CI, COP = it->COP;
    const int pitch = 32 * nt + 1;
    const bool co_rows = it->s_co > it->s_ci;        // conv: rows = co (stride CI*T); transposed conv: rows = ci
    const int co0 = tco * 32, ci0 = tci * 32;
    const int R0 = co_rows ? co0 : ci0, C0 = co_rows ? ci0 : co0;
    const int RN = co_rows ? CO : CI, CN = co_rows ? CI : CO;
    const long long s_r = co_rows ? it->s_co : it->s_ci;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const int cvalid = min(32, CN - C0) * nt;                 // valid floats of a slab row
    if constexpr (NT > 0) {
        float v[4][NT];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = ty + 8 * j;
            const float* src = it->master + (size_t)(R0 + r) * s_r + (size_t)C0 * NT;
#pragma unroll
            for (int i = 0; i < NT; ++i) v[j][i] = (R0 + r < RN && tx + 32 * i < cvalid) ? src[tx + 32 * i] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < NT; ++i) tile[(ty + 8 * j) * pitch + tx + 32 * i] = v[j][i];
    } else {
        for (int r = ty; r < 32; r += 8) {
            const float* src = it->master + (size_t)(R0 + r) * s_r + (size_t)C0 * nt;
            for (int k = tx; k < 32 * nt; k += 32)
                tile[r * pitch + k] = (R0 + r < RN && k < cvalid) ? src[k] : 0.f;
        }
    }
    __syncthreads();
    T* wf = reinterpret_cast<T*>(it->w_fwd);
    T* wd = reinterpret_cast<T*>(it->w_dgrad);
    if (NT > 0 && !(CI & 1) && !(COP & 1)) {
        const int q = (threadIdx.x & 15) * 2, a0 = threadIdx.x >> 4;      // fast-index pair q, q + 1 of slow index a
#pragma unroll
        for (int t = 0; t < (NT ? NT : 1); ++t) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int a = a0 + 16 * k;
                if (wf) {                                // wf[t][co][ci]: co = a, ci = q, q + 1
                    const int co = co0 + a, ci = ci0 + q;
                    const float v0 = co_rows ? tile[a * pitch + q * nt + t] : tile[q * pitch + a * nt + t];
                    const float v1 = co_rows ? tile[a * pitch + (q + 1) * nt + t] : tile[(q + 1) * pitch + a * nt + t];
                    if (co < COP && ci < CI) store_pair<T>(wf + ((size_t)t * COP + co) * CI + ci, v0, v1);
                }
                if (wd) {                                // wd[t][ci][co]: ci = a, co = q, q + 1
                    const int ci = ci0 + a, co = co0 + q;
                    const float v0 = co_rows ? tile[q * pitch + a * nt + t] : tile[a * pitch + q * nt + t];
                    const float v1 = co_rows ? tile[(q + 1) * pitch + a * nt + t] : tile[a * pitch + (q + 1) * nt + t];
                    if (co < COP && ci < CI) store_pair<T>(wd + ((size_t)t * CI + ci) * COP + co, v0, v1);
                }
            }
        }
        return;
    }
    for (int t = 0; t < nt; ++t) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int a = ty + 8 * k;                // slow index of the output row, tx = fast index
            if (wf) {                                // wf[t][co][ci]: co = a, ci = tx
                const int co = co0 + a, ci = ci0 + tx;
                const float v = co_rows ? tile[a * pitch + tx * nt + t] : tile[tx * pitch + a * nt + t];
                if (co < COP && ci < CI) Elem<T>::st(wf + ((size_t)t * COP + co) * CI + ci, v);
            }
            if (wd) {                                // wd[t][ci][co]: ci = a, co = tx
                const int ci = ci0 + a, co = co0 + tx;
                const float v = co_rows ? tile[tx * pitch + a * nt + t] : tile[a * pitch + tx * nt + t];
                if (co < COP && ci < CI) Elem<T>::st(wd + ((size_t)t * CI + ci) * COP + co, v);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void weight_prep_batch_kernel(const cu_prep_item* __restrict__ items, int n) {
    __shared__ float tile[32 * (32 * PREP_MAXT + 1)];
    const cu_prep_item* it = find_item(items, n, blockIdx.x);
    const int b = blockIdx.x - it->blk0;
    switch (it->T) {                                  // uniform per workgroup
        case 9: prep_tile<T, 9>(it, b, tile); break;
        case 4: prep_tile<T, 4>(it, b, tile); break;
        case 1: prep_tile<T, 1>(it, b, tile); break;
        default: prep_tile<T, 0>(it, b, tile); break;
    }
}

// ---------------------------------------------------------------------------------------- Adam
// torch.optim.Adam (no amsgrad, weight_decay folded into the gradient) -- reference vital/vital/system.py:82-115 with
// vital/vital/config/task/optim/adam.yaml:1-4.  One 16-byte-per-lane streaming pass over the flat parameter buffer.
__global__ __launch_bounds__(256) void adam_kernel(size_t n, float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, float lr, float b1,
                                                   float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                   float gscale, const int* __restrict__ step_dev) {
    const size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (step_dev) {          // capturable form: the step count lives on the device (a replayed hipGraph has no host side)
        const float t = (float)(*step_dev + 1);
        bc1 = 1.f - powf(b1, t);
        bc2_sqrt = sqrtf(1.f - powf(b2, t));
    }
    const float step = lr / bc1;
    if (i4 + 4 <= n) {
        f32x4 pv = *reinterpret_cast<f32x4*>(p + i4);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i4);
        f32x4 mv = *reinterpret_cast<f32x4*>(m + i4), vv = *reinterpret_cast<f32x4*>(v + i4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = gv[e] * gscale + wd * pv[e];
            mv[e] = b1 * mv[e] + (1.f - b1) * gg;
            vv[e] = b2 * vv[e] + (1.f - b2) * gg * gg;
            pv[e] -= step * mv[e] / (sqrtf(vv[e]) / bc2_sqrt + eps);
        }
        *reinterpret_cast<f32x4*>(p + i4) = pv;
        *reinterpret_cast<f32x4*>(m + i4) = mv;
        *reinterpret_cast<f32x4*>(v + i4) = vv;
    } else {
        for (size_t i = i4; i < n; ++i) {
            const float gg = g[i] * gscale + wd * p[i];
            m[i] = b1 * m[i] + (1.f - b1) * gg;
            v[i] = b2 * v[i] + (1.f - b2) * gg * gg;
            p[i] -= step * m[i] / (sqrtf(v[i]) / bc2_sqrt + eps);
        }
    }
}

// ---------------------------------------------------------------------------------------- MaxPool2d(2, 2)
// nn.MaxPool2d(kernel_size=2, stride=2) of the `vital` U-Net (reference vital/vital/models/segmentation/unet.py:137-139)
// on NHWC tensors: one thread = 4 consecutive channels of one output pixel.  idx[n][oy][ox][c] in {0,1,2,3} = window
// position (dy * 2 + dx) of the maximum, the FIRST one in row-major window order on ties like ATen; NaN propagates.
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                           unsigned char* __restrict__ idx, int N, int OH, int OW, int C) {
    const size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const size_t total = (size_t)N * OH * OW * C;
    if (i4 >= total) return;
    const int c = (int)(i4 % C);
    const size_t pix = i4 / C;
    const int ox = (int)(pix % OW);
    const size_t t = pix / OW;
    const int oy = (int)(t % OH), n = (int)(t / OH);
    const int W = 2 * OW;
    const T* base = x + (((size_t)n * 2 * OH + 2 * oy) * W + 2 * ox) * C + c;
    float best[4];
    unsigned char bi[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 4; ++e) best[e] = Elem<T>::ld(base + e);
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        const T* q = base + ((size_t)(k >> 1) * W + (k & 1)) * C;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = Elem<T>::ld(q + e);
            if (v > best[e] || v != v) { best[e] = v; bi[e] = (unsigned char)k; }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        Elem<T>::st(y + i4 + e, best[e]);
        idx[i4 + e] = bi[e];
    }
}

// dx[window position idx] = dy, the other three positions 0 (every input pixel belongs to exactly one window)
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                           T* __restrict__ dx, int N, int OH, int OW, int C) {
    const size_t i4 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const size_t total = (size_t)N * OH * OW * C;
    if (i4 >= total) return;
    const int c = (int)(i4 % C);
    const size_t pix = i4 / C;
    const int ox = (int)(pix % OW);
    const size_t t = pix / OW;
    const int oy = (int)(t % OH), n = (int)(t / OH);
    const int W = 2 * OW;
    T* base = dx + (((size_t)n * 2 * OH + 2 * oy) * W + 2 * ox) * C + c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        T* q = base + ((size_t)(k >> 1) * W + (k & 1)) * C;
#pragma unroll
        for (int e = 0; e < 4; ++e) Elem<T>::st(q + e, idx[i4 + e] == k ? Elem<T>::ld(dy + i4 + e) : 0.f);
    }
}

}  // namespace

extern "C" int cu_conv_c1_fwd(int dtype, int N, int H, int W, int CO, const float* img, const float* w, const float* bias,
                              void* dst, void* stream) {
    CU_CHECK_ARG(dtype == CU_F32 || dtype == CU_BF16, "cu_conv_c1_fwd: bad dtype");
    const int PIECE = dtype == CU_BF16 ? 8 : 4;
    CU_CHECK_ARG(N > 0 && H > 0 && W > 0 && CO > 0 && CO % PIECE == 0 && img && w && dst, "cu_conv_c1_fwd: bad argument");
    const size_t total = (size_t)N * H * ((W + C1_PX - 1) / C1_PX) * (CO / PIECE);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == CU_BF16)
        hipLaunchKernelGGL((conv_c1_fwd_kernel<bf16_t, 0>), dim3(blocks), dim3(256), 0, st, img, w, bias, (bf16_t*)dst, N, H,
                           W, CO, (const float*)nullptr, 0.f, (bf16_t*)nullptr);
    else
        hipLaunchKernelGGL((conv_c1_fwd_kernel<float, 0>), dim3(blocks), dim3(256), 0, st, img, w, bias, (float*)dst, N, H, W,
                           CO, (const float*)nullptr, 0.f, (float*)nullptr);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t cu_conv_c1_norm_ws_floats(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    const size_t nwg = ((size_t)H * ((W + C1M_PX - 1) / C1M_PX) + 255) / 256;
    return (size_t)N * nwg * C1M_N;
}

extern "C" int cu_conv_c1_fwd_norm(int dtype, int N, int H, int W, int CO, const float* img, const float* w,
                                   const float* bias, const float* gamma, const float* beta, float eps, float slope,
                                   float* ws, float* stats, void* z, void* a, void* stream) {
    CU_CHECK_ARG(dtype == CU_F32 || dtype == CU_BF16, "cu_conv_c1_fwd_norm: bad dtype");
    const int PIECE = dtype == CU_BF16 ? 8 : 4;
    CU_CHECK_ARG(N > 0 && N < 65536 && H > 0 && W > 0 && CO > 0 && CO % PIECE == 0 && img && w && ws && stats && a,
                 "cu_conv_c1_fwd_norm: bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned nwg = (unsigned)(((size_t)H * ((W + C1M_PX - 1) / C1M_PX) + 255) / 256);
    hipLaunchKernelGGL(c1_moments_kernel, dim3(nwg, N), dim3(256), 0, st, img, ws, H, W);
    CU_LAUNCH_CHECK();
    hipLaunchKernelGGL(c1_stats_kernel, dim3(N), dim3(64), 0, st, (const float*)ws, (int)nwg, w, bias, gamma, beta, eps, stats,
                       N, H * W, CO);
    CU_LAUNCH_CHECK();
    const size_t total = (size_t)N * H * ((W + C1_PX - 1) / C1_PX) * (CO / PIECE);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (!z) {          // activation only: cu_conv_c1_bwd recomputes z
        if (dtype == CU_BF16)
            hipLaunchKernelGGL((conv_c1_fwd_kernel<bf16_t, 3>), dim3(blocks), dim3(256), 0, st, img, w, bias, (bf16_t*)nullptr, N, H,
                               W, CO, (const float*)stats, slope, (bf16_t*)a);
        else
            hipLaunchKernelGGL((conv_c1_fwd_kernel<float, 3>), dim3(blocks), dim3(256), 0, st, img, w, bias, (float*)nullptr, N, H, W,
                               CO, (const float*)stats, slope, (float*)a);
    } else if (dtype == CU_BF16)
        hipLaunchKernelGGL((conv_c1_fwd_kernel<bf16_t, 2>), dim3(blocks), dim3(256), 0, st, img, w, bias, (bf16_t*)z, N, H, W,
                           CO, (const float*)stats, slope, (bf16_t*)a);
    else
        hipLaunchKernelGGL((conv_c1_fwd_kernel<float, 2>), dim3(blocks), dim3(256), 0, st, img, w, bias, (float*)z, N, H, W, CO,
                           (const float*)stats, slope, (float*)a);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_conv_c1_bwd(int dtype, int N, int H, int W, int CO, const float* img, const float* w, const float* bias,
                              const float* stats, const float* gamma, float slope, const void* g, float* sums, float* dw,
                              float* dgamma, float* dbeta, void* stream) {
    CU_CHECK_ARG(dtype == CU_F32 || dtype == CU_BF16, "cu_conv_c1_bwd: bad dtype");
    const int PIECE = dtype == CU_BF16 ? 8 : 4;
    CU_CHECK_ARG(N > 0 && N < 65536 && H > 0 && W > 0 && CO > 0 && CO % PIECE == 0 && img && w && stats && g && sums && dw,
                 "cu_conv_c1_bwd: bad argument");
    const int ppp = CO / PIECE;
    CU_CHECK_ARG((ppp & (ppp - 1)) == 0 && ppp <= 16, "cu_conv_c1_bwd: CO / piece must be a power of two <= 16 (CO=%d)", CO);
    const int want = cdiv(1024, N);
    int R = cdiv(H, want < H ? want : H);
    while (R > 1 && (size_t)(R + 2) * (W + 2) * 4 > 48 * 1024) --R;
    const size_t img_b = sizeof(float) * (size_t)(R + 2) * (W + 2), red_b = sizeof(float) * (size_t)64 * 9 * PIECE;
    const size_t lds = img_b > red_b ? img_b : red_b;
    CU_CHECK_ARG(lds <= 64 * 1024, "cu_conv_c1_bwd: image rows of %d pixels do not fit the LDS", W);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(cdiv(H, R), N);
    if (dtype == CU_BF16) {
        hipLaunchKernelGGL((c1_bwd_kernel<bf16_t, 1>), grid, dim3(256), lds, st, img, (const bf16_t*)g, w, bias, stats, gamma, slope,
                           sums, dw, dgamma, dbeta, N, H, W, CO, R);
        hipLaunchKernelGGL((c1_bwd_kernel<bf16_t, 2>), grid, dim3(256), lds, st, img, (const bf16_t*)g, w, bias, stats, gamma, slope,
                           sums, dw, dgamma, dbeta, N, H, W, CO, R);
    } else {
        hipLaunchKernelGGL((c1_bwd_kernel<float, 1>), grid, dim3(256), lds, st, img, (const float*)g, w, bias, stats, gamma, slope,
                           sums, dw, dgamma, dbeta, N, H, W, CO, R);
        hipLaunchKernelGGL((c1_bwd_kernel<float, 2>), grid, dim3(256), lds, st, img, (const float*)g, w, bias, stats, gamma, slope,
                           sums, dw, dgamma, dbeta, N, H, W, CO, R);
    }
    CU_LAUNCH_CHECK();
    return 0;
}

static int conv_c1_wgrad_impl(int dtype, int N, int H, int W, int CO, const float* img, const void* dz, float* dw,
                              float* part, size_t part_floats, void* stream);

extern "C" int cu_conv_c1_wgrad(int dtype, int N, int H, int W, int CO, const float* img, const void* dz, float* dw,
                                void* stream) {
    return conv_c1_wgrad_impl(dtype, N, H, W, CO, img, dz, dw, nullptr, 0, stream);
}

extern "C" int cu_conv_c1_wgrad_det(int dtype, int N, int H, int W, int CO, const float* img, const void* dz, float* dw,
                                    float* ws, size_t ws_floats, void* stream) {
    CU_CHECK_ARG(ws != nullptr, "cu_conv_c1_wgrad_det: null workspace");
    return conv_c1_wgrad_impl(dtype, N, H, W, CO, img, dz, dw, ws, ws_floats, stream);
}

static int conv_c1_wgrad_impl(int dtype, int N, int H, int W, int CO, const float* img, const void* dz, float* dw,
                              float* part, size_t part_floats, void* stream) {
    CU_CHECK_ARG(dtype == CU_F32 || dtype == CU_BF16, "cu_conv_c1_wgrad: bad dtype");
    const int PIECE = dtype == CU_BF16 ? 8 : 4;
    CU_CHECK_ARG(N > 0 && H > 0 && W > 0 && CO > 0 && CO % PIECE == 0 && CO / PIECE <= 256 && img && dz && dw,
                 "cu_conv_c1_wgrad: bad argument");
    const int ppp = CO / PIECE, rows = 256 / ppp;
    int want = cdiv(1024, N);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if ((ppp & (ppp - 1)) == 0 && ppp <= 16 && !cu_env_set("CU_C1_WGRAD_OLD")) {
        int R = cdiv(H, want < H ? want : H);
        while (R > 1 && (size_t)(R + 2) * (W + 2) * 4 > 48 * 1024) --R;
        const size_t img_b = sizeof(float) * (size_t)(R + 2) * (W + 2), red_b = sizeof(float) * (size_t)64 * 9 * PIECE;
        const size_t lds_r = img_b > red_b ? img_b : red_b;
        if (lds_r <= 64 * 1024) {
            dim3 grid_r(cdiv(H, R), N);
            const size_t nwg = (size_t)grid_r.x * grid_r.y;
            CU_CHECK_ARG(!part || nwg * 9 * CO <= part_floats, "cu_conv_c1_wgrad_det: workspace of %zu floats, need %zu",
                         part_floats, nwg * 9 * CO);
            if (dtype == CU_BF16)
                hipLaunchKernelGGL(conv_c1_wgrad_rows_kernel<bf16_t>, grid_r, dim3(256), lds_r, st, img, (const bf16_t*)dz, dw, N, H, W, CO, R, part);
            else
                hipLaunchKernelGGL(conv_c1_wgrad_rows_kernel<float>, grid_r, dim3(256), lds_r, st, img, (const float*)dz, dw, N, H, W, CO, R, part);
            if (part)
                hipLaunchKernelGGL(conv_c1_wgrad_finish_kernel, dim3(cdiv(9 * CO, 256)), dim3(256), 0, st, part, (int)nwg, 9 * CO, dw);
            CU_LAUNCH_CHECK();
            return 0;
        }
    }
    CU_CHECK_ARG(!part, "cu_conv_c1_wgrad_det: CO / piece must be a power of two <= 16 (CO=%d)", CO);
    int chunk = cdiv(cdiv(H * W, want), rows) * rows;
    if (chunk < rows) chunk = rows;
    dim3 grid(cdiv(H * W, chunk), N);
    const size_t lds = sizeof(float) * (size_t)rows * ppp * 9 * PIECE;
    if (dtype == CU_BF16)
        hipLaunchKernelGGL(conv_c1_wgrad_kernel<bf16_t>, grid, dim3(256), lds, st, img, (const bf16_t*)dz, dw, N, H, W, CO, chunk);
    else
        hipLaunchKernelGGL(conv_c1_wgrad_kernel<float>, grid, dim3(256), lds, st, img, (const float*)dz, dw, N, H, W, CO, chunk);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_weight_prep(int dtype, int T, int CO, int CI, int COP, long s_co, long s_ci, const float* master,
                              void* w_fwd, void* w_dgrad, void* stream) {
    CU_CHECK_ARG(dtype == CU_F32 || dtype == CU_BF16, "cu_weight_prep: bad dtype");
    CU_CHECK_ARG(T > 0 && CO > 0 && CI > 0 && COP >= CO && master && (w_fwd || w_dgrad), "cu_weight_prep: bad argument");
    dim3 grid(cdiv(CI, 32), cdiv(COP, 32), T);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == CU_BF16)
        hipLaunchKernelGGL(weight_prep_kernel<bf16_t>, grid, dim3(256), 0, st, master, (bf16_t*)w_fwd, (bf16_t*)w_dgrad, CO,
                           CI, COP, s_co, s_ci);
    else
        hipLaunchKernelGGL(weight_prep_kernel<float>, grid, dim3(256), 0, st, master, (float*)w_fwd, (float*)w_dgrad, CO, CI,
                           COP, s_co, s_ci);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_grad_unprep(int T, int CO, int CI, int COP, long s_co, long s_ci, float* dwk, float* grad,
                              int accumulate, void* stream) {
    CU_CHECK_ARG(T > 0 && CO > 0 && CI > 0 && COP >= CO && dwk && grad, "cu_grad_unprep: bad argument");
    const bool co_rows = s_co > s_ci;
    if (T <= UNPREP_MAXT && ((co_rows && s_ci == T) || (!co_rows && s_co == T))) {
        dim3 grid(cdiv(co_rows ? CI : CO, 32), cdiv(co_rows ? CO : CI, 8));
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        if (T == 9) hipLaunchKernelGGL(grad_unprep_rows_kernel<9>, grid, dim3(256), 0, st, dwk, grad, T, CO, CI, COP, s_co, s_ci, accumulate);
        else if (T == 4) hipLaunchKernelGGL(grad_unprep_rows_kernel<4>, grid, dim3(256), 0, st, dwk, grad, T, CO, CI, COP, s_co, s_ci, accumulate);
        else if (T == 1) hipLaunchKernelGGL(grad_unprep_rows_kernel<1>, grid, dim3(256), 0, st, dwk, grad, T, CO, CI, COP, s_co, s_ci, accumulate);
        else hipLaunchKernelGGL(grad_unprep_rows_kernel<0>, grid, dim3(256), 0, st, dwk, grad, T, CO, CI, COP, s_co, s_ci, accumulate);
        CU_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid(cdiv(CI, 32), cdiv(CO, 32), T);
    hipLaunchKernelGGL(grad_unprep_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), dwk, grad, CO, CI,
                       COP, s_co, s_ci, accumulate);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_grad_unprep_parts(int T, int CO, int CI, int COP, long s_co, long s_ci, float* parts, size_t parts_floats,
                                    int nparts, int layout, float* grad, int accumulate, void* stream) {
    CU_CHECK_ARG(T > 0 && CO > 0 && CI > 0 && COP >= CO && parts && grad && nparts >= 1, "cu_grad_unprep_parts: bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (layout == CU_PARTS_PLAIN) {      // slabs already in the plain [T][COP][CI] layout (gemm_tn.hip): element-wise sums
        const size_t E = (size_t)T * COP * CI;
        CU_CHECK_ARG(E % 4 == 0 && parts_floats >= (size_t)nparts * E, "cu_grad_unprep_parts: workspace of %zu floats < %d slabs of %zu", parts_floats, nparts, E);
        const unsigned gx = (unsigned)((E / 4 + 255) / 256);
        const bool rows_ok = (s_co > s_ci) ? s_ci == T : s_co == T;        // taps innermost in the logical layout
        if (rows_ok && (T == 9 || T == 4 || T == 1)) {
            // first level (many slabs of a small tile: sums of G consecutive slabs by the whole chip), then ONE launch that adds
            // the group sums in order and writes the logical rows (before: strided sum + un-preparation, two launches)
            size_t stride = E;
            if (nparts > 16) {
                const int G = cdiv(nparts, 16), groups = cdiv(nparts, G);
                hipLaunchKernelGGL(parts_reduce_kernel, dim3(gx, groups), dim3(256), 0, st, parts, E / 4, nparts, G);
                CU_LAUNCH_CHECK();
                nparts = groups;
                stride = (size_t)G * E;
            }
            launch_sum_unprep<false>(T, parts, stride, nparts, grad, CO, CI, COP, s_co, s_ci, 1, 1, 1, accumulate & 1, st);
            CU_LAUNCH_CHECK();
            return 0;
        }
        if (nparts > 16) {
            const int G = cdiv(nparts, 16), groups = cdiv(nparts, G);
            hipLaunchKernelGGL(parts_reduce_kernel, dim3(gx, groups), dim3(256), 0, st, parts, E / 4, nparts, G);
            CU_LAUNCH_CHECK();
            hipLaunchKernelGGL(parts_reduce_strided_kernel, dim3(gx), dim3(256), 0, st, parts, E / 4, (size_t)G * (E / 4), groups);
            CU_LAUNCH_CHECK();
        } else if (nparts > 1) {
            hipLaunchKernelGGL(parts_reduce_strided_kernel, dim3(gx), dim3(256), 0, st, parts, E / 4, E / 4, nparts);
            CU_LAUNCH_CHECK();
        }
        return cu_grad_unprep(T, CO, CI, COP, s_co, s_ci, parts, grad, accumulate & 1, stream);
    }
    const int NBLK = layout >> 8, CBLK = layout & 255;
    CU_CHECK_ARG(NBLK >= 1 && NBLK <= 4 && CBLK >= 1 && CBLK <= 2, "cu_grad_unprep_parts: layout %d is not one cu_conv_wgrad_parts returns", layout);
    const int ctiles = cdiv(CI, 32 * CBLK), ntn = cdiv(COP, 32 * NBLK);
    const size_t E = (size_t)ntn * ctiles * NBLK * CBLK * T * 1024, plain = (size_t)T * COP * CI;
    CU_CHECK_ARG(parts_floats >= (size_t)nparts * E + plain, "cu_grad_unprep_parts: workspace of %zu floats < %d slabs of %zu + %zu",
                 parts_floats, nparts, E, plain);
    size_t stride4 = E / 4;
    const unsigned gx = (unsigned)((E / 4 + 255) / 256);
    if (nparts > 16) {          // first level: <= 16 group sums, by the whole chip
        const int G = cdiv(nparts, 16), groups = cdiv(nparts, G);
        hipLaunchKernelGGL(parts_reduce_kernel, dim3(gx, groups), dim3(256), 0, st, parts, E / 4, nparts, G);
        CU_LAUNCH_CHECK();
        nparts = groups;
        stride4 = (size_t)G * (E / 4);
    }
    if (s_co > s_ci && s_ci == T && (T == 9 || T == 4 || T == 1)) {
        // conv layouts (rows = co, taps innermost): slab sums + un-preparation in ONE launch straight from the native slabs
        launch_sum_unprep<true>(T, parts, stride4 * 4, nparts, grad, CO, CI, COP, s_co, s_ci, NBLK, CBLK, ctiles, accumulate & 1, st);
        CU_LAUNCH_CHECK();
        return 0;
    }
    float* sum = parts + parts_floats - plain;         // the plain tile lives at the END of the workspace
    hipLaunchKernelGGL(parts_finish_kernel, dim3(gx), dim3(256), 0, st, (const float*)parts, E / 4, stride4, nparts, sum, T, COP,
                       CI, NBLK, CBLK, ctiles);
    CU_LAUNCH_CHECK();
    return cu_grad_unprep(T, CO, CI, COP, s_co, s_ci, sum, grad, accumulate & 1, stream);
}

extern "C" int cu_adam_step(size_t n, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                            float eps, float weight_decay, int step, float grad_scale, void* stream) {
    CU_CHECK_ARG(n > 0 && p && g && m && v && step >= 1, "cu_adam_step: bad argument");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    const size_t blocks = (n + 1023) / 1024;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n, p, g, m,
                       v, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale, (const int*)nullptr);
    CU_LAUNCH_CHECK();
    return 0;
}

__global__ void step_advance_kernel(int* step) { *step += 1; }

extern "C" int cu_adam_step_dev(size_t n, float* p, const float* g, float* m, float* v, float lr, float beta1,
                                float beta2, float eps, float weight_decay, const int* steps_done, float grad_scale,
                                void* stream) {
    CU_CHECK_ARG(n > 0 && p && g && m && v && steps_done, "cu_adam_step_dev: bad argument");
    const size_t blocks = (n + 1023) / 1024;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), n, p, g, m,
                       v, lr, beta1, beta2, eps, weight_decay, 1.f, 1.f, grad_scale, steps_done);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_step_advance(int* steps_done, void* stream) {
    CU_CHECK_ARG(steps_done != nullptr, "cu_step_advance: null pointer");
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), steps_done);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_weight_prep_batch(int dtype, int n_items, const cu_prep_item* items, int total_blocks, void* stream) {
    CU_CHECK_ARG(dtype == CU_F32 || dtype == CU_BF16, "cu_weight_prep_batch: bad dtype");
    CU_CHECK_ARG(n_items > 0 && items && total_blocks > 0, "cu_weight_prep_batch: bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == CU_BF16)
        hipLaunchKernelGGL(weight_prep_batch_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, st, items, n_items);
    else
        hipLaunchKernelGGL(weight_prep_batch_kernel<float>, dim3(total_blocks), dim3(256), 0, st, items, n_items);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_maxpool2_fwd(int dtype, int N, int OH, int OW, int C, const void* x, void* y, unsigned char* idx, void* stream) {
    CU_CHECK_ARG(dtype == CU_F32 || dtype == CU_BF16, "cu_maxpool2_fwd: bad dtype");
    CU_CHECK_ARG(N > 0 && OH > 0 && OW > 0 && C > 0 && C % 4 == 0 && x && y && idx, "cu_maxpool2_fwd: bad argument (C % 4 == 0)");
    const size_t n4 = (size_t)N * OH * OW * C / 4;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned blocks = (unsigned)((n4 + 255) / 256);
    if (dtype == CU_BF16)
        hipLaunchKernelGGL(maxpool2_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, idx, N, OH, OW, C);
    else
        hipLaunchKernelGGL(maxpool2_fwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, (float*)y, idx, N, OH, OW, C);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_maxpool2_bwd(int dtype, int N, int OH, int OW, int C, const void* dy, const unsigned char* idx, void* dx,
                               void* stream) {
    CU_CHECK_ARG(dtype == CU_F32 || dtype == CU_BF16, "cu_maxpool2_bwd: bad dtype");
    CU_CHECK_ARG(N > 0 && OH > 0 && OW > 0 && C > 0 && C % 4 == 0 && dy && dx && idx, "cu_maxpool2_bwd: bad argument (C % 4 == 0)");
    const size_t n4 = (size_t)N * OH * OW * C / 4;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const unsigned blocks = (unsigned)((n4 + 255) / 256);
    if (dtype == CU_BF16)
        hipLaunchKernelGGL(maxpool2_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)dy, idx, (bf16_t*)dx, N, OH, OW, C);
    else
        hipLaunchKernelGGL(maxpool2_bwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)dy, idx, (float*)dx, N, OH, OW, C);
    CU_LAUNCH_CHECK();
    return 0;
}
