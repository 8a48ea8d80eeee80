// DSNT head and NLL losses (gfx950).  See include/contour_hip.h.
//
// cu_dsnt_head_fwd : flat_softmax + dsnt + pixel rescale + get_cov_matrix of the reference
//                    (task/regression/dsnt/utils.py:7-47,71-77,95-105; dsnt_al.py:52-60; aleatoric.py:138-144).
//   One workgroup per heat map.  Pass 1: max and arg-max (wave shuffles + LDS).  Pass 2: softmax-weighted moments taken
//   about the arg-max cell so that the centred second moments do not cancel (the reference centres on the mean with 8
//   full-size temporaries; here the map is read twice, 16 bytes per lane, and nothing is written but 13 floats).
// cu_dsnt_head_bwd : analytic gradient, one streaming pass:  dlogit_i = p_i * (q_i - sum_j p_j q_j) with
//   q_i = gx X_i + gy Y_i + gxx (X_i-xbar)^2 + gyy (Y_i-ybar)^2 + gxy (X_i-xbar)(Y_i-ybar); written NHWC for the MFMA
//   gradient kernels of the 1x1 output conv.
// cu_nll_fwd_bwd   : Gaussian NLL (dsnt_al.py:64-74) / skew-normal NLL (distributions/bivariateskewnormal.py:36-61)
//   with closed-form 2x2 algebra and analytic gradients; one workgroup, deterministic reductions.
#include "common.h"

namespace {

constexpr int HT = 256;

__device__ __forceinline__ float block_sum(float v, float* lds) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    float r = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += lds[w];
    return r;
}

__device__ __forceinline__ double block_sum_f64(double v, double* lds) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double r = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += lds[w];
    return r;
}

// One workgroup per heat map, ONE streaming pass (round 1 read every map twice: maximum, then moments about the arg-max,
// 704 MB per step at batch 64).  Every thread keeps a running maximum m and the raw moments of w = exp(v - m) in f64,
// rescaled when its maximum grows (online softmax); at the end the threads' moments are brought to the common maximum
// and summed.  f64 for the accumulation and the final E[x^2] - E[x]^2 is what replaces the centring: the variance of a
// sharp off-centre peak (sigma ~ 1e-2 in normalised units) loses 2500x in that subtraction, 1e-16 * 2500 is nothing.
// The f32 rounding of expf and of the rescaling factors only re-weights pixels by 1e-7 (no cancellation involved).
__global__ __launch_bounds__(HT) void dsnt_fwd_kernel(const float* __restrict__ logits, int H, int W, int use_covar,
                                                      float* __restrict__ mu, float* __restrict__ sigma,
                                                      float* __restrict__ aux) {
    __shared__ float lds[16];
    __shared__ double ldd[16];
    const int map = blockIdx.x;
    const int HWn = H * W;
    const float* lp = logits + (size_t)map * HWn;
    const int tid = threadIdx.x;
    const int nvec = HWn >> 2;
    const float invW = 1.f / (float)W;       // normalised coordinates (utils.py:50-68): lin[j] = (2j + 1)/W - 1

    float m = -INFINITY;
    double s0 = 0.0, sx = 0.0, sy = 0.0, sxx = 0.0, syy = 0.0, sxy = 0.0;
    for (int i = tid; i < nvec; i += HT) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lp + 4 * i);
        const float lm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        if (lm > m) {
            const double sc = m == -INFINITY ? 0.0 : (double)expf(m - lm);
            s0 *= sc; sx *= sc; sy *= sc; sxx *= sc; syy *= sc; sxy *= sc;
            m = lm;
        }
        const int p = 4 * i;
        const int yy = p / W, xx = p - yy * W;      // W % 4 == 0: the 4 elements share a row
        const double Y = (double)((2.f * yy + 1.f) * invW - 1.f);
        double r0 = 0.0, rx = 0.0, rxx = 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double w = (double)expf(v[e] - m);
            const double X = (double)((2.f * (xx + e) + 1.f) * invW - 1.f);
            r0 += w; rx += w * X; rxx += w * X * X;
        }
        s0 += r0; sx += rx; sxx += rxx;
        sy += r0 * Y; syy += r0 * Y * Y; sxy += rx * Y;
    }
    // ---- common maximum, then the sums
    float mx = m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) lds[tid >> 6] = mx;
    __syncthreads();
    mx = lds[0];
    for (int w = 1; w < HT / 64; ++w) mx = fmaxf(mx, lds[w]);
    const double sc = m == -INFINITY ? 0.0 : (double)expf(m - mx);
    s0 = block_sum_f64(s0 * sc, ldd); sx = block_sum_f64(sx * sc, ldd); sy = block_sum_f64(sy * sc, ldd);
    sxx = block_sum_f64(sxx * sc, ldd); syy = block_sum_f64(syy * sc, ldd); sxy = block_sum_f64(sxy * sc, ldd);
    if (tid == 0) {
        const double inv = 1.0 / s0;
        const double xbar = sx * inv, ybar = sy * inv;
        const double vx = sxx * inv - xbar * xbar, vy = syy * inv - ybar * ybar;
        const double cv = sxy * inv - xbar * ybar;
        const double half = 0.5 * (double)W;                    // image_size / 2
        // normalized_to_pixel_coordinates: 0.5*((c + 1)*size - 1)
        mu[2 * map] = (float)(0.5 * ((xbar + 1.0) * (double)W - 1.0));
        mu[2 * map + 1] = (float)(0.5 * ((ybar + 1.0) * (double)H - 1.0));
        sigma[3 * map] = (float)(vx * half * half);
        sigma[3 * map + 1] = (float)(vy * half * half);
        sigma[3 * map + 2] = use_covar ? (float)(cv * half * half) : 0.f;
        float* a = aux + 8 * (size_t)map;
        a[0] = mx; a[1] = (float)inv; a[2] = (float)xbar; a[3] = (float)ybar; a[4] = (float)vx; a[5] = (float)vy;
        a[6] = (float)cv; a[7] = 0.f;
    }
}

// One workgroup per heat map, one streaming pass: dlogit_i = p_i * (q_i - sum_j p_j q_j).
__global__ __launch_bounds__(HT) void dsnt_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ aux,
                                                      const float* __restrict__ gmu, const float* __restrict__ gsigma,
                                                      int use_covar, float* __restrict__ dl, int H, int W) {
    const int map = blockIdx.x;
    const int HWn = H * W;
    const float* a = aux + 8 * (size_t)map;
    const float half = 0.5f * (float)W;
    // chain rule from pixel units back to normalised units
    const float gx = gmu[2 * map] * 0.5f * (float)W, gy = gmu[2 * map + 1] * 0.5f * (float)H;
    const float gxx = gsigma[3 * map] * half * half, gyy = gsigma[3 * map + 1] * half * half;
    const float gxy = use_covar ? gsigma[3 * map + 2] * half * half : 0.f;
    const float mx = a[0], inv = a[1], xbar = a[2], ybar = a[3];
    const float pq = gx * xbar + gy * ybar + gxx * a[4] + gyy * a[5] + gxy * a[6];   // sum_j p_j q_j in closed form
    const float invW = 1.f / (float)W;
    const float* lp = logits + (size_t)map * HWn;
    float* op = dl + (size_t)map * HWn;
    const int nvec = HWn >> 2;
    for (int i = blockIdx.y * HT + threadIdx.x; i < nvec; i += HT * gridDim.y) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(lp + 4 * i);
        const int p = 4 * i;
        const int yy = p / W, xx = p - yy * W;
        const float Y = (2.f * yy + 1.f) * invW - 1.f, dy = Y - ybar;
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float X = (2.f * (xx + e) + 1.f) * invW - 1.f, dx = X - xbar;
            const float pr = expf(v[e] - mx) * inv;
            const float q = gx * X + gy * Y + gxx * dx * dx + gyy * dy * dy + gxy * dx * dy;
            o[e] = pr * (q - pq);
        }
        *reinterpret_cast<f32x4*>(op + 4 * i) = o;
    }
}

// The same gradient written where its consumers read it: NHWC with CP (= 32) channels per pixel in the engine's element
// type -- the Z operand of the 1x1 output convolution's weight / input gradients -- instead of NCHW f32 plus a layout
// pass (saves one f32 write and one f32 read of N*K*H*W).  One thread per pixel walks the K maps (loads coalesced per
// map across the wave), one 64-byte (bf16) row store per pixel; maps K..CP-1 are zero.
template <typename T>
__global__ __launch_bounds__(256) void dsnt_bwd_nhwc_kernel(const float* __restrict__ logits, const float* __restrict__ aux,
                                                            const float* __restrict__ gmu, const float* __restrict__ gsigma,
                                                            int use_covar, T* __restrict__ dl, int K, int H, int W) {
    constexpr int CP = 32;
    __shared__ float sc[CP][12];
    const int n = blockIdx.y, HWn = H * W;
    const float half = 0.5f * (float)W;
    if ((int)threadIdx.x < K) {
        const int map = n * K + threadIdx.x;
        const float* a = aux + 8 * (size_t)map;
        const float gx = gmu[2 * map] * 0.5f * (float)W, gy = gmu[2 * map + 1] * 0.5f * (float)H;
        const float gxx = gsigma[3 * map] * half * half, gyy = gsigma[3 * map + 1] * half * half;
        const float gxy = use_covar ? gsigma[3 * map + 2] * half * half : 0.f;
        float* o = sc[threadIdx.x];
        o[0] = gx; o[1] = gy; o[2] = gxx; o[3] = gyy; o[4] = gxy;
        o[5] = a[0]; o[6] = a[1]; o[7] = a[2]; o[8] = a[3];
        o[9] = gx * a[2] + gy * a[3] + gxx * a[4] + gyy * a[5] + gxy * a[6];        // sum_j p_j q_j in closed form
    }
    __syncthreads();
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HWn) return;
    const int yy = p / W, xx = p - yy * W;
    const float invW = 1.f / (float)W;
    const float X = (2.f * xx + 1.f) * invW - 1.f, Y = (2.f * yy + 1.f) * invW - 1.f;
    const float* lp = logits + (size_t)n * K * HWn + p;
    float v[CP];
#pragma unroll
    for (int k = 0; k < CP; ++k) v[k] = k < K ? lp[(size_t)k * HWn] : 0.f;
#pragma unroll
    for (int k = 0; k < CP; ++k) {
        if (k < K) {
            const float* o = sc[k];
            const float dx = X - o[7], dy = Y - o[8];
            const float pr = expf(v[k] - o[5]) * o[6];
            const float q = o[0] * X + o[1] * Y + o[2] * dx * dx + o[3] * dy * dy + o[4] * dx * dy;
            v[k] = pr * (q - o[9]);
        }
    }
    T* op = dl + ((size_t)n * HWn + p) * CP;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            u32x4 w;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                w[e] = (unsigned)f32_to_bf16(v[8 * g + 2 * e]) | ((unsigned)f32_to_bf16(v[8 * g + 2 * e + 1]) << 16);
            *reinterpret_cast<u32x4*>(op + 8 * g) = w;
        }
    } else {
#pragma unroll
        for (int g = 0; g < 8; ++g)
            *reinterpret_cast<f32x4*>(op + 4 * g) = f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
    }
}

// -------------------------------------------------------------------------------------------------- NLL
__global__ __launch_bounds__(HT) void nll_kernel(int M, int skew, float w_mse, float w_log, const float* __restrict__ mu,
                                                 const float* __restrict__ sigma, const float* __restrict__ y,
                                                 const float* __restrict__ alpha, float* __restrict__ logs,
                                                 float* __restrict__ gmu, float* __restrict__ gsigma,
                                                 float* __restrict__ galpha, float* __restrict__ terms) {
    __shared__ float lds[16];
    const float invM = 1.f / (float)M;
    float l_loss = 0.f, l_dist = 0.f, l_t1 = 0.f, l_t2 = 0.f, l_t3 = 0.f, l_an = 0.f;
    for (int i = threadIdx.x; i < M; i += HT) {
        const float a = sigma[3 * i], b = sigma[3 * i + 1], c = sigma[3 * i + 2];
        // d = mu - y for the quadratic form (dsnt_al.py:68-69); the skew affine uses (y - mu) (bivariateskewnormal.py:57)
        const float d1 = mu[2 * i] - y[2 * i], d2 = mu[2 * i + 1] - y[2 * i + 1];
        const float det = a * b - c * c;
        const float idet = 1.f / det;
        const float logdet = logf(det);
        const float quad = (b * d1 * d1 - 2.f * c * d1 * d2 + a * d2 * d2) * idet;
        // d quad / d(a,b,c):  quad = num/det
        const float num = quad * det;
        const float dq_da = (d2 * d2 - num * b * idet) * idet;
        const float dq_db = (d1 * d1 - num * a * idet) * idet;
        const float dq_dc = (-2.f * d1 * d2 + num * 2.f * c * idet) * idet;
        const float dq_d1 = 2.f * (b * d1 - c * d2) * idet, dq_d2 = 2.f * (a * d2 - c * d1) * idet;
        const float dl_da = b * idet, dl_db = a * idet, dl_dc = -2.f * c * idet;
        l_dist += sqrtf(d1 * d1 + d2 * d2);
        float g1, g2, ga, gb, gc;
        if (!skew) {
            const float t1 = w_log * logdet, t2 = w_mse * quad;
            l_t1 += t1; l_t2 += t2; l_loss += t1 + t2;
            if (terms) { terms[4 * i] = t1 + t2; terms[4 * i + 1] = t1; terms[4 * i + 2] = t2; terms[4 * i + 3] = 0.f; }
            g1 = w_mse * dq_d1; g2 = w_mse * dq_d2;
            ga = w_log * dl_da + w_mse * dq_da; gb = w_log * dl_db + w_mse * dq_db; gc = w_log * dl_dc + w_mse * dq_dc;
        } else {
            const float al1 = alpha[2 * i], al2 = alpha[2 * i + 1];
            const float e1 = -d1, e2 = -d2;                        // y - mu
            const float s = sqrtf(det), t = sqrtf(a + b + 2.f * s);
            // Sigma^-1/2 = 1/(s t) [[b+s, -c], [-c, a+s]]
            const float u = al1 * ((b + s) * e1 - c * e2) + al2 * ((a + s) * e2 - c * e1);
            const float ist = 1.f / (s * t);
            const float zz = u * ist;
            const float cdf = 0.5f * (1.f + erff(zz * 0.70710678118654752f));
            const float t3 = logf(cdf + 1e-7f);
            const float nll = 0.5f * logdet + 0.5f * quad - t3;
            l_t1 += logdet; l_t2 += quad; l_t3 += t3; l_loss += nll;
            if (terms) { terms[4 * i] = nll; terms[4 * i + 1] = logdet; terms[4 * i + 2] = quad; terms[4 * i + 3] = t3; }
            l_an += fabsf(al1) + fabsf(al2);                       // torch.norm(alpha_flat, dim=-1) over a size-1 dim
            // d(-t3)/dz = -phi(z)/(cdf + 1e-7)
            const float phi = 0.3989422804014327f * expf(-0.5f * zz * zz);
            const float gz = -phi / (cdf + 1e-7f);
            const float ae = al1 * e1 + al2 * e2;
            const float ds_da = 0.5f * b / s, ds_db = 0.5f * a / s, ds_dc = -c / s;
            const float dt_da = (1.f + 2.f * ds_da) / (2.f * t), dt_db = (1.f + 2.f * ds_db) / (2.f * t),
                        dt_dc = ds_dc / t;
            const float du_da = al2 * e2 + ae * ds_da, du_db = al1 * e1 + ae * ds_db,
                        du_dc = -al1 * e2 - al2 * e1 + ae * ds_dc;
            const float k2 = u * ist * ist;
            const float dz_da = du_da * ist - k2 * (t * ds_da + s * dt_da);
            const float dz_db = du_db * ist - k2 * (t * ds_db + s * dt_db);
            const float dz_dc = du_dc * ist - k2 * (t * ds_dc + s * dt_dc);
            // dz/d(mu) = -dz/d(e)
            const float dz_de1 = (al1 * (b + s) - al2 * c) * ist, dz_de2 = (al2 * (a + s) - al1 * c) * ist;
            g1 = 0.5f * dq_d1 - gz * dz_de1; g2 = 0.5f * dq_d2 - gz * dz_de2;
            ga = 0.5f * dl_da + 0.5f * dq_da + gz * dz_da;
            gb = 0.5f * dl_db + 0.5f * dq_db + gz * dz_db;
            gc = 0.5f * dl_dc + 0.5f * dq_dc + gz * dz_dc;
            if (galpha) {
                galpha[2 * i] = gz * ((b + s) * e1 - c * e2) * ist * invM;
                galpha[2 * i + 1] = gz * ((a + s) * e2 - c * e1) * ist * invM;
            }
        }
        if (gmu) { gmu[2 * i] = g1 * invM; gmu[2 * i + 1] = g2 * invM; }
        if (gsigma) { gsigma[3 * i] = ga * invM; gsigma[3 * i + 1] = gb * invM; gsigma[3 * i + 2] = gc * invM; }
    }
    l_loss = block_sum(l_loss, lds); l_dist = block_sum(l_dist, lds); l_t1 = block_sum(l_t1, lds);
    l_t2 = block_sum(l_t2, lds); l_t3 = block_sum(l_t3, lds); l_an = block_sum(l_an, lds);
    if (threadIdx.x == 0) {
        logs[0] = l_loss * invM; logs[1] = l_dist * invM; logs[2] = l_t1 * invM; logs[3] = l_t2 * invM;
        logs[4] = l_t3 * invM; logs[5] = l_an * invM * 0.5f; logs[6] = 0.f; logs[7] = 0.f;
    }
}

// ConfidenceNet Linear(512, 2K*) (unet2.py:28-29): tiny.  One wave per output, the lanes stride over the inputs (both rows
// are read in 256-byte lines; one thread per output walked w with a stride of IN floats: 50 us for 2688 outputs).
__global__ __launch_bounds__(256) void linear_fwd_kernel(int N, int IN, int OUT, const float* __restrict__ x,
                                                         const float* __restrict__ w, const float* __restrict__ b,
                                                         float* __restrict__ out) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= N * OUT) return;                       // wave-uniform
    const int n = i / OUT, o = i - n * OUT;
    const float* xr = x + (size_t)n * IN;
    const float* wr = w + (size_t)o * IN;
    float acc = 0.f;
    for (int k = lane; k < IN; k += 64) acc = fmaf(xr[k], wr[k], acc);
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) out[i] = acc + (b ? b[o] : 0.f);
}
__global__ void linear_bwd_x_kernel(int N, int IN, int OUT, const float* __restrict__ w, const float* __restrict__ go,
                                    float* __restrict__ gx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * IN) return;
    const int n = i / IN, k = i - n * IN;
    float acc = 0.f;
    for (int o = 0; o < OUT; ++o) acc += go[(size_t)n * OUT + o] * w[(size_t)o * IN + k];
    gx[i] = acc;
}
__global__ void linear_bwd_w_kernel(int N, int IN, int OUT, const float* __restrict__ x, const float* __restrict__ go,
                                    float* __restrict__ gw, float* __restrict__ gb) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= OUT * IN) return;
    const int o = i / IN, k = i - o * IN;
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc += go[(size_t)n * OUT + o] * x[(size_t)n * IN + k];
    gw[i] += acc;
    if (k == 0 && gb) {
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += go[(size_t)n * OUT + o];
        gb[o] += s;
    }
}

}  // namespace

extern "C" int cu_dsnt_head_fwd(int NK, int H, int W, const float* logits, int use_covar, float* mu, float* sigma,
                                float* aux, void* stream) {
    CU_CHECK_ARG(NK > 0 && H > 0 && W > 0 && H == W, "cu_dsnt_head_fwd: maps must be square (reference utils.py:9), got %dx%d",
                 H, W);
    CU_CHECK_ARG(W % 4 == 0, "cu_dsnt_head_fwd: W=%d must be a multiple of 4", W);
    CU_CHECK_ARG(logits && mu && sigma && aux, "cu_dsnt_head_fwd: null pointer");
    hipLaunchKernelGGL(dsnt_fwd_kernel, dim3(NK), dim3(HT), 0, reinterpret_cast<hipStream_t>(stream), logits, H, W,
                       use_covar, mu, sigma, aux);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_dsnt_head_bwd(int NK, int H, int W, const float* logits, const float* aux, const float* gmu,
                                const float* gsigma, int use_covar, float* dlogits, void* stream) {
    CU_CHECK_ARG(NK > 0 && H == W && W % 4 == 0, "cu_dsnt_head_bwd: bad shape %dx%d", H, W);
    CU_CHECK_ARG(logits && aux && gmu && gsigma && dlogits, "cu_dsnt_head_bwd: null pointer");
    int split = 1;                      // several workgroups per map when there are few maps
    while (NK * split < 1024 && (H * W) / (4 * HT * split) > 4) split *= 2;
    hipLaunchKernelGGL(dsnt_bwd_kernel, dim3(NK, split), dim3(HT), 0, reinterpret_cast<hipStream_t>(stream), logits, aux,
                       gmu, gsigma, use_covar, dlogits, H, W);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_dsnt_head_bwd_nhwc(int dtype, int N, int K, int H, int W, const float* logits, const float* aux,
                                     const float* gmu, const float* gsigma, int use_covar, void* dl, void* stream) {
    CU_CHECK_ARG(dtype == CU_F32 || dtype == CU_BF16, "cu_dsnt_head_bwd_nhwc: bad dtype %d", dtype);
    CU_CHECK_ARG(N > 0 && K > 0 && K <= 32 && H > 0 && W > 0 && H == W, "cu_dsnt_head_bwd_nhwc: bad shape (K <= 32, square maps)");
    CU_CHECK_ARG(logits && aux && gmu && gsigma && dl, "cu_dsnt_head_bwd_nhwc: null pointer");
    const dim3 grid(cdiv(H * W, 256), N);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == CU_BF16)
        hipLaunchKernelGGL(dsnt_bwd_nhwc_kernel<bf16_t>, grid, dim3(256), 0, st, logits, aux, gmu, gsigma, use_covar,
                           (bf16_t*)dl, K, H, W);
    else
        hipLaunchKernelGGL(dsnt_bwd_nhwc_kernel<float>, grid, dim3(256), 0, st, logits, aux, gmu, gsigma, use_covar,
                           (float*)dl, K, H, W);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_nll_fwd_bwd(int M, int skew, float w_mse, float w_log, const float* mu, const float* sigma,
                              const float* y, const float* alpha, float* logs, float* gmu, float* gsigma,
                              float* galpha, float* terms, void* stream) {
    CU_CHECK_ARG(M > 0 && mu && sigma && y && logs, "cu_nll_fwd_bwd: bad argument");
    CU_CHECK_ARG(!skew || alpha, "cu_nll_fwd_bwd: skew NLL needs alpha");
    hipLaunchKernelGGL(nll_kernel, dim3(1), dim3(HT), 0, reinterpret_cast<hipStream_t>(stream), M, skew, w_mse, w_log, mu,
                       sigma, y, alpha, logs, gmu, gsigma, galpha, terms);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_linear_fwd(int N, int IN, int OUT, const float* x, const float* w, const float* b, float* out,
                             void* stream) {
    CU_CHECK_ARG(N > 0 && IN > 0 && OUT > 0 && x && w && out, "cu_linear_fwd: bad argument");
    hipLaunchKernelGGL(linear_fwd_kernel, dim3(cdiv(N * OUT, 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), N,
                       IN, OUT, x, w, b, out);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_linear_bwd(int N, int IN, int OUT, const float* x, const float* w, const float* gout, float* gx,
                             float* gw, float* gb, void* stream) {
    CU_CHECK_ARG(N > 0 && IN > 0 && OUT > 0 && x && w && gout, "cu_linear_bwd: bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (gx) hipLaunchKernelGGL(linear_bwd_x_kernel, dim3(cdiv(N * IN, 128)), dim3(128), 0, st, N, IN, OUT, w, gout, gx);
    if (gw) hipLaunchKernelGGL(linear_bwd_w_kernel, dim3(cdiv(OUT * IN, 128)), dim3(128), 0, st, N, IN, OUT, x, gout, gw, gb);
    CU_LAUNCH_CHECK();
    return 0;
}
