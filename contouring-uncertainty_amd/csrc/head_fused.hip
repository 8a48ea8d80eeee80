// Fused head of the DSNT path (gfx950, bf16 production mode; round 3): everything between the LAST ConvLayer's raw output z
// and the landmark moments, and everything between the moments' gradients and dL/d(activation) of that layer, without the
// activation, the logits or dL/dlogits ever existing in HBM.
//
// Replaces, for the 32-channel full-resolution level (reference models/nnUnet/layers.py:192-205 InstanceNorm + LeakyReLU of the
// last ConvLayer, :441-463 OutputBlock 1x1 conv, task/regression/dsnt/utils.py:7-47,71-77,95-105 flat_softmax + dsnt + pixel
// rescale, dsnt_al.py:52-60 / aleatoric.py:138-144 covariance):
//   forward  (cu_head_fused_fwd):  apply pass (read z, write a) + 1x1 conv (read a, write f32 logits) + cu_dsnt_head_fwd
//            (read logits) = 1.8 GB per step at batch 64  ->  ONE read of z (268 MB) + 8 MB of per-tile partial moments;
//   backward (cu_head_fused_bwd):  cu_dsnt_head_bwd_nhwc (read logits, write dL/dlogits) + 1x1 input gradient (read it,
//            write g) + 1x1 weight gradient (read it and a) + the reduction pass of the last layer's InstanceNorm backward
//            (read g and z) = 2.7 GB  ->  read z, write g (536 MB); the logits are recomputed (2 MFMAs per 32 pixels).
// The kernels are HBM streams with a VALU body, no LDS-DMA and no workgroup barriers in the loop: a WAVE owns a tile of 32
// pixels x R rows and walks it row by row; the next two rows' loads are in flight while a row is processed.
//
// Data flow per row of 32 pixels (lane = pixel r + 32 h; MFMA 32x32x16 bf16, weights as A, pixels as B):
//   z row (two 16-byte loads per lane) -> two v_permlane32_swap pairs give the lane channels (i & 3) + 8 (i >> 2) + 4 h,
//   i = 0..15: exactly the rows the MFMA output layout hands a lane, so the SAME 16 channels serve the normalisation on the
//   way in and the InstanceNorm-backward sums on the way out; k-slot j of step s = channel index i = 8 s + j (the weight
//   fragments are loaded with the same permutation: any k order works if both operands agree)
//   -> a = bf16(LeakyReLU(z scale + shift)) (the apply pass's arithmetic)  -> logits = W a   (2 MFMAs; lane: 16 classes)
//   forward : online-softmax moments per (lane, class) about the TILE centre, the lane's x being constant over the tile
//             (s0, sum e dy, sum e dy^2; x moments follow from the lane's dx at the end), a reference logit per (half wave,
//             class) that only moves when some logit exceeds it by 40 (so no cross-lane maximum per row);
//   backward: dl = softmax * (q - sum p q) (cu_dsnt_head_bwd's formula, per-class constants from a wave-local LDS table)
//             -> bf16 -> g = W^T dl (accumulator tile as the next MFMA's B operand, no lane movement)
//             -> sums of gl = g LeakyReLU'(y) and gl z per channel (-> the two sums cu_instnorm_bwd_given consumes)
//             -> dW += dl^T a: both tiles go through a wave-local 4-KiB LDS image (8-byte pieces, XOR-swizzled) and come
//                back k-major with ds_read_b64_tr_b16.
#include "common.h"

namespace {

struct HfArgs {
    const bf16_t* z; const float* stats; const bf16_t* w_cls; const bf16_t* w_ch;
    int N, H, W, K;
    float slope;
    int use_covar;
    int tiles_x, tiles_y, ntiles;       // forward: tiles of 32 x 16 pixels; backward: runs of 32 x rows pixels
    int rows;
    float* partials;                    // forward: [ntiles][32 classes][8] = {ref, s0, sx, sy, sxx, syy, sxy, -} about the tile centre
    const float* aux; const float* gmu; const float* gsigma;
    bf16_t* g; float* sums; float* parts;
};

constexpr float L2E = 1.4426950408889634f;
constexpr int FTR = 16;                 // rows of a forward tile (moments are taken about its centre)

__device__ __forceinline__ bf16x8 as_frag(unsigned a, unsigned b, unsigned c, unsigned d) {
    u32x4 v = {a, b, c, d};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x4 tr_read64(unsigned lds_byte_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(size_t)lds_byte_addr);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    return (unsigned)f32_to_bf16(a) | ((unsigned)f32_to_bf16(b) << 16);
}

// weight fragment of a lane: row `row` of a [32][32] bf16 matrix, elements 4 h + (j & 3) + 8 (j >> 2) + 16 s
__device__ __forceinline__ void load_wfrag(const bf16_t* w, int row, int h, bf16x8 (&f)[2]) {
    const bf16_t* p = w + row * 32 + 4 * h;
    const u32x2 a0 = *reinterpret_cast<const u32x2*>(p), a1 = *reinterpret_cast<const u32x2*>(p + 8);
    const u32x2 a2 = *reinterpret_cast<const u32x2*>(p + 16), a3 = *reinterpret_cast<const u32x2*>(p + 24);
    f[0] = as_frag(a0[0], a0[1], a1[0], a1[1]);
    f[1] = as_frag(a2[0], a2[1], a3[0], a3[1]);
}

// the lane's 16 values of a per-channel f32 row ([32] floats): index i -> channel (i & 3) + 8 (i >> 2) + 4 h
__device__ __forceinline__ void load_chan16(const float* row, int h, float (&v)[16]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(row + 8 * g + 4 * h);
        v[4 * g] = t[0]; v[4 * g + 1] = t[1]; v[4 * g + 2] = t[2]; v[4 * g + 3] = t[3];
    }
}

// two 16-byte pieces of a pixel (channels 8 h .. 8 h + 7 and 16 + 8 h ..) -> the lane's 16 channels, packed pairs q[0..7]
__device__ __forceinline__ void swap_in(const u32x4 p0, const u32x4 p1, unsigned (&q)[8]) {
    const auto a = __builtin_amdgcn_permlane32_swap(p0[0], p0[2], false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(p0[1], p0[3], false, false);
    q[0] = a[0]; q[1] = b[0]; q[2] = a[1]; q[3] = b[1];
    const auto c = __builtin_amdgcn_permlane32_swap(p1[0], p1[2], false, false);
    const auto d = __builtin_amdgcn_permlane32_swap(p1[1], p1[3], false, false);
    q[4] = c[0]; q[5] = d[0]; q[6] = c[1]; q[7] = d[1];
}

// a = bf16(LeakyReLU(z * scale + shift)) of the lane's 16 channels -> the two B fragments; pos: bit i = (y_i > 0)
__device__ __forceinline__ void normalise(const unsigned (&q)[8], const float (&sc)[16], const float (&sh)[16], float slope,
                                          unsigned (&a)[8], unsigned& pos) {
    pos = 0;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const float z0 = __uint_as_float(q[d] << 16), z1 = __uint_as_float(q[d] & 0xffff0000u);
        const float y0 = fmaf(z0, sc[2 * d], sh[2 * d]), y1 = fmaf(z1, sc[2 * d + 1], sh[2 * d + 1]);
        pos |= (y0 > 0.f ? 1u : 0u) << (2 * d) | (y1 > 0.f ? 1u : 0u) << (2 * d + 1);
        a[d] = pack2(y0 > 0.f ? y0 : y0 * slope, y1 > 0.f ? y1 : y1 * slope);
    }
}

// ------------------------------------------------------------------------------------------------------------ forward
// NG = ceil(K / 8): a lane carries 4 NG class slots (slot i = class (i & 3) + 8 (i >> 2) + 4 h).
template <int NG>
__global__ __launch_bounds__(256) void head_fwd_kernel(const HfArgs p) {
    constexpr int NS = 4 * NG;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= p.ntiles) return;                      // whole waves; no workgroup barrier below
    // (a persistent loop over tiles with a one-round grid was measured: 95 us against 82 us for one tile per wave)
    const int tx = tile % p.tiles_x, rest = tile / p.tiles_x;
    const int ty = rest % p.tiles_y, n = rest / p.tiles_y;

    bf16x8 wA[2];
    load_wfrag(p.w_cls, r, h, wA);
    float sc[16], sh[16];
    load_chan16(p.stats + ((size_t)2 * p.N + n) * 32, h, sc);
    load_chan16(p.stats + ((size_t)3 * p.N + n) * 32, h, sh);

    const size_t row_el = (size_t)p.W * 32;
    const bf16_t* zp = p.z + (((size_t)n * p.H + ty * FTR) * p.W + tx * 32 + r) * 32 + 8 * h;
    u32x4 c0 = *reinterpret_cast<const u32x4*>(zp), c1 = *reinterpret_cast<const u32x4*>(zp + 16);
    u32x4 n0 = *reinterpret_cast<const u32x4*>(zp + row_el), n1 = *reinterpret_cast<const u32x4*>(zp + row_el + 16);

    float ref[NS], s0[NS], sy[NS], syy[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) { ref[i] = 0.f; s0[i] = 0.f; sy[i] = 0.f; syy[i] = 0.f; }
    const float step = 2.f / (float)p.W;

#pragma unroll 1
    for (int b = 0; b < FTR; ++b) {
        u32x4 m0 = n0, m1 = n1;
        if (b + 2 < FTR) {
            m0 = *reinterpret_cast<const u32x4*>(zp + (size_t)(b + 2) * row_el);
            m1 = *reinterpret_cast<const u32x4*>(zp + (size_t)(b + 2) * row_el + 16);
        }
        unsigned q[8], a[8], pos;
        swap_in(c0, c1, q);
        normalise(q, sc, sh, p.slope, a, pos);
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wA[0], as_frag(a[0], a[1], a[2], a[3]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wA[1], as_frag(a[4], a[5], a[6], a[7]), acc, 0, 0, 0);
        if (b == 0) {
            // reference logit of a (half wave, class): the half's first pixel; it only has to be within e^40 of the values
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc[i]), 0));
                const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, acc[i]), 32));
                ref[i] = h ? hi : lo;
            }
        }
        float over = -INFINITY;
#pragma unroll
        for (int i = 0; i < NS; ++i) over = fmaxf(over, acc[i] - ref[i]);
        if (__builtin_amdgcn_ballot_w64(over > 40.f)) {       // rare: move the references up (wave-uniform branch)
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                float d = acc[i] - ref[i];
#pragma unroll
                for (int m = 1; m < 32; m <<= 1) d = fmaxf(d, __shfl_xor(d, m, 64));
                if (d > 40.f) {
                    const float f = __builtin_amdgcn_exp2f(-d * L2E);
                    ref[i] += d; s0[i] *= f; sy[i] *= f; syy[i] *= f;
                }
            }
        }
        const float dy = ((float)b - 0.5f * (FTR - 1)) * step;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const float e = __builtin_amdgcn_exp2f((acc[i] - ref[i]) * L2E);
            const float t = e * dy;
            s0[i] += e;
            sy[i] += t;
            syy[i] = fmaf(t, dy, syy[i]);
        }
        c0 = n0; c1 = n1; n0 = m0; n1 = m1;
    }
    // ---- the tile's moments about its centre: the lane's dx is constant, so sx = dx s0, sxx = dx^2 s0, sxy = dx sy
    const float dx = ((float)r - 15.5f) * step;
    float* out = p.partials + (size_t)tile * 256;
    if (r == 0) {
#pragma unroll
        for (int i = 0; i < NS; ++i) out[((i & 3) + 8 * (i >> 2) + 4 * h) * 8] = ref[i];
    }
    const int reg = ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 8) >> 2) | ((lane & 16) >> 4);
    float* mine = out + ((reg & 3) + 8 * (reg >> 2) + 4 * h) * 8;
    const bool writer = !(lane & 4) && reg < NS;
    float v[16];
#define HF_REDUCE(expr, slot)                                   \
    _Pragma("unroll") for (int i = 0; i < 16; ++i) v[i] = 0.f;  \
    _Pragma("unroll") for (int i = 0; i < NS; ++i) v[i] = (expr); \
    { const float t = lane_reduce16(v, lane); if (writer) mine[slot] = t; }
    HF_REDUCE(s0[i], 1)
    HF_REDUCE(dx * s0[i], 2)
    HF_REDUCE(sy[i], 3)
    HF_REDUCE(dx * dx * s0[i], 4)
    HF_REDUCE(syy[i], 5)
    HF_REDUCE(dx * sy[i], 6)
#undef HF_REDUCE
}

// One wave per heat map: the tiles' partial moments -> mu, Sigma, aux of cu_dsnt_head_fwd.  f64 for the shift to the image
// origin and the final E[x^2] - E[x]^2 (see dsnt_fwd_kernel in head.hip); the f32 partials are taken about tile centres, so
// their rounding is relative to (1/8 of the image)^2, not to 1.
__global__ __launch_bounds__(64) void head_fwd_finish_kernel(const float* __restrict__ partials, int K, int H, int W,
                                                            int tiles_x, int tiles_y, int use_covar, float* __restrict__ mu,
                                                            float* __restrict__ sigma, float* __restrict__ aux) {
    const int map = blockIdx.x, n = map / K, k = map - n * K, lane = threadIdx.x;
    const int per = tiles_x * tiles_y;
    const float* base = partials + ((size_t)n * per * 32 + k) * 8;
    float mx = -INFINITY;
    for (int t = lane; t < per; t += 64) mx = fmaxf(mx, base[(size_t)t * 256]);
    mx = wave_max(mx);
    double s0 = 0, sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
    const double invW = 1.0 / (double)W;
    for (int t = lane; t < per; t += 64) {
        const float* q = base + (size_t)t * 256;
        const f32x4 a = *reinterpret_cast<const f32x4*>(q), b = *reinterpret_cast<const f32x4*>(q + 4);
        const int tx = t % tiles_x, ty = t / tiles_x;
        const double X = (64.0 * tx + 32.0) * invW - 1.0, Y = (2.0 * FTR * ty + FTR) * invW - 1.0;
        const double w = (double)expf(a[0] - mx);
        const double m0 = w * a[1], mx1 = w * a[2], my1 = w * a[3], mxx = w * b[0], myy = w * b[1], mxy = w * b[2];
        s0 += m0;
        sx += mx1 + X * m0;
        sy += my1 + Y * m0;
        sxx += mxx + 2.0 * X * mx1 + X * X * m0;
        syy += myy + 2.0 * Y * my1 + Y * Y * m0;
        sxy += mxy + X * my1 + Y * mx1 + X * Y * m0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s0 += __shfl_xor(s0, o, 64); sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64);
        sxx += __shfl_xor(sxx, o, 64); syy += __shfl_xor(syy, o, 64); sxy += __shfl_xor(sxy, o, 64);
    }
    if (lane == 0) {
        const double inv = 1.0 / s0;
        const double xbar = sx * inv, ybar = sy * inv;
        const double vx = sxx * inv - xbar * xbar, vy = syy * inv - ybar * ybar, cv = sxy * inv - xbar * ybar;
        const double half = 0.5 * (double)W;
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

// ----------------------------------------------------------------------------------------------------------- backward
template <int NG>
__global__ __launch_bounds__(256, 2) void head_bwd_kernel(const HfArgs p) {
    constexpr int NS = 4 * NG;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 4096 + 4 * 32 * 12 * 4 + 4 * 256 + 4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    unsigned char* img = smem + wave * 4096;                                   // [32 pixels][64 B] dl, then the same for a
    float* tab = reinterpret_cast<float*>(smem + 4 * 4096) + wave * (32 * 12);   // [32 classes][12]
    float* nrm = reinterpret_cast<float*>(smem + 4 * 4096 + 4 * 32 * 12 * 4) + wave * 64;      // [32 channels]{scale, shift}
    const unsigned img_a = lds_addr(img);

    // weight fragments, lane-linear in LDS (re-read every row: 16 registers less): [frag 0..3][lane] x 16 B
    bf16x8* wfr = reinterpret_cast<bf16x8*>(smem + 4 * 4096 + 4 * 32 * 12 * 4 + 4 * 256);
    if (wave == 0) {
        bf16x8 wA[2], wT[2];
        load_wfrag(p.w_cls, r, h, wA);          // logits = W a   : rows = classes, k = channels
        load_wfrag(p.w_ch, r, h, wT);           // g = W^T dl     : rows = channels, k = classes (same k permutation)
        wfr[lane] = wA[0]; wfr[64 + lane] = wA[1]; wfr[128 + lane] = wT[0]; wfr[192 + lane] = wT[1];
    }
    __syncthreads();
    // LDS addressing of the two transposes.  Write: piece pc = 2 g + h of row r in slot pc ^ ((r >> 1) & 7); read
    // (ds_read_b64_tr_b16: the lane supplies row q of its group's 4 x 16 block, columns 4 pp ..): piece 4 chalf + pp
    const unsigned wr_row = (unsigned)(r * 64), wr_x = (unsigned)((r >> 1) & 7);
    const int q4 = (lane >> 2) & 3, pp = lane & 3, chalf = (lane >> 4) & 1;
    unsigned rd[2][2];                                                         // [k-step][lo / hi]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = 16 * ks + 8 * h + 4 * u + q4;
            rd[ks][u] = (unsigned)(row * 64 + 8 * ((4 * chalf + pp) ^ ((row >> 1) & 7)));
        }

    f32x16 dw;
#pragma unroll
    for (int i = 0; i < 16; ++i) dw[i] = 0.f;
    const int reg = ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 8) >> 2) | ((lane & 16) >> 4);
    const size_t row_el = (size_t)p.W * 32;
    const int total_waves = gridDim.x * 4;

#pragma unroll 1
    for (int run = blockIdx.x * 4 + wave; run < p.ntiles; run += total_waves) {
        const int tx = run % p.tiles_x, rest = run / p.tiles_x;
        const int ty = rest % p.tiles_y, n = rest / p.tiles_y;
        // ---- per-class constants of this image (lane = class) -> wave-local table
        {
            float c[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) c[i] = 0.f;
            if (r < p.K) {
                const int map = n * p.K + r;
                const float* a = p.aux + 8 * (size_t)map;
                const float half = 0.5f * (float)p.W;
                const float gx = p.gmu[2 * map] * 0.5f * (float)p.W, gy = p.gmu[2 * map + 1] * 0.5f * (float)p.H;
                const float gxx = p.gsigma[3 * map] * half * half, gyy = p.gsigma[3 * map + 1] * half * half;
                const float gxy = p.use_covar ? p.gsigma[3 * map + 2] * half * half : 0.f;
                const float xbar = a[2], ybar = a[3];
                const float pq = gx * xbar + gy * ybar + gxx * a[4] + gyy * a[5] + gxy * a[6];
                c[0] = gx; c[1] = gxx; c[2] = gxy; c[3] = gy;
                c[4] = gyy; c[5] = a[0] - __logf(a[1]); c[6] = xbar; c[7] = ybar;      // c[5]: log-sum-exp, p = exp(l - lse)
                c[8] = gy * ybar - pq;
            }
            if (h == 0) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    *reinterpret_cast<f32x4*>(tab + r * 12 + 4 * i) = f32x4{c[4 * i], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3]};
                *reinterpret_cast<f32x2*>(nrm + 2 * r) = f32x2{p.stats[((size_t)2 * p.N + n) * 32 + r], p.stats[((size_t)3 * p.N + n) * 32 + r]};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // q - sum p q = A + dy (B + gyy dy), dy = Y - ybar:  A = gx X + gxx dx^2 + gy ybar - pq,  B = gy + gxy dx
        const float X = (2.f * (float)(tx * 32 + r) + 1.f) / (float)p.W - 1.f;
        // (A and B live in registers; gyy, lse, ybar are re-read from the table every row: 36 registers less)
        float cA[NS], cB[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const float* t = tab + ((i & 3) + 8 * (i >> 2) + 4 * h) * 12;
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(t), t1 = *reinterpret_cast<const f32x4*>(t + 4);
            const float c0 = t[8];
            const float dx = X - t1[2];
            cA[i] = fmaf(t0[1] * dx, dx, fmaf(t0[0], X, c0));
            cB[i] = fmaf(t0[2], dx, t0[3]);
        }
        float sgl[16], sgz[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { sgl[i] = 0.f; sgz[i] = 0.f; }

        const size_t pix0 = ((size_t)n * p.H + (size_t)ty * p.rows) * p.W + tx * 32 + r;
        const bf16_t* zp = p.z + pix0 * 32 + 8 * h;
        bf16_t* gp = p.g + pix0 * 32 + 16 * h;
        u32x4 c0 = *reinterpret_cast<const u32x4*>(zp), c1 = *reinterpret_cast<const u32x4*>(zp + 16);
        u32x4 n0 = c0, n1 = c1;
        if (p.rows > 1) { n0 = *reinterpret_cast<const u32x4*>(zp + row_el); n1 = *reinterpret_cast<const u32x4*>(zp + row_el + 16); }
#pragma unroll 1
        for (int b = 0; b < p.rows; ++b) {
            u32x4 m0 = n0, m1 = n1;
            if (b + 2 < p.rows) {
                m0 = *reinterpret_cast<const u32x4*>(zp + (size_t)(b + 2) * row_el);
                m1 = *reinterpret_cast<const u32x4*>(zp + (size_t)(b + 2) * row_el + 16);
            }
            unsigned q[8], a[8], pos = 0;
            swap_in(c0, c1, q);
            // a = bf16(LeakyReLU(z scale + shift)); scale / shift of the channel pair from the wave's table (not 32 registers)
#pragma unroll
            for (int d2 = 0; d2 < 8; ++d2) {
                const f32x4 ss = *reinterpret_cast<const f32x4*>(nrm + 2 * (2 * (d2 & 1) + 8 * (d2 >> 1) + 4 * h));
                const float z0 = __uint_as_float(q[d2] << 16), z1 = __uint_as_float(q[d2] & 0xffff0000u);
                const float y0 = fmaf(z0, ss[0], ss[1]), y1 = fmaf(z1, ss[2], ss[3]);
                pos |= (y0 > 0.f ? 1u : 0u) << (2 * d2) | (y1 > 0.f ? 1u : 0u) << (2 * d2 + 1);
                a[d2] = pack2(y0 > 0.f ? y0 : y0 * p.slope, y1 > 0.f ? y1 : y1 * p.slope);
            }
            // (dW[class][channel] += sum over the row's 32 pixels dl a needs both tiles k-major: they go through the wave's LDS
            //  image as soon as they exist, so that their registers die early)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<u32x2*>(img + 2048 + wr_row + 8u * ((unsigned)(2 * g4 + h) ^ wr_x)) = u32x2{a[2 * g4], a[2 * g4 + 1]};
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[lane], as_frag(a[0], a[1], a[2], a[3]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[64 + lane], as_frag(a[4], a[5], a[6], a[7]), acc, 0, 0, 0);
            // ---- dl = p (q - sum p q), rounded to bf16 (what cu_dsnt_head_bwd_nhwc stores)
            const float Y = (2.f * (float)(ty * p.rows + b) + 1.f) / (float)p.W - 1.f;
            float dl[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if ((i & 3) == 0) asm volatile("" ::: "memory");      // keep the table reads of a class group next to their use
                if (i < NS) {
                    const f32x4 t1 = *reinterpret_cast<const f32x4*>(tab + ((i & 3) + 8 * (i >> 2) + 4 * h) * 12 + 4);   // gyy, lse, xbar, ybar
                    const float pr = __builtin_amdgcn_exp2f((acc[i] - t1[1]) * L2E);
                    const float dy = Y - t1[3];
                    dl[i] = pr * fmaf(dy, fmaf(t1[0], dy, cB[i]), cA[i]);
                } else {
                    dl[i] = 0.f;
                }
            }
            unsigned d[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] = pack2(dl[2 * k], dl[2 * k + 1]);
            // ---- g = W^T dl: the accumulator layout IS the B operand (k-slot j of step s = class slot 8 s + j)
            f32x16 gacc;
#pragma unroll
            for (int i = 0; i < 16; ++i) gacc[i] = 0.f;
            gacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[128 + lane], as_frag(d[0], d[1], d[2], d[3]), gacc, 0, 0, 0);
            gacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr[192 + lane], as_frag(d[4], d[5], d[6], d[7]), gacc, 0, 0, 0);
            // ---- dW += dl^T a
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<u32x2*>(img + wr_row + 8u * ((unsigned)(2 * g4 + h) ^ wr_x)) = u32x2{d[2 * g4], d[2 * g4 + 1]};
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x4 alo = tr_read64(img_a + rd[ks][0]), ahi = tr_read64(img_a + rd[ks][1]);
                const bf16x4 blo = tr_read64(img_a + 2048 + rd[ks][0]), bhi = tr_read64(img_a + 2048 + rd[ks][1]);
                dw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7),
                                                             __builtin_shufflevector(blo, bhi, 0, 1, 2, 3, 4, 5, 6, 7), dw, 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the reads are done before the next row's writes
            // ---- InstanceNorm-backward sums of the last ConvLayer: gl = g LeakyReLU'(y); sum gl, sum gl z
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float zf = (i & 1) ? __uint_as_float(q[i >> 1] & 0xffff0000u) : __uint_as_float(q[i >> 1] << 16);
                const float gl = (pos >> i) & 1u ? gacc[i] : gacc[i] * p.slope;
                sgl[i] += gl;
                sgz[i] = fmaf(gl, zf, sgz[i]);
            }
            // ---- g row store: two 16-byte pieces per lane (channels 16 h .. 16 h + 15)
            {
                unsigned w[4][2];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    w[g4][0] = pack2(gacc[4 * g4], gacc[4 * g4 + 1]);
                    w[g4][1] = pack2(gacc[4 * g4 + 2], gacc[4 * g4 + 3]);
                }
#pragma unroll
                for (int w2 = 0; w2 < 2; ++w2) {
                    const auto r02 = __builtin_amdgcn_permlane32_swap(w[0][w2], w[2][w2], false, false);
                    w[0][w2] = r02[0]; w[2][w2] = r02[1];
                    const auto r13 = __builtin_amdgcn_permlane32_swap(w[1][w2], w[3][w2], false, false);
                    w[1][w2] = r13[0]; w[3][w2] = r13[1];
                }
                bf16_t* o = gp + (size_t)b * row_el;
                *reinterpret_cast<u32x4*>(o) = u32x4{w[0][0], w[0][1], w[2][0], w[2][1]};
                *reinterpret_cast<u32x4*>(o + 8) = u32x4{w[1][0], w[1][1], w[3][0], w[3][1]};
            }
            c0 = n0; c1 = n1; n0 = m0; n1 = m1;
        }
        // ---- flush the run's sums: (sum gl, sum gl zhat) with zhat = (z - mean) rstd -- linear in the raw sums
        {
            const float tg = lane_reduce16(sgl, lane), tz = lane_reduce16(sgz, lane);
            if (!(lane & 4)) {
                const int ch = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                const float mean = p.stats[(size_t)n * 32 + ch], rstd = p.stats[((size_t)p.N + n) * 32 + ch];
                float* o = p.sums + ((size_t)n * 32 + ch) * 2;
                unsafeAtomicAdd(o, tg);
                unsafeAtomicAdd(o + 1, rstd * (tz - mean * tg));
            }
        }
    }
    // ---- the workgroup's dW slab, plain [32 classes][32 channels]: lane (channel r, half h) holds classes (i&3)+8(i>>2)+4h
    __syncthreads();
    float* stage = reinterpret_cast<float*>(smem) + wave * 1024;
#pragma unroll
    for (int i = 0; i < 16; ++i) stage[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = dw[i];
    __syncthreads();
    const float* s = reinterpret_cast<const float*>(smem);
    float* slab = p.parts + (size_t)blockIdx.x * 1024;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = threadIdx.x + 256 * j;
        slab[e] = (s[e] + s[1024 + e]) + (s[2048 + e] + s[3072 + e]);
    }
}

inline bool hf_shape_ok(int N, int H, int W, int K) {
    return N > 0 && K > 0 && K <= 32 && H == W && W % 32 == 0 && H % FTR == 0 && (size_t)N * H * W * 64 < 0x7fff0000ull * 4;
}

}  // namespace

extern "C" size_t cu_head_fused_ws_floats(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)N * (size_t)((W + 31) / 32) * (size_t)((H + FTR - 1) / FTR) * 256;
}

extern "C" int cu_head_fused_fwd(int N, int H, int W, int K, const void* z, const float* stats, float slope, const void* w_cls,
                                 int use_covar, float* ws, size_t ws_floats, float* mu, float* sigma, float* aux, void* stream) {
    CU_CHECK_ARG(hf_shape_ok(N, H, W, K), "cu_head_fused_fwd: needs square maps with W %% 32 == 0, H %% 16 == 0 and K <= 32, got %dx%d K=%d",
                 H, W, K);
    CU_CHECK_ARG(z && stats && w_cls && ws && mu && sigma && aux, "cu_head_fused_fwd: null pointer");
    CU_CHECK_ARG(ws_floats >= cu_head_fused_ws_floats(N, H, W), "cu_head_fused_fwd: workspace of %zu floats, needs %zu", ws_floats,
                 cu_head_fused_ws_floats(N, H, W));
    HfArgs a;
    memset(&a, 0, sizeof(a));
    a.z = (const bf16_t*)z; a.stats = stats; a.w_cls = (const bf16_t*)w_cls; a.N = N; a.H = H; a.W = W; a.K = K;
    a.slope = slope; a.use_covar = use_covar; a.tiles_x = W / 32; a.tiles_y = H / FTR; a.ntiles = N * a.tiles_x * a.tiles_y;
    a.partials = ws;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(cdiv(a.ntiles, 4));
    const int ng = (K + 7) / 8;
    if (ng <= 1) hipLaunchKernelGGL(head_fwd_kernel<1>, grid, dim3(256), 0, st, a);
    else if (ng == 2) hipLaunchKernelGGL(head_fwd_kernel<2>, grid, dim3(256), 0, st, a);
    else if (ng == 3) hipLaunchKernelGGL(head_fwd_kernel<3>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(head_fwd_kernel<4>, grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL(head_fwd_finish_kernel, dim3(N * K), dim3(64), 0, st, ws, K, H, W, a.tiles_x, a.tiles_y, use_covar, mu,
                       sigma, aux);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_head_fused_bwd(int N, int H, int W, int K, const void* z, const float* stats, float slope, const void* w_cls,
                                 const void* w_ch, const float* aux, const float* gmu, const float* gsigma, int use_covar, void* g,
                                 float* sums, float* parts, size_t parts_floats, int* nparts, void* stream) {
    CU_CHECK_ARG(hf_shape_ok(N, H, W, K), "cu_head_fused_bwd: needs square maps with W %% 32 == 0, H %% 16 == 0 and K <= 32, got %dx%d K=%d",
                 H, W, K);
    CU_CHECK_ARG(z && stats && w_cls && w_ch && aux && gmu && gsigma && g && sums && parts && nparts, "cu_head_fused_bwd: null pointer");
    HfArgs a;
    memset(&a, 0, sizeof(a));
    a.z = (const bf16_t*)z; a.stats = stats; a.w_cls = (const bf16_t*)w_cls; a.w_ch = (const bf16_t*)w_ch;
    a.N = N; a.H = H; a.W = W; a.K = K; a.slope = slope; a.use_covar = use_covar;
    a.aux = aux; a.gmu = gmu; a.gsigma = gsigma; a.g = (bf16_t*)g; a.sums = sums; a.parts = parts;
    // runs of 32 x rows pixels per wave: one round of 512 workgroups (two per CU at this kernel's 2 waves per SIMD) with one run
    // per wave where the problem is large enough.  Measured at 64 x 256^2 (tools/head_bench.py, tuning knobs CU_HF_ROWS /
    // CU_HF_WGS): rows 64 x 512 workgroups 155 us; 32 x 1024: 161 - 164; 16: 173 - 177; 128: 184; 256: 293
    int rows = H;
    while (rows > 16 && rows % 2 == 0 && (long)N * (W / 32) * (H / rows) < 2048) rows /= 2;
    if (cu_env_int("CU_HF_ROWS", 0) > 0 && H % cu_env_int("CU_HF_ROWS", 0) == 0) rows = cu_env_int("CU_HF_ROWS", 0);
    a.rows = rows; a.tiles_x = W / 32; a.tiles_y = H / rows; a.ntiles = N * a.tiles_x * a.tiles_y;
    int wgs = cdiv(a.ntiles, 4);
    const int wg_cap = cu_env_int("CU_HF_WGS", 512) <= 1024 ? cu_env_int("CU_HF_WGS", 512) : 1024;
    if (wgs > wg_cap) wgs = wg_cap;
    CU_CHECK_ARG(parts_floats >= (size_t)(wgs + 1) * 1024, "cu_head_fused_bwd: workspace of %zu floats, needs %zu", parts_floats,
                 (size_t)(wgs + 1) * 1024);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int ng = (K + 7) / 8;
    if (ng <= 1) hipLaunchKernelGGL(head_bwd_kernel<1>, dim3(wgs), dim3(256), 0, st, a);
    else if (ng == 2) hipLaunchKernelGGL(head_bwd_kernel<2>, dim3(wgs), dim3(256), 0, st, a);
    else if (ng == 3) hipLaunchKernelGGL(head_bwd_kernel<3>, dim3(wgs), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(head_bwd_kernel<4>, dim3(wgs), dim3(256), 0, st, a);
    CU_LAUNCH_CHECK();
    *nparts = wgs;
    return 0;
}
