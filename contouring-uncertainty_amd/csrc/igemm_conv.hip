// Generic implicit-GEMM gather convolution on MFMA (gfx950).  See include/contour_hip.h : cu_conv_gemm.
//
//   D[p, n] = bias[n] + sum_t sum_c act(S[p*IS + off_t, c]) * W[t][n][c]
//
// Replaces (reference ThierryJudge/contouring-uncertainty): nn.Conv2d 3x3 s1/s2 (models/nnUnet/layers.py:55-80,192) and
// its input gradient, nn.ConvTranspose2d k2 s2 (layers.py:83-109,415-417) and its input gradient, the 1x1 output conv
// (layers.py:456-463), torch.cat of the skip (layers.py:436) and the InstanceNorm2d+LeakyReLU of the producing layer
// (layers.py:193-194) fused into the operand load, and the ConfidenceNet convs (unet2.py:21-27).
//
// Structure (one workgroup = 4 waves, one wave per SIMD):
//   - a workgroup owns BM = 128*MA loop pixels (an IMGS x TH x TW patch) x BN = 32*NB output channels;
//   - per channel chunk (CK = 32 bf16 / 16 f32) the source halo patch and all taps' weights are staged in LDS in
//     "k-plane" order (16-byte entries of 8 bf16 / 4-byte f32 planes) so that every MFMA fragment read is a
//     conflict-free ds_read_b128 / ds_read_b32; the halo is read once and reused by all taps (9x less LDS fill than
//     im2col);
//   - the next chunk's global loads are issued before the current chunk's MFMAs (register prefetch);
//   - bf16: v_mfma_f32_32x32x16_bf16; f32 (parity mode): v_mfma_f32_32x32x2_f32 (exact f32 FMA chain).
#include "common.h"

namespace {

constexpr int MAXI = 10;   // halo pieces per thread per chunk (host checks halo_px*4 <= 256*MAXI)

struct ConvKArgs {
    const void* src0; const void* src1;
    const float* sc0; const float* sh0; const float* sc1; const float* sh1;
    const void* w; const float* bias; void* dst0; void* dst1;
    int N, PH, PW, SH, SW, C0, C1, IS, OH, OW, OS, OY0, OX0, CO, D0, DC0, DC1, ntaps;
    int tap_off[CU_MAX_TAPS];
    int tap_w[CU_MAX_TAPS];
    int dymin, dxmin, HH, HW, halo_px, XP, WP;
    int twl, thl, iml;          // log2 of the tile's TW, TH and image count
    int tiles_x, tiles_y;
    float slope0, slope1;
    int accum0, accum1, out_nchw;
    int imgs;                   // images per tile (TW*TH*imgs <= 128; rows beyond are idle)
};

template <typename T> struct Cfg;
template <> struct Cfg<bf16_t> {
    static constexpr int CK = 32;        // channels per chunk
    static constexpr int PIECE = 8;      // elements per 16-byte piece
    static constexpr int PPP = 1;        // LDS planes per piece
    static constexpr int KSTEPS = 2;     // MFMA k-steps per chunk (16 channels each)
    static constexpr int WPLANES = 4;    // weight planes per tap
};
template <> struct Cfg<float> {
    static constexpr int CK = 16;
    static constexpr int PIECE = 4;
    static constexpr int PPP = 4;
    static constexpr int KSTEPS = 8;     // 2 channels each
    static constexpr int WPLANES = 16;
};

template <typename T, int MA, int NB>
__global__ __launch_bounds__(256) void igemm_conv_kernel(const ConvKArgs p) {
    using C = Cfg<T>;
    constexpr int CK = C::CK, PIECE = C::PIECE, PPP = C::PPP;
    constexpr int BN = 32 * NB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int TW = 1 << p.twl, TH = 1 << p.thl;

    int bx = blockIdx.x;
    const int tile_x = bx % p.tiles_x; bx /= p.tiles_x;
    const int tile_y = bx % p.tiles_y;
    const int ig = bx / p.tiles_y;
    const int py0 = tile_y << p.thl, px0 = tile_x << p.twl, img0 = ig << p.iml;
    const int n0 = blockIdx.y * BN;
    const int CI = p.C0 + p.C1;
    const int sy0 = py0 * p.IS + p.dymin, sx0 = px0 * p.IS + p.dxmin;
    const int hpi = p.HH * p.HW;   // halo pixels per image

    // LDS carve: X planes then W planes.  Units: 16-byte entries (bf16) / floats (f32).
    T* Xs = reinterpret_cast<T*>(smem);
    const int x_elems = (CK / PIECE) * PPP * p.XP * (PPP == 1 ? PIECE : 1);
    T* Ws = Xs + x_elems;

    // ---- per-thread halo staging items (independent of the chunk)
    int pix[MAXI], nimg[MAXI], ldsx[MAXI];
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
        const int i = tid + j * 256;
        pix[j] = -2; nimg[j] = 0; ldsx[j] = 0;
        if (i < p.halo_px * 4) {
            const int piece = i & 3, hp = i >> 2;
            const int im = hp / hpi;
            const int rem = hp - im * hpi;
            const int hy = rem / p.HW, hx = rem - hy * p.HW;
            const int n = img0 + im, sy = sy0 + hy, sx = sx0 + hx;
            const bool inb = (n < p.N) && (sy >= 0) && (sy < p.SH) && (sx >= 0) && (sx < p.SW);
            pix[j] = inb ? (n * p.SH + sy) * p.SW + sx : -1;
            nimg[j] = n;
            ldsx[j] = piece * PPP * p.XP + hp;
        }
    }

    // ---- A-fragment base halo index per M block
    int hpA[MA];
#pragma unroll
    for (int a = 0; a < MA; ++a) {
        const int m = wave * 32 * MA + a * 32 + r;
        const int tx = m & (TW - 1), ty = (m >> p.twl) & (TH - 1), im = m >> (p.twl + p.thl);
        hpA[a] = im < p.imgs ? im * hpi + ty * p.IS * p.HW + tx * p.IS : 0;
    }

    f32x16 acc[MA][NB];
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    const int w_items = p.ntaps * BN * 4;
    constexpr int MAXW = (CU_MAX_TAPS * BN * 4 + 255) / 256;
    u32x4 xreg[MAXI];
    u32x4 wreg[MAXW];

    auto prefetch = [&](int c0) {
        const bool s1 = c0 >= p.C0;
        const T* src = reinterpret_cast<const T*>(s1 ? p.src1 : p.src0);
        const int Cs = s1 ? p.C1 : p.C0;
        const int cc = s1 ? c0 - p.C0 : c0;
#pragma unroll
        for (int j = 0; j < MAXI; ++j) {
            if (pix[j] >= 0) {
                const int piece = (tid + j * 256) & 3;
                xreg[j] = *reinterpret_cast<const u32x4*>(src + (size_t)pix[j] * Cs + cc + piece * PIECE);
            }
        }
        const T* wp = reinterpret_cast<const T*>(p.w);
#pragma unroll
        for (int j = 0; j < MAXW; ++j) {
            const int i = tid + j * 256;
            if (i < w_items) {
                const int piece = i & 3;
                const int col = (i >> 2) % BN;
                const int t = (i >> 2) / BN;
                const int n = n0 + col;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (n < p.CO)
                    v = *reinterpret_cast<const u32x4*>(wp + ((size_t)p.tap_w[t] * p.CO + n) * CI + c0 + piece * PIECE);
                wreg[j] = v;
            }
        }
    };

    auto commit = [&](int c0) {
        const bool s1 = c0 >= p.C0;
        const int Cs = s1 ? p.C1 : p.C0;
        const int cc = s1 ? c0 - p.C0 : c0;
        const float* sc = s1 ? p.sc1 : p.sc0;
        const float* sh = s1 ? p.sh1 : p.sh0;
        const float slope = s1 ? p.slope1 : p.slope0;
#pragma unroll
        for (int j = 0; j < MAXI; ++j) {
            if (pix[j] >= -1) {
                float v[PIECE];
                if (pix[j] >= 0) {
                    const int piece = (tid + j * 256) & 3;
                    if constexpr (PIECE == 8) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[2 * e] = __uint_as_float(xreg[j][e] << 16);
                            v[2 * e + 1] = __uint_as_float(xreg[j][e] & 0xffff0000u);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = __uint_as_float(xreg[j][e]);
                    }
                    if (sc != nullptr) {
                        const float* scp = sc + (size_t)nimg[j] * Cs + cc + piece * PIECE;
                        const float* shp = sh + (size_t)nimg[j] * Cs + cc + piece * PIECE;
#pragma unroll
                        for (int e = 0; e < PIECE; ++e) v[e] = v[e] * scp[e] + shp[e];
                    }
#pragma unroll
                    for (int e = 0; e < PIECE; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
                } else {
#pragma unroll
                    for (int e = 0; e < PIECE; ++e) v[e] = 0.f;
                }
                if constexpr (PIECE == 8) {
                    store_piece<bf16_t>(reinterpret_cast<bf16_t*>(Xs) + (size_t)ldsx[j] * 8, v);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) reinterpret_cast<float*>(Xs)[ldsx[j] + e * p.XP] = v[e];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < MAXW; ++j) {
            const int i = tid + j * 256;
            if (i < w_items) {
                const int piece = i & 3;
                const int col = (i >> 2) % BN;
                const int t = (i >> 2) / BN;
                if constexpr (PIECE == 8) {
                    *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(Ws) +
                                              ((size_t)(t * 4 + piece) * p.WP + col) * 8) = wreg[j];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        reinterpret_cast<float*>(Ws)[(t * 16 + piece * 4 + e) * p.WP + col] = __uint_as_float(wreg[j][e]);
                }
            }
        }
    };

    prefetch(0);
    for (int c0 = 0; c0 < CI; c0 += CK) {
        __syncthreads();          // previous chunk's fragment reads are done
        commit(c0);
        __syncthreads();
        if (c0 + CK < CI) prefetch(c0 + CK);

        for (int t = 0; t < p.ntaps; ++t) {
            const int toff = p.tap_off[t];
#pragma unroll
            for (int kk = 0; kk < C::KSTEPS; ++kk) {
                if constexpr (PIECE == 8) {
                    const bf16_t* X16 = reinterpret_cast<const bf16_t*>(Xs);
                    const bf16_t* W16 = reinterpret_cast<const bf16_t*>(Ws);
                    bf16x8 af[MA], bfr[NB];
#pragma unroll
                    for (int a = 0; a < MA; ++a)
                        af[a] = *reinterpret_cast<const bf16x8*>(X16 + ((size_t)(2 * kk + h) * p.XP + hpA[a] + toff) * 8);
#pragma unroll
                    for (int b = 0; b < NB; ++b)
                        bfr[b] = *reinterpret_cast<const bf16x8*>(
                            W16 + ((size_t)(t * 4 + 2 * kk + h) * p.WP + b * 32 + r) * 8);
#pragma unroll
                    for (int a = 0; a < MA; ++a)
#pragma unroll
                        for (int b = 0; b < NB; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
                } else {
                    const float* Xf = reinterpret_cast<const float*>(Xs);
                    const float* Wf = reinterpret_cast<const float*>(Ws);
                    float af[MA], bfr[NB];
#pragma unroll
                    for (int a = 0; a < MA; ++a) af[a] = Xf[(2 * kk + h) * p.XP + hpA[a] + toff];
#pragma unroll
                    for (int b = 0; b < NB; ++b) bfr[b] = Wf[(t * 16 + 2 * kk + h) * p.WP + b * 32 + r];
#pragma unroll
                    for (int a = 0; a < MA; ++a)
#pragma unroll
                        for (int b = 0; b < NB; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a], bfr[b], acc[a][b], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int col = n0 + b * 32 + r;
        if (col >= p.CO) continue;
        const float bias = p.bias ? p.bias[col] : 0.f;
        const bool d1 = col >= p.D0;
        const int dcol = d1 ? col - p.D0 : col;
        const int DC = d1 ? p.DC1 : p.DC0;
        const int accum = d1 ? p.accum1 : p.accum0;
        T* dst = reinterpret_cast<T*>(d1 ? p.dst1 : p.dst0);
        if (p.out_nchw && dcol >= p.DC0) continue;
#pragma unroll
        for (int a = 0; a < MA; ++a) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
                const int m = wave * 32 * MA + a * 32 + row;
                const int tx = m & (TW - 1), ty = (m >> p.twl) & (TH - 1), im = m >> (p.twl + p.thl);
                const int n = img0 + im;
                const int py = py0 + ty, px = px0 + tx;
                if (im >= p.imgs || n >= p.N || py >= p.PH || px >= p.PW) continue;
                const int oy = py * p.OS + p.OY0, ox = px * p.OS + p.OX0;
                const float v = acc[a][b][i] + bias;
                if (p.out_nchw) {
                    float* o = reinterpret_cast<float*>(p.dst0) + (((size_t)n * p.DC0 + dcol) * p.OH + oy) * p.OW + ox;
                    *o = accum ? *o + v : v;
                } else {
                    T* o = dst + (((size_t)n * p.OH + oy) * p.OW + ox) * DC + dcol;
                    Elem<T>::st(o, accum ? Elem<T>::ld(o) + v : v);
                }
            }
        }
    }
}

template <typename T, int MA, int NB>
int launch(const ConvKArgs& a, int tiles, hipStream_t st) {
    using C = Cfg<T>;
    const size_t x_bytes = (size_t)(C::CK / C::PIECE) * C::PPP * a.XP * (C::PPP == 1 ? 16 : 4);
    const size_t w_bytes = (size_t)a.ntaps * C::WPLANES * a.WP * (C::PPP == 1 ? 16 : 4);
    const size_t lds = x_bytes + w_bytes;
    CU_CHECK_ARG(lds <= 160 * 1024, "cu_conv_gemm: LDS %zu bytes exceeds 160 KiB", lds);
    auto k = igemm_conv_kernel<T, MA, NB>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        CU_CHECK_ARG(e == hipSuccess, "cu_conv_gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    dim3 grid(tiles, cdiv(a.CO, 32 * NB));
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
    CU_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int cu_conv_gemm(const cu_conv_desc* d, const void* src0, const float* scale0, const float* shift0,
                            const void* src1, const float* scale1, const float* shift1, const void* w,
                            const float* bias, void* dst0, void* dst1, void* stream) {
    CU_CHECK_ARG(d != nullptr, "cu_conv_gemm: null descriptor");
    CU_CHECK_ARG(d->dtype == CU_F32 || d->dtype == CU_BF16, "cu_conv_gemm: bad dtype %d", d->dtype);
    const int CK = d->dtype == CU_BF16 ? 32 : 16;
    CU_CHECK_ARG(d->ntaps >= 1 && d->ntaps <= CU_MAX_TAPS, "cu_conv_gemm: ntaps %d", d->ntaps);
    CU_CHECK_ARG(d->C0 > 0 && d->C0 % CK == 0 && d->C1 >= 0 && d->C1 % CK == 0,
                 "cu_conv_gemm: channel counts %d,%d must be multiples of %d", d->C0, d->C1, CK);
    CU_CHECK_ARG(src0 && w && dst0 && (d->C1 == 0 || src1), "cu_conv_gemm: null pointer");
    CU_CHECK_ARG((scale0 == nullptr) == (shift0 == nullptr) && (scale1 == nullptr) == (shift1 == nullptr),
                 "cu_conv_gemm: scale/shift must come in pairs");
    CU_CHECK_ARG(d->D0 > 0 && d->D0 <= d->CO && (d->D0 == d->CO || dst1), "cu_conv_gemm: bad destination split");
    CU_CHECK_ARG(d->D0 == d->CO || d->D0 % 32 == 0, "cu_conv_gemm: split point must be a multiple of 32");
    CU_CHECK_ARG(d->N > 0 && d->PH > 0 && d->PW > 0 && d->IS >= 1 && d->OS >= 1, "cu_conv_gemm: bad geometry");

    ConvKArgs a;
    memset(&a, 0, sizeof(a));
    a.src0 = src0; a.src1 = src1; a.sc0 = scale0; a.sh0 = shift0; a.sc1 = scale1; a.sh1 = shift1;
    a.w = w; a.bias = bias; a.dst0 = dst0; a.dst1 = dst1;
    a.N = d->N; a.PH = d->PH; a.PW = d->PW; a.SH = d->SH; a.SW = d->SW; a.C0 = d->C0; a.C1 = d->C1; a.IS = d->IS;
    a.OH = d->OH; a.OW = d->OW; a.OS = d->OS; a.OY0 = d->OY0; a.OX0 = d->OX0; a.CO = d->CO; a.D0 = d->D0;
    a.DC0 = d->DC0; a.DC1 = d->DC1; a.ntaps = d->ntaps;
    a.slope0 = d->slope0; a.slope1 = d->slope1; a.accum0 = d->accum0; a.accum1 = d->accum1;
    a.out_nchw = d->out_nchw_f32;

    // destination bounds: every loop pixel must land inside the destination image
    CU_CHECK_ARG((d->PH - 1) * d->OS + d->OY0 < d->OH && (d->PW - 1) * d->OS + d->OX0 < d->OW && d->OY0 >= 0 &&
                     d->OX0 >= 0,
                 "cu_conv_gemm: destination pixel out of range");

    // tile geometry: BM = 128 loop pixels = IMGS x TH x TW
    const int BM = 128;
    int tw = d->PW < 32 ? d->PW : 32;
    int th = BM / tw;
    if (th > d->PH) th = d->PH;
    int imgs = BM / (tw * th);
    a.twl = ilog2_exact(tw); a.thl = ilog2_exact(th); a.iml = ilog2_exact(imgs);
    CU_CHECK_ARG(a.twl >= 0 && a.thl >= 0 && a.iml >= 0 && d->PW % tw == 0 && d->PH % th == 0,
                 "cu_conv_gemm: loop grid %dx%d must be powers of two", d->PH, d->PW);
    a.tiles_x = d->PW / tw; a.tiles_y = d->PH / th;

    int dymin = 1 << 20, dxmin = 1 << 20, dymax = -(1 << 20), dxmax = -(1 << 20);
    for (int t = 0; t < d->ntaps; ++t) {
        dymin = d->tap_dy[t] < dymin ? d->tap_dy[t] : dymin; dymax = d->tap_dy[t] > dymax ? d->tap_dy[t] : dymax;
        dxmin = d->tap_dx[t] < dxmin ? d->tap_dx[t] : dxmin; dxmax = d->tap_dx[t] > dxmax ? d->tap_dx[t] : dxmax;
    }
    a.dymin = dymin; a.dxmin = dxmin;
    a.HH = (th - 1) * d->IS + (dymax - dymin) + 1;
    a.HW = (tw - 1) * d->IS + (dxmax - dxmin) + 1;
    while (imgs > 1 && imgs * a.HH * a.HW * 4 > 256 * MAXI) imgs >>= 1;   // stride-2 gathers on tiny maps: fewer images per tile
    a.imgs = imgs; a.iml = ilog2_exact(imgs);
    a.halo_px = imgs * a.HH * a.HW;
    CU_CHECK_ARG(a.halo_px * 4 <= 256 * MAXI, "cu_conv_gemm: halo of %d pixels too large", a.halo_px);
    const int igroups = cdiv(d->N, imgs);
    for (int t = 0; t < d->ntaps; ++t) {
        a.tap_off[t] = (d->tap_dy[t] - dymin) * a.HW + (d->tap_dx[t] - dxmin);
        a.tap_w[t] = d->tap_w[t];
        CU_CHECK_ARG(d->tap_w[t] >= 0, "cu_conv_gemm: negative weight tap index");
    }
    const int mod = d->dtype == CU_BF16 ? 16 : 32;   // plane stride == 2 (mod 16 entries / 32 floats): conflict-free fills
    a.XP = a.halo_px + ((2 - a.halo_px % mod) + mod) % mod;

    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int tiles = a.tiles_x * a.tiles_y * igroups;
    // column tile: the widest of {128, 96, 64, 32} that divides the work without waste
    int nb;
    if (d->CO % 128 == 0) nb = 4;
    else if (d->CO % 96 == 0) nb = 3;
    else if (d->CO % 64 == 0) nb = 2;
    else nb = 1;
    if (d->D0 != d->CO && d->D0 % (32 * nb) != 0) nb = (d->D0 % 64 == 0 && d->CO % 64 == 0) ? 2 : 1;
    if (d->dtype == CU_F32 && nb > 2) nb = (d->CO % 64 == 0) ? 2 : 1;   // keep the f32 weight tile within LDS
    a.WP = 32 * nb + 2;
#define CU_GO(T, NBv) return launch<T, 1, NBv>(a, tiles, st)
    if (d->dtype == CU_BF16) {
        switch (nb) { case 4: CU_GO(bf16_t, 4); case 3: CU_GO(bf16_t, 3); case 2: CU_GO(bf16_t, 2); default: CU_GO(bf16_t, 1); }
    } else {
        switch (nb) { case 2: CU_GO(float, 2); default: CU_GO(float, 1); }
    }
#undef CU_GO
}
