// Weight gradient of the 2x2 stride-2 transposed convolution as ONE pixel-major GEMM (gfx950, bf16; round 3, VERDICT r2
// item 2a).  Reference: autograd of nn.ConvTranspose2d(kernel 2, stride 2) in models/nnUnet/layers.py:83-109,415-417.
//
//   dW[t][co][ci] = sum_p  dU[n][2y + ty][2x + tx][co] * S[n][y][x][ci],      t = ty * 2 + tx,  p = (n, y, x)
//
// Every tap reads a different parity plane of dU and all four share S: with v = t * CO + co this is the plain product
// dW'[v][ci] = sum_p A[p][v] B[p][ci], a "TN" GEMM (both operands pixel-major, the reduction index slowest) with M = 4 CO
// rows.  Row p of A is two contiguous runs of 2 CO elements (image rows 2y and 2y + 1, pixels 2x and 2x + 1 adjacent), so
// the gather is a per-lane LDS-DMA source address and nothing is strided in LDS -- the generic kernel ran these layers
// with 64-pixel tiles, 4 k-steps per barrier and 128-byte row strides at 5-12 % of the MFMA peak.
//
// Structure (the producer / consumer form of igemm_wgrad.hip with a GEMM-shaped register block):
//   * workgroup = 8 waves: waves 0..3 compute, one per SIMD, each a (32 MT) x (32 NT) block of the (64 MT) x (64 NT) tile
//     -- MT x NT accumulators, MT + NT fragment reads per MT * NT MFMAs; waves 4..7 only issue LDS-DMA;
//   * a ring of THREE K-tiles of KT pixels: the copy of tile i+2 is issued when tile i starts, so one whole tile is always in
//     flight beside the one being waited for (the chip's per-CU fetch rate is bytes in flight / latency); the producers'
//     counted s_waitcnt vmcnt(IT) waits for tile i+1 only (every tile issues the same IT instructions per wave: rows
//     past the end of K or of the matrix get out-of-range offsets = zero fill); ONE barrier per tile;
//   * LDS images [32-column plane][pixel][64 B], read k-major with ds_read_b64_tr_b16 (as igemm_wgrad.hip);
//   * split-K over grid.y; every split STORES its partial [M][N] tile (plain layout) into its own slab -- no atomics --
//     and cu_grad_unprep_parts adds the slabs in slab order (bit-identical run to run).
// Algorithmic intensity: 2 * 64MT * 64NT / (2 (64MT + 64NT)) FLOP per staged byte = 85 for the 256 x 128 tile; at the
// ~10 B/clk a CU can fetch that is ~0.5 PFLOP/s chip-wide: these launches are fetch-bound, not MFMA-bound, by shape.
#include "common.h"

namespace {

struct GtArgs {
    const void* a; const void* b; float* parts;
    unsigned a_bytes, b_bytes;
    int K;                          // low-resolution pixels N * h * w
    int h, w;                       // low-resolution image
    unsigned mg_hw, mg_w;           // ceil(2^32 / (h w)), ceil(2^32 / w)
    int CO, CI;                     // M = 4 CO rows, CI columns
    int tiles_m, tiles_n, ktiles, splits;
    size_t part_stride;             // floats per slab = 4 CO * CI
};

__device__ __forceinline__ bf16x4 tr_read64(unsigned lds_byte_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(size_t)lds_byte_addr);
}

template <int MT, int NT, int KT>
__global__ __launch_bounds__(512) void gemm_tn_kernel(const GtArgs p) {
    constexpr int BM = 64 * MT, BN = 64 * NT;
    constexpr int A_PIECES = BM / 8 * KT, B_PIECES = BN / 8 * KT;         // 16-byte pieces per K-tile
    constexpr int IT = (A_PIECES + B_PIECES) / 256;                       // DMA instructions per producer wave and tile
    constexpr int IMG_B = (A_PIECES + B_PIECES) * 16;
    static_assert((A_PIECES + B_PIECES) % 256 == 0 && 3 * IMG_B <= 160 * 1024, "gemm_tn: bad instance");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool producer = wave >= 4;
    const unsigned lds0 = lds_addr(smem);

    // XCD-aware placement: all (m, n) tiles of a K split on one XCD (they share the split's A / B panels through its L2)
    int bxi = blockIdx.x, byi = blockIdx.y;
    if ((p.splits & 7) == 0) {
        const int gx = gridDim.x, hid = blockIdx.x + gx * blockIdx.y;
        const int xcd = hid & 7, slot = hid >> 3;
        bxi = slot % gx;
        byi = xcd + 8 * (slot / gx);
    }
    const int tm = bxi / p.tiles_n, tn = bxi - tm * p.tiles_n;
    const int m_base = tm * BM, n_base = tn * BN;
    const int M = 4 * p.CO;

    const i32x4 ra = make_rsrc(p.a, p.a_bytes), rb = make_rsrc(p.b, p.b_bytes);
    constexpr unsigned OOB = 0x7ffffff0u;
    const int ptid = threadIdx.x & 255;

    auto issue = [&](int kt, int slot) {      // producers: K-tile kt (or IT out-of-range instructions) -> ring slot
        const bool live = kt < p.ktiles;
        const int k0 = kt * KT;
#pragma unroll
        for (int j = 0; j < IT; ++j) {
            const int it = ptid + j * 256;                    // piece index: A pieces first ([plane][pixel][4]), then B
            const bool isb = __builtin_amdgcn_readfirstlane(it) >= A_PIECES;     // 256 | A_PIECES: wave-uniform
            const int q = isb ? it - A_PIECES : it;
            const int plane = q / (KT * 4), rem = q - plane * (KT * 4);
            const int px = rem >> 2, s = rem & 3;
            const int pk = k0 + px;
            unsigned off = OOB;
            if (isb) {
                const int col = n_base + plane * 32 + s * 8;
                if (live && pk < p.K && col < p.CI) off = (unsigned)(pk * p.CI + col) * 2u;
            } else {
                const int col = m_base + plane * 32 + s * 8;              // virtual column v = t * CO + co
                if (live && pk < p.K && col < M) {
                    const int n = __umulhi((unsigned)pk, p.mg_hw), r2 = pk - n * (p.h * p.w);
                    const int y = __umulhi((unsigned)r2, p.mg_w), x = r2 - y * p.w;
                    const int zy = col >= 2 * p.CO ? 1 : 0, c2 = col - zy * 2 * p.CO;
                    off = (unsigned)((((n * 2 * p.h + 2 * y + zy) * 2 * p.w + 2 * x)) * p.CO + c2) * 2u;
                }
            }
            const unsigned dst = lds0 + (unsigned)(slot * IMG_B + j * 4096 + (wave - 4) * 1024);
            if (isb) dma16(rb, off, dst);
            else dma16(ra, off, dst);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    // fragment addressing: lane l = 16 g + 4 q + pp supplies row q of its group's 4 x 16 transpose block
    const int g = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
    const int hh = g >> 1, chalf = g & 1;
    const int wm = (wave & 3) >> 1, wn = wave & 1;
    const unsigned lane_off = (unsigned)((8 * hh + q4) * 64 + (16 * chalf + 4 * pp) * 2);
    const unsigned a_plane0 = (unsigned)(wm * MT) * (KT * 64), b_plane0 = (unsigned)(A_PIECES * 16) + (unsigned)(wn * NT) * (KT * 64);

    int kt = byi;
    if (producer) {
        issue(kt, 0);
        issue(kt + p.splits, 1);
    }
    for (int i = 0; kt < p.ktiles; ++i, kt += p.splits) {
        if (producer) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IT) : "memory");      // tile i landed; tile i+1 may still fly
        __syncthreads();      // tile i is complete, and nobody reads the slot of tile i-1 any more
        if (producer) {
            issue(kt + 2 * p.splits, (i + 2) % 3);
            continue;
        }
        const unsigned img = lds0 + (unsigned)((i % 3) * IMG_B);
#pragma unroll 2
        for (int ks = 0; ks < KT / 16; ++ks) {
            bf16x8 af[MT], bfr[NT];
            const unsigned row = img + (unsigned)(ks * 16 * 64) + lane_off;
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                const bf16x4 lo = tr_read64(row + a_plane0 + a * (KT * 64)), hi = tr_read64(row + a_plane0 + a * (KT * 64) + 4 * 64);
                af[a] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                const bf16x4 lo = tr_read64(row + b_plane0 + b * (KT * 64)), hi = tr_read64(row + b_plane0 + b * (KT * 64) + 4 * 64);
                bfr[b] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
    }
    if (producer) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing out-of-range instructions target live LDS
        return;
    }
    // ---- this split's partial tile, plain [M][CI] layout
    float* slab = p.parts + (size_t)byi * p.part_stride;
    const int r = lane & 31, h2 = lane >> 5;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int col = n_base + (wn * NT + b) * 32 + r;
            if (col >= p.CI) continue;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = m_base + (wm * MT + a) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h2;
                if (m < M) slab[(size_t)m * p.CI + col] = acc[a][b][i];
            }
        }
}

template <int MT, int NT, int KT>
int launch_gt(GtArgs& a, size_t parts_floats, int* nparts, hipStream_t st) {
    constexpr int BM = 64 * MT, BN = 64 * NT;
    constexpr size_t LDS = (size_t)3 * (BM / 8 + BN / 8) * KT * 16;
    a.tiles_m = cdiv(4 * a.CO, BM); a.tiles_n = cdiv(a.CI, BN);
    a.ktiles = cdiv(a.K, KT);
    const int tiles = a.tiles_m * a.tiles_n;
    int splits = cdiv(256, tiles);
    if (splits > a.ktiles) splits = a.ktiles;
    const size_t plain = a.part_stride;
    CU_CHECK_ARG(parts_floats >= 2 * plain, "cu_conv_wgrad_parts: workspace of %zu floats; this shape needs >= %zu", parts_floats, 2 * plain);
    const size_t cap = parts_floats / plain - 1;           // + the plain sum cu_grad_unprep_parts forms at the end
    if ((size_t)splits > cap) splits = (int)cap;
    a.splits = splits;
    auto k = gemm_tn_kernel<MT, NT, KT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    CU_CHECK_ARG(e == hipSuccess, "cu_conv_wgrad_parts: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k, dim3(tiles, splits), dim3(512), LDS, st, a);
    CU_LAUNCH_CHECK();
    *nparts = splits;
    return 1;
}

}  // namespace

// Called by cu_conv_wgrad_parts.  1 = launched (slabs in the plain [4][CO][CI] layout, *nparts of them), 0 = not this
// kernel's shape, < 0 = error.
int cu_gemm_tn_try(const cu_wgrad_desc* d, const void* src0, const void* z, float* parts, size_t parts_floats, int* nparts,
                   void* stream) {
    if (d->dtype != CU_BF16 || d->ntaps != 4 || d->IS != 1 || d->ZS != 2 || d->C1 || d->slope0 != 1.0f) return 0;
    if (d->ZH != 2 * d->PH || d->ZW != 2 * d->PW || d->SH != d->PH || d->SW != d->PW || d->CO != d->ZC || d->CO % 8 || d->C0 % 8) return 0;
    for (int t = 0; t < 4; ++t)
        if (d->tap_dy[t] || d->tap_dx[t] || d->tap_zy[t] != (t >> 1) || d->tap_zx[t] != (t & 1) || d->tap_w[t] != t) return 0;
    const size_t ab = (size_t)d->N * d->ZH * d->ZW * d->ZC * 2, bb = (size_t)d->N * d->SH * d->SW * d->C0 * 2;
    if (ab >= 0x7fff0000ull || bb >= 0x7fff0000ull) return 0;
    GtArgs a;
    memset(&a, 0, sizeof(a));
    a.a = z; a.b = src0; a.parts = parts; a.a_bytes = (unsigned)ab; a.b_bytes = (unsigned)bb;
    a.K = d->N * d->PH * d->PW; a.h = d->PH; a.w = d->PW; a.CO = d->CO; a.CI = d->C0;
    a.mg_hw = (unsigned)((0x100000000ull + (unsigned)(a.h * a.w) - 1) / (unsigned)(a.h * a.w));
    a.mg_w = (unsigned)((0x100000000ull + (unsigned)a.w - 1) / (unsigned)a.w);
    a.part_stride = (size_t)4 * d->CO * d->C0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (4 * d->CO <= 128) return launch_gt<2, 1, 128>(a, parts_floats, nparts, st);      // 128 x 64 tile, 128-pixel K-tiles
    return launch_gt<4, 2, 64>(a, parts_floats, nparts, st);                             // 256 x 128 tile, 64-pixel K-tiles
}
