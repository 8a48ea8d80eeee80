// Lean gather-GEMM for the few-tap members of the convolution family (gfx950, bf16 production mode).
//
//   Out[m, n] (+)= bias[n] + sum_t sum_c  S[row_t(m), c] * W[wrow(t, group(n)) + n', c]
//
// m = loop pixel (image, py, px); row_t(m) = source pixel (py * IS + dy_t, px * IS + dx_t) (zero outside the image);
// the GEMM columns are `ngroups` groups of `gcols`: one group for plain outputs, four output parities for the
// one-pass forms (group g -> output pixel (2 py + (g >> 1), 2 px + (g & 1))).  Served by this kernel (dispatched inside
// cu_conv_gemm, same descriptor, reference layers as listed in igemm_conv.hip):
//   * ConvTranspose2d k2 s2 forward           1 tap,  4 parity groups, weight row block = parity
//   * its input gradient                      4 taps (dy, dx in {0, 1}), IS = 2
//   * input gradient of Conv2d 3x3 stride 2   4 gather taps, 4 parity groups, 9 of the 16 (tap, parity) pairs exist
//   * Conv2d 3x3 stride 2 forward             9 taps, IS = 2
//   * 1x1 convolutions (NHWC out)             1 tap
//
// Why a second kernel: on these layers the tile-generic igemm_conv_kernel issues ~37 vector instructions per MFMA
// (rocprofv3 PMC, profiles/r02_pmc_convT.txt: halo geometry, per-piece 64-bit addresses, predicated staging, parity
// epilogue) with one workgroup per CU: with loads, MFMAs and stores all disabled it still takes 70 % of its time.  Here:
//   * every operand byte reaches LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no commit pass; a
//     tap is just another row offset of the same per-lane source offsets, zero padding = an out-of-range offset;
//   * LDS images are [128 rows][128 bytes] (64 channels of a pixel / of a weight row), 16-byte piece p of row r stored at
//     slot p ^ ((r >> 1) & 7) (swizzle on the DMA source address and on the read): conflict-free ds_read_b128 fragments;
//   * the weights stay resident in LDS when the whole slice is small (thin layers), else they stream with the pixels;
//   * two LDS stages: the DMA of iteration i + 1 flies under the MFMAs of iteration i; <= 128 VGPRs, 2+ workgroups per CU.
#include "common.h"

namespace {

constexpr int PBM = 128;        // loop pixels per tile
constexpr int PBN = 128;        // GEMM columns per workgroup
constexpr int PKC = 64;         // channels per iteration (128-byte rows)
constexpr int PSTAGE = PBM * PKC * 2;       // bytes of one staged block (16 KiB)
constexpr int PMAXT = 9;

struct PcArgs {
    const void* src; const void* w; const float* bias; void* dst;
    unsigned src_bytes, w_bytes;
    int M, pwl, phl;                    // loop pixels; log2 of the loop grid's width / height
    int SH, SW, C, IS;
    int ntaps, tap_dy[PMAXT], tap_dx[PMAXT];
    int wrow[PMAXT * 4];                // first weight row of (tap, group), -1 = this pair does not exist
    int ngroups, gcols, CO;             // CO = GEMM columns = ngroups * gcols
    int OH, OW, OS, OY0, OX0, DC, accum;
    int kchunks;                        // ceil(C / 64)
    int mtiles, coltiles;
    int wres;                           // whole weight slice of the column tile resident in LDS
};

// block b (32 columns) of column tile ct -> (group, first column inside the group); cols_here = valid columns
__device__ __forceinline__ void col_block(const PcArgs& p, int ct, int b, int& group, int& c0) {
    if (p.gcols >= PBN) {
        const int tpg = (p.gcols + PBN - 1) / PBN;
        group = ct / tpg;
        c0 = (ct - group * tpg) * PBN + b * 32;
    } else {
        const int col = ct * PBN + b * 32;          // gcols in {32, 64}: a tile holds whole groups
        group = col / p.gcols;
        c0 = col - group * p.gcols;
    }
}

template <bool WRES>
__global__ __launch_bounds__(256, 2) void pconv_kernel(const PcArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int ct = blockIdx.y;
    const unsigned a_base = lds_addr(smem);                                  // 2 stages of pixels
    const unsigned w_base = a_base + 2 * PSTAGE;                             // weights: resident slice or 2 stages
    const i32x4 rs = make_rsrc(p.src, p.src_bytes);
    const i32x4 rw = make_rsrc(p.w, p.w_bytes);
    constexpr unsigned OOB = 0x7ffffff0u;
    const int niter = p.ntaps * p.kchunks;
    // Tap tables go to LDS through STATIC indices: a runtime index into the by-value kernel argument would force the
    // whole argument struct into scratch memory.  s_tab: [0, 9) dy, [9, 18) dx, [18, 54) first weight row of (tap, group)
    __shared__ int s_tab[PMAXT * 6];
#pragma unroll
    for (int i = 0; i < PMAXT; ++i)
        if (tid == i) { s_tab[i] = p.tap_dy[i]; s_tab[PMAXT + i] = p.tap_dx[i]; }
#pragma unroll
    for (int i = 0; i < PMAXT * 4; ++i)
        if (tid == 64 + i) s_tab[2 * PMAXT + i] = p.wrow[i];
    __syncthreads();
    const int* s_wrow = s_tab + 2 * PMAXT;

    // ---- this thread's 4 staging pieces of a block: piece i = s * 256 + tid -> row i >> 3, slot i & 7
    int prow[4], ppiece[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int i = s * 256 + tid;
        prow[s] = i >> 3;
        ppiece[s] = (i & 7) ^ ((prow[s] >> 1) & 7);          // source piece that lands in this slot
    }
    // ---- weight rows of this column tile: block b = row / 32
    int wgrp[4], wc0[4];
    unsigned long long benable = 0;                          // bit t * 4 + b: MFMAs of (tap, block) exist
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        col_block(p, ct, b, wgrp[b], wc0[b]);
        for (int t = 0; t < p.ntaps; ++t)
            if (wgrp[b] < p.ngroups && wc0[b] < p.gcols && s_wrow[t * 4 + wgrp[b]] >= 0) benable |= 1ull << (t * 4 + b);
    }
    auto issue_w = [&](int t, int kc, unsigned dst) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int row = prow[s], b = row >> 5;
            const int col = wc0[b] + (row & 31);
            const int k = kc * PKC + ppiece[s] * 8;
            const int wr = s_wrow[t * 4 + (wgrp[b] < p.ngroups ? wgrp[b] : 0)];
            const bool ok = ((benable >> (t * 4 + b)) & 1u) && col < p.gcols && k < p.C;
            const unsigned off = ok ? (unsigned)(((wr + col) * p.C + k) * 2) : OOB;
            dma16(rw, off, dst + (unsigned)(s * 4096 + wave * 1024));
        }
    };
    if constexpr (WRES) {
        for (int it = 0; it < niter; ++it) issue_w(it / p.kchunks, it % p.kchunks, w_base + (unsigned)it * PSTAGE);
    }

    // ---- fragment read offsets (bytes inside a block): row = 32 * wave + r (pixels) / 32 * b + r (weights)
    const int arow = wave * 32 + r;
    const int asw = (arow >> 1) & 7;
    int wsw[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) wsw[b] = ((b * 32 + r) >> 1) & 7;

    const int pw = 1 << p.pwl, ph = 1 << p.phl;
    for (int tile = blockIdx.x; tile < p.mtiles; tile += gridDim.x) {
        // ---- per-tile geometry of this thread's 4 staging rows
        int sy0[4], sx0[4], nb[4];
        unsigned rvalid = 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = tile * PBM + prow[s];
            const int px = m & (pw - 1), py = (m >> p.pwl) & (ph - 1), n = m >> (p.pwl + p.phl);
            sy0[s] = py * p.IS; sx0[s] = px * p.IS; nb[s] = n * p.SH;
            rvalid |= (m < p.M ? 1u : 0u) << s;
        }
        auto issue_a = [&](int t, int kc, unsigned dst) {
            const int dy = s_tab[t], dx = s_tab[PMAXT + t];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int sy = sy0[s] + dy, sx = sx0[s] + dx;
                const int k = kc * PKC + ppiece[s] * 8;
                const bool ok = ((rvalid >> s) & 1u) && sy >= 0 && sy < p.SH && sx >= 0 && sx < p.SW && k < p.C;
                const unsigned off = ok ? (unsigned)((((nb[s] + sy) * p.SW + sx) * p.C + k) * 2) : OOB;
                dma16(rs, off, dst + (unsigned)(s * 4096 + wave * 1024));
            }
        };
        f32x16 acc[4];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;

        __syncthreads();                 // the previous tile's fragment reads are done: stage 0 may be refilled
        issue_a(0, 0, a_base);
        if constexpr (!WRES) issue_w(0, 0, w_base);
        for (int it = 0; it < niter; ++it) {
            const int st = it & 1;
            const int t = it / p.kchunks;
            dma_wait();                  // this wave's share of iteration `it` (and of the resident weights) has landed
            __syncthreads();             // ... everybody's has; the other stage is no longer being read
            if (it + 1 < niter) {
                const int t1 = (it + 1) / p.kchunks, k1 = (it + 1) - t1 * p.kchunks;
                issue_a(t1, k1, a_base + (unsigned)((st ^ 1) * PSTAGE));
                if constexpr (!WRES) issue_w(t1, k1, w_base + (unsigned)((st ^ 1) * PSTAGE));
            }
            const unsigned char* A = smem + st * PSTAGE;
            const unsigned char* W = smem + 2 * PSTAGE + (WRES ? it : st) * PSTAGE;
            const unsigned en = (unsigned)(benable >> (t * 4)) & 15u;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(A + arow * 128 + (((2 * kk + h) ^ asw) << 4));
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if ((en >> b) & 1u) {            // uniform
                        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(W + (b * 32 + r) * 128 + (((2 * kk + h) ^ wsw[b]) << 4));
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[b], 0, 0, 0);
                    }
                }
            }
        }

        // ---- epilogue.  D[row = column][col = pixel]: lane -> pixel (lane & 31); register i -> column
        //      (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5); two v_permlane32_swap give every lane 16 consecutive channels.
        const int m = tile * PBM + wave * 32 + r;
        const bool mvalid = m < p.M;
        const int px = m & (pw - 1), py = (m >> p.pwl) & (ph - 1), n = m >> (p.pwl + p.phl);
        const long obase = (((long)n * p.OH + py * p.OS + p.OY0) * p.OW + px * p.OS + p.OX0) * p.DC;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (!((benable >> b) & 0x111111111ull)) continue;             // uniform: no tap feeds this block
            unsigned q[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[b][4 * g + e];
                if (p.bias) {
                    const int cb = wgrp[b] * p.gcols + wc0[b] + 8 * g + 4 * h;
                    if (wc0[b] + 8 * g + 4 * h < p.gcols) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + cb);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += bv[e];
                    }
                }
                q[g][0] = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                q[g][1] = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            }
#pragma unroll
            for (int w2 = 0; w2 < 2; ++w2) {
                auto r02 = __builtin_amdgcn_permlane32_swap(q[0][w2], q[2][w2], false, false);
                q[0][w2] = r02[0]; q[2][w2] = r02[1];
                auto r13 = __builtin_amdgcn_permlane32_swap(q[1][w2], q[3][w2], false, false);
                q[1][w2] = r13[0]; q[3][w2] = r13[1];
            }
            const int c = wc0[b] + 16 * h;                               // this lane's 16 channels inside the group
            if (mvalid && c < p.gcols) {
                const long poff = p.ngroups > 1 ? ((long)(wgrp[b] >> 1) * p.OW + (wgrp[b] & 1)) * p.DC : 0;
                bf16_t* o = reinterpret_cast<bf16_t*>(p.dst) + obase + poff + c;
                u32x4 lo = u32x4{q[0][0], q[0][1], q[2][0], q[2][1]};
                u32x4 hi = u32x4{q[1][0], q[1][1], q[3][0], q[3][1]};
                if (p.accum) {
                    const u32x4 ol = *reinterpret_cast<const u32x4*>(o), oh = *reinterpret_cast<const u32x4*>(o + 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a0 = __uint_as_float(lo[e] << 16) + __uint_as_float(ol[e] << 16);
                        const float a1 = __uint_as_float(lo[e] & 0xffff0000u) + __uint_as_float(ol[e] & 0xffff0000u);
                        lo[e] = (unsigned)f32_to_bf16(a0) | ((unsigned)f32_to_bf16(a1) << 16);
                        const float b0 = __uint_as_float(hi[e] << 16) + __uint_as_float(oh[e] << 16);
                        const float b1 = __uint_as_float(hi[e] & 0xffff0000u) + __uint_as_float(oh[e] & 0xffff0000u);
                        hi[e] = (unsigned)f32_to_bf16(b0) | ((unsigned)f32_to_bf16(b1) << 16);
                    }
                }
                *reinterpret_cast<u32x4*>(o) = lo;
                if (c + 8 < p.gcols) *reinterpret_cast<u32x4*>(o + 8) = hi;
            }
        }
    }
}

}  // namespace

// Called by cu_conv_gemm before its own tiling: returns 1 if the launch was taken, 0 if the shape is not this kernel's,
// < 0 on error.
int cu_pconv_try(const cu_conv_desc* d, const void* src0, const void* w, const float* bias, void* dst0, void* stream) {
    if (d->dtype != CU_BF16 || d->C1 != 0 || d->out_nchw_f32 || d->D0 != d->CO) return 0;
    if (d->slope0 != 1.0f) return 0;
    const bool parity = d->par_co > 0;
    const bool few = d->ntaps <= 4, s2fwd = d->ntaps == 9 && d->IS == 2 && !parity;
    if (!few && !s2fwd) return 0;
    if (d->C0 % 32 != 0 || d->C0 < 64) return 0;
    const int gcols = parity ? d->par_co : d->CO;
    if (gcols % 16 != 0 || (gcols < PBN && gcols != 32 && gcols != 64)) return 0;
    if (d->DC0 % 8 != 0) return 0;
    if (ilog2_exact(d->PW) < 0 || ilog2_exact(d->PH) < 0) return 0;
    if (parity && (d->OS != 2 || d->OY0 != 0 || d->OX0 != 0)) return 0;
    const size_t sb = (size_t)d->N * d->SH * d->SW * d->C0 * 2;
    int wtaps = 0;                                   // weight taps addressed: the weight tensor has at least that many
    for (int t = 0; t < d->ntaps; ++t) {
        if (parity && d->par_taps) { for (int g = 0; g < 4; ++g) wtaps = d->par_tap_w[t * 4 + g] + 1 > wtaps ? d->par_tap_w[t * 4 + g] + 1 : wtaps; }
        else wtaps = (parity ? 4 * (d->tap_w[t] + 1) : d->tap_w[t] + 1) > wtaps ? (parity ? 4 * (d->tap_w[t] + 1) : d->tap_w[t] + 1) : wtaps;
    }
    const int wrows_total = wtaps * gcols;
    const size_t wb = (size_t)wrows_total * d->C0 * 2;
    if (sb >= 0x7fff0000ull || wb >= 0x7fff0000ull) return 0;
    const long M = (long)d->N * d->PH * d->PW;
    if (M < PBM || M >= (1l << 30)) return 0;

    PcArgs a;
    memset(&a, 0, sizeof(a));
    a.src = src0; a.w = w; a.bias = bias; a.dst = dst0;
    a.src_bytes = (unsigned)sb; a.w_bytes = (unsigned)wb;
    a.M = (int)M; a.pwl = ilog2_exact(d->PW); a.phl = ilog2_exact(d->PH);
    a.SH = d->SH; a.SW = d->SW; a.C = d->C0; a.IS = d->IS;
    a.ntaps = d->ntaps;
    a.ngroups = parity ? 4 : 1; a.gcols = gcols; a.CO = a.ngroups * gcols;
    for (int t = 0; t < d->ntaps; ++t) {
        a.tap_dy[t] = d->tap_dy[t]; a.tap_dx[t] = d->tap_dx[t];
        for (int g = 0; g < 4; ++g) {
            int wr = -1;
            if (g < a.ngroups) {
                if (parity && d->par_taps) { const int tw = d->par_tap_w[t * 4 + g]; wr = tw >= 0 ? tw * gcols : -1; }
                else if (parity) wr = (d->tap_w[t] * 4 + g) * gcols;          // transposed conv: W viewed as [1][4 * co][ci]
                else wr = d->tap_w[t] * gcols;
            }
            a.wrow[t * 4 + g] = wr;
        }
    }
    a.OH = d->OH; a.OW = d->OW; a.OS = d->OS; a.OY0 = d->OY0; a.OX0 = d->OX0; a.DC = d->DC0; a.accum = d->accum0;
    a.kchunks = cdiv(d->C0, PKC);
    a.mtiles = cdiv((int)M, PBM);
    a.coltiles = gcols >= PBN ? a.ngroups * cdiv(gcols, PBN) : cdiv(a.CO, PBN);
    const int niter = a.ntaps * a.kchunks;
    a.wres = niter * PSTAGE <= 64 * 1024;
    const size_t lds = (size_t)2 * PSTAGE + (size_t)(a.wres ? niter : 2) * PSTAGE;
    int grid_x = a.mtiles;
    const int cap = 256 * 4 / (a.coltiles < 4 ? a.coltiles : 4);
    if (grid_x > cap && cap > 0) grid_x = cap;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    auto k = a.wres ? pconv_kernel<true> : pconv_kernel<false>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        CU_CHECK_ARG(e == hipSuccess, "cu_conv_gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k, dim3(grid_x, a.coltiles), dim3(256), lds, st, a);
    CU_LAUNCH_CHECK();
    return 1;
}
