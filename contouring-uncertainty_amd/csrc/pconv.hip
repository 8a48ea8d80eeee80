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
//   * a ring of LDS stages with counted waits: three blocks are always in flight, across taps, chunks and tiles.
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

template <bool WRES, int PRING>
__global__ __launch_bounds__(256, 2) void pconv_kernel(const PcArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int ct = blockIdx.y;
    const unsigned a_base = lds_addr(smem);                                  // ring of pixel blocks
    const unsigned w_base = a_base + PRING * PSTAGE;                         // weights: resident slice or their own ring
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
    // resident weights are PACKED: only the (iteration, block) pairs that exist get a 4-KiB slot (the stride-2 input gradient
    // uses 9 of its 16 (gather tap, parity) pairs), which is what lets two workgroups share a CU on the thin layers
    __shared__ unsigned char s_wslot[PMAXT * 8 * 4];
    if constexpr (WRES) {
        if (tid == 0) {
            int cnt = 0;
            for (int i = 0; i < niter; ++i)
                for (int b = 0; b < 4; ++b)
                    s_wslot[i * 4 + b] = (unsigned char)(((benable >> ((i / p.kchunks) * 4 + b)) & 1ull) ? cnt++ : 0);
        }
        __syncthreads();
    }
    auto issue_w = [&](int t, int kc, unsigned dst) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {                    // slot s of this thread = block s of the 128 weight rows
            if (WRES && !((benable >> (t * 4 + s)) & 1ull)) continue;       // uniform
            const int row = prow[s], b = row >> 5;
            const int col = wc0[b] + (row & 31);
            const int k = kc * PKC + ppiece[s] * 8;
            const int wr = s_wrow[t * 4 + (wgrp[b] < p.ngroups ? wgrp[b] : 0)];
            const bool ok = ((benable >> (t * 4 + b)) & 1u) && col < p.gcols && k < p.C;
            const unsigned off = ok ? (unsigned)(((wr + col) * p.C + k) * 2) : OOB;
            const unsigned slot = WRES ? (unsigned)s_wslot[(t * p.kchunks + kc) * 4 + s] : (unsigned)s;
            dma16(rw, off, dst + slot * 4096u + (unsigned)(wave * 1024));
        }
    };
    if constexpr (WRES) {
        for (int it = 0; it < niter; ++it) issue_w(it / p.kchunks, it % p.kchunks, w_base);
    }

    // ---- fragment read offsets (bytes inside a block): row = 32 * wave + r (pixels) / 32 * b + r (weights)
    const int arow = wave * 32 + r;
    const int asw = (arow >> 1) & 7;
    int wsw[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) wsw[b] = ((b * 32 + r) >> 1) & 7;

    // ---- the workgroup's work as ONE flat sequence of iterations (tile k, tap t, chunk kc), staged through a ring of
    //      PRING LDS stages: the DMA of iteration j + PRING - 1 is issued when iteration j starts, so PRING - 1 blocks are
    //      always in flight -- across taps, chunks AND tile boundaries (the epilogue of a tile runs under the next tile's
    //      loads).  Waits are COUNTED: vmcnt(D * younger) leaves the younger iterations' DMAs in flight.  (First version:
    //      two stages and s_waitcnt vmcnt(0) per iteration = one exposed L2/HBM round trip per 16 KiB block; the stride-2
    //      input gradient at 128^2 took 347 us against 306 us for four launches of the tile-generic kernel.)
    constexpr int D = WRES ? 4 : 8;                  // DMA wave-instructions per iteration and wave
    const int pw = 1 << p.pwl, ph = 1 << p.phl;
    const int my_tiles = blockIdx.x < p.mtiles ? (p.mtiles - 1 - blockIdx.x) / gridDim.x + 1 : 0;
    const int T = my_tiles * niter;
    // issue-side cursor
    int iss = 0, iss_it = 0, iss_tile = blockIdx.x, iss_st = 0;
    int sy0[4], sx0[4], nb[4];
    unsigned rvalid = 0;
    auto issue_geo = [&]() {
        rvalid = 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = iss_tile * PBM + prow[s];
            const int px = m & (pw - 1), py = (m >> p.pwl) & (ph - 1), n = m >> (p.pwl + p.phl);
            sy0[s] = py * p.IS; sx0[s] = px * p.IS; nb[s] = n * p.SH;
            rvalid |= (m < p.M ? 1u : 0u) << s;
        }
    };
    auto issue_next = [&]() {
        const int t = iss_it / p.kchunks, kc = iss_it - t * p.kchunks;
        const unsigned stg = (unsigned)iss_st * PSTAGE;
        iss_st = iss_st + 1 == PRING ? 0 : iss_st + 1;
        const int dy = s_tab[t], dx = s_tab[PMAXT + t];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int sy = sy0[s] + dy, sx = sx0[s] + dx;
            const int k = kc * PKC + ppiece[s] * 8;
            const bool ok = ((rvalid >> s) & 1u) && sy >= 0 && sy < p.SH && sx >= 0 && sx < p.SW && k < p.C;
            const unsigned off = ok ? (unsigned)((((nb[s] + sy) * p.SW + sx) * p.C + k) * 2) : OOB;
            dma16(rs, off, a_base + stg + (unsigned)(s * 4096 + wave * 1024));
        }
        if constexpr (!WRES) issue_w(t, kc, w_base + stg);
        ++iss;
        if (++iss_it == niter) { iss_it = 0; iss_tile += gridDim.x; issue_geo(); }
    };
    if (T > 0) issue_geo();
    for (int k = 0; k < PRING - 1 && iss < T; ++k) issue_next();

    f32x16 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
    u32x4 old_lo[4], old_hi[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) old_lo[b] = old_hi[b] = u32x4{0u, 0u, 0u, 0u};
    int tile = blockIdx.x, it = 0, st = 0;
    for (int j = 0; j < T; ++j) {
        const int younger = iss - j - 1;          // iterations issued after j: 0 .. PRING - 2
        if (PRING > 3 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * D) : "memory");
        else if (PRING > 2 && younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();             // iteration j has landed for every wave; stage (j - 1) % PRING is no longer being read
        if (iss < T) issue_next();
        if (it == 0 && p.accum) {
            // accumulate: the old destination values are fetched NOW, under the tile's MFMAs (fetched in the epilogue they
            // cost one exposed HBM round trip per tile)
            const int m = tile * PBM + wave * 32 + r;
            const int px = m & (pw - 1), py = (m >> p.pwl) & (ph - 1), n = m >> (p.pwl + p.phl);
            const long obase = (((long)n * p.OH + py * p.OS + p.OY0) * p.OW + px * p.OS + p.OX0) * p.DC;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int c = wc0[b] + 16 * h;
                const bool ok = ((benable >> b) & 0x111111111ull) && m < p.M && c < p.gcols;
                const long poff = p.ngroups > 1 ? ((long)(wgrp[b] >> 1) * p.OW + (wgrp[b] & 1)) * p.DC : 0;
                const bf16_t* o = reinterpret_cast<const bf16_t*>(p.dst) + (ok ? obase + poff + c : 0);
                old_lo[b] = *reinterpret_cast<const u32x4*>(o);
                old_hi[b] = *reinterpret_cast<const u32x4*>(o + 8);
            }
        }
        {
            const int t = it / p.kchunks;
            const unsigned char* A = smem + st * PSTAGE;
            const unsigned char* W = smem + PRING * PSTAGE + (WRES ? 0 : st) * PSTAGE;
            const unsigned en = (unsigned)(benable >> (t * 4)) & 15u;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(A + arow * 128 + (((2 * kk + h) ^ asw) << 4));
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if ((en >> b) & 1u) {            // uniform
                        const int wb = WRES ? (int)s_wslot[it * 4 + b] : b;
                        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(W + (wb * 32 + r) * 128 + (((2 * kk + h) ^ wsw[b]) << 4));
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[b], 0, 0, 0);
                    }
                }
            }
        }
        st = st + 1 == PRING ? 0 : st + 1;
        if (++it < niter) continue;
        it = 0;

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
                    const u32x4 ol = old_lo[b], oh = old_hi[b];
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
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
        }
        tile += gridDim.x;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 256 x 256 tile form for the MID levels (16^2 ... 64^2 maps, 128 ... 480 channels).  There the kernel above is bound by
// its LDS-DMA staging, not by HBM or MFMA: a 128 x 128 tile stages (128 + 128) rows per 128 x 128 x 64 MACs, and the
// stride-2 input gradient at 16^2 moves 1 GB through the DMA path in 156 us (6.5 TB/s, the chip's LDS-DMA ceiling,
// MI355X_MICROARCH.md).  A 256 x 256 tile stages (256 + 256) rows per four times the MACs: half the bytes per FLOP.
// Eight waves: wave = (pixel group of 64 rows, column half of 128) -> 2 x 4 MFMA blocks (128 accumulator registers), two
// A-fragment and four W-fragment reads per eight MFMAs.  Two stages of 64 KiB (A 32 KiB | W 32 KiB), one workgroup per
// CU, the same flat iteration ring across taps, chunks and tiles.  Few-tap forms only (<= 4 taps), weights streamed.
constexpr int P2M = 256, P2N = 256, P2STAGE = (P2M + P2N) * PKC * 2;      // 64 KiB

__device__ __forceinline__ void col_block2(const PcArgs& p, int ct, int b, int& group, int& c0) {
    if (p.gcols >= P2N) {
        const int tpg = (p.gcols + P2N - 1) / P2N;
        group = ct / tpg;
        c0 = (ct - group * tpg) * P2N + b * 32;
    } else {
        const int col = ct * P2N + b * 32;          // gcols in {32, 64, 128}: a tile holds whole groups
        group = col / p.gcols;
        c0 = col - group * p.gcols;
    }
}

__global__ __launch_bounds__(512) void pconv2_kernel(const PcArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wr = wave & 3, wcol = wave >> 2;            // pixel group (64 rows), column half (4 blocks)
    const int ct = blockIdx.y;
    const unsigned a_base = lds_addr(smem);
    const i32x4 rs = make_rsrc(p.src, p.src_bytes);
    const i32x4 rw = make_rsrc(p.w, p.w_bytes);
    constexpr unsigned OOB = 0x7ffffff0u;
    const int niter = p.ntaps * p.kchunks;
    __shared__ int s_tab[4 * 2 + 4 * 4];                  // dy[4], dx[4], wrow[tap][group]
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (tid == i) { s_tab[i] = p.tap_dy[i]; s_tab[4 + i] = p.tap_dx[i]; }
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (tid == 64 + i) s_tab[8 + i] = p.wrow[i];
    __syncthreads();
    const int* s_wrow = s_tab + 8;

    // ---- staging pieces of a stage half (A or W): piece i = s * 512 + tid -> row i >> 3 (0..255), slot i & 7
    int prow[4], ppiece[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int i = s * 512 + tid;
        prow[s] = i >> 3;
        ppiece[s] = (i & 7) ^ ((prow[s] >> 1) & 7);
    }
    int wgrp[8], wc0[8];
    unsigned benable = 0;                                 // bit t * 8 + b: MFMAs of (tap, column block) exist
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        col_block2(p, ct, b, wgrp[b], wc0[b]);
        for (int t = 0; t < p.ntaps; ++t)
            if (wgrp[b] < p.ngroups && wc0[b] < p.gcols && s_wrow[t * 4 + wgrp[b]] >= 0) benable |= 1u << (t * 8 + b);
    }
    auto issue_w = [&](int t, int kc, unsigned dst) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int row = prow[s], b = row >> 5;
            int grp = 0, c0 = 0;
#pragma unroll
            for (int q = 0; q < 8; ++q) { grp = b == q ? wgrp[q] : grp; c0 = b == q ? wc0[q] : c0; }
            const int col = c0 + (row & 31);
            const int k = kc * PKC + ppiece[s] * 8;
            const int wrw = s_wrow[t * 4 + (grp < p.ngroups ? grp : 0)];
            const bool ok = ((benable >> (t * 8 + b)) & 1u) && col < p.gcols && k < p.C;
            const unsigned off = ok ? (unsigned)(((wrw + col) * p.C + k) * 2) : OOB;
            dma16(rw, off, dst + (unsigned)(s * 8192 + wave * 1024));
        }
    };

    int arow[2], asw[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) { arow[a] = wr * 64 + a * 32 + r; asw[a] = (arow[a] >> 1) & 7; }
    int wsw[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) wsw[b] = (((wcol * 4 + b) * 32 + r) >> 1) & 7;

    constexpr int D = 8;                                  // DMA wave-instructions per iteration and wave
    const int pw = 1 << p.pwl, ph = 1 << p.phl;
    const int my_tiles = (int)blockIdx.x < p.mtiles ? (p.mtiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int T = my_tiles * niter;
    int iss = 0, iss_it = 0, iss_tile = blockIdx.x, iss_st = 0;
    int sy0[4], sx0[4], nb[4];
    unsigned rvalid = 0;
    auto issue_geo = [&]() {
        rvalid = 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int m = iss_tile * P2M + prow[s];
            const int px = m & (pw - 1), py = (m >> p.pwl) & (ph - 1), n = m >> (p.pwl + p.phl);
            sy0[s] = py * p.IS; sx0[s] = px * p.IS; nb[s] = n * p.SH;
            rvalid |= (m < p.M ? 1u : 0u) << s;
        }
    };
    auto issue_next = [&]() {
        const int t = iss_it / p.kchunks, kc = iss_it - t * p.kchunks;
        const unsigned stg = (unsigned)iss_st * P2STAGE;
        iss_st ^= 1;
        const int dy = s_tab[t], dx = s_tab[4 + t];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int sy = sy0[s] + dy, sx = sx0[s] + dx;
            const int k = kc * PKC + ppiece[s] * 8;
            const bool ok = ((rvalid >> s) & 1u) && sy >= 0 && sy < p.SH && sx >= 0 && sx < p.SW && k < p.C;
            const unsigned off = ok ? (unsigned)((((nb[s] + sy) * p.SW + sx) * p.C + k) * 2) : OOB;
            dma16(rs, off, a_base + stg + (unsigned)(s * 8192 + wave * 1024));
        }
        issue_w(t, kc, a_base + stg + 32768u);
        ++iss;
        if (++iss_it == niter) { iss_it = 0; iss_tile += gridDim.x; issue_geo(); }
    };
    if (T > 0) { issue_geo(); issue_next(); }

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    int tile = blockIdx.x, it = 0, st = 0;
    for (int j = 0; j < T; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // two stages: iteration j is the only one in flight
        __syncthreads();             // iteration j has landed for every wave; the other stage is no longer being read
        if (iss < T) issue_next();
        {
            const int t = it / p.kchunks;
            const unsigned char* A = smem + st * P2STAGE;
            const unsigned char* W = A + 32768;
            const unsigned en = (benable >> (t * 8 + wcol * 4)) & 15u;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 xf[2];
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    xf[a] = *reinterpret_cast<const bf16x8*>(A + arow[a] * 128 + (((2 * kk + h) ^ asw[a]) << 4));
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if ((en >> b) & 1u) {            // uniform
                        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(
                            W + ((wcol * 4 + b) * 32 + r) * 128 + (((2 * kk + h) ^ wsw[b]) << 4));
#pragma unroll
                        for (int a = 0; a < 2; ++a)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf[a], acc[a][b], 0, 0, 0);
                    }
                }
            }
        }
        st ^= 1;
        if (++it < niter) continue;
        it = 0;

        // ---- epilogue (as pconv_kernel; the old values of an accumulating destination are read here)
        unsigned any = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) any |= (benable >> (t * 8 + wcol * 4)) & 15u;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int m = tile * P2M + wr * 64 + a * 32 + r;
            const bool mvalid = m < p.M;
            const int px = m & (pw - 1), py = (m >> p.pwl) & (ph - 1), n = m >> (p.pwl + p.phl);
            const long obase = (((long)n * p.OH + py * p.OS + p.OY0) * p.OW + px * p.OS + p.OX0) * p.DC;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (!((any >> b) & 1u)) continue;             // uniform: no tap feeds this block
                const int bb = wcol * 4 + b;
                int grp = 0, c0 = 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) { grp = bb == q ? wgrp[q] : grp; c0 = bb == q ? wc0[q] : c0; }
                unsigned q[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * g + e];
                    if (p.bias && c0 + 8 * g + 4 * h < p.gcols) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + grp * p.gcols + c0 + 8 * g + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += bv[e];
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
                const int c = c0 + 16 * h;
                if (mvalid && c < p.gcols) {
                    const long poff = p.ngroups > 1 ? ((long)(grp >> 1) * p.OW + (grp & 1)) * p.DC : 0;
                    bf16_t* o = reinterpret_cast<bf16_t*>(p.dst) + obase + poff + c;
                    u32x4 lo = u32x4{q[0][0], q[0][1], q[2][0], q[2][1]};
                    u32x4 hi = u32x4{q[1][0], q[1][1], q[3][0], q[3][1]};
                    if (p.accum) {
                        const u32x4 ol = *reinterpret_cast<const u32x4*>(o);
                        const u32x4 oh = c + 8 < p.gcols ? *reinterpret_cast<const u32x4*>(o + 8) : u32x4{0u, 0u, 0u, 0u};
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
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
            }
        }
        tile += gridDim.x;
    }
}

}  // namespace

// Called by cu_conv_gemm before its own tiling: returns 1 if the launch was taken, 0 if the shape is not this kernel's,
// < 0 on error.
int cu_pconv_try(const cu_conv_desc* d, const void* src0, const void* w, const float* bias, void* dst0, void* stream) {
    if (d->dtype != CU_BF16 || d->C1 != 0 || d->out_nchw_f32 || d->D0 != d->CO) return 0;
    if (d->slope0 != 1.0f) return 0;
    const bool parity = d->par_co > 0;
    // 3x3 stride-2 forward (9 taps, IS = 2) computes correctly here (tests/test_kernels_gpu.py runs it under
    // CU_PCONV_S2FWD=1 in the tuning build) but re-fetches every pixel row 9 times through L2 and lost to the halo kernels
    // on all levels but one (profiles/r02_layers_pconv.txt): off.
    const bool few = d->ntaps <= 4, s2fwd = d->ntaps == 9 && d->IS == 2 && !parity && cu_env_set("CU_PCONV_S2FWD");
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
    // few pixel tiles: every workgroup stages its weight slice for a handful of tiles and the tile-generic kernel's
    // narrow-column tiling wins (measured in the step, 64 images: transposed conv forward from 8x8 up, the gradients
    // from 16x16 up)
    const long min_m = (parity && !d->par_taps && d->ntaps == 1) ? 4096 : 16384;
    if (M < min_m || M >= (1l << 30)) return 0;

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
    // resident weights: the largest packed slice over the column tiles (4 KiB per existing (iteration, block) pair)
    int max_blocks = 0;
    for (int ct = 0; ct < a.coltiles; ++ct) {
        int blocks = 0;
        for (int b = 0; b < 4; ++b) {
            int group, c0;
            if (gcols >= PBN) { const int tpg = cdiv(gcols, PBN); group = ct / tpg; c0 = (ct - group * tpg) * PBN + b * 32; }
            else { const int col = ct * PBN + b * 32; group = col / gcols; c0 = col - group * gcols; }
            if (group >= a.ngroups || c0 >= gcols) continue;
            for (int t = 0; t < a.ntaps; ++t) blocks += a.wrow[t * 4 + group] >= 0 ? a.kchunks : 0;
        }
        max_blocks = blocks > max_blocks ? blocks : max_blocks;
    }
    a.wres = max_blocks * 4096 <= 48 * 1024 && niter <= PMAXT * 8;
    // ---- mid levels: the 256 x 256 tile form (half the staged bytes per FLOP; see pconv2_kernel)
    if (few && d->ntaps <= 4 && !a.wres && d->C0 >= 128 && a.CO >= 256 && (gcols >= 256 || 256 % gcols == 0) &&
        M >= 16384 && M <= (1l << 19) && !cu_env_set("CU_PCONV_NO256")) {
        a.mtiles = cdiv((int)M, P2M);
        a.coltiles = gcols >= P2N ? a.ngroups * cdiv(gcols, P2N) : cdiv(a.CO, P2N);
        const size_t lds2 = 2 * (size_t)P2STAGE;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pconv2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds2);
        CU_CHECK_ARG(e == hipSuccess, "cu_conv_gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        int gx = a.mtiles;
        const int cap2 = 256 / (a.coltiles < 4 ? a.coltiles : 4);
        if (gx > cap2 && cap2 > 0) gx = cap2;
        hipLaunchKernelGGL(pconv2_kernel, dim3(gx, a.coltiles), dim3(512), lds2, reinterpret_cast<hipStream_t>(stream), a);
        CU_LAUNCH_CHECK();
        return 1;
    }
    // ring depth: measured (profiles/r02_pconv_ring.txt) -- occupancy beats depth: two stages with 2-3 workgroups per CU
    // (their loads interleave) outrun four stages with one workgroup per CU
    int ring = cu_env_int("CU_PCONV_RING", 2);
    if (ring != 2 && ring != 3 && ring != 4) ring = 2;
    const size_t lds = (size_t)ring * PSTAGE + (a.wres ? (size_t)max_blocks * 4096 : (size_t)ring * PSTAGE);
    int grid_x = a.mtiles;
    const int cap = 256 * cu_env_int("CU_PCONV_CAP", 2) / (a.coltiles < 4 ? a.coltiles : 4);
    if (grid_x > cap && cap > 0) grid_x = cap;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    auto k = a.wres ? (ring == 2 ? pconv_kernel<true, 2> : ring == 3 ? pconv_kernel<true, 3> : pconv_kernel<true, 4>)
                    : (ring == 2 ? pconv_kernel<false, 2> : ring == 3 ? pconv_kernel<false, 3> : pconv_kernel<false, 4>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        CU_CHECK_ARG(e == hipSuccess, "cu_conv_gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k, dim3(grid_x, a.coltiles), dim3(256), lds, st, a);
    CU_LAUNCH_CHECK();
    return 1;
}
