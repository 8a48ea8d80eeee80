// Streaming 3x3 stride-1 convolution for the THIN, large layers (gfx950, bf16 production mode): 256^2 x 32 channels and
// 128^2 x 64 channels of the U-Net's two top levels -- forward, concat forward, input gradient and the input gradient
// with two destinations (reference models/nnUnet/layers.py:55-80,192-205,436; dispatched inside cu_conv_gemm).
//
// These layers are HBM-bound (32 -> 32 at 256^2, batch 64: 536 MB in + out = ~100 us at 5.4 TB/s against 31 us of
// MFMA work), so the kernel is built as a stream, not as a GEMM:
//   * persistent workgroups (one per CU, 8 waves); the whole weight set (<= 73 KiB) is staged ONCE and stays in LDS;
//   * a ring of R halo tiles ((TH+2) x 34 pixels, all input channels) filled by LDS-DMA (buffer_load_dwordx4 ... lds) one or
//     two tiles ahead: the copy of tile i+1 / i+2 flies under the MFMAs and the stores of tile i, and ONE barrier per
//     tile both publishes tile i and frees the slot of tile i-1;
//   * counted waits: loads, LDS-DMA and stores retire in issue order, so s_waitcnt vmcnt(N) with N = the (compile-time)
//     number of younger operations waits for exactly the tile that is needed and never for the stores just issued
//     (every iteration issues the same number of DMA instructions -- out-of-range ones when the tiles run out -- and of
//     stores, which is why the kernel only takes whole tiles: PW % 32 == 0, PH % TH == 0);
//   * LDS images are lane-linear [32-channel plane][row][64 bytes] (row = halo pixel / weight row); 16-byte piece p of
//     row r sits in slot p ^ ((r >> 2) & 3) (applied on the DMA source address and on the read): conflict-free
//     ds_read_b128 fragments, and every DMA instruction (16 rows of one plane) reads ONE source of a concat;
//   * the MFMA takes the weights as A and the pixels as B: a lane owns one pixel and 16 consecutive channels after two
//     v_permlane32_swap -> two 16-byte stores per 32-channel block; zero padding = the buffer range check.
#include "common.h"

namespace {

struct TcArgs {
    const void* src0; const void* src1; const void* w; const float* bias; void* dst0; void* dst1;
    unsigned src0_bytes, src1_bytes, w_bytes;
    int N, H, W, C0, C1;            // source = destination grid; channels of the two sources (C1 may be 0)
    int tap_off[9];                 // halo row offset of gather tap t: (dy + 1) * 34 + (dx + 1)
    int tap_w[9];                   // weight tap of gather tap t
    int D0, DC0, DC1;               // columns [0, D0) -> dst0, the rest -> dst1; channel strides
    int ntiles, tiles_x, tiles_y;   // tiles of TH x 32 pixels
    float* stat_sums;               // MODE 1: [N][CO][2] += (sum, sum of squares) of the outputs minus their bias
                                    // MODE 2: [N][CO][2] += the two sums of the InstanceNorm backward (see below)
    int run;                        // MODE 1, 2: consecutive tiles per workgroup (all of one image)
    const void* nz; const float* nstats; float nslope;      // MODE 2: raw output z, statistics planes, LeakyReLU slope of
                                                            // the layer whose output gradient this launch produces
    // XF (round 3): the source is the RAW output z of the producing layer and its InstanceNorm +
    // LeakyReLU (a = LeakyReLU(scale z + shift), reference layers.py:193-194) is applied to every halo tile IN PLACE in
    // LDS, once per staged element, before the MFMAs read it -- the materialised activation and its apply pass are gone.
    const float* xscale;            // [N][C0] f32; the shifts follow N * C0 floats later (planes 2 and 3 of cu_instnorm_stats)
    unsigned xstat_bytes;           // bytes of the two planes
    float xslope;
};

__device__ __forceinline__ int swz(int row) { return (row >> 2) & 3; }

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// CIP = input channels / 32, NB = output channels / 32, TH = tile rows, R = ring slots.
// MODE 1: the InstanceNorm statistics of the layer are gathered here: a workgroup walks `run`
// CONSECUTIVE tiles of one image, every lane keeps running sums of its 16 channels' f32 accumulators (one add and one
// fma per accumulator register and tile -- the stream has the vector slack), and ONE 32-lane reduction + one set of
// atomics per workgroup hands sum / sum of squares of (z - bias) to the norm kernels' finalize step.
// MODE 2 (this launch is an input gradient g = dL/da of a layer a = LeakyReLU(scale z + shift); item 1(ii)): the reduction
// pass of that layer's InstanceNorm backward happens here -- the lane reads its pixel's z (8-byte loads issued by hand
// right after the barrier, so that they are OLDER than this iteration's LDS-DMA and waiting for them does not wait for
// the prefetch), forms gl = g * LeakyReLU'(y) and keeps running sums of gl and gl * zhat; same flush as MODE 1.
// XF: normalise + activate the staged source in LDS (see TcArgs).  One more LDS-DMA instruction per wave and tile brings the
// image's scale / shift rows into a 1-KiB table beside the ring slot (waves 1..7 issue theirs out of range into a shared dump
// kilobyte, so every wave's counted waits stay equal); after the barrier that publishes the tile every thread rewrites the
// pieces it staged itself -- skipping out-of-image halo pixels, which must stay zero (the zero padding of the conv) -- and a
// second barrier hands the tile to the MFMAs.  Rounding: bf16(LeakyReLU(z * scale + shift)), the apply pass's own arithmetic.
template <int CIP, int NB, int TH, int R, int MODE, bool XF = false>
__global__ __launch_bounds__(512) void tconv_kernel(const TcArgs p) {
    constexpr bool STATS = MODE != 0;
    constexpr int CI = 32 * CIP, CO = 32 * NB;
    constexpr int HW34 = 34, HALO = (TH + 2) * HW34, HPAD = (HALO + 15) / 16 * 16;     // plane stride: whole DMA instructions
    constexpr int XPIECES = CIP * HPAD * 4, D = (XPIECES + 511) / 512, SLOT_B = D * 8192;
    constexpr int WROWS = 9 * CO;                                                        // a multiple of 16
    constexpr int WPIECES = CIP * WROWS * 4, WD = (WPIECES + 511) / 512, W_B = WD * 8192;
    constexpr int NBW = NB * TH / 8;            // 32-column blocks per wave
    constexpr int S = 2 * NBW;                  // 16-byte stores per thread and tile
    constexpr int ZL = MODE == 2 ? 4 * NBW : 0; // hand-issued 8-byte loads of z per thread and tile
    // MODE 1 (a run of tiles of ONE image per workgroup): the table is loaded once and every thread rewrites the pieces it
    // staged itself BEFORE the tile's barrier -- no second barrier; MODE 0 (tiles of any image): table by LDS-DMA per tile
    constexpr bool XF1 = XF && MODE == 1;
    constexpr int XT = (XF && !XF1) ? 1 : 0;    // table DMA instructions per wave and tile
    constexpr int TAB0 = W_B + R * SLOT_B + 1024;        // XF: R tables of 1 KiB, then the dump kilobyte
    static_assert(NBW >= 1 && R >= 2 && R <= 4 && W_B + R * SLOT_B + 1024 + (XF ? (R + 1) * 1024 : 0) <= 160 * 1024, "tconv: bad instance");
    static_assert(!XF || (MODE != 2 && CI <= 128), "tconv: XF serves the forward instances");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const unsigned w_base = lds_addr(smem), x_base = w_base + W_B;
    const i32x4 rs0 = make_rsrc(p.src0, p.src0_bytes);
    const i32x4 rs1 = make_rsrc(p.src1 ? p.src1 : p.src0, p.src1 ? p.src1_bytes : 0u);
    const i32x4 rw = make_rsrc(p.w, p.w_bytes);
    const i32x4 rx = make_rsrc(XF ? (const void*)p.xscale : p.src0, XF ? p.xstat_bytes : 0u);
    constexpr unsigned OOB = 0x7ffffff0u;

    // ---- weights, once: LDS row (t * CO + n) of plane pl = channels [32 pl, 32 pl + 32) of weight row (tap_w[t], n);
    //      static tap indices only (a runtime index into the by-value argument block would move it to scratch)
#pragma unroll
    for (int j = 0; j < WD; ++j) {
        const int q = tid + j * 512;
        const int pl = q / (WROWS * 4), qq = q - pl * (WROWS * 4);
        const int row = qq >> 2, s = qq & 3;
        const int t = row / CO, n = row - t * CO;
        int tw = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) tw = t == k ? p.tap_w[k] : tw;
        const int piece = s ^ swz(row);
        const unsigned off = q < WPIECES ? (unsigned)(((tw * CO + n) * CI + pl * 32 + piece * 8) * 2) : OOB;
        dma16(rw, off, w_base + (unsigned)(j * 8192 + wave * 1024));
    }

    // ---- per-thread halo staging geometry (tile-invariant): plane, halo pixel, piece.  A DMA instruction covers 16 rows
    //      of ONE plane (HPAD is a multiple of 16), so its source is wave-uniform
    int hy[D], hx[D];
    unsigned coff[D];               // byte offset of the piece inside its source pixel; ~0u: padding row / no item
    const bool cat = p.C1 > 0;      // plane 1 = source 1 (C0 = C1 = 32)
#pragma unroll
    for (int j = 0; j < D; ++j) {
        const int q = tid + j * 512;
        const int pl = q / (HPAD * 4), qq = q - pl * (HPAD * 4);
        const int hp = qq >> 2, s = qq & 3;
        hy[j] = hp / HW34; hx[j] = hp - hy[j] * HW34;
        const int c = (cat ? 0 : pl * 32) + (s ^ swz(hp)) * 8;
        coff[j] = (q < XPIECES && hp < HALO) ? (unsigned)(c * 2) : 0xffffffffu;
    }
    const unsigned pix0_b = (unsigned)p.C0 * 2u, pix1_b = (unsigned)p.C1 * 2u;

    // XCD-aware tile order (workgroups are dealt round-robin to the 8 XCDs): logical tile L -> (L % 8) * ntiles/8 + L / 8,
    // so one XCD's L2 sees a contiguous eighth of the tiles and the halo rows shared by vertical neighbours are L2 hits
    const bool xcd = !STATS && (p.ntiles & 7) == 0 && (gridDim.x & 7) == 0;
    auto tile_of = [&](int l) { return xcd ? (l & 7) * (p.ntiles >> 3) + (l >> 3) : l; };
    const int l_step = STATS ? 1 : (int)gridDim.x;
    const int l_begin = STATS ? (int)blockIdx.x * p.run : (int)blockIdx.x;
    const int l_end = STATS ? l_begin + p.run : p.ntiles;
    auto issue = [&](int l, int slot) {          // DMA of logical tile l (or D out-of-range instructions) into ring slot
        const bool live = l < l_end;
        const int tile = live ? tile_of(l) : 0;
        const int tx = tile % p.tiles_x, rest = tile / p.tiles_x;
        const int ty = rest % p.tiles_y, n = rest / p.tiles_y;
        const int y0 = ty * TH - 1, x0 = tx * 32 - 1;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int sy = y0 + hy[j], sx = x0 + hx[j];
            const bool ok = live && coff[j] != 0xffffffffu && sy >= 0 && sy < p.H && sx >= 0 && sx < p.W;
            // plane of this instruction (wave-uniform, made scalar so that exactly ONE DMA instruction is issued)
            const bool s1 = cat && __builtin_amdgcn_readfirstlane((tid + j * 512) / (HPAD * 4)) != 0;
            const unsigned pix = (unsigned)((n * p.H + sy) * p.W + sx);
            const unsigned off = ok ? pix * (s1 ? pix1_b : pix0_b) + coff[j] : OOB;
            const unsigned dst = x_base + (unsigned)(slot * SLOT_B + j * 8192 + wave * 1024);
            if (s1) dma16(rs1, off, dst);
            else dma16(rs0, off, dst);
        }
        if constexpr (XF && !XF1) {        // lanes [0, CI/4): 16-byte pieces of scale[n][:], lanes [CI/4, CI/2): of shift[n][:]
            const bool sh = lane >= CI / 4;
            const unsigned off = (live && wave == 0 && lane < CI / 2)
                ? (unsigned)(((sh ? p.N : 0) + n) * CI + 4 * (lane - (sh ? CI / 4 : 0))) * 4u : OOB;
            dma16(rx, off, w_base + (unsigned)(TAB0 + (wave == 0 ? slot : R) * 1024));
        }
    };

    // ---- this wave's output: row `row` of the tile, column blocks b0 .. b0 + NBW - 1
    const int row = wave % TH, b0 = (wave / TH) * NBW;
    const int hp0 = row * HW34 + r;              // + tap_off[t] = the pixel's halo row for tap t
    f32x4 bv[NBW][4];
#pragma unroll
    for (int b = 0; b < NBW; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            bv[b][g] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + (b0 + b) * 32 + 8 * g + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
    float* s_par = reinterpret_cast<float*>(smem + W_B + R * SLOT_B);      // MODE 2: [CO][4] = scale, shift, rstd, -mean*rstd
    if constexpr (MODE == 2) {
        if (tid < CO) {
            const int n_img = ((int)blockIdx.x * p.run) / (p.tiles_x * p.tiles_y);
            const size_t NC = (size_t)p.N * CO, i = (size_t)n_img * CO + tid;
            const float mean = p.nstats[i], rstd = p.nstats[NC + i];
            *reinterpret_cast<f32x4*>(s_par + 4 * tid) = f32x4{p.nstats[2 * NC + i], p.nstats[3 * NC + i], rstd, -mean * rstd};
        }
    }
    if constexpr (XF1) {        // scale[CI], shift[CI] of this workgroup's image -> table 0, once
        if (tid < 2 * CI) {
            const int n_img = ((int)blockIdx.x * p.run) / (p.tiles_x * p.tiles_y);
            const bool sh = tid >= CI;
            reinterpret_cast<float*>(smem + TAB0)[tid] = p.xscale[(size_t)((sh ? p.N : 0) + n_img) * CI + (tid - (sh ? CI : 0))];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // weights + bias have landed (before any counted wait below)
    if constexpr (XF1) __syncthreads();                    // ... and the table is visible to every wave

    const bf16_t* W16 = reinterpret_cast<const bf16_t*>(smem);
    float ssum[STATS ? NBW : 1][16], ssq[STATS ? NBW : 1][16];
    if constexpr (STATS) {
#pragma unroll
        for (int b = 0; b < NBW; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) { ssum[b][i] = 0.f; ssq[b][i] = 0.f; }
    }
    // XF: rewrite the pieces THIS thread staged into ring slot `slot` (tile origin y0, x0) with table `tb`
    auto xform = [&](int slot, int tb, int y0, int x0) {
        const float* tab = reinterpret_cast<const float*>(smem + TAB0 + tb * 1024);     // scale[CI], shift[CI]
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int sy = y0 + hy[j], sx = x0 + hx[j];
            if (coff[j] != 0xffffffffu && sy >= 0 && sy < p.H && sx >= 0 && sx < p.W) {
                const int c0 = (int)(coff[j] >> 1);      // first channel of the piece (single source)
                u32x4* pp = reinterpret_cast<u32x4*>(smem + W_B + slot * SLOT_B + j * 8192 + tid * 16);
                u32x4 v = *pp;
                const f32x4 sa = *reinterpret_cast<const f32x4*>(tab + c0), sb = *reinterpret_cast<const f32x4*>(tab + c0 + 4);
                const f32x4 ha = *reinterpret_cast<const f32x4*>(tab + CI + c0), hb = *reinterpret_cast<const f32x4*>(tab + CI + c0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float z0 = __uint_as_float(v[e] << 16), z1 = __uint_as_float(v[e] & 0xffff0000u);
                    const float sc0 = e < 2 ? sa[2 * e] : sb[2 * e - 4], sc1 = e < 2 ? sa[2 * e + 1] : sb[2 * e - 3];
                    const float sh0 = e < 2 ? ha[2 * e] : hb[2 * e - 4], sh1 = e < 2 ? ha[2 * e + 1] : hb[2 * e - 3];
                    float a0 = z0 * sc0 + sh0, a1 = z1 * sc1 + sh1;
                    a0 = a0 > 0.f ? a0 : a0 * p.xslope;
                    a1 = a1 > 0.f ? a1 : a1 * p.xslope;
                    v[e] = (unsigned)f32_to_bf16(a0) | ((unsigned)f32_to_bf16(a1) << 16);
                }
                *pp = v;
            }
        }
    };
    int l = l_begin;
#pragma unroll
    for (int k = 0; k < R - 1; ++k) issue(l + k * l_step, k);
    for (int it = 0; l < l_end; ++it, l += l_step) {
        // DMA(it) must have landed.  An iteration issues, in this order: its ZL hand-issued loads (MODE 2), the DMA of tile
        // it + R - 1 (DI instructions), its S stores.  Younger than DMA(it): the R - 2 - it prologue DMAs behind it and the `it`
        // whole iterations so far (it <= R - 2); in the steady state the stores of iteration it - R + 1 and R - 2 whole iterations
        constexpr int DI = D + XT, ITER = ZL + DI + S;
        if (it >= R - 1) wait_vm<S + (R - 2) * ITER>();
        else if (it == 0) wait_vm<(R - 2) * DI>();
        else if (it == 1) wait_vm<(R >= 3 ? (R - 3) * DI + ITER : 0)>();
        else wait_vm<(R >= 4 ? (R - 4) * DI + 2 * ITER : 0)>();          // it == 2 (R = 4)
        const int tile = tile_of(l);
        const int tx = tile % p.tiles_x, rest = tile / p.tiles_x;
        const int ty = rest % p.tiles_y, n = rest / p.tiles_y;
        if constexpr (XF1) xform(it % R, 0, ty * TH - 1, tx * 32 - 1);      // this thread's own pieces: they have landed
        __syncthreads();        // tile `it` is complete for every wave, and nobody reads the slot of tile it-1 any more
        const size_t opix = ((size_t)n * p.H + ty * TH + row) * p.W + tx * 32 + r;
        u32x2 zq[MODE == 2 ? NBW : 1][4];
        if constexpr (MODE == 2) {
            const bf16_t* zp = reinterpret_cast<const bf16_t*>(p.nz) + opix * CO + b0 * 32 + 4 * h;
#pragma unroll
            for (int b = 0; b < NBW; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(zq[b][g]) : "v"(zp + b * 32 + 8 * g) : "memory");
        }
        issue(l + (R - 1) * l_step, (it + R - 1) % R);
        const bf16_t* X16 = reinterpret_cast<const bf16_t*>(smem + W_B + (it % R) * SLOT_B);
        if constexpr (XF && !XF1) {
            xform(it % R, it % R, ty * TH - 1, tx * 32 - 1);
            __syncthreads();      // the rewritten tile is complete for every wave
        }

        f32x16 acc[NBW];
#pragma unroll
        for (int b = 0; b < NBW; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int xr = hp0 + p.tap_off[t];
            const int xs = swz(xr);
#pragma unroll
            for (int kk = 0; kk < CI / 16; ++kk) {
                const int pl = kk >> 1, pc = 2 * (kk & 1) + h;
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(X16 + ((size_t)(pl * HPAD + xr) * 4 + (pc ^ xs)) * 8);
#pragma unroll
                for (int b = 0; b < NBW; ++b) {
                    const int wr = t * CO + (b0 + b) * 32 + r;
                    const bf16x8 bf = *reinterpret_cast<const bf16x8*>(W16 + ((size_t)(pl * WROWS + wr) * 4 + (pc ^ swz(wr))) * 8);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf, af, acc[b], 0, 0, 0);
                }
            }
        }

        if constexpr (MODE == 1) {
#pragma unroll
            for (int b = 0; b < NBW; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    ssum[b][i] += acc[b][i];
                    ssq[b][i] = fmaf(acc[b][i], acc[b][i], ssq[b][i]);
                }
        }
        if constexpr (MODE == 2) {
            // the z loads are older than this iteration's D DMA instructions: wait for them only (the asm ties the
            // registers to the wait, so no use is scheduled above it)
#pragma unroll
            for (int b = 0; b < NBW; ++b)
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(zq[b][0]), "+v"(zq[b][1]), "+v"(zq[b][2]), "+v"(zq[b][3]) : "n"(D) : "memory");
#pragma unroll
            for (int b = 0; b < NBW; ++b)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned wd = zq[b][g][e >> 1];
                        const float zf = __uint_as_float((e & 1) ? (wd & 0xffff0000u) : (wd << 16));
                        const f32x4 pr = *reinterpret_cast<const f32x4*>(s_par + 4 * ((b0 + b) * 32 + 8 * g + 4 * h + e));
                        const float gv = acc[b][4 * g + e];
                        const float gl = zf * pr[0] + pr[1] > 0.f ? gv : gv * p.nslope;
                        ssum[b][4 * g + e] += gl;
                        ssq[b][4 * g + e] = fmaf(gl, fmaf(zf, pr[2], pr[3]), ssq[b][4 * g + e]);
                    }
                }
        }
        // ---- epilogue: register i of a block = channel (i & 3) + 8 * (i >> 2) + 4 * h of pixel r
#pragma unroll
        for (int b = 0; b < NBW; ++b) {
            const int colb = (b0 + b) * 32;
            const bool d1 = colb >= p.D0;
            unsigned q[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[b][4 * g + e] + bv[b][g][e];
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
            bf16_t* o = reinterpret_cast<bf16_t*>(d1 ? p.dst1 : p.dst0) + opix * (d1 ? p.DC1 : p.DC0) +
                        (d1 ? colb - p.D0 : colb) + 16 * h;
            *reinterpret_cast<u32x4*>(o) = u32x4{q[0][0], q[0][1], q[2][0], q[2][1]};
            *reinterpret_cast<u32x4*>(o + 8) = u32x4{q[1][0], q[1][1], q[3][0], q[3][1]};
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing out-of-range DMA instructions target live LDS
    if constexpr (STATS) {
        // lanes with the same h hold the same 16 channels of 32 different pixels: butterfly over the pixel lanes
        const int n = (l_begin / (p.tiles_x * p.tiles_y));
#pragma unroll
        for (int b = 0; b < NBW; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float a = ssum[b][i], q2 = ssq[b][i];
#pragma unroll
                for (int m = 1; m < 32; m <<= 1) {
                    a += __shfl_xor(a, m, 64);
                    q2 += __shfl_xor(q2, m, 64);
                }
                if (r == 0) {
                    const int col = (b0 + b) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                    float* o = p.stat_sums + ((size_t)n * CO + col) * 2;
                    unsafeAtomicAdd(o, a);
                    unsafeAtomicAdd(o + 1, q2);
                }
            }
    }
}

template <int CIP, int NB, int TH, int R, bool XFI = false>
int launch_tc(TcArgs& a, hipStream_t st) {
    constexpr int HPAD = ((TH + 2) * 34 + 15) / 16 * 16;
    constexpr int D = (CIP * HPAD * 4 + 511) / 512, WD = (CIP * 9 * 32 * NB * 4 + 511) / 512;
    const bool xf = XFI && a.xscale != nullptr;
    const size_t lds = (size_t)WD * 8192 + (size_t)R * D * 8192 + 1024 +     // + the MODE 2 parameter table
                       (xf ? (size_t)(R + 1) * 1024 : 0);                    // + XF: scale / shift tables and the dump
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;       // workgroups that fit a CU (LDS; <= 128 VGPRs in every instance)
    int grid = a.ntiles < 256 * per_cu ? a.ntiles : 256 * per_cu;
    // statistics: every workgroup needs a run of consecutive tiles inside ONE image
    const int per_img = a.tiles_x * a.tiles_y;
    const bool stats = a.stat_sums && a.ntiles % grid == 0 && per_img % (a.ntiles / grid) == 0;
    if (!stats) a.stat_sums = nullptr;
    a.run = stats ? a.ntiles / grid : 0;
    auto k = !stats ? tconv_kernel<CIP, NB, TH, R, 0> : (a.nz ? tconv_kernel<CIP, NB, TH, R, 2> : tconv_kernel<CIP, NB, TH, R, 1>);
    if constexpr (XFI) {
        if (xf) {
            CU_CHECK_ARG(!a.nz, "cu_conv_gemm: the normalise-on-load source is a forward form");
            k = stats ? tconv_kernel<CIP, NB, TH, R, 1, true> : tconv_kernel<CIP, NB, TH, R, 0, true>;
        }
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    CU_CHECK_ARG(e == hipSuccess, "cu_conv_gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), lds, st, a);
    CU_LAUNCH_CHECK();
    return stats ? 2 : 1;
}

}  // namespace

// 2 = launched and the sums were gathered, 1 = launched, 0 = not this kernel's shape, < 0 = error.  Called by cu_conv_gemm
// for plain bf16 operands.  ep (or NULL): the epilogue extension of cu_conv_gemm_ex.
// scale0 / shift0 (or NULL): source 0 is a RAW conv output whose InstanceNorm + LeakyReLU (slope0) this launch applies in
// LDS (XF); served for one source, with the shift plane N * C0 floats behind the scale plane (cu_instnorm_stats' layout).
int cu_tconv_try(const cu_conv_desc* d, const void* src0, const void* src1, const void* w, const float* bias, void* dst0,
                 void* dst1, const cu_conv_epilogue* ep, const float* scale0, const float* shift0, void* stream) {
    const bool xf = scale0 != nullptr;
    if (d->dtype != CU_BF16 || d->ntaps != 9 || d->IS != 1 || d->OS != 1 || d->OY0 || d->OX0 || d->out_nchw_f32 ||
        d->par_co || d->accum0 || d->accum1 || (!xf && d->slope0 != 1.0f) || (d->C1 && d->slope1 != 1.0f))
        return 0;
    if (xf && (d->C1 || shift0 != scale0 + (size_t)d->N * d->C0 || (ep && ep->mode == 2))) return 0;
    if (d->PH != d->SH || d->PW != d->SW || d->OH != d->PH || d->OW != d->PW || d->PW % 32 || d->PH % 8) return 0;
    const int CI = d->C0 + d->C1;
    if (!((CI == 32 && d->C1 == 0) || (CI == 64 && (d->C1 == 0 || d->C0 == 32)))) return 0;
    if (d->CO != 32 && d->CO != 64) return 0;
    const bool two = d->D0 != d->CO;
    if (two ? !(d->D0 == 32 && d->DC0 == 32 && d->DC1 == d->CO - 32 && dst1) : d->DC0 != d->CO) return 0;
    if (CI == 64 && d->CO == 64 && d->C1) return 0;                     // 128-byte rows x 64 columns: no concat instance
    if ((long)d->N * d->PH * d->PW < (1L << 20)) return 0;             // the large maps only (256^2, 128^2 at batch >= 16 / 64)
    const size_t b0 = (size_t)d->N * d->SH * d->SW * d->C0 * 2, b1 = (size_t)d->N * d->SH * d->SW * d->C1 * 2;
    if (b0 >= 0x7fff0000ull || b1 >= 0x7fff0000ull) return 0;
    TcArgs a;
    memset(&a, 0, sizeof(a));
    a.src0 = src0; a.src1 = d->C1 ? src1 : nullptr; a.w = w; a.bias = bias; a.dst0 = dst0; a.dst1 = dst1;
    a.src0_bytes = (unsigned)b0; a.src1_bytes = (unsigned)b1; a.w_bytes = (unsigned)((size_t)9 * d->CO * CI * 2);
    a.N = d->N; a.H = d->PH; a.W = d->PW; a.C0 = d->C0; a.C1 = d->C1;
    a.D0 = d->D0; a.DC0 = d->DC0; a.DC1 = d->DC1;
    if (xf) {
        if (two || !((CI == 32 && d->CO == 32) || (CI == 64 && d->CO == 64))) return 0;      // instances built with XF
        a.xscale = scale0; a.xstat_bytes = (unsigned)((size_t)2 * d->N * d->C0 * 4); a.xslope = d->slope0;
    }
    if (ep && !two && ep->sums && (ep->mode == 1 || (ep->mode == 2 && ep->z && ep->stats && !bias))) {
        a.stat_sums = ep->sums;
        if (ep->mode == 2) { a.nz = ep->z; a.nstats = ep->stats; a.nslope = ep->slope; }
    }
    for (int t = 0; t < 9; ++t) {
        if (d->tap_dy[t] < -1 || d->tap_dy[t] > 1 || d->tap_dx[t] < -1 || d->tap_dx[t] > 1 || d->tap_w[t] < 0 || d->tap_w[t] > 8)
            return 0;
        a.tap_off[t] = (d->tap_dy[t] + 1) * 34 + (d->tap_dx[t] + 1);
        a.tap_w[t] = d->tap_w[t];
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int th = (CI == 64 && d->CO == 64) ? 4 : 8;
    a.tiles_x = d->PW / 32; a.tiles_y = d->PH / th; a.ntiles = a.tiles_x * a.tiles_y * d->N;
    // Measured at 256^2 x 32, batch 64 (tools/thin_bench.py, profiles/r02_thin_bench.txt): two ring slots and two workgroups
    // per CU beat three slots and one workgroup (32 -> 32: 129 vs 143 us forward, 130 vs 162 us input gradient); for the
    // two-destination input gradient the 4-row tile that would fit two workgroups loses to 8 rows x 3 slots (248 vs 237)
    // Round 4: INSIDE the training step the choice reverses -- three slots / one workgroup per CU: 12.26 against 12.43 ms per step
    // over five alternating pairs (profiles/r04_knob_sweep.txt, r04_tconv_var_in_step.txt): two workgroups per CU take the LDS that
    // the launches of the other streams beside them (weight gradients, operand copies) need.  The lone-launch winner stays behind
    // the knob.
    const int var = cu_env_int("CU_TCONV_VAR", 0);      // tuning knob: 1 = two slots, two workgroups per CU
    // (the norm-backward epilogue needs 146 registers: one workgroup per CU either way, so it takes the three-slot ring)
    // (the normalise-on-load form, XF, exists in the two-slot instance only)
    // (tuning: 2 = the two-slot form for launches with a statistics epilogue (forward) only, 3 = for the others only)
    const bool two_slot = var == 1 || (var == 2 && a.stat_sums) || (var == 3 && !a.stat_sums);
    if (CI == 32 && d->CO == 32 && var == 4) return launch_tc<1, 1, 8, 4>(a, st);      // (tuning: four slots)
    if (CI == 32 && d->CO == 32) return ((two_slot || a.xscale) && !a.nz) ? launch_tc<1, 1, 8, 2, true>(a, st) : launch_tc<1, 1, 8, 3>(a, st);
    if (CI == 32 && d->CO == 64) return launch_tc<1, 2, 8, 3>(a, st);
    if (CI == 64 && d->CO == 32) return launch_tc<2, 1, 8, 2>(a, st);
    return launch_tc<2, 2, 4, 2, true>(a, st);
}
