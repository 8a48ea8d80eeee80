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
#include <type_traits>

namespace {

constexpr int MAXI = 10;   // largest NX instantiated: halo pieces per thread per chunk (halo_px*4 <= 256*MAXI)

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
    int par_co;                 // > 0: column group g = col / par_co goes to destination parity (g >> 1, g & 1)
    int par_taps;               // with par_co: weight tap of (gather tap t, group g) = nibble t * 4 + g of par_pack, minus 1
    unsigned long long par_pack;
    int imgs;                   // images per tile (TW*TH*imgs <= 128*MA; rows beyond are idle)
    int txl, tyl, ntiles;       // log2 of tiles per row / column, total pixel tiles
    int tap_lds;                // byte offset of the tap table inside the dynamic LDS
    int ksplit, kspan;          // split-K (tiny feature maps): grid.z = ksplit workgroups per tile, each walks kspan input
    float* ws0; float* ws1;     // channels and stores its f32 partial tile into slice blockIdx.z of ws (laid out like
    size_t ws_slice;            // dst0 | dst1; ws_slice floats per split)
    int dbg;                    // CU_CONV_DBG bits (timing experiments only): 1 no stores, 2 no MFMA, 4 no commit, 8 no loads
    float* stat_sums;           // LDS-DMA kernels, cu_conv_epilogue mode 1: [N][CO][2] += {sum, sum of squares} of (out - bias)
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

// Template parameters
//   MA   : 32-pixel MFMA blocks per wave (tile = 128*MA loop pixels)
//   NB   : 32-column MFMA blocks per workgroup (all waves share the columns)
//   NX   : halo pieces per thread per chunk, NT : tap capacity.  Both compile-time so that every global load of the
//          staging is UNCONDITIONAL (clamped address, value masked later): hipcc serialises predicated loads with a
//          vmcnt(0) each (cdna_hip_programming.md, "three .s-level traps" (c)).
//   WRES : the whole weight slice (all chunks, all taps) of this column tile stays resident in LDS; the workgroup is
//          persistent over pixel tiles, so thin layers (C <= 64 at 256^2/128^2) stage their weights once per CU.
// The MFMA is issued with the WEIGHTS as the A operand and the PIXELS as the B operand: D[row = column n][col = pixel],
// so a lane owns one pixel and 4 consecutive output channels per register quad -> 8/16-byte epilogue stores.
//   PLAIN: the sources need no affine / activation (materialised activations): staging is a pure 16-byte copy.
template <typename T, int MA, int NB, int NX, int NT, bool WRES, bool PLAIN>
__global__ __launch_bounds__(256, WRES ? 2 : 1) void igemm_conv_kernel(const ConvKArgs p) {
    using C = Cfg<T>;
    constexpr int CK = C::CK, PIECE = C::PIECE, PPP = C::PPP, WPL = C::WPLANES;
    constexpr int BN = 32 * NB;
    constexpr int NTCAP = NT > 0 ? NT : 4;            // NT > 0: exactly NT taps, unrolled; NT == 0: <= 4 taps, runtime loop
    constexpr int NW = (NTCAP * BN * 4 + 255) / 256;  // weight pieces per thread per chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int TW = 1 << p.twl, TH = 1 << p.thl;
    const int n0 = blockIdx.y * BN;
    const int CI = p.C0 + p.C1;
    const int hpi = p.HH * p.HW;   // halo pixels per image
    const int piece = tid & 3;     // 256 % 4 == 0: a thread always stages the same 16-byte piece of a pixel / weight row

    // LDS carve: X planes then W planes.  Units: 16-byte entries (bf16) / floats (f32).
    T* Xs = reinterpret_cast<T*>(smem);
    const int x_elems = (CK / PIECE) * PPP * p.XP * (PPP == 1 ? PIECE : 1);
    T* Ws = Xs + x_elems;
    const int w_chunk_units = p.ntaps * WPL * p.WP;          // 16-byte entries (bf16) / floats (f32) per chunk
    // Tap tables go to LDS through STATIC indices: a runtime index into the by-value kernel argument would force the
    // whole argument struct into scratch memory (every p.field access then becomes a scratch load).
    int* s_tap = reinterpret_cast<int*>(smem + p.tap_lds);
#pragma unroll
    for (int t = 0; t < CU_MAX_TAPS; ++t)
        if (tid == t) { s_tap[t] = p.tap_off[t]; s_tap[16 + t] = p.tap_w[t]; }
    __syncthreads();

    // ---- tile-invariant halo staging geometry: (image, row, col) inside the halo patch
    int geo[NX], ldsx[NX];
    unsigned exist = 0;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        const int i = tid + j * 256;
        const bool ex = i < p.halo_px * 4;
        const int hp = ex ? (i >> 2) : 0;
        const int im = hp / hpi;
        const int rem = hp - im * hpi;
        const int hy = rem / p.HW, hx = rem - hy * p.HW;
        geo[j] = (im << 20) | (hy << 10) | hx;
        ldsx[j] = piece * PPP * p.XP + hp;
        exist |= (ex ? 1u : 0u) << j;
    }
    // ---- weight staging items
    int wbase[NW], wlds[NW];
    unsigned wexist = 0, wvalid = 0;
    const int w_items = p.ntaps * BN * 4;
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const int i = tid + j * 256;
        const bool ex = i < w_items;
        const int col = (i >> 2) % BN;
        int t = (i >> 2) / BN;
        t = t < p.ntaps ? t : 0;
        const int n = n0 + col;
        bool ok = ex && n < p.CO;
        if (p.par_taps) {        // stride-2 input gradient: the weight tap depends on (gather tap, output parity)
            const int par = ok ? n / p.par_co : 0;
            const int tw = (int)((p.par_pack >> (4 * ((t & 3) * 4 + par))) & 15ull) - 1;
            ok = ok && tw >= 0;
            wbase[j] = ((ok ? tw : 0) * p.par_co + (ok ? n - par * p.par_co : 0)) * CI + piece * PIECE;
        } else {
            wbase[j] = (s_tap[16 + t] * p.CO + (ok ? n : 0)) * CI + piece * PIECE;
        }
        wlds[j] = (PPP == 1) ? (t * 4 + piece) * p.WP + col : (t * 16 + piece * 4) * p.WP + col;
        wexist |= (ex ? 1u : 0u) << j;
        wvalid |= (ok ? 1u : 0u) << j;
    }
    auto store_w = [&](const u32x4 (&wreg)[NW], int unit_off) {
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            if ((wexist >> j) & 1u) {
                u32x4 v = wreg[j];
                if (!((wvalid >> j) & 1u)) v = u32x4{0u, 0u, 0u, 0u};
                if constexpr (PIECE == 8) {
                    *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(Ws) + (size_t)(unit_off + wlds[j]) * 8) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        reinterpret_cast<float*>(Ws)[unit_off + wlds[j] + e * p.WP] = __uint_as_float(v[e]);
                }
            }
        }
    };
    if constexpr (WRES) {   // one-time staging of the whole weight slice
        for (int c0 = 0, ci = 0; c0 < CI; c0 += CK, ++ci) {
            u32x4 wreg[NW];
            const T* wp = reinterpret_cast<const T*>(p.w) + c0;
#pragma unroll
            for (int j = 0; j < NW; ++j) wreg[j] = *reinterpret_cast<const u32x4*>(wp + wbase[j]);
            store_w(wreg, ci * w_chunk_units);
        }
    }

    // ---- tile-invariant fragment geometry: this lane's pixel in each of its MA blocks
    int hpA[MA], ptx[MA], pty[MA], pim[MA];
#pragma unroll
    for (int a = 0; a < MA; ++a) {
        const int m = wave * 32 * MA + a * 32 + r;
        ptx[a] = m & (TW - 1); pty[a] = (m >> p.twl) & (TH - 1); pim[a] = m >> (p.twl + p.thl);
        hpA[a] = pim[a] < p.imgs ? pim[a] * hpi + pty[a] * p.IS * p.HW + ptx[a] * p.IS : 0;
    }

    const int txl = p.txl, tyl = p.tyl;             // log2(tiles_x), log2(tiles_y)
    const int ntiles = p.ntiles;
    const bool one_img = p.imgs == 1;

    // per-tile staging state
    int pix[NX], nimg[NX];
    unsigned inb_mask = 0;
    u32x4 xreg[NX];
    u32x4 wreg[NW];
    float scv[PIECE], shv[PIECE];
    int t_img0 = 0;

    auto tile_geometry = [&](int tile) {
        const int tile_x = tile & ((1 << txl) - 1);
        const int tile_y = (tile >> txl) & ((1 << tyl) - 1);
        const int ig = tile >> (txl + tyl);
        const int py0 = tile_y << p.thl, px0 = tile_x << p.twl;
        t_img0 = ig << p.iml;
        const int sy0 = py0 * p.IS + p.dymin, sx0 = px0 * p.IS + p.dxmin;
        inb_mask = 0;
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            const int im = geo[j] >> 20, hy = (geo[j] >> 10) & 1023, hx = geo[j] & 1023;
            const int n = t_img0 + im, sy = sy0 + hy, sx = sx0 + hx;
            const bool inb = ((exist >> j) & 1u) && (n < p.N) && (sy >= 0) && (sy < p.SH) && (sx >= 0) && (sx < p.SW);
            pix[j] = inb ? (n * p.SH + sy) * p.SW + sx : 0;
            nimg[j] = inb ? n : 0;
            inb_mask |= (inb ? 1u : 0u) << j;
        }
    };

    auto prefetch = [&](int c0) {
        const bool s1 = c0 >= p.C0;
        const T* src = reinterpret_cast<const T*>(s1 ? p.src1 : p.src0);
        const int Cs = s1 ? p.C1 : p.C0;
        const int cc = (s1 ? c0 - p.C0 : c0) + piece * PIECE;
#pragma unroll
        for (int j = 0; j < NX; ++j)
            xreg[j] = *reinterpret_cast<const u32x4*>(src + (size_t)pix[j] * Cs + cc);
        if constexpr (!WRES) {
            const T* wp = reinterpret_cast<const T*>(p.w) + c0;
#pragma unroll
            for (int j = 0; j < NW; ++j) wreg[j] = *reinterpret_cast<const u32x4*>(wp + wbase[j]);
        }
        if (!PLAIN && one_img) {
            const float* sc = s1 ? p.sc1 : p.sc0;
            const float* sh = s1 ? p.sh1 : p.sh0;
            const bool aff = sc != nullptr;
            const int nn = t_img0 < p.N ? t_img0 : 0;
            // no affine: read any valid memory and select the identity afterwards (keeps the loads unconditional)
            const float* scp = aff ? sc + (size_t)nn * Cs + cc : reinterpret_cast<const float*>(p.w);
            const float* shp = aff ? sh + (size_t)nn * Cs + cc : reinterpret_cast<const float*>(p.w);
#pragma unroll
            for (int e = 0; e < PIECE; ++e) {
                const float a = scp[e], b = shp[e];
                scv[e] = aff ? a : 1.f;
                shv[e] = aff ? b : 0.f;
            }
        }
    };

    auto commit = [&](int c0) {
        const bool s1 = c0 >= p.C0;
        const int Cs = s1 ? p.C1 : p.C0;
        const int cc = (s1 ? c0 - p.C0 : c0) + piece * PIECE;
        const float* sc = s1 ? p.sc1 : p.sc0;
        const float* sh = s1 ? p.sh1 : p.sh0;
        const float slope = s1 ? p.slope1 : p.slope0;
        if constexpr (PLAIN) {
#pragma unroll
            for (int j = 0; j < NX; ++j) {
                if ((exist >> j) & 1u) {
                    u32x4 v = xreg[j];
                    if (!((inb_mask >> j) & 1u)) v = u32x4{0u, 0u, 0u, 0u};
                    if constexpr (PIECE == 8) {
                        *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(Xs) + (size_t)ldsx[j] * 8) = v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            reinterpret_cast<float*>(Xs)[ldsx[j] + e * p.XP] = __uint_as_float(v[e]);
                    }
                }
            }
        } else {
#pragma unroll
        for (int j = 0; j < NX; ++j) {
            float v[PIECE];
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
            if (one_img) {
#pragma unroll
                for (int e = 0; e < PIECE; ++e) v[e] = v[e] * scv[e] + shv[e];
            } else if (sc != nullptr) {      // several small images per tile: per-item affine (tiny layers only)
                const float* scp = sc + (size_t)nimg[j] * Cs + cc;
                const float* shp = sh + (size_t)nimg[j] * Cs + cc;
#pragma unroll
                for (int e = 0; e < PIECE; ++e) v[e] = v[e] * scp[e] + shp[e];
            }
            const bool inb = (inb_mask >> j) & 1u;
#pragma unroll
            for (int e = 0; e < PIECE; ++e) {
                const float y = v[e] > 0.f ? v[e] : v[e] * slope;
                v[e] = inb ? y : 0.f;          // zero padding applies to the ACTIVATED tensor
            }
            if ((exist >> j) & 1u) {
                if constexpr (PIECE == 8) {
                    store_piece<bf16_t>(reinterpret_cast<bf16_t*>(Xs) + (size_t)ldsx[j] * 8, v);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) reinterpret_cast<float*>(Xs)[ldsx[j] + e * p.XP] = v[e];
                }
            }
        }
        }
        if constexpr (!WRES) store_w(wreg, 0);
    };

    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (own L2 each), so logical tile L -> tile
    // (L % 8) * ntiles/8 + L / 8: the workgroups running on one XCD at a time walk one contiguous eighth of the tiles and
    // the halo rows shared by vertically adjacent tiles are L2 hits instead of second HBM reads.
    const bool xcd_order = (ntiles & 7) == 0 && (gridDim.x & 7) == 0 && !CU_DBG(p, 64);
    auto tile_of = [&](int l) { return xcd_order ? (l & 7) * (ntiles >> 3) + (l >> 3) : l; };
    int ltile = blockIdx.x;
    int tile = tile_of(ltile);
    const int c_begin = p.ksplit > 1 ? (int)blockIdx.z * p.kspan : 0;
    const int c_end = p.ksplit > 1 ? (c_begin + p.kspan < CI ? c_begin + p.kspan : CI) : CI;
    if (ltile < ntiles) {
        tile_geometry(tile);
        prefetch(c_begin);
    }
    while (ltile < ntiles) {
        // origin of the tile being computed (the staging state may move on to the next tile before the epilogue)
        const int cur_tile_x = tile & ((1 << txl) - 1), cur_tile_y = (tile >> txl) & ((1 << tyl) - 1);
        const int cur_img0 = (tile >> (txl + tyl)) << p.iml;
        const int py0 = cur_tile_y << p.thl, px0 = cur_tile_x << p.twl;
        const int lnext = ltile + gridDim.x;
        const int next = tile_of(lnext);

        f32x16 acc[MA][NB];
#pragma unroll
        for (int a = 0; a < MA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

        for (int c0 = c_begin, ci = c_begin / CK; c0 < c_end; c0 += CK, ++ci) {
            __syncthreads();          // previous chunk's fragment reads are done
            if (!CU_DBG(p, 4)) commit(c0);
            __syncthreads();
            if (!CU_DBG(p, 8)) {
            if (c0 + CK < c_end) {
                prefetch(c0 + CK);
            } else if (lnext < ntiles) {  // cross-tile prefetch: the next tile's first chunk flies under these MFMAs
                tile_geometry(next);
                prefetch(c_begin);
            }
            }
            const int wbase_units = WRES ? ci * w_chunk_units : 0;
            auto tap_step = [&](int t, int kk) {
                const int toff = s_tap[t];
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
                            W16 + ((size_t)wbase_units + (t * 4 + 2 * kk + h) * p.WP + b * 32 + r) * 8);
#pragma unroll
                    for (int a = 0; a < MA; ++a)
#pragma unroll
                        for (int b = 0; b < NB; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[b], af[a], acc[a][b], 0, 0, 0);
                } else {
                    const float* Xf = reinterpret_cast<const float*>(Xs);
                    const float* Wf = reinterpret_cast<const float*>(Ws);
                    float af[MA], bfr[NB];
#pragma unroll
                    for (int a = 0; a < MA; ++a) af[a] = Xf[(2 * kk + h) * p.XP + hpA[a] + toff];
#pragma unroll
                    for (int b = 0; b < NB; ++b)
                        bfr[b] = Wf[wbase_units + (t * 16 + 2 * kk + h) * p.WP + b * 32 + r];
#pragma unroll
                    for (int a = 0; a < MA; ++a)
#pragma unroll
                        for (int b = 0; b < NB; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(bfr[b], af[a], acc[a][b], 0, 0, 0);
                }
            };
            if (!CU_DBG(p, 2)) {
                if constexpr (NT > 0) {      // exact tap count: branch-free, the scheduler hoists the fragment reads
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int kk = 0; kk < C::KSTEPS; ++kk) tap_step(t, kk);
                } else {
                    for (int t = 0; t < p.ntaps; ++t)
#pragma unroll
                        for (int kk = 0; kk < C::KSTEPS; ++kk) tap_step(t, kk);
                }
            }
        }

        // ---- epilogue.  D[row = column][col = pixel]: lane -> pixel (lane & 31); register i -> column
        //      (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5): each register quad is 4 consecutive output channels.
#pragma unroll
        for (int a = 0; a < MA; ++a) {
            const int n = cur_img0 + pim[a];
            const int py = py0 + pty[a], px = px0 + ptx[a];
            const bool pvalid = !(pim[a] >= p.imgs || n >= p.N || py >= p.PH || px >= p.PW || CU_DBG(p, 1));
            const int oy = py * p.OS + p.OY0, ox = px * p.OS + p.OX0;
            const size_t opix0 = pvalid ? ((size_t)n * p.OH + oy) * p.OW + ox : 0;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                int colb = n0 + b * 32;
                if (colb >= p.CO) continue;                       // uniform
                size_t opix = opix0;
                if (p.par_co > 0) {          // transposed conv: column group -> output parity
                    const int par = colb / p.par_co;
                    colb -= par * p.par_co;
                    if (pvalid) opix += (size_t)(par >> 1) * p.OW + (par & 1);
                }
                const bool d1 = colb >= p.D0;
                const int accum = d1 ? p.accum1 : p.accum0;
                // split-K partial tile: plain f32 stores into this split's slice of the workspace (laid out like the
                // destination); the finish pass sums the slices in a fixed order and applies bias / rounding / accumulate
                if (p.ksplit > 1) {
                    if (pvalid) {
                        const int DC = d1 ? p.DC1 : p.DC0;
                        float* o = (d1 ? p.ws1 : p.ws0) + (size_t)blockIdx.z * p.ws_slice + opix * DC +
                                   (d1 ? colb - p.D0 : colb) + 4 * h;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            // the arithmetic no-op moves the value out of the accumulator file: storing the accumulator
                            // registers of this block directly makes hipcc emit an illegal instruction in some instances
                            f32x4 ov;
#pragma unroll
                            for (int e = 0; e < 4; ++e) ov[e] = acc[a][b][4 * g + e] + 0.f * (float)p.ksplit;
                            *reinterpret_cast<f32x4*>(o + 8 * g) = ov;
                        }
                    }
                    continue;
                }
                if constexpr (sizeof(T) == 2) {
                    if (!p.out_nchw && !accum) {
                        // bf16 fast path: lanes l and l+32 hold interleaved channel quads of the same pixel; two
                        // v_permlane32_swap per quad pair give each lane 16 CONSECUTIVE channels -> two 16-byte stores
                        // (cdna_hip_programming.md T21).  EXEC is full here; only the stores are predicated.
                        unsigned q[4][2];
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * g + e];
                            if (p.bias) {
                                const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + colb + 8 * g + 4 * h);
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
                        if (pvalid) {
                            const int dcol = (d1 ? colb - p.D0 : colb) + 16 * h;
                            const int DC = d1 ? p.DC1 : p.DC0;
                            bf16_t* o = reinterpret_cast<bf16_t*>(d1 ? p.dst1 : p.dst0) + opix * DC + dcol;
                            *reinterpret_cast<u32x4*>(o) = u32x4{q[0][0], q[0][1], q[2][0], q[2][1]};
                            *reinterpret_cast<u32x4*>(o + 8) = u32x4{q[1][0], q[1][1], q[3][0], q[3][1]};
                        }
                        continue;
                    }
                }
                if (!pvalid) continue;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int col = colb + 8 * g + 4 * h;
                    if (col >= p.CO) continue;        // ragged last block (f32: 16-channel layers of the `vital` U-Net)
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * g + e];
                    if (p.bias) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + col);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += bv[e];
                    }
                    if (p.out_nchw) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (col + e < p.DC0) {
                                float* o = reinterpret_cast<float*>(p.dst0) +
                                           (((size_t)n * p.DC0 + col + e) * p.OH + oy) * p.OW + ox;
                                *o = p.accum0 ? *o + v[e] : v[e];
                            }
                        }
                    } else {
                        const int dcol = d1 ? col - p.D0 : col;
                        const int DC = d1 ? p.DC1 : p.DC0;
                        T* o = reinterpret_cast<T*>(d1 ? p.dst1 : p.dst0) + opix * DC + dcol;
                        if constexpr (sizeof(T) == 2) {
                            if (accum) {
                                const u32x2 old = *reinterpret_cast<const u32x2*>(o);
                                v[0] += __uint_as_float(old[0] << 16); v[1] += __uint_as_float(old[0] & 0xffff0000u);
                                v[2] += __uint_as_float(old[1] << 16); v[3] += __uint_as_float(old[1] & 0xffff0000u);
                            }
                            u32x2 pk;
                            pk[0] = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                            pk[1] = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                            *reinterpret_cast<u32x2*>(o) = pk;
                        } else {
                            f32x4 ov = {v[0], v[1], v[2], v[3]};
                            if (accum) {
                                const f32x4 old = *reinterpret_cast<const f32x4*>(o);
                                ov += old;
                            }
                            *reinterpret_cast<f32x4*>(o) = ov;
                        }
                    }
                }
            }
        }
        tile = next;
        ltile = lnext;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Wide layers (bf16, 3x3 stride-1 / stride-2 gathers, weights too large to stay in LDS): 8 waves, tile = 512 (256 for
// stride 2) pixels x 128 columns.
//   * the weight slice of a 32-channel chunk (9 x 128 x 64 B = 72 KiB) is streamed once per 512 pixels (the 4-wave
//     kernel above re-streams it per 256: on these layers its run time is the L2 traffic of the weights);
//   * staging is LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, so a wave needs < 256 VGPRs and two waves
//     share a SIMD - one wave's MFMAs cover the other wave's LDS latency and DMA issue;
//   * LDS images are lane-linear rows of 64 bytes ([pixel][4 pieces] and [tap][column][4 pieces]); slot s of row i
//     holds piece s ^ ((i >> 2) & 3), which makes the ds_read_b128 fragment reads of 32 consecutive rows conflict-free
//     (lane groups of ds_read_b128: MI355X_MICROARCH.md, LDS); zero padding comes from the buffer range check.
constexpr int DW = 8;                 // waves

// ---- InstanceNorm statistics of the finished tile (cu_conv_epilogue mode 1 on the LDS-DMA kernels) ------------------------
// A lane owns one pixel; register i of a column block is channel (i & 3) + 8 (i >> 2) + 4 h.  Summing a register over the
// 32 pixel lanes with a plain butterfly costs 5 cross-lane steps per register (80 per block and statistic, each an LDS
// permute: measured +12 us on a 64^2 x 128-channel layer, as much as the statistics pass it removed).  The halving
// butterfly below exchanges HALF the registers per step (lane pairs split the register set between them): 8 + 4 + 2 + 1
// exchanges bring the 16 registers down to one per lane -- in DPP / permlane-swap form (no LDS round trip), one last LDS
// permute joins the lane pairs the DPP patterns cannot reach.  The totals end up one channel per lane.
// Lane l then holds register i = (l & 1) << 3 | (l & 2) << 1 | (l & 8) >> 2 | (l & 16) >> 4 (bit 2 of l: replicated).
// All 8 waves call this after their MFMAs (it starts with a barrier: the LDS is reused).  valid[a]: block a holds pixels
// of an existing image; img_l: the wave's image inside the tile (wave-uniform: the host only enables the statistics when
// a wave's 32 * MA pixels lie in one image).  Every wave stores its 2 x BN column totals into a slot of its own; after one
// barrier the workgroup adds the slots of each image and issues one global atomic per (image, column, statistic).
template <int MA, int NB>
__device__ __forceinline__ void tile_stats(const f32x16 (&acc)[MA][NB], const bool (&valid)[MA], int img_l, float* lds,
                                           float* __restrict__ sums, int img0, int imgs, int N, int CO, int n0, int wpi) {
    constexpr int BN = 32 * NB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    __syncthreads();
    const int reg = ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 8) >> 2) | ((lane & 16) >> 4);
    float* mine = lds + wave * (2 * BN);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float s[16], q[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s[i] = q[i] = 0.f;
#pragma unroll
            for (int a = 0; a < MA; ++a) {
                const float v = valid[a] ? acc[a][b][i] : 0.f;
                s[i] += v;
                q[i] = fmaf(v, v, q[i]);
            }
        }
        const float st = lane_reduce16(s, lane), qt = lane_reduce16(q, lane);
        if (!(lane & 4)) {
            const int col = b * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            mine[2 * col] = st;
            mine[2 * col + 1] = qt;
        }
    }
    __syncthreads();
    // wpi waves per image (8 / imgs': 8 when the tile is one image or part of one, 4 with two images per tile)
    for (int i = tid; i < (DW / wpi) * 2 * BN; i += 64 * DW) {
        const int img = i / (2 * BN), rem = i - img * 2 * BN, col = rem >> 1;
        const int n = img0 + img;
        float t = 0.f;
        for (int w = 0; w < wpi; ++w) t += lds[(img * wpi + w) * (2 * BN) + rem];
        if (img < imgs && n < N && n0 + col < CO) unsafeAtomicAdd(sums + ((size_t)n * CO + n0 + col) * 2 + (rem & 1), t);
    }
}
constexpr int RING_MF16_DEFAULT = 0;   // round 4 A/B: profiles/r04_mf16_ring.txt
constexpr int DMA_MAXX = 9;           // halo items per thread: halo_px * 4 <= 512 * 9 (stride-2 gathers: 17 x 65 pixels)
// MA = 32-pixel blocks per wave (tile = 256 * MA pixels): 2 for stride-1 gathers, 1 for stride-2 gathers, whose halo patch
// is four times larger per pixel.
template <int NTAPS, int NB, int MA>
__global__ __launch_bounds__(64 * DW) void igemm_conv_dma_kernel(const ConvKArgs p, unsigned src0_bytes, unsigned src1_bytes,
                                                                 unsigned w_bytes) {
    constexpr int BN = 32 * NB, CK = 32;
    constexpr int WROWS = NTAPS * BN, WROUNDS = (WROWS + 127) / 128;     // 128 weight rows of 64 B per staging round
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int TW = 1 << p.twl, TH = 1 << p.thl;
    const int n0 = blockIdx.y * BN;
    const int CI = p.C0 + p.C1;
    const int hpi = p.HH * p.HW;
    const unsigned x_base = lds_addr(smem);
    const unsigned w_base = x_base + (unsigned)p.halo_px * 64u;     // host rounds halo_px up to a multiple of 16 rows
    const i32x4 rs0 = make_rsrc(p.src0, src0_bytes);
    const i32x4 rs1 = make_rsrc(p.src1 ? p.src1 : p.src0, p.src1 ? src1_bytes : 0u);
    const i32x4 rw = make_rsrc(p.w, w_bytes);
    constexpr unsigned OOB = 0x7ffffff0u;

    // ---- tile origin (one tile per workgroup)
    const int tile = blockIdx.x;
    const int tile_x = tile & ((1 << p.txl) - 1), tile_y = (tile >> p.txl) & ((1 << p.tyl) - 1);
    const int img0 = (tile >> (p.txl + p.tyl)) << p.iml;
    const int py0 = tile_y << p.thl, px0 = tile_x << p.twl;

    // ---- halo staging items of this thread: pixel (tid >> 2) + 128 j, slot tid & 3
    const int slot = tid & 3;
    unsigned xoff[DMA_MAXX];          // byte offset of the pixel inside a source with 1 channel (x Cs later), or OOB
    int xpiece[DMA_MAXX];
#pragma unroll
    for (int j = 0; j < DMA_MAXX; ++j) {
        const int hp = (tid >> 2) + 128 * j;
        const int im = hp / hpi, rem = hp - im * hpi;
        const int hy = rem / p.HW, hx = rem - hy * p.HW;
        const int n = img0 + im, sy = py0 * p.IS + p.dymin + hy, sx = px0 * p.IS + p.dxmin + hx;
        const bool ok = hp < p.imgs * hpi && n < p.N && sy >= 0 && sy < p.SH && sx >= 0 && sx < p.SW;
        xoff[j] = ok ? (unsigned)((n * p.SH + sy) * p.SW + sx) : 0xffffffffu;
        xpiece[j] = slot ^ ((hp >> 2) & 3);
    }
    // ---- weight staging items: row (tid >> 2) + 128 j of the [tap][column] image, slot tid & 3
    const int wrow0 = tid >> 2;
    const int wcol = wrow0 & (BN - 1);
    const bool wok = n0 + wcol < p.CO;
    const int wpiece = slot ^ ((wrow0 >> 2) & 3);          // 128 and BN are multiples of 16: the swizzle is round-invariant

    // ---- fragment geometry: this lane's pixel in each of its MA blocks; weight rows of its NB column blocks
    int hpA[MA], ptx[MA], pty[MA], pim[MA];
#pragma unroll
    for (int a = 0; a < MA; ++a) {
        const int m = wave * 32 * MA + a * 32 + r;
        ptx[a] = m & (TW - 1); pty[a] = (m >> p.twl) & (TH - 1); pim[a] = m >> (p.twl + p.thl);
        hpA[a] = pim[a] < p.imgs ? pim[a] * hpi + pty[a] * p.IS * p.HW + ptx[a] * p.IS : 0;
    }
    const int wsw = (r >> 2) & 3;     // swizzle of weight row (b * 32 + r): ((b * 32 + r) >> 2) & 3

    f32x16 acc[MA][NB];
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    const bf16_t* X16 = reinterpret_cast<const bf16_t*>(smem);
    const bf16_t* W16 = reinterpret_cast<const bf16_t*>(smem + (size_t)p.halo_px * 64);

    for (int c0 = 0; c0 < CI; c0 += CK) {
        __syncthreads();              // the previous chunk's fragment reads are done
        {
            const bool s1 = c0 >= p.C0;
            const unsigned Cs = (unsigned)(s1 ? p.C1 : p.C0);
            const unsigned cc = (unsigned)(s1 ? c0 - p.C0 : c0);
#pragma unroll
            for (int j = 0; j < DMA_MAXX; ++j) {
                if (j * 128 < p.halo_px && !(CU_DBG(p, 8) && c0)) {       // uniform
                    const unsigned off = xoff[j] == 0xffffffffu ? OOB : (xoff[j] * Cs + cc + (unsigned)xpiece[j] * 8u) * 2u;
                    const unsigned dst = x_base + (unsigned)(j * 8192 + wave * 1024);
                    if (s1) dma16(rs1, off, dst);
                    else dma16(rs0, off, dst);
                }
            }
#pragma unroll
            for (int j = 0; j < WROUNDS; ++j) {
                // tap of row wrow0 + 128 j: static indices only (a runtime index would spill the argument block)
                int tw;
                if constexpr (NB == 4) {
                    tw = p.tap_w[j < NTAPS ? j : 0];
                } else if constexpr (NB == 1) {      // four taps per round of 128 rows: tap 4 j + (tid >> 7)
                    const int T0 = 4 * j < NTAPS ? 4 * j : 0, T1 = 4 * j + 1 < NTAPS ? 4 * j + 1 : 0;
                    const int T2 = 4 * j + 2 < NTAPS ? 4 * j + 2 : 0, T3 = 4 * j + 3 < NTAPS ? 4 * j + 3 : 0;
                    const int q4 = tid >> 7;
                    tw = q4 == 0 ? p.tap_w[T0] : (q4 == 1 ? p.tap_w[T1] : (q4 == 2 ? p.tap_w[T2] : p.tap_w[T3]));
                } else {
                    const int T0 = 2 * j < NTAPS ? 2 * j : 0, T1 = 2 * j + 1 < NTAPS ? 2 * j + 1 : 0;   // constants after unrolling
                    tw = (tid >> 8) ? p.tap_w[T1] : p.tap_w[T0];
                }
                const bool ok = wok && wrow0 + 128 * j < WROWS;
                const unsigned off = ok ? (unsigned)(((tw * p.CO + n0 + wcol) * CI + c0 + wpiece * 8)) * 2u : OOB;
                if (!(CU_DBG(p, 4) && c0)) dma16(rw, off, w_base + (unsigned)(j * 8192 + wave * 1024));
            }
        }
        dma_wait();
        __syncthreads();
        if (CU_DBG(p, 2)) continue;
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            int xrow[MA], xsw[MA];
#pragma unroll
            for (int a = 0; a < MA; ++a) {
                xrow[a] = hpA[a] + p.tap_off[t];
                xsw[a] = (xrow[a] >> 2) & 3;
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 af[MA], bfr[NB];
#pragma unroll
                for (int a = 0; a < MA; ++a)
                    af[a] = *reinterpret_cast<const bf16x8*>(X16 + ((size_t)xrow[a] * 4 + ((2 * kk + h) ^ xsw[a])) * 8);
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    bfr[b] = *reinterpret_cast<const bf16x8*>(
                        W16 + ((size_t)(t * BN + b * 32 + r) * 4 + ((2 * kk + h) ^ wsw)) * 8);
#pragma unroll
                for (int a = 0; a < MA; ++a)
#pragma unroll
                    for (int b = 0; b < NB; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[b], af[a], acc[a][b], 0, 0, 0);
            }
        }
    }

    // ---- epilogue (same register -> channel mapping as igemm_conv_kernel)
#pragma unroll
    for (int a = 0; a < MA; ++a) {
        const int n = img0 + pim[a];
        const int py = py0 + pty[a], px = px0 + ptx[a];
        const bool pvalid = !(pim[a] >= p.imgs || n >= p.N || py >= p.PH || px >= p.PW || CU_DBG(p, 1));
        const int oy = py * p.OS + p.OY0, ox = px * p.OS + p.OX0;
        const size_t opix = pvalid ? ((size_t)n * p.OH + oy) * p.OW + ox : 0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int colb = n0 + b * 32;
            if (colb >= p.CO) continue;                       // uniform
            const bool d1 = colb >= p.D0;
            const int accum = d1 ? p.accum1 : p.accum0;
            const int DC = d1 ? p.DC1 : p.DC0;
            bf16_t* dstp = reinterpret_cast<bf16_t*>(d1 ? p.dst1 : p.dst0);
            if (!accum) {
                unsigned q[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * g + e];
                    if (p.bias) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + colb + 8 * g + 4 * h);
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
                if (pvalid) {
                    const int dcol = (d1 ? colb - p.D0 : colb) + 16 * h;
                    bf16_t* o = dstp + opix * DC + dcol;
                    *reinterpret_cast<u32x4*>(o) = u32x4{q[0][0], q[0][1], q[2][0], q[2][1]};
                    *reinterpret_cast<u32x4*>(o + 8) = u32x4{q[1][0], q[1][1], q[3][0], q[3][1]};
                }
                continue;
            }
            if (!pvalid) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = colb + 8 * g + 4 * h;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * g + e];
                if (p.bias) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + col);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += bv[e];
                }
                bf16_t* o = dstp + opix * DC + (d1 ? col - p.D0 : col);
                const u32x2 old = *reinterpret_cast<const u32x2*>(o);
                v[0] += __uint_as_float(old[0] << 16); v[1] += __uint_as_float(old[0] & 0xffff0000u);
                v[2] += __uint_as_float(old[1] << 16); v[3] += __uint_as_float(old[1] & 0xffff0000u);
                u32x2 pk;
                pk[0] = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                pk[1] = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                *reinterpret_cast<u32x2*>(o) = pk;
            }
        }
    }
    if (p.stat_sums) {
        bool valid[MA];
#pragma unroll
        for (int a = 0; a < MA; ++a) valid[a] = pim[a] < p.imgs;
        tile_stats<MA, NB>(acc, valid, (wave * 32 * MA) >> (p.twl + p.thl), reinterpret_cast<float*>(smem), p.stat_sums, img0,
                           p.imgs, p.N, p.CO, n0, p.imgs == 2 ? DW / 2 : DW);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same tile (512 pixels x 128 columns, 8 waves, 3x3 stride-1 taps) with the staging of chunk c+1 in flight under the
// MFMAs of chunk c (tools/dma_abl.py: the kernel above spends 14 of 71 us on 32^2 x 256 waiting for its
// per-chunk LDS-DMA because no second 111-KiB image fits).  What does fit in 160 KiB: TWO halo images (<= 40 KiB each)
// and a ring of THREE weight slots of one tap row (3 taps x 128 columns x 64 B = 24 KiB).  An iteration = (chunk, tap
// row g): wait for its stage with a counted vmcnt, ONE barrier (publishes the stage, frees slot (g+2) % 3), issue the
// weights of the iteration after next plus a third of the next chunk's halo image -- always six DMA instructions per
// thread, out-of-range ones (into an 8-KiB dump area) where there is nothing to fetch, so the wait count is a constant.
// NB = 4: 128 columns, one staging round (128 rows) per tap.  NB = 2 (round 3): 64 columns -- the layers whose column count or
// workgroup count rules the 128-column tile out (128^2 concat 64+64 -> 64; 16^2 x 480 channels, where 128 columns leave half the
// CUs without a workgroup) ran on the non-overlapped kernel above at ~700 TFLOP/s with their per-chunk LDS-DMA (3.4 us) and
// MFMAs (3.4 us) in series.  A tap row of 64 columns is 12 KiB: two staging rounds per tap row (taps 0, 1 | tap 2 + padding),
// five DMA instructions per thread and iteration instead of six.  NXR = staging rounds of the halo image (5: one image of
// 16 x 32 pixels; 6: two images of 16 x 16).
//
// MF16 (round 4): the same tile on v_mfma_f32_16x16x32_bf16 -- one MFMA takes the whole 32-channel chunk of a
// 16-pixel x 16-column sub-block (four per 32 x 32 block: the same MFMA cycles and the same twelve 16-byte fragment reads per
// tap as the 32x32x16 form, but the chip holds a higher clock on this shape: MI355X_MICROARCH.md, DVFS give-back item 7).
// Lane l reads LDS row (l & 15) of its sub-block and k-group q = l >> 4; on the SAME swizzled images the four k-groups take the
// logical 16-byte pieces {0, 3, 1, 2} (for the weights and the pixels alike, so the contraction is unchanged): with that
// assignment every one of ds_read_b128's four hardware lane groups ({0-3, 12-15, 20-27}, ...) touches all 64 banks once.
// C/D: lane = pixel (l & 15), registers = columns 4 q .. 4 q + 3 -> 8-byte epilogue stores (16-byte ones in the 32x32 form).
template <int NB, int NXR, bool MF16 = false>
__global__ __launch_bounds__(64 * DW) void igemm_conv_dma_ring_kernel(const ConvKArgs p, unsigned src0_bytes,
                                                                     unsigned src1_bytes, unsigned w_bytes) {
    constexpr int MA = 2, BN = 32 * NB, CK = 32;
    constexpr int WR = NB == 4 ? 3 : 2;                            // weight staging rounds (DMA instructions) per tap row
    constexpr int IT = WR + 3;                                     // DMA instructions per thread and iteration
    constexpr unsigned XB = NXR * 8192, WSLOT = WR * 8192;
    static_assert((NB == 4 || NB == 2) && NXR >= 3 && NXR <= 6, "ring kernel: instances");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int TW = 1 << p.twl, TH = 1 << p.thl;
    const int n0 = blockIdx.y * BN;
    const int CI = p.C0 + p.C1;
    const int hpi = p.HH * p.HW;
    const unsigned x_base = lds_addr(smem), w_base = x_base + 2 * XB, dump = w_base + 3 * WSLOT;
    const i32x4 rs0 = make_rsrc(p.src0, src0_bytes);
    const i32x4 rs1 = make_rsrc(p.src1 ? p.src1 : p.src0, p.src1 ? src1_bytes : 0u);
    const i32x4 rw = make_rsrc(p.w, w_bytes);
    constexpr unsigned OOB = 0x7ffffff0u;

    const int tile = blockIdx.x;
    const int tile_x = tile & ((1 << p.txl) - 1), tile_y = (tile >> p.txl) & ((1 << p.tyl) - 1);
    const int img0 = (tile >> (p.txl + p.tyl)) << p.iml;
    const int py0 = tile_y << p.thl, px0 = tile_x << p.twl;

    const int slot = tid & 3;
    unsigned xoff[NXR];
    int xpiece[NXR];
#pragma unroll
    for (int j = 0; j < NXR; ++j) {
        const int hp = (tid >> 2) + 128 * j;
        const int im = hp / hpi, rem = hp - im * hpi;
        const int hy = rem / p.HW, hx = rem - hy * p.HW;
        const int n = img0 + im, sy = py0 + p.dymin + hy, sx = px0 + p.dxmin + hx;
        const bool ok = hp < p.imgs * hpi && n < p.N && sy >= 0 && sy < p.SH && sx >= 0 && sx < p.SW;
        xoff[j] = ok ? (unsigned)((n * p.SH + sy) * p.SW + sx) : 0xffffffffu;
        xpiece[j] = slot ^ ((hp >> 2) & 3);
    }
    // weight rows of a staging round: NB = 4: 128 columns of one tap; NB = 2: row (tid >> 2) + 128 j = (tap, column) pair
    const int wcol = (tid >> 2) & (BN - 1);
    const bool wok = n0 + wcol < p.CO;
    const int wpiece = slot ^ ((wcol >> 2) & 3);

    int hpA[MA], ptx[MA], pty[MA], pim[MA];
#pragma unroll
    for (int a = 0; a < MA; ++a) {
        const int m = wave * 32 * MA + a * 32 + r;
        ptx[a] = m & (TW - 1); pty[a] = (m >> p.twl) & (TH - 1); pim[a] = m >> (p.twl + p.thl);
        hpA[a] = pim[a] < p.imgs ? pim[a] * hpi + pty[a] * p.HW + ptx[a] : 0;
    }
    const int wsw = (r >> 2) & 3;

    f32x16 acc[MF16 ? 1 : MA][MF16 ? 1 : NB];
    if constexpr (!MF16) {
#pragma unroll
        for (int a = 0; a < MA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    }
    // MF16: sub-block (a, ph) = pixels a * 32 + ph * 16 + (lane & 15) of the wave's 64; (b, ch) = columns b * 32 + ch * 16 + ...
    const int l15 = lane & 15, kq = lane >> 4;
    const int kpiece = kq == 0 ? 0 : (kq == 1 ? 3 : (kq == 2 ? 1 : 2));
    const int wlane16 = (l15 * 4 + (kpiece ^ ((l15 >> 2) & 3))) * 8;      // element offset of this lane's piece in a weight row block
    int hp16[MA][2];
    f32x4 acc16[MF16 ? MA : 1][2][MF16 ? NB : 1][2];
    if constexpr (MF16) {
#pragma unroll
        for (int a = 0; a < MA; ++a)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const int m = wave * 32 * MA + a * 32 + ph * 16 + l15;
                const int tx = m & (TW - 1), ty = (m >> p.twl) & (TH - 1), im = m >> (p.twl + p.thl);
                hp16[a][ph] = im < p.imgs ? im * hpi + ty * p.HW + tx : 0;
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) acc16[a][ph][b][ch] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    }

    const int nch = CI / CK;
    // rounds [j0, j1) of chunk c's halo image into image c & 1; rounds beyond the image / chunks beyond the last go,
    // out of range, to the dump area: the instruction count stays the same
    auto issue_x = [&](int c, int j0, int j1) {
        const bool live = c < nch;
        const int c0 = live ? c * CK : 0;
        const bool s1 = c0 >= p.C0;
        const unsigned Cs = (unsigned)(s1 ? p.C1 : p.C0);
        const unsigned cc = (unsigned)(s1 ? c0 - p.C0 : c0);
#pragma unroll
        for (int j = 0; j < NXR; ++j) {
            if (j >= j0 && j < j1) {
                const unsigned off = (!live || xoff[j] == 0xffffffffu) ? OOB : (xoff[j] * Cs + cc + (unsigned)xpiece[j] * 8u) * 2u;
                const unsigned dst = x_base + (unsigned)((c & 1) * XB + j * 8192 + wave * 1024);
                if (s1) dma16(rs1, off, dst);
                else dma16(rs0, off, dst);
            }
        }
    };
    auto issue_dump = [&](int n) {
        for (int j = 0; j < n; ++j) dma16(rw, OOB, dump + (unsigned)(wave * 1024));
    };
    // the three taps of tap row G (static) of chunk c into weight slot G
    auto issue_w = [&](int c, auto G) {
        constexpr int g = decltype(G)::value;
        const bool live = c < nch;
#pragma unroll
        for (int j = 0; j < WR; ++j) {
            // the tap this thread's row of round j belongs to (NB = 4: j; NB = 2: two taps per round, the last half round empty)
            // (static indices into the by-value argument block only: a lane-dependent one would move it to scratch)
            const int tl = NB == 4 ? j : 2 * j + (tid >> 8);
            const int tw_a = p.tap_w[3 * g + (NB == 4 ? j : (2 * j < 3 ? 2 * j : 0))];
            const int tw_b = p.tap_w[3 * g + (NB == 4 ? j : (2 * j + 1 < 3 ? 2 * j + 1 : 0))];
            const int tw = (NB == 4 || tid < 256) ? tw_a : tw_b;
            const unsigned off = (live && wok && tl < 3) ? (unsigned)(((tw * p.CO + n0 + wcol) * CI + c * CK + wpiece * 8)) * 2u : OOB;
            dma16(rw, off, w_base + (unsigned)(g * WSLOT + j * 8192 + wave * 1024));
        }
    };
    auto compute = [&](int c, auto G) {
        constexpr int g = decltype(G)::value;
        const bf16_t* X16 = reinterpret_cast<const bf16_t*>(smem + (size_t)(c & 1) * XB);
        const bf16_t* W16 = reinterpret_cast<const bf16_t*>(smem + 2 * (size_t)XB + (size_t)g * WSLOT);
        if constexpr (MF16) {
#pragma unroll
            for (int tl = 0; tl < 3; ++tl) {
                bf16x8 xf[MA][2];
#pragma unroll
                for (int a = 0; a < MA; ++a)
#pragma unroll
                    for (int ph = 0; ph < 2; ++ph) {
                        // (opaque per iteration: hoisted out of the chunk loop, the 36 row addresses of a chunk spilled the tile)
                        int hb = hp16[a][ph];
                        asm volatile("" : "+v"(hb));
                        const int row = hb + p.tap_off[3 * g + tl];
                        xf[a][ph] = *reinterpret_cast<const bf16x8*>(X16 + ((size_t)row * 4 + (kpiece ^ ((row >> 2) & 3))) * 8);
                    }
                // one 32-column block at a time: 4 pixel + 2 weight fragments live (all 8 weight fragments of a 128-column tile
                // at once spilled: profiles/r04_mf16_ring_v1_all_fragments_live.txt, 1.9x slower).  Weight rows: the swizzle of row
                // (tap, column block, 16 ch + l15) is ((l15 >> 2) & 3) whatever the block: one lane base + immediate offsets
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    bf16x8 wf[2];
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch)
                        wf[ch] = *reinterpret_cast<const bf16x8*>(W16 + wlane16 + (size_t)(tl * BN + b * 32 + ch * 16) * 32);
#pragma unroll
                    for (int a = 0; a < MA; ++a)
#pragma unroll
                        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                            for (int ch = 0; ch < 2; ++ch)
                                acc16[a][ph][b][ch] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ch], xf[a][ph], acc16[a][ph][b][ch], 0, 0, 0);
                }
            }
            return;
        }
#pragma unroll
        for (int tl = 0; tl < 3; ++tl) {
            int xrow[MA], xsw[MA];
#pragma unroll
            for (int a = 0; a < MA; ++a) {
                xrow[a] = hpA[a] + p.tap_off[3 * g + tl];
                xsw[a] = (xrow[a] >> 2) & 3;
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 af[MA], bfr[NB];
#pragma unroll
                for (int a = 0; a < MA; ++a)
                    af[a] = *reinterpret_cast<const bf16x8*>(X16 + ((size_t)xrow[a] * 4 + ((2 * kk + h) ^ xsw[a])) * 8);
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    bfr[b] = *reinterpret_cast<const bf16x8*>(
                        W16 + ((size_t)(tl * BN + b * 32 + r) * 4 + ((2 * kk + h) ^ wsw)) * 8);
#pragma unroll
                for (int a = 0; a < MA; ++a)
#pragma unroll
                    for (int b = 0; b < NB; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[b], af[a], acc[a][b], 0, 0, 0);
            }
        }
    };
    using G0 = std::integral_constant<int, 0>;
    using G1 = std::integral_constant<int, 1>;
    using G2 = std::integral_constant<int, 2>;

    // prologue: X(0), W(0, 0), W(0, 1)
    issue_x(0, 0, NXR);
    issue_w(0, G0{});
    issue_w(0, G1{});
    for (int c = 0; c < nch; ++c) {
        // ---- tap row 0: needs X(c) and W(c, 0); younger: W(c, 1) [first chunk] / the IT instructions of the iteration before
        if (c == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WR) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IT) : "memory");
        __syncthreads();
        issue_w(c, G2{});
        issue_x(c + 1, 0, 3);
        compute(c, G0{});
        // ---- tap row 1
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IT) : "memory");
        __syncthreads();
        issue_w(c + 1, G0{});
        issue_x(c + 1, 3, NXR);
        issue_dump(3 - (NXR - 3));
        compute(c, G1{});
        // ---- tap row 2
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IT) : "memory");
        __syncthreads();
        issue_w(c + 1, G1{});
        issue_dump(3);
        compute(c, G2{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing out-of-range DMA instructions target this LDS

    if constexpr (MF16) {      // lane = pixel (l & 15) of a sub-block, registers = 4 consecutive columns: 8-byte stores
#pragma unroll
        for (int a = 0; a < MA; ++a)
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                const int m = wave * 32 * MA + a * 32 + ph * 16 + l15;       // (the pixel geometry is recomputed here: kept live
                const int im = m >> (p.twl + p.thl);                          //  through the main loop it spilled the 128-column tile)
                const int n = img0 + im;
                const int py = py0 + ((m >> p.twl) & (TH - 1)), px = px0 + (m & (TW - 1));
                const bool pvalid = !(im >= p.imgs || n >= p.N || py >= p.PH || px >= p.PW);
                const size_t opix = pvalid ? ((size_t)n * p.OH + py) * p.OW + px : 0;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const int colb = n0 + b * 32;
                    if (colb >= p.CO) continue;                       // uniform
                    const bool d1 = colb >= p.D0;
                    const int accum = d1 ? p.accum1 : p.accum0;
                    const int DC = d1 ? p.DC1 : p.DC0;
                    bf16_t* dstp = reinterpret_cast<bf16_t*>(d1 ? p.dst1 : p.dst0);
                    if (!pvalid) continue;
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        const int col = colb + ch * 16 + 4 * kq;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc16[a][ph][b][ch][e];
                        if (p.bias) {
                            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + col);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] += bv[e];
                        }
                        bf16_t* o = dstp + opix * DC + (d1 ? col - p.D0 : col);
                        if (accum) {
                            const u32x2 old = *reinterpret_cast<const u32x2*>(o);
                            v[0] += __uint_as_float(old[0] << 16); v[1] += __uint_as_float(old[0] & 0xffff0000u);
                            v[2] += __uint_as_float(old[1] << 16); v[3] += __uint_as_float(old[1] & 0xffff0000u);
                        }
                        u32x2 pk;
                        pk[0] = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                        pk[1] = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                        *reinterpret_cast<u32x2*>(o) = pk;
                    }
                }
            }
        return;
    }
    // ---- epilogue (as igemm_conv_dma_kernel)
#pragma unroll
    for (int a = 0; a < MA; ++a) {
        const int n = img0 + pim[a];
        const int py = py0 + pty[a], px = px0 + ptx[a];
        const bool pvalid = !(pim[a] >= p.imgs || n >= p.N || py >= p.PH || px >= p.PW);
        const size_t opix = pvalid ? ((size_t)n * p.OH + py) * p.OW + px : 0;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int colb = n0 + b * 32;
            if (colb >= p.CO) continue;                       // uniform
            const bool d1 = colb >= p.D0;
            const int accum = d1 ? p.accum1 : p.accum0;
            const int DC = d1 ? p.DC1 : p.DC0;
            bf16_t* dstp = reinterpret_cast<bf16_t*>(d1 ? p.dst1 : p.dst0);
            if (!accum) {
                unsigned q[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * g + e];
                    if (p.bias) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + colb + 8 * g + 4 * h);
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
                if (pvalid) {
                    const int dcol = (d1 ? colb - p.D0 : colb) + 16 * h;
                    bf16_t* o = dstp + opix * DC + dcol;
                    *reinterpret_cast<u32x4*>(o) = u32x4{q[0][0], q[0][1], q[2][0], q[2][1]};
                    *reinterpret_cast<u32x4*>(o + 8) = u32x4{q[1][0], q[1][1], q[3][0], q[3][1]};
                }
                continue;
            }
            if (!pvalid) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = colb + 8 * g + 4 * h;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * g + e];
                if (p.bias) {
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + col);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += bv[e];
                }
                bf16_t* o = dstp + opix * DC + (d1 ? col - p.D0 : col);
                const u32x2 old = *reinterpret_cast<const u32x2*>(o);
                v[0] += __uint_as_float(old[0] << 16); v[1] += __uint_as_float(old[0] & 0xffff0000u);
                v[2] += __uint_as_float(old[1] << 16); v[3] += __uint_as_float(old[1] & 0xffff0000u);
                u32x2 pk;
                pk[0] = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                pk[1] = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                *reinterpret_cast<u32x2*>(o) = pk;
            }
        }
    }
    if constexpr (!MF16) if (p.stat_sums) {
        bool valid[MA];
#pragma unroll
        for (int a = 0; a < MA; ++a) valid[a] = pim[a] < p.imgs;
        tile_stats<MA, NB>(acc, valid, (wave * 32 * MA) >> (p.twl + p.thl), reinterpret_cast<float*>(smem), p.stat_sums, img0,
                           p.imgs, p.N, p.CO, n0, p.imgs == 2 ? DW / 2 : DW);
    }
}

template <typename T, int MA, int NB, int NX, int NT, bool WRES, bool PLAIN>
int launch(const ConvKArgs& a, size_t lds, int grid_x, int grid_y, hipStream_t st) {
    auto k = igemm_conv_kernel<T, MA, NB, NX, NT, WRES, PLAIN>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        CU_CHECK_ARG(e == hipSuccess, "cu_conv_gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k, dim3(grid_x, grid_y, a.ksplit > 1 ? a.ksplit : 1), dim3(256), lds, st, a);
    CU_LAUNCH_CHECK();
    return 0;
}

// split-K finish: dst = T(sum of the splits' slices, in split order + bias [+ dst]).  4 elements per thread; DC % 4 == 0.
template <typename T>
__global__ __launch_bounds__(256) void ksplit_finish_kernel(const float* __restrict__ ws, size_t slice, int ksplit,
                                                            T* __restrict__ dst, const float* __restrict__ bias, size_t n4,
                                                            int DC, int accum) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = *reinterpret_cast<const f32x4*>(ws + 4 * i);
    for (int k = 1; k < ksplit; ++k) v += *reinterpret_cast<const f32x4*>(ws + (size_t)k * slice + 4 * i);
    if (bias) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + (int)((4 * i) % (size_t)DC));
        v += b;
    }
    T* o = dst + 4 * i;
    if constexpr (sizeof(T) == 2) {
        if (accum) {
            const u32x2 old = *reinterpret_cast<const u32x2*>(o);
            v[0] += __uint_as_float(old[0] << 16); v[1] += __uint_as_float(old[0] & 0xffff0000u);
            v[2] += __uint_as_float(old[1] << 16); v[3] += __uint_as_float(old[1] & 0xffff0000u);
        }
        u32x2 pk;
        pk[0] = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        pk[1] = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        *reinterpret_cast<u32x2*>(o) = pk;
    } else {
        if (accum) v += *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = v;
    }
}

// split-K finish of a TINY feature map (HW <= 64 pixels per image) that also carries the layer's InstanceNorm + LeakyReLU
// (layers.py:193-194), forward (BWD = false) or backward (BWD = true): at 8x8 and below the separate norm launch is pure
// latency (13-22 us for kilobytes), and a workgroup here holds whole images -- every (image, channel) statistic is a
// reduction inside the workgroup, no second pass, no cross-workgroup exchange.
//   workgroup = (image n, 32 channels); thread = (channel quad q, pixel lane pl of 32) with pixels pl, pl + 32 in registers.
//   forward : z = T(sum of slices + bias) -> dst; mean / variance of the ROUNDED z (two-pass, in registers) -> the four
//             statistics planes; a = LeakyReLU(z * scale + shift) -> act_out.
//   backward: g = T(sum of slices [+ dst]) = dL/da; dst = dL/dz = gamma rstd (gl - mean(gl) - zhat mean(gl zhat)),
//             gl = g * LeakyReLU'(scale z + shift); dbeta += sum gl, dgamma += sum gl zhat (atomics over the images).
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void ksplit_finish_norm_kernel(const float* __restrict__ ws, size_t slice, int ksplit,
                                                                 T* __restrict__ dst, const float* __restrict__ bias, int N,
                                                                 int HW, int C, int accum, const cu_conv_epilogue ep) {
    __shared__ float red[2][4][64];
    const int n = blockIdx.y, cb = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 7, pl = wave * 8 + (lane >> 3);
    const int c = cb + q * 4;
    const size_t NC = (size_t)N * C, si = (size_t)n * C + c;
    // sum of K values per channel over the workgroup's 32 pixel lanes, returned to every thread of the channel quad
    auto reduce = [&](float* x, int K, int buf) {
        for (int j = 0; j < K; ++j) {
            float v = x[j];
            v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (lane < 8) red[buf][wave][q * K + j] = v;
        }
        __syncthreads();
        for (int j = 0; j < K; ++j)
            x[j] = (red[buf][0][q * K + j] + red[buf][1][q * K + j]) + (red[buf][2][q * K + j] + red[buf][3][q * K + j]);
    };
    auto round_t = [](float x) {
        if constexpr (sizeof(T) == 2) return bf16_to_f32(f32_to_bf16(x));
        else return x;
    };
    auto load_t = [](const T* ptr, float (&o)[4]) {
        if constexpr (sizeof(T) == 2) {
            const u32x2 r = *reinterpret_cast<const u32x2*>(ptr);
            o[0] = __uint_as_float(r[0] << 16); o[1] = __uint_as_float(r[0] & 0xffff0000u);
            o[2] = __uint_as_float(r[1] << 16); o[3] = __uint_as_float(r[1] & 0xffff0000u);
        } else {
            const f32x4 r = *reinterpret_cast<const f32x4*>(ptr);
            o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3];
        }
    };
    auto store_t = [](T* ptr, const float (&o)[4]) {
        if constexpr (sizeof(T) == 2) {
            u32x2 pk;
            pk[0] = (unsigned)f32_to_bf16(o[0]) | ((unsigned)f32_to_bf16(o[1]) << 16);
            pk[1] = (unsigned)f32_to_bf16(o[2]) | ((unsigned)f32_to_bf16(o[3]) << 16);
            *reinterpret_cast<u32x2*>(ptr) = pk;
        } else {
            *reinterpret_cast<f32x4*>(ptr) = f32x4{o[0], o[1], o[2], o[3]};
        }
    };
    // every load below is UNCONDITIONAL (clamped pixel / slice, masked afterwards): predicated loads are serialised with a
    // full wait each, and this kernel is nothing but one round trip of loads (all 2 x 8 slice loads in flight together)
    float v[2][4];
    bool live[2];
    size_t off[2];
    f32x4 part[2][8];
    float old[2][4];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int px = pl + 32 * it;
        live[it] = px < HW;
        off[it] = ((size_t)n * HW + (live[it] ? px : HW - 1)) * C + c;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            part[it][k] = *reinterpret_cast<const f32x4*>(ws + (size_t)(k < ksplit ? k : ksplit - 1) * slice + off[it]);
        if (accum) load_t(dst + off[it], old[it]);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        f32x4 a = part[it][0];
#pragma unroll
        for (int k = 1; k < 8; ++k) a += k < ksplit ? part[it][k] : f32x4{0.f, 0.f, 0.f, 0.f};
        if (bias) a += *reinterpret_cast<const f32x4*>(bias + c);
        if (accum) a += f32x4{old[it][0], old[it][1], old[it][2], old[it][3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[it][e] = live[it] ? round_t(a[e]) : 0.f;
    }
    const float inv = 1.f / (float)HW;
    if constexpr (!BWD) {
        float s[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = (live[0] ? v[0][e] : 0.f) + (live[1] ? v[1][e] : 0.f);
        reduce(s, 4, 0);
        float mean[4], ss[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mean[e] = s[e] * inv;
            const float d0 = live[0] ? v[0][e] - mean[e] : 0.f, d1 = live[1] ? v[1][e] - mean[e] : 0.f;
            ss[e] = d0 * d0 + d1 * d1;
        }
        reduce(ss, 4, 1);
        float sc[4], sh[4], rstd[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            rstd[e] = 1.f / sqrtf(ss[e] * inv + ep.eps);
            sc[e] = (ep.gamma ? ep.gamma[c + e] : 1.f) * rstd[e];
            sh[e] = (ep.beta ? ep.beta[c + e] : 0.f) - mean[e] * sc[e];
        }
        if (pl == 0) {
            *reinterpret_cast<f32x4*>(ep.stats + si) = f32x4{mean[0], mean[1], mean[2], mean[3]};
            *reinterpret_cast<f32x4*>(ep.stats + NC + si) = f32x4{rstd[0], rstd[1], rstd[2], rstd[3]};
            *reinterpret_cast<f32x4*>(ep.stats + 2 * NC + si) = f32x4{sc[0], sc[1], sc[2], sc[3]};
            *reinterpret_cast<f32x4*>(ep.stats + 3 * NC + si) = f32x4{sh[0], sh[1], sh[2], sh[3]};
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            if (!live[it]) continue;
            store_t(dst + off[it], v[it]);
            float a[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = v[it][e] * sc[e] + sh[e];
                a[e] = y > 0.f ? y : y * ep.slope;
            }
            store_t(reinterpret_cast<T*>(ep.act_out) + off[it], a);
        }
    } else {
        const f32x4 mean = *reinterpret_cast<const f32x4*>(ep.stats + si);
        const f32x4 rstd = *reinterpret_cast<const f32x4*>(ep.stats + NC + si);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(ep.stats + 2 * NC + si);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(ep.stats + 3 * NC + si);
        float gl[2][4], xh[2][4], s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = 0.f;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            float zv[4];
            load_t(reinterpret_cast<const T*>(ep.z) + off[it], zv);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = zv[e] * sc[e] + sh[e];
                gl[it][e] = live[it] ? (y > 0.f ? v[it][e] : v[it][e] * ep.slope) : 0.f;
                xh[it][e] = (zv[e] - mean[e]) * rstd[e];
                s[2 * e] += gl[it][e];
                s[2 * e + 1] += gl[it][e] * xh[it][e];
            }
        }
        reduce(s, 8, 0);
        if (pl == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (ep.mode == 5) {      // per-image planes [N][C], sole writer: plain stores (cu_norm_param_grads_batch adds the images)
                    if (ep.dbeta) ep.dbeta[si + e] = s[2 * e];
                    if (ep.dgamma) ep.dgamma[si + e] = s[2 * e + 1];
                } else {
                    if (ep.dbeta) unsafeAtomicAdd(ep.dbeta + c + e, s[2 * e]);
                    if (ep.dgamma) unsafeAtomicAdd(ep.dgamma + c + e, s[2 * e + 1]);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            if (!live[it]) continue;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[e] = (ep.gamma ? ep.gamma[c + e] : 1.f) * rstd[e] * (gl[it][e] - s[2 * e] * inv - xh[it][e] * s[2 * e + 1] * inv);
            store_t(dst + off[it], o);
        }
    }
}

}  // namespace

int cu_pconv_try(const cu_conv_desc* d, const void* src0, const void* w, const float* bias, void* dst0, void* stream);
int cu_tconv_try(const cu_conv_desc* d, const void* src0, const void* src1, const void* w, const float* bias, void* dst0,
                 void* dst1, const cu_conv_epilogue* ep, const float* scale0, const float* shift0, void* stream);

extern "C" int cu_conv_gemm(const cu_conv_desc* d, const void* src0, const float* scale0, const float* shift0,
                            const void* src1, const float* scale1, const float* shift1, const void* w, const float* bias,
                            void* dst0, void* dst1, void* stream) {
    return cu_conv_gemm_stats(d, src0, scale0, shift0, src1, scale1, shift1, w, bias, dst0, dst1, nullptr, 0, nullptr,
                              nullptr, stream);
}

extern "C" int cu_conv_gemm_ws(const cu_conv_desc* d, const void* src0, const float* scale0, const float* shift0,
                               const void* src1, const float* scale1, const float* shift1, const void* w,
                               const float* bias, void* dst0, void* dst1, float* ws, size_t ws_floats, void* stream) {
    return cu_conv_gemm_stats(d, src0, scale0, shift0, src1, scale1, shift1, w, bias, dst0, dst1, ws, ws_floats, nullptr,
                              nullptr, stream);
}

extern "C" int cu_conv_gemm_stats(const cu_conv_desc* d, const void* src0, const float* scale0, const float* shift0,
                                  const void* src1, const float* scale1, const float* shift1, const void* w,
                                  const float* bias, void* dst0, void* dst1, float* ws, size_t ws_floats,
                                  float* stat_sums, int* stats_done, void* stream) {
    cu_conv_epilogue ep;
    memset(&ep, 0, sizeof(ep));
    ep.mode = 1; ep.sums = stat_sums;
    return cu_conv_gemm_ex(d, src0, scale0, shift0, src1, scale1, shift1, w, bias, dst0, dst1, ws, ws_floats,
                           stat_sums && stats_done ? &ep : nullptr, stats_done, stream);
}

extern "C" int cu_conv_gemm_ex(const cu_conv_desc* d, const void* src0, const float* scale0, const float* shift0,
                               const void* src1, const float* scale1, const float* shift1, const void* w,
                               const float* bias, void* dst0, void* dst1, float* ws, size_t ws_floats,
                               const cu_conv_epilogue* ep, int* stats_done, void* stream) {
    if (stats_done) *stats_done = 0;
    CU_CHECK_ARG(d != nullptr, "cu_conv_gemm: null descriptor");
    CU_CHECK_ARG(d->dtype == CU_F32 || d->dtype == CU_BF16, "cu_conv_gemm: bad dtype %d", d->dtype);
    const bool bf = d->dtype == CU_BF16;
    const int CK = bf ? 32 : 16;
    CU_CHECK_ARG(d->ntaps == 9 || (d->ntaps >= 1 && d->ntaps <= 4), "cu_conv_gemm: ntaps must be 9 or <= 4 (got %d)", d->ntaps);
    CU_CHECK_ARG(d->C0 > 0 && d->C0 % CK == 0 && d->C1 >= 0 && d->C1 % CK == 0,
                 "cu_conv_gemm: channel counts %d,%d must be multiples of %d", d->C0, d->C1, CK);
    CU_CHECK_ARG(src0 && w && dst0 && (d->C1 == 0 || src1), "cu_conv_gemm: null pointer");
    CU_CHECK_ARG((scale0 == nullptr) == (shift0 == nullptr) && (scale1 == nullptr) == (shift1 == nullptr),
                 "cu_conv_gemm: scale/shift must come in pairs");
    CU_CHECK_ARG(d->D0 > 0 && d->D0 <= d->CO && (d->D0 == d->CO || dst1), "cu_conv_gemm: bad destination split");
    CU_CHECK_ARG(d->D0 == d->CO || d->D0 % 32 == 0, "cu_conv_gemm: split point must be a multiple of 32");
    CU_CHECK_ARG(d->CO % 4 == 0 && d->DC0 % 4 == 0 || d->out_nchw_f32, "cu_conv_gemm: CO and DC0 must be multiples of 4");
    CU_CHECK_ARG(d->N > 0 && d->PH > 0 && d->PW > 0 && d->IS >= 1 && d->OS >= 1, "cu_conv_gemm: bad geometry");
    CU_CHECK_ARG((d->PH - 1) * d->OS + d->OY0 < d->OH && (d->PW - 1) * d->OS + d->OX0 < d->OW && d->OY0 >= 0 &&
                     d->OX0 >= 0,
                 "cu_conv_gemm: destination pixel out of range");

    ConvKArgs a;
    memset(&a, 0, sizeof(a));
    a.src0 = src0; a.src1 = src1; a.sc0 = scale0; a.sh0 = shift0; a.sc1 = scale1; a.sh1 = shift1;
    a.w = w; a.bias = bias; a.dst0 = dst0; a.dst1 = dst1;
    a.N = d->N; a.PH = d->PH; a.PW = d->PW; a.SH = d->SH; a.SW = d->SW; a.C0 = d->C0; a.C1 = d->C1; a.IS = d->IS;
    a.OH = d->OH; a.OW = d->OW; a.OS = d->OS; a.OY0 = d->OY0; a.OX0 = d->OX0; a.CO = d->CO; a.D0 = d->D0;
    a.DC0 = d->DC0; a.DC1 = d->DC1; a.ntaps = d->ntaps;
    a.slope0 = d->slope0; a.slope1 = d->slope1; a.accum0 = d->accum0; a.accum1 = d->accum1;
    a.out_nchw = d->out_nchw_f32;
    a.par_co = d->par_co;
    a.par_taps = d->par_co > 0 ? d->par_taps : 0;
    a.par_pack = 0;
    for (int i = 0; i < 16 && a.par_taps; ++i) {
        CU_CHECK_ARG(d->par_tap_w[i] >= -1 && d->par_tap_w[i] < 15, "cu_conv_gemm: bad parity tap %d", d->par_tap_w[i]);
        a.par_pack |= (unsigned long long)(d->par_tap_w[i] + 1) << (4 * i);
    }
    CU_CHECK_ARG(!a.par_taps || (d->ntaps <= 4 && !bias), "cu_conv_gemm: parity taps need ntaps <= 4 and no bias");
    CU_CHECK_ARG(d->par_co == 0 || (d->par_co > 0 && d->par_co % 32 == 0 && d->CO == 4 * d->par_co && d->D0 == d->CO &&
                                    d->OS == 2 && d->OY0 == 0 && d->OX0 == 0 && !d->out_nchw_f32 &&
                                    (!d->accum0 || d->par_taps) &&
                                    (d->PH - 1) * 2 + 1 < d->OH && (d->PW - 1) * 2 + 1 < d->OW),
                 "cu_conv_gemm: bad parity-column mode (par_co=%d)", d->par_co);
    const int CI = d->C0 + d->C1;
    a.dbg = cu_env_int("CU_CONV_DBG", 0);

    // ---- few-tap / strided bf16 gathers of plain operands: the lean gather-GEMM (pconv.hip)
    if (bf && !scale0 && !scale1 && !cu_env_set("CU_CONV_NOPCONV")) {
        const int rc = cu_pconv_try(d, src0, w, bias, dst0, stream);
        if (rc != 0) return rc < 0 ? rc : 0;
    }

    // ---- thin, large bf16 3x3 stride-1 layers (256^2 x 32, 128^2 x 64 channels): the streaming kernel (tconv.hip)
    // (a raw source 0 -- scale0 / shift0 given -- is normalised + activated in LDS by the kernel's XF instances)
    if (bf && !scale1 && (!scale0 || shift0) && !cu_env_set("CU_CONV_NOTCONV")) {
        const int rc = cu_tconv_try(d, src0, src1, w, bias, dst0, dst1, stats_done ? ep : nullptr, scale0, shift0, stream);
        if (rc == 2) *stats_done = 1;
        if (rc != 0) return rc < 0 ? rc : 0;
    }

    // ---- wide bf16 3x3 stride-1 layers whose weight slice cannot stay in LDS: 8-wave LDS-DMA kernel
    {
        const bool plain0 = !scale0 && d->slope0 == 1.0f && (d->C1 == 0 || (!scale1 && d->slope1 == 1.0f));
        const size_t b0 = (size_t)d->N * d->SH * d->SW * d->C0 * 2, b1 = (size_t)d->N * d->SH * d->SW * d->C1 * 2;
        const size_t bw = (size_t)9 * d->CO * CI * 2, lim = 0x7fff0000ull;
        const long px_all = (long)d->N * d->PH * d->PW;
        const int dbm = d->IS == 1 ? 512 : 256;                        // loop pixels per tile
        int dnb = d->CO % 128 == 0 || d->CO > 256 ? 4 : 2;             // 128-column tiles unless that wastes half a tile
        if (dnb == 4 && (px_all / dbm) * cdiv(d->CO, 128) < 256)
            dnb = 2;                                                     // ... or leaves CUs without a workgroup
        // (64 columns for the two-chunk 64 -> 128-column input gradient at 128^2, which is mostly prologue + epilogue on the
        //  128-column ring: measured slower, 229 / 259 us with / without the 64-column ring against 195 us)
        if (d->IS == 2) dnb = 4;
        if (d->IS == 1 && d->CO <= 32) dnb = 1;                          // thin layers (256^2 x 32 channels)
        if (d->IS == 1 && dnb > 1 && cu_env_int("CU_CONV_DNB", 0)) dnb = cu_env_int("CU_CONV_DNB", 0);
        // Thin layers too, from 64 input channels on (measured, tools/thin_bench.py, profiles/r02_thin_bench.txt: 128^2 x 64
        // concat forward 313 -> 227 us, 256^2 x 32+32 -> 32 322 -> 286 us; with 32 input channels the one-tile-per-workgroup
        // structure loses to the persistent register-staged kernel, 173 vs 150 us)
        const int min_ci = d->IS == 1 ? cu_env_int("CU_CONV_DMA_MINC", 64) : 64;
        if (bf && plain0 && (d->IS == 1 || (d->IS == 2 && (d->CO % 128 == 0 || (d->CO > 256 && !cu_env_set("CU_CONV_S2_RAGGED_OFF"))))) && d->ntaps == 9 &&
            !d->out_nchw_f32 && d->CO >= (d->IS == 1 ? 32 : 128) && CI >= min_ci && d->C0 % 32 == 0 &&
            d->D0 % 32 == 0 && d->CO % 16 == 0 && !d->par_co && b0 < lim && b1 < lim && bw < lim &&
            // (64 workgroups are enough for the 960-column input gradient at 8x8: 78 vs 112 us on the generic kernel; the
            // 480-column layers there lose on this kernel, 79 vs 39 us: tools/small_dma.py)
            (px_all / dbm) * cdiv(d->CO, 32 * dnb) >= cu_env_int("CU_CONV_DMA_MINWG", d->CO >= 768 ? 64 : 128) && !cu_env_set("CU_CONV_NODMA")) {
            int tw = d->PW < 32 ? d->PW : 32;
            int th = dbm / tw;
            if (th > d->PH) th = d->PH;
            int imgs = dbm / (tw * th);
            a.twl = ilog2_exact(tw); a.thl = ilog2_exact(th); a.iml = ilog2_exact(imgs);
            int dymin = 1 << 20, dxmin = 1 << 20, dymax = -(1 << 20), dxmax = -(1 << 20);
            for (int t = 0; t < 9; ++t) {
                dymin = d->tap_dy[t] < dymin ? d->tap_dy[t] : dymin; dymax = d->tap_dy[t] > dymax ? d->tap_dy[t] : dymax;
                dxmin = d->tap_dx[t] < dxmin ? d->tap_dx[t] : dxmin; dxmax = d->tap_dx[t] > dxmax ? d->tap_dx[t] : dxmax;
            }
            a.dymin = dymin; a.dxmin = dxmin;
            a.HH = (th - 1) * d->IS + (dymax - dymin) + 1; a.HW = (tw - 1) * d->IS + (dxmax - dxmin) + 1;
            const int halo = imgs * a.HH * a.HW;
            const int halo_pad = cdiv(halo, 128) * 128;          // whole 8-KiB staging rounds (128 rows of 64 B)
            const size_t lds = (size_t)halo_pad * 64 + (size_t)cdiv(9 * 32 * dnb, 128) * 8192;
            if (a.twl >= 0 && a.thl >= 0 && a.iml >= 0 && d->PW % tw == 0 && d->PH % th == 0 && halo_pad <= 128 * DMA_MAXX &&
                lds <= 160 * 1024 && ilog2_exact(d->PW / tw) >= 0 && ilog2_exact(d->PH / th) >= 0) {
                a.imgs = imgs; a.halo_px = halo_pad;
                // the tile's InstanceNorm statistics in the epilogue (mode 1): a wave's pixels must lie in one image
                const int wave_px = 32 * (d->IS == 1 ? 2 : 1);
                if (ep && stats_done && ep->mode == 1 && ep->sums && d->D0 == d->CO && !d->accum0 && (tw * th) % wave_px == 0 &&
                    imgs <= 2 && !cu_env_set("CU_CONV_NO_DMA_STATS")) {
                    a.stat_sums = ep->sums;
                    *stats_done = 1;
                }
                a.tiles_x = d->PW / tw; a.tiles_y = d->PH / th;
                a.txl = ilog2_exact(a.tiles_x); a.tyl = ilog2_exact(a.tiles_y);
                a.ntiles = a.tiles_x * a.tiles_y * cdiv(d->N, imgs);
                for (int t = 0; t < 9; ++t) {
                    a.tap_off[t] = (d->tap_dy[t] - dymin) * a.HW + (d->tap_dx[t] - dxmin);
                    a.tap_w[t] = d->tap_w[t];
                    CU_CHECK_ARG(d->tap_w[t] >= 0, "cu_conv_gemm: negative weight tap index");
                }
                const int nxr = halo_pad / 128;
                // two images per tile (16 x 16 maps) and 128 columns: the two halo images + three 24-KiB tap rows exceed the
                // LDS; the 64-column ring fits (tuning knob CU_CONV_RING2_WIDE: 0 keeps the non-overlapped 128-column kernel)
                if (d->IS == 1 && dnb == 4 && nxr == 6 && cu_env_int("CU_CONV_RING2_WIDE", 1)) dnb = 2;
                const bool ring4 = d->IS == 1 && dnb == 4 && nxr == 5;
                const bool ring2 = d->IS == 1 && dnb == 2 && (nxr == 5 || nxr == 6) && !cu_env_set("CU_CONV_NORING2");
                if ((ring4 || ring2) && !cu_env_set("CU_CONV_NORING")) {
                    // two halo images, three tap rows (3 / 2 staging rounds each), dump
                    const size_t rl = (size_t)2 * nxr * 8192 + (size_t)3 * (ring4 ? 3 : 2) * 8192 + 8192;
                    // MF16: the 16x16x32 MFMA form (no epilogue statistics); CU_CONV_MF16 = 0 / 1 forces it off / on in the tuning build
                    const bool mf16 = !a.stat_sums && cu_env_int("CU_CONV_MF16", RING_MF16_DEFAULT) != 0;
                    auto kr = ring4 ? (mf16 ? igemm_conv_dma_ring_kernel<4, 5, true> : igemm_conv_dma_ring_kernel<4, 5, false>)
                                    : (nxr == 5 ? (mf16 ? igemm_conv_dma_ring_kernel<2, 5, true> : igemm_conv_dma_ring_kernel<2, 5, false>)
                                                : (mf16 ? igemm_conv_dma_ring_kernel<2, 6, true> : igemm_conv_dma_ring_kernel<2, 6, false>));
                    hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void*>(kr),
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl);
                    CU_CHECK_ARG(er == hipSuccess, "cu_conv_gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(er));
                    hipLaunchKernelGGL(kr, dim3(a.ntiles, cdiv(d->CO, 32 * dnb)), dim3(64 * DW), rl,
                                       reinterpret_cast<hipStream_t>(stream), a, (unsigned)b0, (unsigned)b1, (unsigned)bw);
                    CU_LAUNCH_CHECK();
                    return 0;
                }
                auto k = d->IS == 2 ? igemm_conv_dma_kernel<9, 4, 1>
                                    : (dnb == 4 ? igemm_conv_dma_kernel<9, 4, 2>
                                                : (dnb == 2 ? igemm_conv_dma_kernel<9, 2, 2> : igemm_conv_dma_kernel<9, 1, 2>));
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                CU_CHECK_ARG(e == hipSuccess, "cu_conv_gemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
                hipLaunchKernelGGL(k, dim3(a.ntiles, cdiv(d->CO, 32 * dnb)), dim3(64 * DW), lds,
                                   reinterpret_cast<hipStream_t>(stream), a, (unsigned)b0, (unsigned)b1, (unsigned)bw);
                CU_LAUNCH_CHECK();
                return 0;
            }
        }
    }

    // ---- column tile: the widest of {128, 96, 64, 32} that divides the work without waste
    int nb;
    if (d->CO % 128 == 0) nb = 4;
    else if (d->CO % 96 == 0) nb = 3;
    else if (d->CO % 64 == 0) nb = 2;
    else nb = 1;
    if (d->D0 != d->CO && d->D0 % (32 * nb) != 0) nb = (d->D0 % 64 == 0 && d->CO % 64 == 0) ? 2 : 1;
    if (!bf && nb > 2) nb = (d->CO % 64 == 0) ? 2 : 1;   // keep the f32 weight tile within LDS
    // (round 4: 64-column tiles at most.  The 128-column ring tile wins as a lone launch (profiles/r03b_layers_conv_one_stream_ring2.txt);
    //  inside the step the narrower one does, by a little: 12.133 -> 12.106 ms, five of five alternating pairs, profiles/r04_knob_sweep2.txt)
    { const int cap = cu_env_int("CU_CONV_NBMAX", 2);
      while (nb > cap) nb = (nb == 4) ? 2 : (nb == 3 ? 1 : 1); }
    // small feature maps: few pixel tiles, and every workgroup walks all channel chunks one L2 round trip at a time --
    // narrower column tiles put more CUs on the same work (4x4 x 480 channels: 20 workgroups with 96 columns each)
    if (!cu_env_set("CU_CONV_NO_NARROW")) {
        const long px_tiles = ((long)d->N * d->PH * d->PW + 127) / 128;
        while (nb > 1 && px_tiles * cdiv(d->CO, 32 * nb) < 192) {
            const int nn = (nb == 4) ? 2 : 1;
            if (d->D0 != d->CO && d->D0 % (32 * nn) != 0) break;
            nb = nn;
        }
    }
    int coltiles = cdiv(d->CO, 32 * nb);
    a.WP = 32 * nb + 2;

    int dymin = 1 << 20, dxmin = 1 << 20, dymax = -(1 << 20), dxmax = -(1 << 20);
    for (int t = 0; t < d->ntaps; ++t) {
        dymin = d->tap_dy[t] < dymin ? d->tap_dy[t] : dymin; dymax = d->tap_dy[t] > dymax ? d->tap_dy[t] : dymax;
        dxmin = d->tap_dx[t] < dxmin ? d->tap_dx[t] : dxmin; dxmax = d->tap_dx[t] > dxmax ? d->tap_dx[t] : dxmax;
    }
    a.dymin = dymin; a.dxmin = dxmin;

    // ---- pixel tile: BM = 128*MA loop pixels = IMGS x TH x TW.  MA = 2 (bf16, stride-1 gathers, 3x3) when there is
    //      enough work to keep every CU busy with 256-pixel tiles.
    const long total_px = (long)d->N * d->PH * d->PW;
    int ma = (bf && d->IS == 1 && d->ntaps == 9 && total_px / 256 * coltiles >= 512) ? 2 : 1;
    // 256 pixels x 128 columns needs 8 accumulator tiles + staging registers per wave: that instance spills (220 bytes
    // of scratch per lane) and runs 1.5x slower than two 64-column tiles (128^2 x 64->128: 452 vs 299 us)
    if (ma == 2 && nb == 4 && (d->D0 == d->CO || d->D0 % 64 == 0)) {
        nb = 2;
        a.WP = 32 * nb + 2;
        coltiles = cdiv(d->CO, 32 * nb);
    }
    int tw = 0, th = 0, imgs = 0;
    for (;; ma = 1) {
        const int BM = 128 * ma;
        tw = d->PW < 32 ? d->PW : 32;
        th = BM / tw;
        if (th > d->PH) th = d->PH;
        imgs = BM / (tw * th);
        a.twl = ilog2_exact(tw); a.thl = ilog2_exact(th);
        CU_CHECK_ARG(a.twl >= 0 && a.thl >= 0 && ilog2_exact(imgs) >= 0 && d->PW % tw == 0 && d->PH % th == 0,
                     "cu_conv_gemm: loop grid %dx%d must be powers of two", d->PH, d->PW);
        a.HH = (th - 1) * d->IS + (dymax - dymin) + 1;
        a.HW = (tw - 1) * d->IS + (dxmax - dxmin) + 1;
        const int nxmax = ma == 2 ? 6 : MAXI;
        while (imgs > 1 && imgs * a.HH * a.HW * 4 > 256 * nxmax) imgs >>= 1;   // stride-2 gathers on tiny maps
        if (imgs * a.HH * a.HW * 4 <= 256 * nxmax || ma == 1) break;
    }
    a.imgs = imgs; a.iml = ilog2_exact(imgs);
    a.halo_px = imgs * a.HH * a.HW;
    CU_CHECK_ARG(a.halo_px * 4 <= 256 * MAXI, "cu_conv_gemm: halo of %d pixels too large", a.halo_px);
    CU_CHECK_ARG(a.HH < 1024 && a.HW < 1024, "cu_conv_gemm: halo too large");
    a.tiles_x = d->PW / tw; a.tiles_y = d->PH / th;
    a.txl = ilog2_exact(a.tiles_x); a.tyl = ilog2_exact(a.tiles_y);
    CU_CHECK_ARG(a.txl >= 0 && a.tyl >= 0, "cu_conv_gemm: tile counts must be powers of two");
    const int igroups = cdiv(d->N, imgs);
    a.ntiles = a.tiles_x * a.tiles_y * igroups;
    for (int t = 0; t < d->ntaps; ++t) {
        a.tap_off[t] = (d->tap_dy[t] - dymin) * a.HW + (d->tap_dx[t] - dxmin);
        a.tap_w[t] = d->tap_w[t];
        CU_CHECK_ARG(d->tap_w[t] >= 0, "cu_conv_gemm: negative weight tap index");
    }
    const int mod = bf ? 16 : 32;   // plane stride == 2 (mod 16 entries / 32 floats): conflict-free fills
    a.XP = a.halo_px + ((2 - a.halo_px % mod) + mod) % mod;

    // ---- LDS budget; resident weights when the whole slice fits beside the halo patch (bf16, <= 2 column blocks)
    const int unit = bf ? 16 : 4;
    const int wplanes = bf ? 4 : 16, xplanes = bf ? 4 : 16;
    const size_t x_bytes = (size_t)xplanes * a.XP * unit;
    const int nchunks = CI / CK;
    auto wbytes = [&](int nbv) { return (size_t)d->ntaps * wplanes * (32 * nbv + 2) * unit; };
    auto wres_ok = [&](int nbv) {
        return bf && nbv <= 2 && x_bytes + wbytes(nbv) * nchunks <= 96 * 1024 && a.ntiles >= 512;
    };
    // a 64-column tile whose weight slice does not fit LDS loses to two resident 32-column tiles (measured: 64->64 at
    // 128^2, 190 -> 150 us)
    int coltiles2 = coltiles;
    if (bf && nb == 2 && !wres_ok(2) && wres_ok(1) && (d->D0 == d->CO || d->D0 % 32 == 0)) {
        nb = 1;
        a.WP = 34;
        coltiles2 = cdiv(d->CO, 32);
    }
    const size_t w_chunk_bytes = wbytes(nb);
    const bool wres = wres_ok(nb);
    const size_t lds_data = x_bytes + w_chunk_bytes * (wres ? nchunks : 1);
    a.tap_lds = (int)((lds_data + 15) / 16 * 16);
    const size_t lds = a.tap_lds + 128;
    CU_CHECK_ARG(lds <= 160 * 1024, "cu_conv_gemm: LDS %zu bytes exceeds 160 KiB", lds);

    // ---- grid: persistent over pixel tiles (a few workgroups per CU), one grid row per column tile
    int grid_x = a.ntiles;
    const int cap = wres ? 256 * 3 : 256 * 8;
    if (grid_x > cap) grid_x = cap;

    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // ---- split-K for tiny feature maps (<= 4x4 at batch 64): with so few pixel tiles every workgroup walks all channel
    //      chunks one L2 round trip at a time (33-36 us whatever the FLOPs).  grid.z workgroups share a tile's chunks and
    //      store f32 partial tiles into their slice of the caller's workspace; a finish pass sums the slices (fixed order:
    //      deterministic), applies bias / rounding / accumulate.  Needs a destination that this launch covers completely.
    size_t fin0 = 0, fin1 = 0;
    {
        const long wgs = (long)grid_x * coltiles2;
        const bool whole = (d->OS == 1 && d->OY0 == 0 && d->OX0 == 0 && d->PH == d->OH && d->PW == d->OW &&
                            d->DC0 == d->D0 && (d->D0 == d->CO || d->DC1 == d->CO - d->D0)) ||
                           (d->par_co > 0 && d->DC0 == d->par_co && d->OH == 2 * d->PH && d->OW == 2 * d->PW);
        fin0 = (size_t)d->N * d->OH * d->OW * d->DC0;
        fin1 = d->D0 == d->CO ? 0 : (size_t)d->N * d->OH * d->OW * d->DC1;
        // a fused norm finish (epilogue modes 3 / 4) is worth a two-way split of its own on 8x8 maps (480 workgroups): the
        // finish pass replaces a norm launch instead of adding one
        const bool norm_fin = ep && stats_done && (ep->mode == 3 || ep->mode == 4 || ep->mode == 5) && d->OH * d->OW <= 64 &&
                              !cu_env_set("CU_CONV_NO_NORMFIN8");
        const int max_wgs = cu_env_int("CU_CONV_KSPLIT_WGS", 128);
        if (ws && !wres && whole && !d->out_nchw_f32 && nchunks >= 4 && (wgs <= max_wgs || (norm_fin && wgs <= 512)) &&
            a.ntiles <= grid_x && !cu_env_set("CU_CONV_NO_KSPLIT")) {
            // workgroups the split aims at: 512; 256 on 4x4 maps, where the partial-tile traffic of a deeper split costs more than
            // the extra workgroups bring (profiles/r04_small_map_sweep.txt: 34.7 -> 28.6 us, 48.3 -> 40.1 us per ConvLayer; 8x8 and
            // 2x2 maps are indifferent or worse)
            const int ks_target = (long)d->PH * d->PW == 16 ? 256 : 512;
            int want = (int)(cu_env_int("CU_CONV_KSPLIT_TARGET", ks_target) / wgs);
            if (norm_fin && want < 2) want = 2;
            if (want > cu_env_int("CU_CONV_KSPLIT_MAX", 8)) want = cu_env_int("CU_CONV_KSPLIT_MAX", 8);
            if (want > nchunks / 2) want = nchunks / 2;
            while (want > 1 && (size_t)want * (fin0 + fin1) > ws_floats) --want;
            if (want > 1) {
                const int per = cdiv(nchunks, want);
                a.ksplit = cdiv(nchunks, per);
                a.kspan = per * CK;
                a.ws0 = ws;
                a.ws1 = ws + fin0;
                a.ws_slice = fin0 + fin1;
            }
        }
    }
    // modes 3 / 4 of the epilogue: the finish pass of a tiny map also does the layer's InstanceNorm + LeakyReLU
    const bool fin_norm = ep && stats_done && (ep->mode == 3 || ep->mode == 4 || ep->mode == 5) && a.ksplit > 1 && fin1 == 0 &&
                          d->OH * d->OW <= 64 && d->DC0 % 32 == 0 && ep->stats &&
                          (ep->mode == 3 ? (ep->act_out && !d->accum0) : (ep->z != nullptr));
    auto finish = [&]() -> int {
        if (a.ksplit <= 1) return 0;
        if (fin_norm) {
            const dim3 g(d->DC0 / 32, d->N);
            const int HW = d->OH * d->OW;
            if (bf) {
                if (ep->mode == 3) hipLaunchKernelGGL((ksplit_finish_norm_kernel<bf16_t, false>), g, dim3(256), 0, st, a.ws0, a.ws_slice, a.ksplit, (bf16_t*)dst0, bias, d->N, HW, d->DC0, d->accum0, *ep);
                else hipLaunchKernelGGL((ksplit_finish_norm_kernel<bf16_t, true>), g, dim3(256), 0, st, a.ws0, a.ws_slice, a.ksplit, (bf16_t*)dst0, bias, d->N, HW, d->DC0, d->accum0, *ep);
            } else {
                if (ep->mode == 3) hipLaunchKernelGGL((ksplit_finish_norm_kernel<float, false>), g, dim3(256), 0, st, a.ws0, a.ws_slice, a.ksplit, (float*)dst0, bias, d->N, HW, d->DC0, d->accum0, *ep);
                else hipLaunchKernelGGL((ksplit_finish_norm_kernel<float, true>), g, dim3(256), 0, st, a.ws0, a.ws_slice, a.ksplit, (float*)dst0, bias, d->N, HW, d->DC0, d->accum0, *ep);
            }
            CU_LAUNCH_CHECK();
            *stats_done = 1;
            return 0;
        }
        for (int k = 0; k < 2; ++k) {
            const size_t n = k ? fin1 : fin0;
            if (!n) continue;
            const float* wsp = k ? a.ws1 : a.ws0;
            void* dp = k ? dst1 : dst0;
            const float* bp = bias ? bias + (k ? d->D0 : 0) : nullptr;
            const int DC = k ? d->DC1 : d->DC0, acc = k ? d->accum1 : d->accum0;
            const unsigned blocks = (unsigned)((n / 4 + 255) / 256);
            if (bf) hipLaunchKernelGGL(ksplit_finish_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, wsp, a.ws_slice, a.ksplit, (bf16_t*)dp, bp, n / 4, DC, acc);
            else hipLaunchKernelGGL(ksplit_finish_kernel<float>, dim3(blocks), dim3(256), 0, st, wsp, a.ws_slice, a.ksplit, (float*)dp, bp, n / 4, DC, acc);
        }
        CU_LAUNCH_CHECK();
        return 0;
    };
    const int nx = ma == 2 ? 6 : (a.halo_px * 4 > 256 * 4 ? 10 : 4);
    const bool plain = !scale0 && d->slope0 == 1.0f && (d->C1 == 0 || (!scale1 && d->slope1 == 1.0f));
#define CU_L(T, MAv, NBv, NXv, NTv, WR)                                                              \
    do {                                                                                             \
        const int rc_ = plain ? launch<T, MAv, NBv, NXv, NTv, WR, true>(a, lds, grid_x, coltiles2, st) \
                              : launch<T, MAv, NBv, NXv, NTv, WR, false>(a, lds, grid_x, coltiles2, st); \
        return rc_ ? rc_ : finish();                                                                 \
    } while (0)
#define CU_NXNT(T, NBv, WR)                                       \
    do {                                                          \
        if (ma == 2) CU_L(T, 2, NBv, 6, 9, WR);                   \
        if (nx == 10) {                                           \
            if (d->ntaps == 9) CU_L(T, 1, NBv, 10, 9, WR);        \
            if (d->ntaps == 4) CU_L(T, 1, NBv, 10, 4, WR);        \
            CU_L(T, 1, NBv, 10, 0, WR);                           \
        }                                                         \
        if (d->ntaps == 9) CU_L(T, 1, NBv, 4, 9, WR);             \
        if (d->ntaps == 4) CU_L(T, 1, NBv, 4, 4, WR);             \
        CU_L(T, 1, NBv, 4, 0, WR);                                \
    } while (0)
    if (bf) {
        switch (nb) {
            case 4: CU_NXNT(bf16_t, 4, false);
            case 3: CU_NXNT(bf16_t, 3, false);
            case 2: if (wres) CU_NXNT(bf16_t, 2, true); CU_NXNT(bf16_t, 2, false);
            default: if (wres) CU_NXNT(bf16_t, 1, true); CU_NXNT(bf16_t, 1, false);
        }
    } else {
        ma = 1;
        switch (nb) {
            case 2: CU_NXNT(float, 2, false);
            default: CU_NXNT(float, 1, false);
        }
    }
#undef CU_NXNT
#undef CU_L
}
