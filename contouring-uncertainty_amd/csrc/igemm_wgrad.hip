// Weight gradient of the gather convolution on MFMA (gfx950).  See include/contour_hip.h : cu_conv_wgrad.
//
//   dW[t][n][c] += sum_p Z[p*ZS + zoff_t, n] * act(S[p*IS + off_t, c])
//
// Replaces autograd's weight gradient of nn.Conv2d / nn.ConvTranspose2d (reference models/nnUnet/layers.py:55-109) with
// the InstanceNorm+LeakyReLU / concat of the forward operand recomputed in the load (never materialised).
//
// The reduction index is the pixel, which is the slow index of NHWC: both operands are staged row-major
// ([pixel][channel]) in LDS and read k-major with the hardware transpose read ds_read_b64_tr_b16 (bf16) so that tap
// shifts are plain row offsets; f32 parity mode uses v_mfma_f32_32x32x2_f32 with scalar LDS reads.
// One workgroup owns a (32*NBLK) x (32*CBLK) block of dW for all taps (<= 9 accumulators of 32x32 per wave) and walks
// pixel tiles split over grid.y; partial sums are added with f32 atomics in 128-byte rows.
#include "common.h"

namespace {

struct WgKArgs {
    const void* src0; const void* src1;
    const float* sc0; const float* sh0; const float* sc1; const float* sh1;
    const void* z; float* dw;
    int N, PH, PW, SH, SW, C0, C1, IS, ZH, ZW, ZC, ZS, CO, ntaps;
    int s_off[CU_MAX_TAPS], z_off[CU_MAX_TAPS], tap_w[CU_MAX_TAPS];
    int sdymin, sdxmin, SHH, SHW, s_halo;      // source halo
    int zdymin, zdxmin, ZHH, ZHW, z_halo;      // Z halo
    int twl, thl, iml, tiles_x, tiles_y, igroups, ntiles, splits;
    int ctiles;                                // number of channel tiles (grid.x = ntile_n * ctiles)
    int tile_px;                               // loop pixels per tile (128, or 64 for stride-2 gathers)
    int wtaps;                                 // weight taps (max tap_w + 1): slab layout of the partial-tile mode
    // LDS-DMA kernel, stride-2 3x3 gathers (s_deint): the source tile is staged as its four PARITY planes
    // S_ab[y][x] = S[2y + a][2x + b], each (th + 1) x (tw + 1) pixels from (py0 - 1, px0 - 1), so that tap (dy, dx) is a
    // plain row offset into plane (dy & 1, dx & 1) and the k-loop walks dense rows (ISL = 1) exactly like a stride-1 layer
    int s_deint, ISL, s_hpi, s_plane;
    unsigned mg_spl;
    unsigned mg_shpi, mg_shw, mg_zhpi, mg_zhw; // ceil(2^32/d) magics for the halo index decode
    float slope0, slope1;
    int zsame;                                 // all taps read the same Z pixel
    unsigned src0_bytes, src1_bytes, z_bytes;  // tensor sizes (LDS-DMA kernel: buffer resources, out-of-range = zero fill)
    int s_iters, z_iters;                      // LDS-DMA kernel: 4-KiB staging blocks per tile for S / Z
    int pc_items;                              // producer/consumer kernel: staging rounds issued by the producer waves
    int pc_early;                              // ... and rounds the computing waves issue before their k-loop
    unsigned xstat_bytes;                      // XF: bytes of the scale + shift planes behind sc0
    int dbg;                                   // CU_CONV_DBG bits (timing experiments): 1 no atomics, 2 no MFMA, 4 no commit, 8 no loads
    // partial-tile mode (cu_conv_wgrad_parts): part_stride != 0 -> every adder of a dW block STORES its partial tile into
    // its own slab dw + part * part_stride (no atomics); the slabs are summed in a fixed order by cu_grad_unprep_parts.
    // Slab layout = the accumulators as they stand ("native"): [block = nt * ctiles + ct][wave block][weight tap][q]
    // [lane][4] floats, register 4q + e of a lane at [q][lane][e] -- every store instruction writes 1 KiB contiguous.
    // parts_floats / nparts_out / layout_out are host-side only.
    size_t part_stride;
    size_t parts_floats;
    int* nparts_out;
    int* layout_out;
};

// partial-tile mode: one wave's accumulators -> its region of the slab (16-byte stores, 1 KiB per instruction)
template <int NTAPS>
__device__ __forceinline__ void store_native(float* slab, int block, int nwb, int blk, int wtaps, const int* tap_w,
                                             const f32x16 (&acc)[NTAPS], int lane) {
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
        float* o = slab + ((size_t)((block * nwb + blk) * wtaps + tap_w[t]) * 4) * 256 + lane * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(o + q * 256) = f32x4{acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
    }
}

template <typename T> struct WCfg;
template <> struct WCfg<bf16_t> { static constexpr int PIECE = 8; static constexpr int KPIX = 16; };
template <> struct WCfg<float> { static constexpr int PIECE = 4; static constexpr int KPIX = 2; };

__device__ __forceinline__ bf16x4 tr_read(const void* lds_ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (__attribute__((address_space(3))) bf16x4*)(const_cast<void*>(lds_ptr)));
}


// NS / NZ: source / Z pieces per thread per tile (compile-time so that every staging load is unconditional, with a
// clamped address and a mask -- predicated loads are serialised by hipcc).  PLAIN: the source needs no affine /
// activation (materialised activations): staging is a pure 16-byte copy.  The next tile's loads are issued before
// the current tile's MFMAs (register prefetch).
// NTAPS is compile-time (9 = 3x3, 4 = 2x2 transposed conv, 1 = 1x1): no branches between the taps, so the scheduler
// issues all fragment reads of a k-step ahead of its MFMAs.
template <typename T, int NBLK, int CBLK, int NS, int NZ, bool PLAIN, int NTAPS>
__global__ __launch_bounds__(256) void igemm_wgrad_kernel(const WgKArgs p) {
    using C = WCfg<T>;
    constexpr int PIECE = C::PIECE, KPIX = C::KPIX;
    constexpr int TN = 32 * NBLK, TC = 32 * CBLK;
    constexpr int NWB = NBLK * CBLK;            // wave blocks
    constexpr int KSPLIT = 4 / NWB;             // waves sharing one block split the k-steps
    constexpr int SPP = TC / PIECE;             // source pieces per pixel
    constexpr int ZPP = TN / PIECE;
    // row pitch in bytes: bf16 rows of 64 B are contiguous per 4-pixel transpose block; 128-B rows get +64 B so that the
    // 4 rows of a transpose block fall on distinct bank quarters
    constexpr int ROWS_B = (sizeof(T) == 2) ? (TC == 32 ? 64 : 192) : TC * 4 + 16;
    constexpr int ROWZ_B = (sizeof(T) == 2) ? (TN == 32 ? 64 : 192) : TN * 4 + 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ss = smem;
    unsigned char* Zs = smem + (size_t)p.s_halo * ROWS_B;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int blk = wave % NWB, kpart = wave / NWB;
    const int nblk = blk / CBLK, cblk = blk % CBLK;
    const int TW = 1 << p.twl, TH = 1 << p.thl;
    const int CI = p.C0 + p.C1;
    const int ct = blockIdx.x % p.ctiles, nt = blockIdx.x / p.ctiles;
    const int c_base = ct * TC, n_base = nt * TN;
    const int s_hpi = p.SHH * p.SHW, z_hpi = p.ZHH * p.ZHW;
    const bool wave_active = (n_base + nblk * 32 < p.CO) && (c_base + cblk * 32 < CI);

    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // ---- tile-invariant staging geometry.  256 % SPP == 0 and 256 % ZPP == 0: a thread always stages the same piece.
    const int s_piece = tid % SPP, z_piece = tid % ZPP;
    const int s_c = c_base + s_piece * PIECE;                 // channel of this thread's source piece
    const bool s_cok = s_c < CI;
    const bool s_src1 = s_cok && s_c >= p.C0;      // out-of-range pieces read (and discard) source 0: never a null source 1
    const T* s_ptr = reinterpret_cast<const T*>(s_src1 ? p.src1 : p.src0);
    const int s_Cs = s_src1 ? p.C1 : p.C0;
    const int s_cc = s_cok ? (s_src1 ? s_c - p.C0 : s_c) : 0;
    const float* s_sc = s_src1 ? p.sc1 : p.sc0;
    const float* s_sh = s_src1 ? p.sh1 : p.sh0;
    const float s_slope = s_src1 ? p.slope1 : p.slope0;
    const int z_col = n_base + z_piece * PIECE;
    const bool z_cok = z_col < p.CO;
    const int z_cc = z_cok ? z_col : 0;
    int s_geo[NS], s_lds[NS], z_geo[NZ], z_lds[NZ];
    unsigned s_exist = 0, z_exist = 0;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        const int i = tid + j * 256;
        const bool ex = i < p.s_halo * SPP;
        const int hp = ex ? i / SPP : 0;
        const int im = hp / s_hpi, rem = hp - im * s_hpi;
        const int hy = rem / p.SHW, hx = rem - hy * p.SHW;
        s_geo[j] = (im << 20) | (hy << 10) | hx;
        s_lds[j] = hp * ROWS_B + s_piece * 16;
        s_exist |= (ex ? 1u : 0u) << j;
    }
#pragma unroll
    for (int j = 0; j < NZ; ++j) {
        const int i = tid + j * 256;
        const bool ex = i < p.z_halo * ZPP;
        const int hp = ex ? i / ZPP : 0;
        const int im = hp / z_hpi, rem = hp - im * z_hpi;
        const int hy = rem / p.ZHW, hx = rem - hy * p.ZHW;
        z_geo[j] = (im << 20) | (hy << 10) | hx;
        z_lds[j] = hp * ROWZ_B + z_piece * 16;
        z_exist |= (ex ? 1u : 0u) << j;
    }

    u32x4 sreg[NS], zreg[NZ];
    int s_img[NS];
    unsigned s_inb = 0, z_inb = 0;

    auto prefetch = [&](int tile) {
        int bx = tile;
        const int tile_x = bx % p.tiles_x; bx /= p.tiles_x;
        const int tile_y = bx % p.tiles_y;
        const int ig = bx / p.tiles_y;
        const int py0 = tile_y << p.thl, px0 = tile_x << p.twl, img0 = ig << p.iml;
        const int sy0 = py0 * p.IS + p.sdymin, sx0 = px0 * p.IS + p.sdxmin;
        const int zy0 = py0 * p.ZS + p.zdymin, zx0 = px0 * p.ZS + p.zdxmin;
        s_inb = 0; z_inb = 0;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int im = s_geo[j] >> 20, hy = (s_geo[j] >> 10) & 1023, hx = s_geo[j] & 1023;
            const int n = img0 + im, sy = sy0 + hy, sx = sx0 + hx;
            const bool inb = ((s_exist >> j) & 1u) && s_cok && n < p.N && sy >= 0 && sy < p.SH && sx >= 0 && sx < p.SW;
            const size_t pix = inb ? (size_t)(n * p.SH + sy) * p.SW + sx : 0;
            s_img[j] = inb ? n : 0;
            s_inb |= (inb ? 1u : 0u) << j;
            sreg[j] = *reinterpret_cast<const u32x4*>(s_ptr + pix * s_Cs + s_cc);
        }
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            const int im = z_geo[j] >> 20, hy = (z_geo[j] >> 10) & 1023, hx = z_geo[j] & 1023;
            const int n = img0 + im, zy = zy0 + hy, zx = zx0 + hx;
            const bool inb = ((z_exist >> j) & 1u) && z_cok && n < p.N && zy >= 0 && zy < p.ZH && zx >= 0 && zx < p.ZW;
            const size_t pix = inb ? (size_t)(n * p.ZH + zy) * p.ZW + zx : 0;
            z_inb |= (inb ? 1u : 0u) << j;
            zreg[j] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.z) + pix * p.ZC + z_cc);
        }
    };

    auto commit = [&]() {
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            if ((s_exist >> j) & 1u) {
                u32x4 v = sreg[j];
                const bool inb = (s_inb >> j) & 1u;
                if constexpr (!PLAIN) {
                    float f[PIECE];
                    if constexpr (PIECE == 8) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            f[2 * e] = __uint_as_float(v[e] << 16);
                            f[2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) f[e] = __uint_as_float(v[e]);
                    }
                    if (s_sc != nullptr) {
                        const float* scp = s_sc + (size_t)s_img[j] * s_Cs + s_cc;
                        const float* shp = s_sh + (size_t)s_img[j] * s_Cs + s_cc;
#pragma unroll
                        for (int e = 0; e < PIECE; ++e) f[e] = f[e] * scp[e] + shp[e];
                    }
#pragma unroll
                    for (int e = 0; e < PIECE; ++e) f[e] = f[e] > 0.f ? f[e] : f[e] * s_slope;
                    if constexpr (PIECE == 8) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            v[e] = (unsigned)f32_to_bf16(f[2 * e]) | ((unsigned)f32_to_bf16(f[2 * e + 1]) << 16);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = __float_as_uint(f[e]);
                    }
                }
                if (!inb) v = u32x4{0u, 0u, 0u, 0u};
                *reinterpret_cast<u32x4*>(Ss + s_lds[j]) = v;
            }
        }
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            if ((z_exist >> j) & 1u) {
                u32x4 v = zreg[j];
                if (!((z_inb >> j) & 1u)) v = u32x4{0u, 0u, 0u, 0u};
                *reinterpret_cast<u32x4*>(Zs + z_lds[j]) = v;
            }
        }
    };

    int tile = blockIdx.y;
    if (tile < p.ntiles) prefetch(tile);
    for (; tile < p.ntiles; tile += p.splits) {
        __syncthreads();   // previous tile's fragment reads are done
        if (!CU_DBG(p, 4)) commit();
        __syncthreads();
        if (tile + p.splits < p.ntiles && !CU_DBG(p, 8)) prefetch(tile + p.splits);   // flies under this tile's MFMAs
        if (!wave_active || CU_DBG(p, 2)) continue;

        const int NK = p.tile_px / KPIX;
        if constexpr (sizeof(T) == 2) {
            // Software-pipelined k-loop: the fragments of k-step i+1 are read from LDS while the MFMAs of k-step i run
            // (ping-pong register buffers).  With one wave per SIMD nothing else hides the LDS latency.
            // lane l = 16g + 4q + pp supplies row q of its group's 4x16 transpose block
            const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
            const int hh = g >> 1, chalf = g & 1;
            const int s_col = (cblk * 32 + 16 * chalf + 4 * pp) * 2, z_col = (nblk * 32 + 16 * chalf + 4 * pp) * 2;
            const int NKW = NK / KSPLIT;                     // k-steps of this wave: kpart, kpart + KSPLIT, ...
            constexpr int NA = NTAPS == 4 ? NTAPS : 1;       // only the 2x2 transposed conv shifts Z per tap
            auto load = [&](int i, bf16x8 (&A)[NA], bf16x8 (&B)[NTAPS]) {
                const int ks = kpart + (i < NKW ? i : NKW - 1) * KSPLIT;    // clamped: the tail re-reads a valid step
                int sbase[2], zbase[2];
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int m = ks * 16 + 8 * hh + 4 * half + q;
                    const int tx = m & (TW - 1), ty = (m >> p.twl) & (TH - 1), im = m >> (p.twl + p.thl);
                    sbase[half] = (im * s_hpi + ty * p.IS * p.SHW + tx * p.IS) * ROWS_B + s_col;
                    zbase[half] = (im * z_hpi + ty * p.ZS * p.ZHW + tx * p.ZS) * ROWZ_B + z_col;
                }
#pragma unroll
                for (int t = 0; t < NA; ++t) {
                    const bf16x4 a0 = tr_read(Zs + zbase[0] + p.z_off[t] * ROWZ_B);
                    const bf16x4 a1 = tr_read(Zs + zbase[1] + p.z_off[t] * ROWZ_B);
                    A[t] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    const bf16x4 b0 = tr_read(Ss + sbase[0] + p.s_off[t] * ROWS_B);
                    const bf16x4 b1 = tr_read(Ss + sbase[1] + p.s_off[t] * ROWS_B);
                    B[t] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            };
            auto mma = [&](const bf16x8 (&A)[NA], const bf16x8 (&B)[NTAPS]) {
#pragma unroll
                for (int t = 0; t < NTAPS; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[NA == 1 ? 0 : t], B[t], acc[t], 0, 0, 0);
            };
            bf16x8 A0[NA], B0[NTAPS], A1[NA], B1[NTAPS];
            // one pipeline stage: issue the reads of step `nxt` while the MFMAs of the current step run; the group
            // barriers pin the order "1 MFMA, 2 LDS reads" so that no MFMA waits on a read issued in the same stage
            auto stage = [&](const bf16x8 (&Ac)[NA], const bf16x8 (&Bc)[NTAPS], int nxt, bf16x8 (&An)[NA], bf16x8 (&Bn)[NTAPS]) {
                __builtin_amdgcn_sched_barrier(0);
                load(nxt, An, Bn);
                mma(Ac, Bc);
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * NA, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            load(0, A0, B0);
            int i = 0;
            for (; i + 2 <= NKW; i += 2) {
                stage(A0, B0, i + 1, A1, B1);
                stage(A1, B1, i + 2, A0, B0);
            }
            if (i < NKW) mma(A0, B0);
        } else {
            for (int ks = kpart; ks < NK; ks += KSPLIT) {
                const int r = lane & 31, hh = lane >> 5;
                const int m = ks * 2 + hh;
                const int tx = m & (TW - 1), ty = (m >> p.twl) & (TH - 1), im = m >> (p.twl + p.thl);
                const int sbase = (im * s_hpi + ty * p.IS * p.SHW + tx * p.IS) * ROWS_B + (cblk * 32 + r) * 4;
                const int zbase = (im * z_hpi + ty * p.ZS * p.ZHW + tx * p.ZS) * ROWZ_B + (nblk * 32 + r) * 4;
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    const float a = *reinterpret_cast<const float*>(Zs + zbase + p.z_off[t] * ROWZ_B);
                    const float b = *reinterpret_cast<const float*>(Ss + sbase + p.s_off[t] * ROWS_B);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                }
            }
        }
    }

    if (CU_DBG(p, 1)) return;
    // ---- atomics: col (lane&31) = c, rows = n
    const int r = lane & 31, hh = lane >> 5;
    const int c = c_base + cblk * 32 + r;
    auto flush = [&]() {
        if (!wave_active || c >= CI) return;
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int n = n_base + nblk * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                if (n < p.CO) unsafeAtomicAdd(p.dw + ((size_t)p.tap_w[t] * p.CO + n) * CI + c, acc[t][i]);
            }
        }
    };
    if (p.part_stride) {      // partial-tile mode: the slab of (pixel split, k-part)
        if (wave_active)
            store_native<NTAPS>(p.dw + (size_t)(blockIdx.y * KSPLIT + kpart) * p.part_stride, blockIdx.x, NWB, blk, p.wtaps,
                                p.tap_w, acc, lane);
        return;
    }
    if constexpr (KSPLIT > 1) {
        // one pixel split (cu_wgrad_desc.splits == 1, the deterministic mode): this workgroup is the only adder of its dW
        // block, and its KSPLIT k-parts add one after the other -- a fixed summation order
        if (p.splits == 1) {
#pragma unroll
            for (int kp = 0; kp < KSPLIT; ++kp) {
                if (kp == kpart) flush();
                __threadfence();
                __syncthreads();
            }
            return;
        }
    }
    flush();
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16 production variant: staging by LDS-DMA (buffer_load_dwordx4 ... lds), two LDS images, software-pipelined k-loop.
//   * no staging registers and no commit phase: the DMA of tile i+1 is in flight during the whole k-loop of tile i and
//     is retired by the vmcnt(0) + barrier at the top of the next tile (one barrier per tile);
//   * zero padding (image border, ragged tiles, channel tail) comes from the buffer range check: lanes that must read
//     zeros get an out-of-range offset;
//   * the LDS image is lane-linear (16-byte slot i <- staging item i).  Items are ordered [32-channel plane][halo pixel]
//     [4 pieces]: 64-byte rows, so the 4 rows of a transpose read fall on 4 distinct bank quarters without padding or
//     swizzle, and a wave's block (cblk / nblk) simply selects its plane.
constexpr int DMA_IMG_BYTES = 78 * 1024;      // per image; two of them per workgroup
constexpr int WGRAD_TWO_DEFAULT = 0;          // round 4 A/B: profiles/r04_wgrad_two_per_cu.txt

// NW = 8 waves: two waves per SIMD own the SAME 32x32 block and split the k-steps; while one is blocked issuing its
// DMA instructions (the queue drains at L2 speed, ~4 us per tile) the other keeps the MFMA pipe busy.  The k-split
// partial sums are combined through LDS before the atomics, so the atomic traffic does not grow with the wave count.
// PC (producer / consumer, NW = 8): waves 0..3 run the k-loop over the whole tile (software-pipelined, one per SIMD),
// waves 4..7 only issue the LDS-DMA of the next tile into the other image.  The DMA queue drains at L2 speed and blocks
// the wave that issues into it; in this split that wave has nothing else to do, and its SIMD keeps issuing the other
// wave's MFMAs, so staging and k-loop overlap instead of adding up.
// XF (round 3): source 0 is the RAW output z of the producing layer; its InstanceNorm + LeakyReLU (sc0 / sh0 = the scale and
// shift planes [N][C0], slope0) is applied to the staged source image IN PLACE in LDS, once per staged element, between the
// barrier that publishes the tile and a second one that hands it to the k-loop.  The image's scale / shift rows arrive by one
// more LDS-DMA instruction per tile (1-KiB table beside each image); out-of-image halo pixels stay zero.  One image per tile.
// IMGB = bytes of one LDS image.  The default fills the CU with ONE workgroup; the thin layers (32-row dW blocks at 256^2 / 128^2,
// HBM-bound) also have a half-size form with four waves -- IMGB = 39 KiB, 170 VGPRs -- of which TWO workgroups share a CU: one
// stages (its waves blocked in the LDS-DMA queue) while the other runs its k-loop (round 4).
template <int NBLK, int CBLK, int NTAPS, int NW, bool PC = false, bool ROWK = false, bool XF = false, int IMGB = DMA_IMG_BYTES>
__global__ __launch_bounds__(64 * NW) void igemm_wgrad_dma_kernel(const WgKArgs p) {
    constexpr int TN = 32 * NBLK, TC = 32 * CBLK;
    constexpr int NWC = PC ? NW / 2 : NW;              // waves that compute
    constexpr int NWI = PC ? NW / 2 : NW;              // waves that issue DMA
    constexpr int NWB = NBLK * CBLK, KSPLIT = NWC / NWB;
    constexpr int NTHR = 64 * NWI, BLK_B = NTHR * 16;  // issuing threads; bytes of one staging round
    constexpr int ROW_B = 64;                          // one pixel of one 32-channel plane
    __shared__ __attribute__((aligned(16))) unsigned char img2[2 * IMGB + (XF ? 3072 : 0)];
    unsigned char* imgA = img2;
    unsigned char* imgB = img2 + IMGB;
    unsigned char* xtab = img2 + 2 * IMGB;     // XF: table of image A, table of image B, dump kilobyte
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool producer = PC && wave >= NWC, issuer = !PC || producer;
    const int tid = PC ? (threadIdx.x & (NTHR - 1)) : threadIdx.x;      // index among the issuing threads
    const int iwave = tid >> 6;
    const int blk = wave % NWB, kpart = producer ? 2 * KSPLIT : wave / NWB;
    const int nblk = blk / CBLK, cblk = blk % CBLK;
    const int TW = 1 << p.twl, TH = 1 << p.thl;
    const int CI = p.C0 + p.C1;
    // XCD-aware placement: workgroups are dealt round-robin to the 8 XCDs (each with its own L2).  Workgroups of the
    // same pixel split read the same S / Z tiles, so when the split count allows it all (n, c) tiles of a split are put
    // on one XCD: hardware id h -> xcd = h % 8, slot = h / 8; that XCD owns splits xcd, xcd + 8, ...
    int bxi = blockIdx.x, byi = blockIdx.y;
    if ((p.splits & 7) == 0 && !CU_DBG(p, 64)) {
        const int gx = gridDim.x;
        const int hid = blockIdx.x + gx * blockIdx.y;
        const int xcd = hid & 7, slot = hid >> 3;
        bxi = slot % gx;
        byi = xcd + 8 * (slot / gx);
    }
    const int ct = bxi % p.ctiles, nt = bxi / p.ctiles;
    const int c_base = ct * TC, n_base = nt * TN;
    const int s_hpi = p.s_hpi, z_hpi = p.ZHH * p.ZHW;
    const bool wave_active = !producer && (n_base + nblk * 32 < p.CO) && (c_base + cblk * 32 < CI);
    const int s_img_bytes = p.s_iters * BLK_B;

    const i32x4 rs0 = make_rsrc(p.src0, p.src0_bytes);
    const i32x4 rs1 = make_rsrc(p.src1 ? p.src1 : p.src0, p.src1 ? p.src1_bytes : 0u);
    const i32x4 rz = make_rsrc(p.z, p.z_bytes);
    constexpr unsigned OOB = 0x7ffffff0u;

    f32x16 acc[NTAPS];
#pragma unroll
    for (int t = 0; t < NTAPS; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    const int slot = tid & 3;       // NTHR % 4 == 0: a thread always stages the same 16-byte piece of a pixel
    // XF: table ring of 3 slots (512 B each: scale[TC], shift[TC] of a tile's image, channels [c_base, c_base + TC)).
    // Wave 0 writes the table of logical tile L into slot (L / splits) % 3 two tiles ahead of its use, so the barrier at the
    // top of the tile in between publishes it: no synchronisation of its own.
    auto tile_img = [&](int ltile) {
        const int bx = ((p.ntiles & 7) == 0 && !CU_DBG(p, 64)) ? (ltile & 7) * (p.ntiles >> 3) + (ltile >> 3) : ltile;
        return (bx / (p.tiles_x * p.tiles_y)) << p.iml;
    };
    auto write_table = [&](int ltile) {
        const int n = tile_img(ltile);
        float* tab = reinterpret_cast<float*>(xtab) + ((ltile / p.splits) % 3) * 128;
#pragma unroll
        for (int k = 0; k < 2 * TC / 64; ++k) {
            const int i = lane + 64 * k;                     // [0, TC): scale, [TC, 2 TC): shift
            const bool sh = i >= TC;
            const int c = c_base + i - (sh ? TC : 0);
            tab[i] = c < p.C0 ? p.sc0[(size_t)((sh ? p.N : 0) + n) * p.C0 + c] : 0.f;
        }
    };

    // staging of one tile = s_iters + z_iters wave-level DMA instructions per wave, issued back to back right after the
    // barrier (measured: spreading them over the k-steps stalls the MFMA pipeline far more than it hides, 157 -> 191 us)
    int g_img0 = 0, g_sy0 = 0, g_sx0 = 0, g_zy0 = 0, g_zx0 = 0;
    // logical tile L -> tile (L % 8) * ntiles/8 + L / 8: the workgroups of one XCD walk a contiguous eighth of the pixel
    // tiles at a time, so the halo rows shared by neighbouring tiles are L2 hits (thin layers: splits = 256 workgroups
    // with consecutive logical tiles)
    const bool xcd_tiles = (p.ntiles & 7) == 0 && !CU_DBG(p, 64);
    auto tile_geo = [&](int ltile) {
        int bx = xcd_tiles ? (ltile & 7) * (p.ntiles >> 3) + (ltile >> 3) : ltile;
        const int tile_x = bx % p.tiles_x; bx /= p.tiles_x;
        const int tile_y = bx % p.tiles_y;
        const int ig = bx / p.tiles_y;
        const int py0 = tile_y << p.thl, px0 = tile_x << p.twl;
        g_img0 = ig << p.iml;
        g_sy0 = py0 * p.IS + p.sdymin; g_sx0 = px0 * p.IS + p.sdxmin;
        g_zy0 = py0 * p.ZS + p.zdymin; g_zx0 = px0 * p.ZS + p.zdxmin;
    };
    auto issue_item = [&](int j, unsigned char* img) {
        if (j < p.s_iters) {
            const int i = (tid + j * NTHR) >> 2;                         // plane-major pixel index
            const int pl = (CBLK == 2 && i >= p.s_halo) ? 1 : 0;
            const int hp = i - pl * p.s_halo;
            const int im = __umulhi((unsigned)hp, p.mg_shpi), rem = hp - im * s_hpi;
            int hy, hx, sy, sx;
            bool used = true;
            if (p.s_deint) {          // parity plane a*2+b, then (hy, hx) inside it; row / column 0 of an even plane is no tap's
                const int par = __umulhi((unsigned)rem, p.mg_spl), r2 = rem - par * p.s_plane;
                hy = __umulhi((unsigned)r2, p.mg_shw); hx = r2 - hy * p.SHW;
                const int pa = par >> 1, pb = par & 1;
                sy = g_sy0 + 2 * hy + pa; sx = g_sx0 + 2 * hx + pb;
                used = hy + pa > 0 && hx + pb > 0;
            } else {
                hy = __umulhi((unsigned)rem, p.mg_shw); hx = rem - hy * p.SHW;
                sy = g_sy0 + hy; sx = g_sx0 + hx;
            }
            const int c = c_base + pl * 32 + slot * 8;
            const int n = g_img0 + im;
            const bool ok = used && hp < p.s_halo && c < CI && n < p.N && sy >= 0 && sy < p.SH && sx >= 0 && sx < p.SW;
            const bool s1 = c >= p.C0;
            const unsigned pix = (unsigned)((n * p.SH + sy) * p.SW + sx);
            const unsigned off = ok ? (pix * (unsigned)(s1 ? p.C1 : p.C0) + (unsigned)(s1 ? c - p.C0 : c)) * 2u : OOB;
            const unsigned dst = lds_addr(img) + iwave * 1024 + j * BLK_B;
            if (s1) dma16(rs1, off, dst);
            else dma16(rs0, off, dst);
        } else {
            const int jz = j - p.s_iters;
            const int i = (tid + jz * NTHR) >> 2;
            int pl = 0;                     // 32-column plane of this item (NBLK planes, z_halo rows each)
#pragma unroll
            for (int q = 1; q < NBLK; ++q) pl += i >= q * p.z_halo ? 1 : 0;
            const int hp = i - pl * p.z_halo;
            const int im = __umulhi((unsigned)hp, p.mg_zhpi), rem = hp - im * z_hpi;
            const int hy = __umulhi((unsigned)rem, p.mg_zhw), hx = rem - hy * p.ZHW;
            const int col = n_base + pl * 32 + slot * 8;
            const int n = g_img0 + im, zy = g_zy0 + hy, zx = g_zx0 + hx;
            const bool ok = hp < p.z_halo && col < p.CO && n < p.N && zy >= 0 && zy < p.ZH && zx >= 0 && zx < p.ZW;
            const unsigned pix = (unsigned)((n * p.ZH + zy) * p.ZW + zx);
            const unsigned off = ok ? (pix * (unsigned)p.ZC + (unsigned)col) * 2u : OOB;
            dma16(rz, off, lds_addr(img) + s_img_bytes + iwave * 1024 + jz * BLK_B);
        }
    };
    const int n_items = p.s_iters + p.z_iters;
    // XF: rewrite source item j of image `img` (the 16-byte piece THIS thread staged) as LeakyReLU(scale z + shift); g_*
    // describe the tile of `img`; out-of-image halo pixels stay zero
    auto xf_item = [&](int j, unsigned char* img, const float* tab) {
        if (j >= p.s_iters) return;
        const int i = (tid + j * NTHR) >> 2;
        const int pl = (CBLK == 2 && i >= p.s_halo) ? 1 : 0;
        const int hp = i - pl * p.s_halo;
        const int hy = __umulhi((unsigned)hp, p.mg_shw), hx = hp - hy * p.SHW;
        const int sy = g_sy0 + hy, sx = g_sx0 + hx, cc = pl * 32 + slot * 8;
        if (hp < p.s_halo && sy >= 0 && sy < p.SH && sx >= 0 && sx < p.SW && c_base + cc < p.C0) {
            u32x4* pp = reinterpret_cast<u32x4*>(img + (size_t)(tid + j * NTHR) * 16);
            u32x4 v = *pp;
            const f32x4 sa = *reinterpret_cast<const f32x4*>(tab + cc), sb = *reinterpret_cast<const f32x4*>(tab + cc + 4);
            const f32x4 ha = *reinterpret_cast<const f32x4*>(tab + TC + cc), hb = *reinterpret_cast<const f32x4*>(tab + TC + cc + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float z0 = __uint_as_float(v[e] << 16), z1 = __uint_as_float(v[e] & 0xffff0000u);
                const float s0 = e < 2 ? sa[2 * e] : sb[2 * e - 4], s1 = e < 2 ? sa[2 * e + 1] : sb[2 * e - 3];
                const float h0 = e < 2 ? ha[2 * e] : hb[2 * e - 4], h1 = e < 2 ? ha[2 * e + 1] : hb[2 * e - 3];
                float a0 = z0 * s0 + h0, a1 = z1 * s1 + h1;
                a0 = a0 > 0.f ? a0 : a0 * p.slope0;
                a1 = a1 > 0.f ? a1 : a1 * p.slope0;
                v[e] = (unsigned)f32_to_bf16(a0) | ((unsigned)f32_to_bf16(a1) << 16);
            }
            *pp = v;
        }
    };
    // the items this wave issued for logical tile `ltile` (same loops as the issue sites below)
    auto xf_own = [&](int ltile, unsigned char* img) {
        const float* tab = reinterpret_cast<const float*>(xtab) + ((ltile / p.splits) % 3) * 128;
        if (ltile == byi) {                       // first image
            if constexpr (PC) {
                const int half = n_items / 2;
                for (int j = producer ? 0 : half; j < (producer ? half : n_items); ++j) xf_item(j, img, tab);
            } else {
                for (int j = 0; j < n_items; ++j) xf_item(j, img, tab);
            }
            return;
        }
        if constexpr (PC) {
            if (producer) {
                for (int j = p.pc_early; j < p.pc_items; ++j) xf_item(j, img, tab);
            } else {
                for (int j = 0; j < p.pc_early; ++j) xf_item(j, img, tab);
                for (int j = p.pc_items; j < n_items; ++j) xf_item(j, img, tab);
            }
        } else {
            for (int j = 0; j < n_items; ++j) xf_item(j, img, tab);
        }
    };
    if (PC && producer && CU_DBG(p, 128)) __builtin_amdgcn_s_setprio(3);

    // lane l = 16g + 4q + pp supplies row q of its group's 4x16 transpose block
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int hh = g >> 1, chalf = g & 1;
    const int s_col = cblk * p.s_halo * ROW_B + (16 * chalf + 4 * pp) * 2;      // plane base + column inside the row
    const int z_colb = nblk * p.z_halo * ROW_B + (16 * chalf + 4 * pp) * 2;
    int s_lane[2], z_lane[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        s_lane[half] = (8 * hh + 4 * half + q) * p.ISL * ROW_B + s_col;
        z_lane[half] = (8 * hh + 4 * half + q) * p.ZS * ROW_B + z_colb;
    }
    const int NK = p.tile_px / 16;
    const int NKW = NK / KSPLIT;
    constexpr int NA = NTAPS == 4 ? NTAPS : 1;

    // one tile: wait for its image, start the DMA of the next tile into the other image, run the k-loop
    long long t_wait = 0, t_bar = 0, t_issue = 0, t_loop = 0;     // CU_CONV_DBG bit 16: phase stamps of one workgroup
    auto run_tile = [&](int tile, unsigned char* cur, unsigned char* other) {
        const long long c0 = CU_DBG(p, 16) ? wall_clock64() : 0;
        dma_wait();           // this wave's share of the tile has landed ...
        if constexpr (XF) xf_own(tile, cur);      // ... and is normalised + activated in place by the wave that staged it
        const long long c1 = CU_DBG(p, 16) ? wall_clock64() : 0;
        __syncthreads();      // ... everybody's has, and the other image is no longer being read
        const long long c2 = CU_DBG(p, 16) ? wall_clock64() : 0;
        const bool more = tile + p.splits < p.ntiles && !CU_DBG(p, 8);
        if constexpr (XF) {   // wave 0: scale / shift rows of the tile after next -> table ring (published by the next barrier)
            if (wave == 0 && tile + 2 * p.splits < p.ntiles) write_table(tile + 2 * p.splits);
        }
        if (more) tile_geo(tile + p.splits);
        if (more && issuer) {
            // all DMA instructions now: spreading them over the k-steps was slower with one wave per SIMD (it stalls
            // the software pipeline, 157 -> 191 us) and with two (157 -> 181 us).  PC: the producers take the first
            // pc_items rounds, the computing waves the rest once their k-loop is done (they would otherwise idle at
            // the barrier: under the k-loop's LDS traffic the DMA queue drains slower than the MFMAs finish).
            const int n_mine = PC ? p.pc_items : n_items;
            for (int item = PC ? p.pc_early : 0; item < n_mine; ++item) issue_item(item, other);
        }
        if constexpr (PC) {       // a few rounds fit the computing waves' empty DMA queue without blocking them
            if (more && !producer)
                for (int item = 0; item < p.pc_early; ++item) issue_item(item, other);
        }
        const long long c3 = CU_DBG(p, 16) ? wall_clock64() : 0;
        t_wait += c1 - c0; t_bar += c2 - c1; t_issue += c3 - c2; t_loop -= c3;
        auto late_items = [&]() {
            if constexpr (PC) {
                if (more && !producer)
                    for (int item = p.pc_items; item < n_items; ++item) issue_item(item, other);
            }
        };
        if (!wave_active || CU_DBG(p, 2)) { late_items(); return; }
        const unsigned char* Ss = cur;
        const unsigned char* Zs = cur + s_img_bytes;
        auto load = [&](int i, bf16x8 (&A)[NA], bf16x8 (&B)[NTAPS]) {
            // the k-part is wave-uniform: in a scalar register the whole k-step addressing below is scalar arithmetic
            // instead of quarter-rate vector multiplies (they, not the MFMAs or the LDS, bounded the k-loop)
            const int kp = KSPLIT == 1 ? 0 : __builtin_amdgcn_readfirstlane(kpart);
            const int ks = (ROWK ? kp : kpart) + (i < NKW ? i : NKW - 1) * KSPLIT;
            int sbase[2], zbase[2];
            if constexpr (ROWK) {   // tile rows of >= 16 pixels: a k-step lies in one row -> scalar row base + per-lane offset
                const int m0 = ks * 16;
                const int tx0 = m0 & (TW - 1), ty = (m0 >> p.twl) & (TH - 1), im = m0 >> (p.twl + p.thl);
                const int srow = (im * s_hpi + ty * p.ISL * p.SHW + tx0 * p.ISL) * ROW_B;
                const int zrow = (im * z_hpi + ty * p.ZS * p.ZHW + tx0 * p.ZS) * ROW_B;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    sbase[half] = srow + s_lane[half];
                    zbase[half] = zrow + z_lane[half];
                }
            } else {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int m = ks * 16 + 8 * hh + 4 * half + q;
                const int tx = m & (TW - 1), ty = (m >> p.twl) & (TH - 1), im = m >> (p.twl + p.thl);
                sbase[half] = (im * s_hpi + ty * p.ISL * p.SHW + tx * p.ISL) * ROW_B + s_col;
                zbase[half] = (im * z_hpi + ty * p.ZS * p.ZHW + tx * p.ZS) * ROW_B + z_colb;
            }
            }
#pragma unroll
            for (int t = 0; t < NA; ++t) {
                const bf16x4 a0 = tr_read(Zs + zbase[0] + p.z_off[t] * ROW_B);
                const bf16x4 a1 = tr_read(Zs + zbase[1] + p.z_off[t] * ROW_B);
                A[t] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                const bf16x4 b0 = tr_read(Ss + sbase[0] + p.s_off[t] * ROW_B);
                const bf16x4 b1 = tr_read(Ss + sbase[1] + p.s_off[t] * ROW_B);
                B[t] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        };
        auto mma = [&](const bf16x8 (&A)[NA], const bf16x8 (&B)[NTAPS]) {
#pragma unroll
            for (int t = 0; t < NTAPS; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[NA == 1 ? 0 : t], B[t], acc[t], 0, 0, 0);
        };
        if constexpr (NW == 4 || PC) {      // one computing wave per SIMD: software pipeline (ping-pong fragment registers)
            bf16x8 A0[NA], B0[NTAPS], A1[NA], B1[NTAPS];
            auto stage = [&](const bf16x8 (&Ac)[NA], const bf16x8 (&Bc)[NTAPS], int nxt, bf16x8 (&An)[NA], bf16x8 (&Bn)[NTAPS]) {
                __builtin_amdgcn_sched_barrier(0);
                load(nxt, An, Bn);
                mma(Ac, Bc);
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * NA, 0);
                __builtin_amdgcn_sched_barrier(0);
            };
            load(0, A0, B0);
            int i = 0;
            for (; i + 2 <= NKW; i += 2) {
                stage(A0, B0, i + 1, A1, B1);
                stage(A1, B1, i + 2, A0, B0);
            }
            if (i < NKW) mma(A0, B0);
        } else {                      // two waves per SIMD cover each other's LDS latency: keep the register count low
            for (int i = 0; i < NKW; ++i) {
                bf16x8 A0[NA], B0[NTAPS];
                load(i, A0, B0);
                mma(A0, B0);
            }
        }
        if (CU_DBG(p, 16)) t_loop += wall_clock64();
        late_items();
    };

    int tile = byi;
    const long long k0 = CU_DBG(p, 16) ? wall_clock64() : 0;
    if (tile < p.ntiles) {
        tile_geo(tile);
        if constexpr (PC) {       // first image: both wave groups issue half of the rounds
            const int half = n_items / 2;
            for (int j = producer ? 0 : half; j < (producer ? half : n_items); ++j) issue_item(j, imgA);
        } else {
            for (int j = 0; j < n_items; ++j) issue_item(j, imgA);
        }
    }
    if constexpr (XF) {           // tables of this workgroup's first two tiles (the later ones: run_tile, two tiles ahead)
        if (wave == 0) {
            if (tile < p.ntiles) write_table(tile);
            if (tile + p.splits < p.ntiles) write_table(tile + p.splits);
        }
        __syncthreads();
    }
    const long long k1 = CU_DBG(p, 16) ? wall_clock64() : 0;
    for (; tile < p.ntiles; tile += 2 * p.splits) {
        run_tile(tile, imgA, imgB);
        if (tile + p.splits < p.ntiles) run_tile(tile + p.splits, imgB, imgA);
    }
    if (CU_DBG(p, 16) && bxi == 0 && byi == 0 && (tid & 63) == 0)
        printf("[wgrad wave %d] 100MHz ticks: first issue %lld, wait %lld, barrier %lld, issue %lld, k-loop %lld, total %lld (tiles %d)\n",
               wave, k1 - k0, t_wait, t_bar, t_issue, t_loop, wall_clock64() - k0, (p.ntiles - byi + p.splits - 1) / p.splits);

    // ---- combine the k-split partial sums of one block through LDS (tree over kpart), then one set of atomics
    if constexpr (KSPLIT > 1) {
        float* red = reinterpret_cast<float*>(img2);
        constexpr int REG_F = NTAPS * 16 * 64;            // floats of one wave's accumulators
        static_assert((size_t)(KSPLIT / 2) * NWB * REG_F * 4 <= 2 * (size_t)IMGB, "reduction scratch exceeds the images");
#pragma unroll
        for (int sp = KSPLIT / 2; sp >= 1; sp >>= 1) {
            __syncthreads();
            if (kpart >= sp && kpart < 2 * sp) {
                float* d = red + (size_t)((kpart - sp) * NWB + blk) * REG_F + lane;
#pragma unroll
                for (int t = 0; t < NTAPS; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) d[(t * 16 + i) * 64] = acc[t][i];
            }
            __syncthreads();
            if (kpart < sp) {
                const float* d = red + (size_t)(kpart * NWB + blk) * REG_F + lane;
#pragma unroll
                for (int t = 0; t < NTAPS; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[t][i] += d[(t * 16 + i) * 64];
            }
        }
    }
    if (!wave_active || kpart != 0 || CU_DBG(p, 1)) return;
    if (p.part_stride) {      // partial-tile mode: the slab of this pixel split
        store_native<NTAPS>(p.dw + (size_t)byi * p.part_stride, bxi, NWB, blk, p.wtaps, p.tap_w, acc, lane);
        return;
    }
    const int r = lane & 31, h2 = lane >> 5;
    const int c = c_base + cblk * 32 + r;
    if (c >= CI) return;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int n = n_base + nblk * 32 + (i & 3) + 8 * (i >> 2) + 4 * h2;
            if (n < p.CO) unsafeAtomicAdd(p.dw + ((size_t)p.tap_w[t] * p.CO + n) * CI + c, acc[t][i]);
        }
    }
}

// partial-tile mode: slab size of this block shape, split cap from the workspace size, results for the caller.
// The workspace must also hold the plain [wtaps][CO][CI] sum cu_grad_unprep_parts forms behind the slabs.
static int parts_setup(WgKArgs& a, int nblk, int cblk, int ntn, int kparts) {
    a.part_stride = (size_t)ntn * a.ctiles * nblk * cblk * a.wtaps * 1024;
    const size_t plain = (size_t)a.wtaps * a.CO * (a.C0 + a.C1);
    CU_CHECK_ARG(a.parts_floats >= plain + (size_t)kparts * a.part_stride,
                 "cu_conv_wgrad_parts: workspace of %zu floats; this shape needs >= %zu", a.parts_floats,
                 plain + (size_t)kparts * a.part_stride);
    size_t cap = (a.parts_floats - plain) / a.part_stride / kparts;
    if (cap > 1024) cap = 1024;
    if ((size_t)a.splits > cap) a.splits = (int)cap;
    *a.nparts_out = a.splits * kparts;
    *a.layout_out = (nblk << 8) | cblk;
    return 0;
}

template <int NBLK, int CBLK, int NTAPS, int NW, bool PC = false, bool ROWK = false, bool XF = false, int IMGB = DMA_IMG_BYTES>
int launch_dma(WgKArgs& a, hipStream_t st) {
    CU_CHECK_ARG((size_t)(a.s_iters + a.z_iters) * 1024 * (PC ? NW / 2 : NW) <= (size_t)IMGB, "cu_conv_wgrad: tile image exceeds %d bytes", IMGB);
    auto k = igemm_wgrad_dma_kernel<NBLK, CBLK, NTAPS, NW, PC, ROWK, XF, IMGB>;
    constexpr int PER_CU = IMGB < DMA_IMG_BYTES ? 2 : 1;          // workgroups that share a CU
    const int CI = a.C0 + a.C1;
    a.ctiles = cdiv(CI, 32 * CBLK);
    const int ntn = cdiv(a.CO, 32 * NBLK);
    if (a.splits <= 0) {
        // one workgroup per CU is resident (one wave per SIMD): every extra split only re-adds the dW tile atomically
        int want = cdiv(256, a.ctiles * ntn);
        a.splits = want < 1 ? 1 : want;
    }
    a.splits *= PER_CU;              // (an explicit split count is a count of CUs: cu_hip.engine's cap for the second stream)
    if (a.splits > a.ntiles) a.splits = a.ntiles;
    if (a.nparts_out) {
        const int rc = parts_setup(a, NBLK, CBLK, ntn, 1);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k, dim3(a.ctiles * ntn, a.splits), dim3(64 * NW), 0, st, a);
    CU_LAUNCH_CHECK();
    return 0;
}

template <typename T, int NBLK, int CBLK, int NS, int NZ, bool PLAIN, int NTAPS>
int launch_k(WgKArgs& a, hipStream_t st) {
    constexpr int TN = 32 * NBLK, TC = 32 * CBLK;
    constexpr int ROWS_B = (sizeof(T) == 2) ? (TC == 32 ? 64 : 192) : TC * 4 + 16;
    constexpr int ROWZ_B = (sizeof(T) == 2) ? (TN == 32 ? 64 : 192) : TN * 4 + 16;
    const size_t lds = (size_t)a.s_halo * ROWS_B + (size_t)a.z_halo * ROWZ_B;
    CU_CHECK_ARG(lds <= 160 * 1024, "cu_conv_wgrad: LDS %zu bytes exceeds 160 KiB", lds);
    auto k = igemm_wgrad_kernel<T, NBLK, CBLK, NS, NZ, PLAIN, NTAPS>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        CU_CHECK_ARG(e == hipSuccess, "cu_conv_wgrad: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    const int CI = a.C0 + a.C1;
    a.ctiles = cdiv(CI, TC);
    const int ntn = cdiv(a.CO, TN);
    if (a.splits <= 0) {
        // enough workgroups to fill 256 CUs a few times over, but never more splits than tiles
        int want = cdiv(512, a.ctiles * ntn);      // ~2 workgroups per CU: every extra split re-adds the whole dW tile atomically
        a.splits = want < 1 ? 1 : want;
    }
    if (a.splits > a.ntiles) a.splits = a.ntiles;
    if (a.nparts_out) {
        const int rc = parts_setup(a, NBLK, CBLK, ntn, 4 / (NBLK * CBLK));      // k-parts of a block: one slab each
        if (rc) return rc;
    }
    dim3 grid(a.ctiles * ntn, a.splits);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
    CU_LAUNCH_CHECK();
    return 0;
}

}  // namespace

static int wgrad_impl(const cu_wgrad_desc* d, const void* src0, const float* scale0, const float* shift0,
                      const void* src1, const float* scale1, const float* shift1, const void* z, float* dw,
                      size_t parts_floats, int* nparts, int* layout, void* stream) {
    CU_CHECK_ARG(d != nullptr, "cu_conv_wgrad: null descriptor");
    CU_CHECK_ARG(d->dtype == CU_F32 || d->dtype == CU_BF16, "cu_conv_wgrad: bad dtype %d", d->dtype);
    CU_CHECK_ARG(d->ntaps >= 1 && d->ntaps <= CU_MAX_TAPS, "cu_conv_wgrad: ntaps %d", d->ntaps);
    const int PIECE = d->dtype == CU_BF16 ? 8 : 4;
    CU_CHECK_ARG(d->C0 > 0 && d->C0 % PIECE == 0 && d->C1 >= 0 && d->C1 % PIECE == 0 && d->ZC % PIECE == 0 &&
                     d->CO % PIECE == 0 && d->CO <= d->ZC,
                 "cu_conv_wgrad: channel counts must be multiples of %d", PIECE);
    CU_CHECK_ARG(src0 && z && dw && (d->C1 == 0 || src1), "cu_conv_wgrad: null pointer");
    CU_CHECK_ARG((scale0 == nullptr) == (shift0 == nullptr) && (scale1 == nullptr) == (shift1 == nullptr),
                 "cu_conv_wgrad: scale/shift must come in pairs");

    WgKArgs a;
    memset(&a, 0, sizeof(a));
    a.src0 = src0; a.src1 = src1; a.sc0 = scale0; a.sh0 = shift0; a.sc1 = scale1; a.sh1 = shift1; a.z = z; a.dw = dw;
    a.N = d->N; a.PH = d->PH; a.PW = d->PW; a.SH = d->SH; a.SW = d->SW; a.C0 = d->C0; a.C1 = d->C1; a.IS = d->IS;
    a.ZH = d->ZH; a.ZW = d->ZW; a.ZC = d->ZC; a.ZS = d->ZS; a.CO = d->CO; a.ntaps = d->ntaps;
    a.slope0 = d->slope0; a.slope1 = d->slope1; a.splits = d->splits;
    a.dbg = cu_env_int("CU_CONV_DBG", 0);
    for (int t = 0; t < d->ntaps; ++t) a.wtaps = d->tap_w[t] + 1 > a.wtaps ? d->tap_w[t] + 1 : a.wtaps;
    if (nparts) {         // partial-tile mode (the launch function sizes the slabs for its block shape)
        a.parts_floats = parts_floats;
        a.nparts_out = nparts;
        a.layout_out = layout;
        a.part_stride = 1;      // != 0: the kernels store; launch_* sets the real stride
    }

    const int CI_all = d->C0 + d->C1;
    // 64-wide blocks of dW unless the loop grid is tiny (<= CU_WGRAD_SMALLPX pixels, tuning knob): then 32 x 32 blocks put
    // four times as many workgroups on a launch whose k-loop is a few steps long
    const long px_total = (long)d->N * d->PH * d->PW;
    const bool tiny = d->dtype == CU_BF16 && px_total <= cu_env_int("CU_WGRAD_SMALLPX", 0);
    const bool wn = d->CO > 32 && !tiny, wc = CI_all > 32 && !tiny;
    const int rows_b = d->dtype == CU_BF16 ? (wc ? 192 : 64) : (wc ? 64 : 32) * 4 + 16;
    const int rowz_b = d->dtype == CU_BF16 ? (wn ? 192 : 64) : (wn ? 64 : 32) * 4 + 16;
    const int kpix = d->dtype == CU_BF16 ? 16 : 2;
    const int piece_elems = d->dtype == CU_BF16 ? 8 : 4;
    const int ns_small = d->dtype == CU_BF16 ? 7 : 13, nz_small = d->dtype == CU_BF16 ? 4 : 8;
    const int ns_big = d->dtype == CU_BF16 ? 11 : 21, nz_big = d->dtype == CU_BF16 ? 8 : 16;
    // loop-pixel tile: 128 pixels (64 for stride-2 gathers: 4x the halo), halved until both staged patches fit in LDS
    // Thin blocks (32x32, 64x32) get bigger pixel tiles so that every wave still has >= 8 k-steps between barriers;
    // first pass keeps the patches within 80 KiB (two workgroups per CU), second pass takes what fits.
    // geometry of a loop-pixel tile of BM pixels (IMGS x TH x TW) and its two halo patches
    auto geometry = [&](int BM) -> int {
        a.tile_px = BM;
        int tw = d->PW < 32 ? d->PW : 32;
        if (tw > BM) tw = BM;
        int th = BM / tw;
        if (th > d->PH) th = d->PH;
        int imgs = BM / (tw * th);
        a.twl = ilog2_exact(tw); a.thl = ilog2_exact(th); a.iml = ilog2_exact(imgs);
        CU_CHECK_ARG(a.twl >= 0 && a.thl >= 0 && a.iml >= 0 && d->PW % tw == 0 && d->PH % th == 0,
                     "cu_conv_wgrad: loop grid %dx%d must be powers of two", d->PH, d->PW);
        a.tiles_x = d->PW / tw; a.tiles_y = d->PH / th; a.igroups = cdiv(d->N, imgs);
        a.ntiles = a.tiles_x * a.tiles_y * a.igroups;

        int ymin = 1 << 20, xmin = 1 << 20, ymax = -(1 << 20), xmax = -(1 << 20);
        int zymin = 1 << 20, zxmin = 1 << 20, zymax = -(1 << 20), zxmax = -(1 << 20);
        for (int t = 0; t < d->ntaps; ++t) {
            ymin = d->tap_dy[t] < ymin ? d->tap_dy[t] : ymin; ymax = d->tap_dy[t] > ymax ? d->tap_dy[t] : ymax;
            xmin = d->tap_dx[t] < xmin ? d->tap_dx[t] : xmin; xmax = d->tap_dx[t] > xmax ? d->tap_dx[t] : xmax;
            zymin = d->tap_zy[t] < zymin ? d->tap_zy[t] : zymin; zymax = d->tap_zy[t] > zymax ? d->tap_zy[t] : zymax;
            zxmin = d->tap_zx[t] < zxmin ? d->tap_zx[t] : zxmin; zxmax = d->tap_zx[t] > zxmax ? d->tap_zx[t] : zxmax;
        }
        a.sdymin = ymin; a.sdxmin = xmin;
        a.SHH = (th - 1) * d->IS + (ymax - ymin) + 1; a.SHW = (tw - 1) * d->IS + (xmax - xmin) + 1;
        a.ISL = d->IS;
        if (a.s_deint) {          // four parity planes of (th + 1) x (tw + 1) pixels from (2 py0 - 2, 2 px0 - 2)
            a.SHH = th + 1; a.SHW = tw + 1; a.s_plane = a.SHH * a.SHW; a.ISL = 1;
            a.sdymin = -2; a.sdxmin = -2;
            a.mg_spl = (unsigned)((0x100000000ull + (unsigned)a.s_plane - 1) / (unsigned)a.s_plane);
        }
        a.s_hpi = (a.s_deint ? 4 : 1) * a.SHH * a.SHW;
        a.s_halo = imgs * a.s_hpi;
        a.zdymin = zymin; a.zdxmin = zxmin;
        a.ZHH = (th - 1) * d->ZS + (zymax - zymin) + 1; a.ZHW = (tw - 1) * d->ZS + (zxmax - zxmin) + 1;
        a.z_halo = imgs * a.ZHH * a.ZHW;
        a.mg_shpi = (unsigned)((0x100000000ull + (unsigned)a.s_hpi - 1) / (unsigned)a.s_hpi);
        a.mg_shw = (unsigned)((0x100000000ull + (unsigned)a.SHW - 1) / (unsigned)a.SHW);
        a.mg_zhpi = (unsigned)((0x100000000ull + (unsigned)(a.ZHH * a.ZHW) - 1) / (unsigned)(a.ZHH * a.ZHW));
        a.mg_zhw = (unsigned)((0x100000000ull + (unsigned)a.ZHW - 1) / (unsigned)a.ZHW);
        a.zsame = 1;
        for (int t = 0; t < d->ntaps; ++t) {
            a.s_off[t] = (d->tap_dy[t] - ymin) * a.SHW + (d->tap_dx[t] - xmin);
            if (a.s_deint) {
                const int pa = d->tap_dy[t] & 1, pb = d->tap_dx[t] & 1;
                a.s_off[t] = (pa * 2 + pb) * a.s_plane + ((d->tap_dy[t] - pa) / 2 + 1) * a.SHW + ((d->tap_dx[t] - pb) / 2 + 1);
            }
            a.z_off[t] = (d->tap_zy[t] - zymin) * a.ZHW + (d->tap_zx[t] - zxmin);
            a.tap_w[t] = d->tap_w[t];
            if (a.z_off[t] != a.z_off[0]) a.zsame = 0;
        }
        return 0;
    };
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool plain = !scale0 && d->slope0 == 1.0f && (d->C1 == 0 || (!scale1 && d->slope1 == 1.0f));
    CU_CHECK_ARG(d->ntaps == 9 || d->ntaps == 4 || d->ntaps == 1, "cu_conv_wgrad: ntaps must be 9, 4 or 1 (got %d)", d->ntaps);
    // raw source 0 (scale0 / shift0 = the statistics planes of the producing layer) normalised + activated in LDS by the
    // LDS-DMA kernel's XF instances: one source, stride 1, the planes adjacent, bf16, large maps (one image per tile)
    const bool xf_ok = d->dtype == CU_BF16 && scale0 && shift0 == scale0 + (size_t)d->N * d->C0 && d->C1 == 0 && d->IS == 1 &&
                       d->ZS == 1 && (d->ntaps == 9 || d->ntaps == 1) && (long)d->PH * d->PW >= 256 && d->PW >= 16 &&
                       ((d->C0 == 32 && d->CO == 32) || (d->C0 == 64 && d->CO == 64 && d->ntaps == 9)) &&
                       !cu_env_set("CU_WGRAD_NOXF");
    CU_CHECK_ARG(plain || xf_ok || d->ntaps == 9, "cu_conv_wgrad: fused-activation sources are only built for 3x3 taps");

    // ---- bf16 production path: LDS-DMA staging, two LDS images (see igemm_wgrad_dma_kernel)
    const size_t b0 = (size_t)d->N * d->SH * d->SW * d->C0 * 2, b1 = (size_t)d->N * d->SH * d->SW * d->C1 * 2;
    const size_t bz = (size_t)d->N * d->ZH * d->ZW * d->ZC * 2;
    const size_t lim = 0x7fff0000ull;
    if (d->dtype == CU_BF16 && (plain || xf_ok) && b0 < lim && b1 < lim && bz < lim && !cu_env_set("CU_WGRAD_NODMA")) {
        const int spp = wc ? 8 : 4, zpp = wn ? 8 : 4;
        int dma_nw = cu_env_int("CU_WGRAD_NW", 8);
        CU_CHECK_ARG(dma_nw == 4 || dma_nw == 8, "CU_WGRAD_NW must be 4 or 8");
        static const int pc_env = cu_env_int("CU_WGRAD_PC", 1);      // 0 off, 2 every shape
        // producer/consumer split (see the kernel); the 32-column x 64-channel tile of the 256^2 decoder layer is faster
        // with eight computing waves (measured 294 vs 320 us)
        const bool pc = pc_env && dma_nw == 8 && d->ntaps == 9 && (pc_env == 2 || !(!wn && wc));
        {   // every wave needs at least one k-step of the smallest tile this shape may end up with
            const int nwb = (wn ? 2 : 1) * (wc ? 2 : 1);
            const int min_nk = ((d->IS > 1 || d->ZS > 1) ? 64 : 256) / 16 / 4;      // the tile loop may halve BM twice
            if (!pc && dma_nw / nwb > min_nk) dma_nw = 4;
        }
        // ---- stride-2 3x3 layers: parity-deinterleaved source planes, dense k-loop (see WgKArgs::s_deint).  The source has
        //      4x the pixels of Z, so the block is wide in n (Z is cheap) and 32 channels narrow: 128 (n) x 32 (c) x 9 taps
        //      = 9 accumulators per computing wave, producer / consumer waves as on the stride-1 layers.
        bool std3 = d->ntaps == 9 && d->IS == 2 && d->ZS == 1 && d->C1 == 0 && d->PW >= 8 && d->CO > 32 &&
                    !cu_env_set("CU_WGRAD_NODEINT");
        for (int t = 0; std3 && t < 9; ++t)
            std3 = d->tap_dy[t] >= -1 && d->tap_dy[t] <= 1 && d->tap_dx[t] >= -1 && d->tap_dx[t] <= 1 && d->tap_zy[t] == 0 &&
                   d->tap_zx[t] == 0;
        if (std3) {
            a.s_deint = 1;
            const int nb = d->CO > 64 ? 4 : 2;
            for (int BM = 128;; BM >>= 1) {
                CU_CHECK_ARG(BM >= 32, "cu_conv_wgrad: patches do not fit in LDS");
                const int rc = geometry(BM);
                if (rc) return rc;
                a.s_iters = cdiv(a.s_halo * 4, 256);
                a.z_iters = cdiv(a.z_halo * 4 * nb, 256);
                if ((size_t)(a.s_iters + a.z_iters) * 4096 <= (size_t)DMA_IMG_BYTES) break;
            }
            a.src0_bytes = (unsigned)b0; a.src1_bytes = (unsigned)b1; a.z_bytes = (unsigned)bz;
            const int n = a.s_iters + a.z_iters;
            static const int dpe = cu_env_int("CU_WGRAD_DPCE", 6), dpf = cu_env_int("CU_WGRAD_DPCF", 100);   // (6 early rounds: 12.133 -> 12.106 ms per step, profiles/r04_knob_sweep2.txt; the stride-1 form keeps 4)
            a.pc_early = dpe < n ? dpe : n;
            a.pc_items = a.pc_early + ((n - a.pc_early) * dpf + 99) / 100;
            if (a.pc_items > n) a.pc_items = n;
            const bool rowk = a.twl >= 4;
            if (nb == 4) return rowk ? launch_dma<4, 1, 9, 8, true, true>(a, st) : launch_dma<4, 1, 9, 8, true, false>(a, st);
            return rowk ? launch_dma<2, 1, 9, 8, true, true>(a, st) : launch_dma<2, 1, 9, 8, true, false>(a, st);
        }
        // ---- thin layers (32-row dW blocks over >= 2^20 loop pixels: 256^2 x 32 -> 32, 32 + 32 -> 32): two half-size workgroups
        //      of four waves per CU (see the kernel's IMGB).  CU_WGRAD_TWO = 0 / 1 in the tuning build
        if (d->ntaps == 9 && d->IS == 1 && d->ZS == 1 && !wn && plain && px_total >= (1l << 20) &&
            cu_env_int("CU_WGRAD_TWO", WGRAD_TWO_DEFAULT)) {
            constexpr int HALF = 39 * 1024;
            for (int BM = 256;; BM >>= 1) {
                CU_CHECK_ARG(BM >= 16, "cu_conv_wgrad: patches do not fit in LDS");
                const int rc = geometry(BM);
                if (rc) return rc;
                a.s_iters = cdiv(a.s_halo * spp, 256);
                a.z_iters = cdiv(a.z_halo * zpp, 256);
                if ((size_t)(a.s_iters + a.z_iters) * 4096 <= (size_t)HALF) break;
            }
            a.src0_bytes = (unsigned)b0; a.src1_bytes = (unsigned)b1; a.z_bytes = (unsigned)bz;
            a.pc_early = a.pc_items = a.s_iters + a.z_iters;
            if (wc) return launch_dma<1, 2, 9, 4, false, false, false, HALF>(a, st);
            return launch_dma<1, 1, 9, 4, false, false, false, HALF>(a, st);
        }
        const int issue_w = pc ? 4 : dma_nw;
        for (int BM = (d->IS > 1 || d->ZS > 1) ? 64 : 256;; BM >>= 1) {
            CU_CHECK_ARG(BM >= 16, "cu_conv_wgrad: patches do not fit in LDS");
            const int rc = geometry(BM);
            if (rc) return rc;
            a.s_iters = cdiv(a.s_halo * spp, 64 * issue_w);
            a.z_iters = cdiv(a.z_halo * zpp, 64 * issue_w);
            if ((size_t)(a.s_iters + a.z_iters) * 1024 * issue_w <= (size_t)DMA_IMG_BYTES) break;
        }
        CU_CHECK_ARG(d->ntaps == 4 || a.zsame, "cu_conv_wgrad: per-tap Z shifts are only built for 4 taps");
        a.src0_bytes = (unsigned)b0; a.src1_bytes = (unsigned)b1; a.z_bytes = (unsigned)bz;
        a.xstat_bytes = (unsigned)((size_t)2 * d->N * d->C0 * 4);
        {
            // round 3 (profiles/r03_wgrad_pc_sweep.txt; in the step, alternating runs: 13.67 / 13.71 ms against 13.83 / 13.85
            // at 0 / 70 %): the computing waves issue 4 rounds BEFORE their k-loop, the producer waves all the others
            static const int pcf = cu_env_int("CU_WGRAD_PCF", 100);     // percent of the remaining rounds issued by the producers
            const int n = a.s_iters + a.z_iters;
            static const int pce = cu_env_int("CU_WGRAD_PCE", 4);
            a.pc_early = pce < n ? pce : n;
            a.pc_items = a.pc_early + ((n - a.pc_early) * pcf + 99) / 100;
            if (a.pc_items > n) a.pc_items = n;
        }
#define CU_WDN(NBv, CBv, NWv)                                                   \
    do {                                                                        \
        if (d->ntaps == 9) return launch_dma<NBv, CBv, 9, NWv>(a, st);          \
        if (d->ntaps == 4) return launch_dma<NBv, CBv, 4, NWv>(a, st);          \
        return launch_dma<NBv, CBv, 1, NWv>(a, st);                             \
    } while (0)
#define CU_WD(NBv, CBv)                                                         \
    do {                                                                        \
        if (pc && a.twl >= 4) return launch_dma<NBv, CBv, 9, 8, true, true>(a, st); \
        if (pc) return launch_dma<NBv, CBv, 9, 8, true>(a, st);                 \
        if (dma_nw == 8) CU_WDN(NBv, CBv, 8);                                   \
        CU_WDN(NBv, CBv, 4);                                                    \
    } while (0)
        // 128 (n) x 64 (c) block per workgroup, eight computing waves (half the Z re-fetch of the 64 x 64
        // block; 128-pixel tiles so that two images still fit): tuning knob CU_WGRAD_N128
        if (wn && wc && d->ntaps == 9 && d->IS == 1 && d->ZS == 1 && d->CO >= 128 && cu_env_int("CU_WGRAD_N128", 0)) {
            for (int BM = 128;; BM >>= 1) {
                CU_CHECK_ARG(BM >= 32, "cu_conv_wgrad: patches do not fit in LDS");
                const int rc = geometry(BM);
                if (rc) return rc;
                a.s_iters = cdiv(a.s_halo * 8, 512);
                a.z_iters = cdiv(a.z_halo * 16, 512);
                if ((size_t)(a.s_iters + a.z_iters) * 8192 <= (size_t)DMA_IMG_BYTES) break;
            }
            if (a.twl >= 4) return launch_dma<4, 2, 9, 8, false, true>(a, st);
            return launch_dma<4, 2, 9, 8, false, false>(a, st);
        }
        if (!plain) {       // XF instances (xf_ok): 32 -> 32 and 64 -> 64 channels, one image per tile
            CU_CHECK_ARG(a.iml == 0 && a.twl >= 4, "cu_conv_wgrad: the normalise-on-load source needs one image per tile");
            if (d->ntaps == 9) {
                CU_CHECK_ARG(pc, "cu_conv_wgrad: normalise-on-load is built for the producer / consumer form");
                if (wn && wc) return launch_dma<2, 2, 9, 8, true, true, true>(a, st);
                return launch_dma<1, 1, 9, 8, true, true, true>(a, st);
            }
            return launch_dma<1, 1, 1, 4, false, false, true>(a, st);
        }
        if (wn && wc) CU_WD(2, 2);
        if (wn) CU_WD(2, 1);
        if (wc) CU_WD(1, 2);
        CU_WD(1, 1);
#undef CU_WD
#undef CU_WDN
    }

    int BM = (d->IS > 1 || d->ZS > 1) ? 64 : 512;
    size_t lds_cap = 80 * 1024;
    for (;; BM >>= 1) {
        if (BM < 128 && lds_cap < 150 * 1024 && d->IS == 1 && d->ZS == 1) { BM = 128; lds_cap = 150 * 1024; }
        CU_CHECK_ARG(BM >= 16 && BM >= kpix, "cu_conv_wgrad: patches do not fit in LDS");
        const int rc = geometry(BM);
        if (rc) return rc;
        const int spp = (wc ? 64 : 32) / piece_elems, zpp = (wn ? 64 : 32) / piece_elems;
        if (d->IS > 1 || d->ZS > 1) lds_cap = 150 * 1024;
        if ((size_t)a.s_halo * rows_b + (size_t)a.z_halo * rowz_b <= lds_cap && a.s_halo * spp <= 256 * ns_big &&
            a.z_halo * zpp <= 256 * nz_big)
            break;
    }
    const int spp_f = (wc ? 64 : 32) / piece_elems, zpp_f = (wn ? 64 : 32) / piece_elems;
    const bool small = a.s_halo * spp_f <= 256 * ns_small && a.z_halo * zpp_f <= 256 * nz_small;

    CU_CHECK_ARG(d->ntaps == 4 || a.zsame, "cu_conv_wgrad: per-tap Z shifts are only built for 4 taps");
#define CU_WT(T, NBv, CBv, NSv, NZv)                                                   \
    do {                                                                               \
        if (!plain) return launch_k<T, NBv, CBv, NSv, NZv, false, 9>(a, st);           \
        if (d->ntaps == 9) return launch_k<T, NBv, CBv, NSv, NZv, true, 9>(a, st);     \
        if (d->ntaps == 4) return launch_k<T, NBv, CBv, NSv, NZv, true, 4>(a, st);     \
        return launch_k<T, NBv, CBv, NSv, NZv, true, 1>(a, st);                        \
    } while (0)
#define CU_W(T, NBv, CBv, NSs, NZs, NSb, NZb)                                          \
    do {                                                                               \
        if (small) CU_WT(T, NBv, CBv, NSs, NZs);                                       \
        CU_WT(T, NBv, CBv, NSb, NZb);                                                  \
    } while (0)
    if (d->dtype == CU_BF16) {
        if (wn && wc) CU_W(bf16_t, 2, 2, 7, 4, 11, 8);
        if (wn) CU_W(bf16_t, 2, 1, 7, 4, 11, 8);
        if (wc) CU_W(bf16_t, 1, 2, 7, 4, 11, 8);
        CU_W(bf16_t, 1, 1, 7, 4, 11, 8);
    } else {
        if (wn && wc) CU_W(float, 2, 2, 13, 8, 21, 16);
        if (wn) CU_W(float, 2, 1, 13, 8, 21, 16);
        if (wc) CU_W(float, 1, 2, 13, 8, 21, 16);
        CU_W(float, 1, 1, 13, 8, 21, 16);
    }
#undef CU_WT
#undef CU_W
}

int cu_gemm_tn_try(const cu_wgrad_desc* d, const void* src0, const void* z, float* parts, size_t parts_floats, int* nparts,
                   void* stream);

extern "C" int cu_conv_wgrad(const cu_wgrad_desc* d, const void* src0, const float* scale0, const float* shift0,
                             const void* src1, const float* scale1, const float* shift1, const void* z, float* dw,
                             void* stream) {
    return wgrad_impl(d, src0, scale0, shift0, src1, scale1, shift1, z, dw, 0, nullptr, nullptr, stream);
}

extern "C" int cu_conv_wgrad_parts(const cu_wgrad_desc* d, const void* src0, const float* scale0, const float* shift0,
                                   const void* src1, const float* scale1, const float* shift1, const void* z, float* parts,
                                   size_t parts_floats, int* nparts, int* layout, void* stream) {
    CU_CHECK_ARG(nparts != nullptr && layout != nullptr && parts != nullptr, "cu_conv_wgrad_parts: null pointer");
    *nparts = 0;
    *layout = 0;
    // 2x2 stride-2 transposed convolution from one plain bf16 source: the pixel-major GEMM of gemm_tn.hip (plain slabs)
    if (d && src0 && z && !scale0 && !cu_env_set("CU_WGRAD_NOGEMMTN")) {
        const int rc = cu_gemm_tn_try(d, src0, z, parts, parts_floats, nparts, stream);
        if (rc < 0) return rc;
        if (rc > 0) { *layout = CU_PARTS_PLAIN; return 0; }
    }
    return wgrad_impl(d, src0, scale0, shift0, src1, scale1, shift1, z, parts, parts_floats, nparts, layout, stream);
}
