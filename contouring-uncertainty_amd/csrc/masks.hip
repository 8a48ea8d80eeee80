// Sampled contour -> filled mask -> entropy map, one workgroup per contour.
//
// Replaces, for batches of contours that are already on the GPU (the MC samplers' output):
//   contour_spline / reconstruction   reference contour_uncertainty/utils/contour.py:9-40
//       scipy splprep(k=3, s=0) + splev at 1000 parameters -> round -> set pixels (upper clip only: negative indices wrap
//       like numpy's), closing skimage.draw.line from the last to the first landmark, scipy.ndimage.binary_fill_holes;
//   USContourToMask (LV-only branch)  reference contour_uncertainty/data/camus/utils.py:31-45   (landmarks rounded first);
//   UncertaintyTask.sample_entropy    reference contour_uncertainty/task/uncertainty.py:107-133 (cu_mask_entropy).
// The interpolating spline is FITPACK's (parcur, s = 0): chord-length parameters, knots u[0] x4, u[2..m-3], u[m-1] x4,
// collocation solve (banded, f64), de Boor evaluation in f64.  Duplicate consecutive landmarks make scipy raise and the
// reference fall back to the raw landmarks; so does this kernel.  binary_fill_holes = the complement of a 4-connected
// flood fill of the background from outside the image: bit-parallel here (a row of <= 256 pixels is four 64-bit words,
// a whole run is filled by one carry chain), alternating row fills with column fills on the transposed bitmap until
// nothing changes.
#include "common.h"

namespace {

constexpr int MT = 256;          // threads = max rows = max columns
constexpr int MAXK = 32;
typedef unsigned long long u64;

struct Row { u64 w[4]; };

__device__ __forceinline__ Row row_and(const Row& a, const Row& b) { return Row{{a.w[0] & b.w[0], a.w[1] & b.w[1], a.w[2] & b.w[2], a.w[3] & b.w[3]}}; }
__device__ __forceinline__ Row row_or(const Row& a, const Row& b) { return Row{{a.w[0] | b.w[0], a.w[1] | b.w[1], a.w[2] | b.w[2], a.w[3] | b.w[3]}}; }
__device__ __forceinline__ bool row_ne(const Row& a, const Row& b) { return ((a.w[0] ^ b.w[0]) | (a.w[1] ^ b.w[1]) | (a.w[2] ^ b.w[2]) | (a.w[3] ^ b.w[3])) != 0; }

// seeds r (subset of m) spread towards higher bit positions through the runs of m: one carry chain
__device__ __forceinline__ Row fill_up(const Row& r, const Row& m) {
    Row o;
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u64 s1 = m.w[i] + r.w[i];
        const u64 c1 = s1 < m.w[i] ? 1ull : 0ull;
        const u64 s2 = s1 + carry;
        const u64 c2 = s2 < s1 ? 1ull : 0ull;
        o.w[i] = ((s2 ^ m.w[i]) & m.w[i]) | r.w[i];
        carry = c1 | c2;
    }
    return o;
}
__device__ __forceinline__ Row row_rev(const Row& a) {
    return Row{{__brevll(a.w[3]), __brevll(a.w[2]), __brevll(a.w[1]), __brevll(a.w[0])}};
}
__device__ __forceinline__ Row fill_both(const Row& r, const Row& m) {
    const Row up = fill_up(r, m);
    const Row dn = row_rev(fill_up(row_rev(up), row_rev(m)));
    return row_or(up, dn);
}

// transpose of a 256 x 256 bitmap stored as rows of 8 x u32 in LDS (bit x of row y = pixel (y, x)): dst row x, bit y = src row
// y, bit x.  All MT = 256 threads; the caller puts a barrier before (src complete) and after (dst complete).  A 32-lane half
// wave holds a 32 x 32 bit block, lane i = row i, and transposes it with five exchange steps (rows i and i ^ j swap their
// off-diagonal j x j blocks: __shfl_xor stays inside the half wave); 64 blocks = 8 rounds of the 8 half waves.  (The first
// version had thread c gather column c bit by bit: 256 LDS reads + extracts per thread and transposition, three to seven
// transpositions per mask -- most of the kernel's time.)
template <int J, unsigned L> __device__ __forceinline__ unsigned tstep(unsigned x, int i) {
    const unsigned other = (unsigned)__shfl_xor((int)x, J, 64);
    return (i & J) ? ((x & ~L) | ((other & ~L) >> J)) : ((x & L) | ((other & L) << J));
}
__device__ __forceinline__ void transpose_bitmap(const unsigned* src, unsigned* dst, int tid) {
    const int half = tid >> 5, i = tid & 31;           // 8 half waves
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int b = t * 8 + half, by = b >> 3, bx = b & 7;
        unsigned x = src[(32 * by + i) * 8 + bx];
        x = tstep<16, 0x0000ffffu>(x, i);
        x = tstep<8, 0x00ff00ffu>(x, i);
        x = tstep<4, 0x0f0f0f0fu>(x, i);
        x = tstep<2, 0x33333333u>(x, i);
        x = tstep<1, 0x55555555u>(x, i);
        dst[(32 * bx + i) * 8 + by] = x;
    }
}
__device__ __forceinline__ double bcast(double v, int lane) {      // lane is wave-uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ void store_row(unsigned* bm, int y, const Row& r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bm[y * 8 + 2 * i] = (unsigned)r.w[i];
        bm[y * 8 + 2 * i + 1] = (unsigned)(r.w[i] >> 32);
    }
}
__device__ __forceinline__ Row load_row(const unsigned* bm, int y) {
    Row r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.w[i] = (u64)bm[y * 8 + 2 * i] | ((u64)bm[y * 8 + 2 * i + 1] << 32);
    return r;
}
__device__ __forceinline__ Row low_bits(int n) {      // bits [0, n)
    Row r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = n - 64 * i;
        r.w[i] = k >= 64 ? ~0ull : (k <= 0 ? 0ull : ((1ull << k) - 1ull));
    }
    return r;
}

__global__ __launch_bounds__(MT) void contour_mask_kernel(int M, int K, int H, int W, const float* __restrict__ contours,
                                                          int round_landmarks, int mode, unsigned* __restrict__ packed,
                                                          unsigned char* __restrict__ bytes, int dbg,
                                                          int* __restrict__ area_out = nullptr,
                                                          float* __restrict__ length_out = nullptr) {
    // A workgroup takes CPW = 4 consecutive contours (round 4): the spline set-up is a serial chain on ONE wave (chord lengths,
    // a 21-step banded elimination and back-substitution with f64 divisions: ~10 us of latency), so the four waves run the four
    // set-ups side by side; drawing, filling and output then take the contours one after the other with all 256 threads.
    constexpr int CPW = MT / 64;
    __shared__ unsigned bmA[MT * 8];        // the drawn curve, then the reached background (rows)
    __shared__ unsigned bmB[MT * 8];        // transposed bitmaps
    __shared__ double px_s[CPW][MAXK], py_s[CPW][MAXK], u_s[CPW][MAXK], t_s[CPW][MAXK + 4], cx_s[CPW][MAXK], cy_s[CPW][MAXK];
    __shared__ int fallback_s[CPW];
    const int tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * CPW;
    const int sc = tid >> 6, sl = tid & 63;                  // set-up: wave sc works on contour base + sc, lane sl
    const bool shas = base + sc < (size_t)M;
    double *px = px_s[sc], *py = py_s[sc], *u = u_s[sc], *t = t_s[sc], *cx = cx_s[sc], *cy = cy_s[sc];
    if (shas && sl < K) {
        const float* pts = contours + (base + sc) * K * 2;
        float x = pts[2 * sl], y = pts[2 * sl + 1];
        if (round_landmarks) { x = rintf(x); y = rintf(y); }      // numpy round: half to even
        px[sl] = x; py[sl] = y;
    }
    __syncthreads();

    // ---- FITPACK interpolation set-up, one wave per contour, lane i = landmark i (K <= 32)
    if (shas) {
        const int i = sl;
        double d = 1.0;
        if (i >= 1 && i < K) d = sqrt((px[i] - px[i - 1]) * (px[i] - px[i - 1]) + (py[i] - py[i - 1]) * (py[i] - py[i - 1]));
        const int bad = (K < 4 || __ballot(!(d > 0.0)) != 0) ? 1 : 0;
        double tot = 0.0, ui = 0.0;
        for (int j = 1; j < K; ++j) {          // sequential sum, like FITPACK's running u[i] = u[i-1] + dist
            tot += bcast(d, j);
            if (i == j) ui = tot;
        }
        ui = i == K - 1 ? 1.0 : ui / tot;
        if (i == 0) fallback_s[sc] = bad | (dbg & 1);
        if (!bad) {
            if (i < K) u[i] = ui;
            if (i < 4) { t[i] = 0.0; t[K + i] = 1.0; }
            if (i >= 2 && i <= K - 3) t[i + 2] = ui;
        }
    }
    __syncthreads();
    // non-zero cubic B-splines at x: span l (t[l] <= x < t[l+1], clamped to [3, K-1]), values N[0..3] of B_{l-3..l}
    auto basis_t = [&](const double* t, double x, double (&N)[4]) -> int {
        int l = 3;
        while (l < K - 1 && x >= t[l + 1]) ++l;
        N[0] = 1.0; N[1] = N[2] = N[3] = 0.0;
#pragma unroll
        for (int deg = 1; deg < 4; ++deg) {
            double saved = 0.0;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if (r < deg) {
                    const double tr = t[l + r + 1], tl = t[l + 1 - deg + r];
                    const double term = N[r] / (tr - tl);
                    N[r] = saved + (tr - x) * term;
                    saved = (x - tl) * term;
                }
            }
            N[deg] = saved;
        }
        return l - 3;
    };
    auto basis = [&](double x, double (&N)[4]) -> int { return basis_t(t, x, N); };
    // collocation solve, one wave per contour: lane i holds row i of the band, B[d] = A(i, i - 3 + d).  Gaussian elimination without
    // pivoting (the collocation matrix of an interpolating spline is totally positive); the pivot row travels by readlane.
    if (shas && !fallback_s[sc]) {
        const int i = sl;
        double B[7] = {0, 0, 0, 0, 0, 0, 0}, rx = 0.0, ry = 0.0;
        if (i < K) {
            double N[4];
            const int j = basis(u[i], N);
#pragma unroll
            for (int dd = 0; dd < 7; ++dd) {
                const int e = dd + i - 3 - j;
                B[dd] = e == 0 ? N[0] : e == 1 ? N[1] : e == 2 ? N[2] : e == 3 ? N[3] : 0.0;
            }
            rx = px[i]; ry = py[i];
        } else B[3] = 1.0;
        for (int c = 0; c < K - 1; ++c) {
            const double p3 = bcast(B[3], c), p4 = bcast(B[4], c), p5 = bcast(B[5], c), p6 = bcast(B[6], c);
            const double qx = bcast(rx, c), qy = bcast(ry, c);
            const int k = i - c;
            if (i < K) {
                if (k == 1) { const double f = B[2] / p3; B[3] -= f * p4; B[4] -= f * p5; B[5] -= f * p6; rx -= f * qx; ry -= f * qy; }
                else if (k == 2) { const double f = B[1] / p3; B[2] -= f * p4; B[3] -= f * p5; B[4] -= f * p6; rx -= f * qx; ry -= f * qy; }
                else if (k == 3) { const double f = B[0] / p3; B[1] -= f * p4; B[2] -= f * p5; B[3] -= f * p6; rx -= f * qx; ry -= f * qy; }
            }
        }
        double x1 = 0, x2 = 0, x3 = 0, y1 = 0, y2 = 0, y3 = 0, solx = 0, soly = 0;
        for (int c = K - 1; c >= 0; --c) {
            const double vx = (rx - B[4] * x1 - B[5] * x2 - B[6] * x3) / B[3];
            const double vy = (ry - B[4] * y1 - B[5] * y2 - B[6] * y3) / B[3];
            if (i == c) { solx = vx; soly = vy; }
            x3 = x2; x2 = x1; x1 = bcast(vx, c);
            y3 = y2; y2 = y1; y1 = bcast(vy, c);
        }
        if (i < K) { cx[i] = solx; cy[i] = soly; }
    }
    __syncthreads();

    for (int c = 0; c < CPW; ++c) {        // ---- the contours of the workgroup, one after the other, all threads
    const size_t m = base + c;
    if (m >= (size_t)M) break;
    double *px = px_s[c], *py = py_s[c], *t = t_s[c], *cx = cx_s[c], *cy = cy_s[c];
    const int fallback = fallback_s[c];
    auto basis = [&](double x, double (&N)[4]) -> int { return basis_t(t, x, N); };
    __syncthreads();                       // the previous contour's bitmaps have been read
    for (int i = tid; i < MT * 8; i += MT) bmA[i] = 0u;
    __syncthreads();

    // ---- clinical measure (cu_contour_measures): length of the open polyline through contour_spline's 1001 points
    // (reference utils/clinical.py:31-72 `perimeter` / `global_longitudinal_strain`; utils/contour.py:9-25: n = 1001, the raw
    // landmarks when splprep raises), summed in f64; per-thread partial sums combined in thread order (deterministic)
    if (length_out) {
        __shared__ double lpart[MT];
        double acc = 0.0;
        auto point = [&](int q, double& sx, double& sy) {
            const double x = (double)q * (1.0 / 1000.0);
            double N[4];
            const int j = basis(q == 1000 ? 1.0 : x, N);
            sx = N[0] * cx[j] + N[1] * cx[j + 1] + N[2] * cx[j + 2] + N[3] * cx[j + 3];
            sy = N[0] * cy[j] + N[1] * cy[j + 1] + N[2] * cy[j + 2] + N[3] * cy[j + 3];
        };
        if (!fallback) {
            for (int q = tid; q < 1000; q += MT) {
                double ax, ay, bx, by;
                point(q, ax, ay);
                point(q + 1, bx, by);
                acc += sqrt((bx - ax) * (bx - ax) + (by - ay) * (by - ay));
            }
        } else if (tid < K - 1) {
            acc = sqrt((px[tid + 1] - px[tid]) * (px[tid + 1] - px[tid]) + (py[tid + 1] - py[tid]) * (py[tid + 1] - py[tid]));
        }
        lpart[tid] = acc;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int i = 0; i < MT; ++i) tot += lpart[i];
            length_out[m] = (float)tot;
        }
        if (!packed && !bytes && !area_out) continue;
    }

    // ---- draw: 1000 spline points (or the raw landmarks), upper clip, negative indices wrap once like numpy
    auto plot = [&](int ix, int iy, bool wrap) {
        if (ix > W - 1) ix = W - 1;
        if (iy > H - 1) iy = H - 1;
        if (wrap) { if (ix < 0) ix += W; if (iy < 0) iy += H; }
        else { if (ix < 0) ix = 0; if (iy < 0) iy = 0; }
        if (ix >= 0 && iy >= 0) atomicOr(&bmA[iy * 8 + (ix >> 5)], 1u << (ix & 31));
    };
    // mode 0: `reconstruction` (1000 points, upper clip + numpy's negative wrap, rounded closing line, fill);
    // mode 1 / 2: the curve of `uncertainty_map` (reference utils/umap.py:24-31: 1001 points, clip to [0, size-1] on both
    // sides, closing line between the TRUNCATED end landmarks (mode 2: no closing line), no fill)
    const int npts = mode == 0 ? 1000 : 1001;
    const bool wrap_pts = mode == 0;
    if (!fallback) {
        for (int q = tid; q < npts; q += MT) {
            const double x = (double)q * (1.0 / (double)(npts - 1));      // np.linspace(0, 1, npts)
            double N[4];
            const int j = basis(q == npts - 1 ? 1.0 : x, N);
            const double sx = N[0] * cx[j] + N[1] * cx[j + 1] + N[2] * cx[j + 2] + N[3] * cx[j + 3];
            const double sy = N[0] * cy[j] + N[1] * cy[j + 1] + N[2] * cy[j + 2] + N[3] * cy[j + 3];
            plot((int)rint(sx), (int)rint(sy), wrap_pts);
        }
    } else if (tid < K) {
        plot((int)rint(px[tid]), (int)rint(py[tid]), wrap_pts);
    }
    {   // closing edge, skimage.draw.line(last, first) with clip(min=0, max): Bresenham in closed form, one step per thread
        int r0 = (int)rint(py[K - 1]), c0 = (int)rint(px[K - 1]), r1 = (int)rint(py[0]), c1 = (int)rint(px[0]);
        if (mode != 0) { r0 = (int)py[K - 1]; c0 = (int)px[K - 1]; r1 = (int)py[0]; c1 = (int)px[0]; }      // astype(int)
        int dr = abs(r1 - r0), dc = abs(c1 - c0);
        int sr = r1 >= r0 ? 1 : -1, sc = c1 >= c0 ? 1 : -1;
        const bool steep = dr > dc;
        if (steep) { int a; a = r0; r0 = c0; c0 = a; a = dr; dr = dc; dc = a; a = sr; sr = sc; sc = a; }
        for (int i = tid; i <= dc && mode != 2; i += MT) {
            const int n = dc > 0 ? (int)((2ll * dr * i + dc) / (2ll * dc)) : 0;     // minor-axis steps before major step i
            const int r = r0 + sr * n, c = c0 + sc * i;
            if (steep) plot(r, c, false); else plot(c, r, false);
        }
    }
    __syncthreads();

    if (mode != 0) {          // curve only
        if (tid < H && packed) {
            unsigned* o = packed + (m * H + tid) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = bmA[tid * 8 + i];
        }
        if (bytes) {
            unsigned char* o = bytes + m * (size_t)H * W;
            for (int i = tid; i < H * W; i += MT) {
                const int y = i / W, x = i - y * W;
                o[i] = (bmA[y * 8 + (x >> 5)] >> (x & 31)) & 1u;
            }
        }
        continue;
    }
    // ---- binary_fill_holes: flood the background from outside the image (4-connectivity), bit-parallel
    const Row wmask = low_bits(W), hmask = low_bits(H);
    Row freeR{{0, 0, 0, 0}}, reach{{0, 0, 0, 0}};
    if (tid < H) {
        const Row curve = load_row(bmA, tid);
        freeR = Row{{~curve.w[0] & wmask.w[0], ~curve.w[1] & wmask.w[1], ~curve.w[2] & wmask.w[2], ~curve.w[3] & wmask.w[3]}};
        if (tid == 0 || tid == H - 1) reach = freeR;
        else {
            const Row lo = low_bits(W - 1);
            Row edge{{(wmask.w[0] ^ lo.w[0]) | 1ull, wmask.w[1] ^ lo.w[1], wmask.w[2] ^ lo.w[2], wmask.w[3] ^ lo.w[3]}};
            reach = row_and(freeR, edge);
        }
        store_row(bmA, tid, freeR);
    }
    __syncthreads();
    transpose_bitmap(bmA, bmB, tid);
    __syncthreads();
    Row freeC{{0, 0, 0, 0}};
    if (tid < W) freeC = row_and(load_row(bmB, tid), hmask);
    __syncthreads();
    // rows >= H of bmA and rows >= W of bmB stay zero from here on (threads >= H / >= W store empty rows)
    for (int iter = 0; iter < ((dbg & 2) ? 0 : 64); ++iter) {
        const Row before = reach;
        reach = fill_both(reach, freeR);
        // the column pass leaves the map closed under vertical moves: a row pass that adds nothing means convergence
        const int grew = __syncthreads_or(row_ne(before, reach) ? 1 : 0);
        if (iter > 0 && !grew) break;
        store_row(bmA, tid, reach);
        __syncthreads();
        transpose_bitmap(bmA, bmB, tid);
        __syncthreads();
        Row col = fill_both(row_and(load_row(bmB, tid), freeC), freeC);
        store_row(bmB, tid, col);            // (a thread's own row: the only one it read)
        __syncthreads();
        transpose_bitmap(bmB, bmA, tid);
        __syncthreads();
        reach = row_and(load_row(bmA, tid), freeR);
    }
    // ---- mask = everything the flood did not reach
    if (tid < H) {
        const Row mk{{~reach.w[0] & wmask.w[0], ~reach.w[1] & wmask.w[1], ~reach.w[2] & wmask.w[2], ~reach.w[3] & wmask.w[3]}};
        store_row(bmA, tid, mk);
        if (packed) {
            unsigned* o = packed + (m * H + tid) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = bmA[tid * 8 + i];
        }
    }
    if (area_out) {          // pixels of the filled mask = EchoMeasure.structure_area of the LV label (reference utils/clinical.py:88-89)
        const Row mk{{~reach.w[0] & wmask.w[0], ~reach.w[1] & wmask.w[1], ~reach.w[2] & wmask.w[2], ~reach.w[3] & wmask.w[3]}};
        int cnt = tid < H ? (__popcll(mk.w[0]) + __popcll(mk.w[1]) + __popcll(mk.w[2]) + __popcll(mk.w[3])) : 0;
        for (int off = 32; off; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
        __syncthreads();
        int* cpart = reinterpret_cast<int*>(bmB);
        if ((tid & 63) == 0) cpart[tid >> 6] = cnt;
        __syncthreads();
        if (tid == 0) area_out[m] = cpart[0] + cpart[1] + cpart[2] + cpart[3];
    }
    if (bytes) {
        __syncthreads();
        unsigned char* o = bytes + m * (size_t)H * W;
        if ((W & 15) == 0) {        // 16 pixels -> one 16-byte store
            const int per_row = W >> 4;
            for (int i = tid; i < H * per_row; i += MT) {
                const int y = i / per_row, xg = i - y * per_row;
                const unsigned half = (bmA[y * 8 + (xg >> 1)] >> ((xg & 1) * 16)) & 0xffffu;
                unsigned v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned n = (half >> (4 * q)) & 0xfu;
                    v[q] = (n & 1u) | ((n & 2u) << 7) | ((n & 4u) << 14) | ((n & 8u) << 21);
                }
                *reinterpret_cast<uint4*>(o + (size_t)y * W + xg * 16) = make_uint4(v[0], v[1], v[2], v[3]);
            }
        } else {
            for (int i = tid; i < H * W; i += MT) {
                const int y = i / W, x = i - y * W;
                o[i] = (bmA[y * 8 + (x >> 5)] >> (x & 31)) & 1u;
            }
        }
    }
    }      // contours of the workgroup
}

// mean over S samples of the packed masks of every frame and its binary entropy (base 2; 0 where the mean is 0 or 1).
// A workgroup owns 32 consecutive words (4 rows) of one frame; thread = (word, one of 8 slices of the samples) keeps the
// 32 pixel counters of its word in registers; the slices are combined through LDS.
constexpr int ENT_SL = 8;
__global__ __launch_bounds__(256) void mask_entropy_kernel(int F, int S, int H, int W, const unsigned* __restrict__ packed,
                                                           float* __restrict__ mean, float* __restrict__ entropy) {
    __shared__ int part[ENT_SL][32][33];
    const int tid = threadIdx.x, wl = tid & 31, sl = tid >> 5;
    const int groups = (H * 8 + 31) / 32;
    const int f = blockIdx.x / groups, g = blockIdx.x - f * groups;
    const int word = g * 32 + wl;                 // word index inside one mask, < H * 8
    int cnt[32];
#pragma unroll
    for (int b = 0; b < 32; ++b) cnt[b] = 0;
    if (word < H * 8) {
        const unsigned* p = packed + (size_t)f * S * H * 8 + word;
        const int per = (S + ENT_SL - 1) / ENT_SL;
        const int s1 = min(S, (sl + 1) * per);
        for (int s = sl * per; s < s1; ++s) {
            const unsigned w = p[(size_t)s * H * 8];
#pragma unroll
            for (int b = 0; b < 32; ++b) cnt[b] += (int)__builtin_amdgcn_ubfe(w, (unsigned)b, 1u);
        }
    }
#pragma unroll
    for (int b = 0; b < 32; ++b) part[sl][wl][b] = cnt[b];
    __syncthreads();
    for (int j = tid; j < 1024; j += 256) {
        const int wj = j >> 5, b = j & 31;
        const int wd = g * 32 + wj;
        const int y = wd >> 3, x = (wd & 7) * 32 + b;
        if (y >= H || x >= W) continue;
        int c = 0;
#pragma unroll
        for (int q = 0; q < ENT_SL; ++q) c += part[q][wj][b];
        const float pr = (float)c / (float)S;
        const size_t o = ((size_t)f * H + y) * W + x;
        if (mean) mean[o] = pr;
        if (entropy) entropy[o] = (c == 0 || c == S) ? 0.f : -(pr * log2f(pr) + (1.f - pr) * log2f(1.f - pr));
    }
}

// weighted mean over S packed masks per frame (weights sum to 1) and its natural-log binary entropy: the reduction of
// skew_umap (reference utils/skew_umap.py:74-79: np.average(rec, weights) then scipy.stats.entropy of [m, 1 - m])
__global__ __launch_bounds__(256) void mask_weighted_entropy_kernel(int F, int S, int H, int W, const unsigned* __restrict__ packed,
                                                                    const float* __restrict__ weights, float* __restrict__ mean,
                                                                    float* __restrict__ entropy) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)F * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H), f = (int)(i / ((size_t)W * H));
    const unsigned* p = packed + ((size_t)f * S * H + y) * 8 + (x >> 5);
    float m = 0.f;
    for (int s = 0; s < S; ++s)
        if ((p[(size_t)s * H * 8] >> (x & 31)) & 1u) m += weights[s];
    m = fminf(fmaxf(m, 0.f), 1.f);
    if (mean) mean[i] = m;
    if (entropy) entropy[i] = -((m > 0.f ? m * logf(m) : 0.f) + (m < 1.f ? (1.f - m) * logf(1.f - m) : 0.f));
}

// out[pixel] = values[s] of the LAST mask s that covers the pixel, 0 if none: the sequential overwrites of
// `uncertainty_map` (reference utils/umap.py:22-31)
__global__ __launch_bounds__(256) void mask_last_value_kernel(int S, int H, int W, const unsigned* __restrict__ packed,
                                                              const float* __restrict__ values, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const int x = i % W, y = i / W;
    const unsigned* p = packed + (size_t)y * 8 + (x >> 5);
    float v = 0.f;
    for (int s = S - 1; s >= 0; --s)
        if ((p[(size_t)s * H * 8] >> (x & 31)) & 1u) { v = values[s]; break; }
    out[i] = v;
}

}  // namespace

extern "C" int cu_contour_masks(int M, int K, int H, int W, const float* contours, int round_landmarks, int mode,
                                unsigned* packed, unsigned char* bytes, void* stream) {
    CU_CHECK_ARG(mode >= 0 && mode <= 2, "cu_contour_masks: bad mode %d", mode);
    CU_CHECK_ARG(M > 0 && K >= 2 && K <= MAXK && H > 0 && H <= MT && W > 0 && W <= MT, "cu_contour_masks: bad sizes M=%d K=%d H=%d W=%d", M, K, H, W);
    CU_CHECK_ARG(contours && (packed || bytes), "cu_contour_masks: null pointer");
    static const int dbg = cu_env_int("CU_MASKS_DBG", 0);      // timing aid (tools/masks_bench.py)
    hipLaunchKernelGGL(contour_mask_kernel, dim3((M + 3) / 4), dim3(MT), 0, reinterpret_cast<hipStream_t>(stream), M, K, H, W, contours,
                       round_landmarks, mode, packed, bytes, dbg);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_contour_measures(int M, int K, int H, int W, const float* contours, int round_landmarks, int* area,
                                   float* length, void* stream) {
    CU_CHECK_ARG(M > 0 && K >= 2 && K <= MAXK && H > 0 && H <= MT && W > 0 && W <= MT, "cu_contour_measures: bad sizes M=%d K=%d H=%d W=%d", M, K, H, W);
    CU_CHECK_ARG(contours && (area || length), "cu_contour_measures: null pointer");
    hipLaunchKernelGGL(contour_mask_kernel, dim3((M + 3) / 4), dim3(MT), 0, reinterpret_cast<hipStream_t>(stream), M, K, H, W, contours,
                       round_landmarks, 0, (unsigned*)nullptr, (unsigned char*)nullptr, 0, area, length);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_mask_entropy(int F, int S, int H, int W, const unsigned* packed, float* mean, float* entropy, void* stream) {
    CU_CHECK_ARG(F > 0 && S > 0 && H > 0 && H <= MT && W > 0 && W <= MT && packed && (mean || entropy), "cu_mask_entropy: bad argument");
    hipLaunchKernelGGL(mask_entropy_kernel, dim3((unsigned)(F * ((H * 8 + 31) / 32))), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), F, S, H, W, packed, mean, entropy);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_mask_weighted_entropy(int F, int S, int H, int W, const unsigned* packed, const float* weights, float* mean,
                                        float* entropy, void* stream) {
    CU_CHECK_ARG(F > 0 && S > 0 && H > 0 && H <= MT && W > 0 && W <= MT && packed && weights && (mean || entropy),
                 "cu_mask_weighted_entropy: bad argument");
    const size_t total = (size_t)F * H * W;
    hipLaunchKernelGGL(mask_weighted_entropy_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), F, S, H, W, packed, weights, mean, entropy);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_mask_last_value(int S, int H, int W, const unsigned* packed, const float* values, float* out, void* stream) {
    CU_CHECK_ARG(S > 0 && H > 0 && H <= MT && W > 0 && W <= MT && packed && values && out, "cu_mask_last_value: bad argument");
    hipLaunchKernelGGL(mask_last_value_kernel, dim3((unsigned)((H * W + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), S, H, W, packed, values, out);
    CU_LAUNCH_CHECK();
    return 0;
}
