/*
 * contour_hip.h -- C ABI of libcontour_hip.so: the MI355X (gfx950) kernels of the DSNT contour-regression hot path.
 *
 * Boundary rules:
 *   - plain pointers and sizes only; every buffer is caller-owned device memory (PyTorch tensors' data_ptr()).
 *   - no allocation, no ownership transfer, no global mutable state (except the thread-local last-error string).
 *   - every call is asynchronous on the hipStream_t passed in (void* here so that C callers need no HIP headers).
 *   - return 0 on success, a negative errno-style code otherwise; cu_last_error() gives the text.
 *
 * Each entry point names the reference (ThierryJudge/contouring-uncertainty) call site it replaces.
 * dtype: 0 = f32 (parity mode, exact-f32 MFMA), 1 = bf16 (production mode, bf16 MFMA, f32 accumulate).
 * Activations are NHWC ("pixel-major, channel-contiguous"); logits are NCHW f32.
 */
#ifndef CONTOUR_HIP_H
#define CONTOUR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CU_F32 0
#define CU_BF16 1
#define CU_MAX_TAPS 9

const char* cu_last_error(void);
int cu_version(void);
const char* cu_arch(void); /* "gfx950" */

/* ------------------------------------------------------------------------------------------------------------------
 * Generic implicit-GEMM "gather convolution":   D[p, n] = bias[n] + sum_t sum_c  act(S[p*IS + off_t, c]) * W[t][n][c]
 * One kernel serves nn.Conv2d 3x3 s1/s2 forward (reference layers.py:55-80,192), its input gradient (autograd of the
 * same), nn.ConvTranspose2d k2 s2 forward/input-gradient (layers.py:83-109,415-417), the 1x1 output conv
 * (layers.py:456-463) and the ConfidenceNet convs (unet2.py:21-27).
 *   - up to two channel-concatenated sources (torch.cat((out, skip), 1), layers.py:436, fused into the load),
 *   - per-(image,channel) affine + LeakyReLU applied while loading (InstanceNorm2d + LeakyReLU of the producing layer,
 *     layers.py:193-194,203-204, fused so the normalised tensor is never materialised),
 *   - up to two channel-split destinations (gradient of the concat), optional accumulate.
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct {
    int dtype;          /* CU_F32 | CU_BF16: element type of sources, weights and NHWC destinations */
    int N;              /* images */
    int PH, PW;         /* loop ("p") grid per image */
    int SH, SW;         /* source image size */
    int C0, C1;         /* channels of source 0 and source 1 (0 = absent); multiples of 32 (bf16) / 16 (f32) */
    int IS;             /* source pixel = p*IS + (dy,dx) */
    int OH, OW;         /* destination image size */
    int OS, OY0, OX0;   /* destination pixel = p*OS + (OY0,OX0) */
    int CO;             /* GEMM columns (rows of W per tap); W is [wtaps][CO][C0+C1] */
    int D0;             /* columns [0,D0) -> dst0, [D0,CO) -> dst1 */
    int DC0, DC1;       /* channel counts (pixel strides) of dst0 / dst1 */
    int ntaps;
    int tap_dy[CU_MAX_TAPS], tap_dx[CU_MAX_TAPS], tap_w[CU_MAX_TAPS];
    float slope0, slope1;   /* LeakyReLU slope applied to source 0/1 after the affine; 1.0f = none */
    int accum0, accum1;     /* 1: dst += result */
    int out_nchw_f32;       /* 1: dst0 is NCHW f32 with DC0 planes (the logits); columns >= DC0 are dropped */
    int par_co;             /* > 0: the CO columns are 4 groups of par_co, group g -> channels [0,par_co) of destination
                               pixel p*OS + (g >> 1, g & 1): all four output parities of a 2x2 stride-2 transposed conv
                               (W viewed as [1][4*par_co][C]) in ONE pass over the source.  Needs D0 == CO, OY0 = OX0 = 0 */
    int par_taps;           /* with par_co: 1 = the weight tap depends on (gather tap t, parity group g):
                               W is [wtaps][par_co][C] and group g of tap t uses weight tap par_tap_w[t*4 + g], or
                               contributes nothing when that is < 0.  This is the input gradient of a stride-2 3x3 conv
                               (4 gather taps = the 2x2 neighbourhood of dz, 9 of the 16 (t, g) pairs in use) in one
                               pass: dz is read once and whole destination rows are written, instead of one launch per
                               output parity.  ntaps <= 4. */
    int par_tap_w[16];
} cu_conv_desc;

int cu_conv_gemm(const cu_conv_desc* d,
                 const void* src0, const float* scale0, const float* shift0,   /* scale/shift: [N][C0] or NULL */
                 const void* src1, const float* scale1, const float* shift1,
                 const void* w, const float* bias /* [CO] or NULL */,
                 void* dst0, void* dst1, void* stream);
/* The same, allowed to split the channel reduction over several workgroups per tile (tiny feature maps: <= 4x4 at
 * batch 64, where one workgroup per tile walks all channel chunks one L2 round trip at a time).  ws: caller-owned f32
 * scratch of ws_floats elements (contents irrelevant on entry, garbage on exit): every split stores its partial tile
 * in a slice of its own and a finish pass sums the slices in a fixed order (deterministic).  Used when the launch
 * covers its destinations completely and >= 2 slices of N*OH*OW*(DC0+DC1) floats fit. */
int cu_conv_gemm_ws(const cu_conv_desc* d,
                    const void* src0, const float* scale0, const float* shift0,
                    const void* src1, const float* scale1, const float* shift1,
                    const void* w, const float* bias, void* dst0, void* dst1, float* ws, size_t ws_floats, void* stream);

/* The same, and -- when the launch goes to the streaming kernel of the thin, large 3x3 layers (tconv.hip) with one
 * destination -- the InstanceNorm statistics of the output gathered in its epilogue (layers.py:192-194: conv -> norm):
 * stat_sums [N][CO][2] f32, ZERO on entry, += {sum, sum of squares} of (output - bias) in f32 before the rounding to the
 * storage type.  *stats_done (host) = 1 when they were gathered, 0 when the launch took another kernel (stat_sums is
 * then untouched and the caller runs the statistics pass).  Consumer: cu_instnorm_fwd_given. */
int cu_conv_gemm_stats(const cu_conv_desc* d,
                       const void* src0, const float* scale0, const float* shift0,
                       const void* src1, const float* scale1, const float* shift1,
                       const void* w, const float* bias, void* dst0, void* dst1, float* ws, size_t ws_floats,
                       float* stat_sums, int* stats_done, void* stream);

/* General form of the epilogue extension.  mode 1 = as cu_conv_gemm_stats.  mode 2 = this launch is the INPUT GRADIENT
 * g = dL/da of a layer a = LeakyReLU(scale * z + shift) (InstanceNorm + LeakyReLU, layers.py:193-194): sums [N][CO][2]
 * (zero on entry) += {sum of gl, sum of gl * zhat} with gl = g * LeakyReLU'(scale z + shift), zhat = (z - mean) * rstd --
 * the reduction pass of that layer's norm backward (z [N][H][W][CO] of the launch's dtype, stats = the four planes of
 * cu_instnorm_stats).  Consumer: cu_instnorm_bwd_given.  *done as in cu_conv_gemm_stats. */
/* modes 3 and 4 (tiny feature maps, <= 64 pixels per image, taken when the launch splits its channel reduction over
 * workgroups -- cu_conv_gemm_ws): the split-K finish pass, which holds whole images per workgroup, carries the layer's
 * WHOLE InstanceNorm + LeakyReLU instead of a separate norm launch.
 *   mode 3 (forward): dst0 = z; stats (written: the four planes of cu_instnorm_stats, of the z as stored);
 *                     act_out = LeakyReLU(z*scale + shift) (dst0's layout and type).  gamma / beta / eps / slope given.
 *   mode 4 (input gradient): dst0 = dL/dz of the layer whose activation this launch differentiates (NOT dL/da): its norm
 *                     backward from z, stats (read), gamma, slope; dgamma / dbeta [C] += (atomics, may be NULL).
 *   mode 5            = mode 4 with dgamma / dbeta as PER-IMAGE planes [N][C], written (not accumulated) by the one workgroup
 *                     that owns an (image, channel): no same-address atomics; cu_norm_param_grads_batch adds the images.
 * *done = 1 when the launch took this form, else nothing of it happened (dst0 holds the plain result). */
typedef struct {
    int mode;
    float* sums;
    const void* z;
    float* stats;
    float slope;
    const float* gamma;
    const float* beta;
    float eps;
    void* act_out;
    float* dgamma;
    float* dbeta;
} cu_conv_epilogue;
int cu_conv_gemm_ex(const cu_conv_desc* d,
                    const void* src0, const float* scale0, const float* shift0,
                    const void* src1, const float* scale1, const float* shift1,
                    const void* w, const float* bias, void* dst0, void* dst1, float* ws, size_t ws_floats,
                    const cu_conv_epilogue* ep, int* done, void* stream);

/* Weight gradient of the same gather convolution:  dW[t][n][c] += sum_p Z[p*ZS + zoff_t, n] * act(S[p*IS + off_t, c])
 * (autograd of nn.Conv2d / nn.ConvTranspose2d weights).  dW is f32, accumulated with atomics. */
typedef struct {
    int dtype;
    int N;
    int PH, PW;
    int SH, SW, C0, C1, IS;        /* activation source(s), as above */
    int ZH, ZW, ZC, ZS;            /* Z = gradient tensor [N][ZH][ZW][ZC] */
    int CO;                        /* columns used (<= ZC) */
    int ntaps;
    int tap_dy[CU_MAX_TAPS], tap_dx[CU_MAX_TAPS];     /* source offsets */
    int tap_zy[CU_MAX_TAPS], tap_zx[CU_MAX_TAPS];     /* Z offsets */
    int tap_w[CU_MAX_TAPS];
    float slope0, slope1;
    int splits;                    /* pixel-range splits (grid.y); 0 = auto; 1 = one workgroup per dW block whose partial
                                    * sums are added in a fixed order: bit-identical results run to run (slow) */
} cu_wgrad_desc;

int cu_conv_wgrad(const cu_wgrad_desc* d,
                  const void* src0, const float* scale0, const float* shift0,
                  const void* src1, const float* scale1, const float* shift1,
                  const void* z, float* dw /* [wtaps][CO][C0+C1] f32 */, void* stream);

/* The same without atomics ("partial tiles"): every adder of a dW block -- one per pixel split, times the k-parts of the
 * register-staged kernel -- STORES its partial tile into a slab of its own in `parts` (f32 scratch of parts_floats
 * elements, contents irrelevant on entry), in the accumulators' own layout (16-byte stores, 1 KiB per instruction).
 * *nparts = slabs written, *layout = their block shape: both host ints, written before the call returns, to be handed to
 * cu_grad_unprep_parts, which adds the slabs in slab order -- the gradient is bit-identical run to run at ANY split count.
 * d->splits is capped so that the slabs plus one plain [wtaps][CO][C0+C1] tile fit parts_floats.
 * (Round 3, the production form: the atomics of cu_conv_wgrad run at the chip's ~1.3 TB/s float-atomic rate, 37.7 MB per
 * launch = 29 us whatever the layer.) */
int cu_conv_wgrad_parts(const cu_wgrad_desc* d,
                        const void* src0, const float* scale0, const float* shift0,
                        const void* src1, const float* scale1, const float* shift1,
                        const void* z, float* parts, size_t parts_floats, int* nparts, int* layout, void* stream);

/* First layer, Cin = 1 (input_block.conv1.conv, unet2.py:113-119): direct 3x3 conv of the f32 image. */
int cu_conv_c1_fwd(int dtype, int N, int H, int W, int CO, const float* img /* [N][H][W] */,
                   const float* w /* [9][CO] f32 */, const float* bias, void* dst /* NHWC */, void* stream);
/* The first layer's conv -> InstanceNorm -> LeakyReLU (layers.py:192-194) without a statistics pass over z and without a
 * separate apply pass.  z is linear in the image, so its per-(image, channel) mean and variance follow from 9 + 45 moments
 * of the image (sums of shifted pixels and of their pairwise products, zero outside the image like the conv's padding):
 * (1) one pass over the IMAGE gathers them (per-workgroup partials in ws, summed in a fixed order: deterministic),
 * (2) the four stats planes of cu_instnorm_stats, (3) one pass writes z AND a = LeakyReLU(z*scale + shift).
 * z is bit-identical to cu_conv_c1_fwd's; the statistics are those of z before its rounding to the storage type (as in
 * cu_conv_gemm_stats).  ws: f32 scratch of cu_conv_c1_norm_ws_floats(N, H, W) elements, contents irrelevant on entry. */
size_t cu_conv_c1_norm_ws_floats(int N, int H, int W);
int cu_conv_c1_fwd_norm(int dtype, int N, int H, int W, int CO, const float* img, const float* w, const float* bias,
                        const float* gamma, const float* beta, float eps, float slope, float* ws, float* stats,
                        void* z, void* a, void* stream);
/* z == NULL: only the activation is written (the backward below recomputes z from the image, 9 FMAs per channel, with
 * the very same expressions -- so both sides decide every LeakyReLU branch on the same value). */
/* The first layer's WHOLE backward from g = dL/da (NHWC, read only) without z and without dz: the layer has no input
 * gradient, so its dz feeds nothing but its own 9 x CO weight gradient.  Pass 1: sums [N][CO][2] += (sum gl, sum gl zhat)
 * with gl = g LeakyReLU'(z scale + shift) (sums must arrive zeroed); pass 2: dz = gamma rstd (gl - S1/HW - zhat S2/HW)
 * (layers.py:192-205 backward) formed in registers, dw [9][CO] += sum_p x[p + t] dz[p], dgamma / dbeta [CO] += (may be NULL).
 * stats: the four planes of cu_conv_c1_fwd_norm.  2 x (g + image) of HBM reads instead of read g, z + write dz + read dz. */
int cu_conv_c1_bwd(int dtype, int N, int H, int W, int CO, const float* img, const float* w, const float* bias,
                   const float* stats, const float* gamma, float slope, const void* g, float* sums, float* dw,
                   float* dgamma, float* dbeta, void* stream);
int cu_conv_c1_wgrad(int dtype, int N, int H, int W, int CO, const float* img, const void* dz /* NHWC */,
                     float* dw /* [9][CO] f32, += */, void* stream);
/* The same with the workgroups' partial sums stored in ws (>= (rows chunks x N) x 9 x CO floats; 2^20 covers every shape
 * of the path) and added to dw in workgroup order by a finish pass: no atomics, bit-identical run to run. */
int cu_conv_c1_wgrad_det(int dtype, int N, int H, int W, int CO, const float* img, const void* dz, float* dw, float* ws,
                         size_t ws_floats, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * InstanceNorm2d(affine) + LeakyReLU (layers.py:193-194) in its fused form.
 * ---------------------------------------------------------------------------------------------------------------- */
/* statistics of z [N][HW][C] -> stats[4][N][C] = planes {mean, rstd, scale = gamma*rstd, shift = beta - mean*gamma*rstd}
 * (planes 2 and 3 are what cu_conv_gemm / cu_conv_wgrad take as scale / shift).
 * ws: f32 workspace [N][C][2], zero-filled by the call. */
int cu_instnorm_stats(int dtype, int N, int HW, int C, const void* z, const float* gamma, const float* beta,
                      float eps, float* stats, float* ws, void* stream);
/* out = LeakyReLU(z*scale + shift) [N][HW][C] (dtype): the activated tensor the next layers' kernels stage as a plain
 * operand (measured: recomputing it in every consumer's load made thin layers VALU-bound). */
int cu_instnorm_apply(int dtype, int N, int HW, int C, const void* z, const float* stats, float slope, void* out,
                      void* stream);
/* backward: g = dL/d(activated output) [N][HW][C] is overwritten in place with dL/dz.
 * dgamma/dbeta/dbias: f32 [C], accumulated (+=), any may be NULL.
 * ws: f32 workspace [N][C][2], zero-filled by the call. */
int cu_instnorm_lrelu_bwd(int dtype, int N, int HW, int C, void* g, const void* z, const float* stats,
                          const float* gamma, float slope, float* dgamma, float* dbeta, float* dbias,
                          float* ws, void* stream);
/* Production entry points of the engine for the same two operations, in ONE call per direction:
 *   forward : stats as above AND out = LeakyReLU(z*scale + shift)
 *   backward: as cu_instnorm_lrelu_bwd without dbias (identically zero behind an InstanceNorm)
 * mode 0 picks per shape between
 *   1  "resident": one launch, every tensor read once -- workgroups keep their pixel chunk in registers while the
 *      per-(image, channel) sums are combined with atomics behind an image-local arrival counter (norm.hip), and
 *   2  the two-pass kernels (statistics / reduction pass, then apply pass).
 * ws: f32 workspace of cu_instnorm_resident_ws_floats(N, C) elements, zero-filled by the call as needed -- unless
 * CU_NORM_WS_CLEAN is or-ed into mode: the caller then hands over a workspace that is already zero (one memset for
 * all layers of a step instead of one launch per layer); the call leaves it dirty.  After the
 * stream has drained, ((unsigned*)ws)[1] != 0 after a mode-1 call reports that its bounded arrival wait gave up; a
 * workgroup whose wait gave up also POISONS its totals with NaN (statistics, outputs and the step's loss turn NaN): the
 * training path does not have to poll the flag to notice. */
#define CU_NORM_WS_CLEAN 16
/* + CU_NORM_DETERMINISTIC: two-pass kernels with one workgroup per image (fixed summation order, every sum has a single
 * adder) and, in the backward, dgamma / dbeta summed over the images by a finish pass: bit-identical results run to run. */
#define CU_NORM_DETERMINISTIC 32
/* + CU_NORM_PARAM_PARTS (cu_instnorm_bwd_fused on maps of <= 1024 pixels, two-pass form): dgamma / dbeta are per-image planes
 * [N][C] that the launch WRITES (one workgroup owns an (image, channel): no atomics, fixed result); the caller adds the
 * images with cu_norm_param_grads_batch -- every layer of a backward pass in one launch. */
#define CU_NORM_PARAM_PARTS 64
/* + CU_NORM_SMALL_RES (cu_instnorm_bwd_fused, bf16, maps of 16 x 16 and 32 x 32): the register-resident form -- g and z read once,
 * a workgroup of 512 threads owns whole (image, 32- or 64-channel) planes.  Faster as a lone launch (20 / 33 us against 30 / 48 at
 * batch 64), slower inside a step whose weight gradients run beside it (profiles/r04_norm_small_res_in_step.txt): opt-in. */
#define CU_NORM_SMALL_RES 128
typedef struct {
    const float* dgamma_parts;     /* [N][C] or NULL */
    const float* dbeta_parts;      /* [N][C] or NULL */
    uint64_t dgamma_off;           /* BYTE offset of dgamma [C] from grad_base (+= sum over the images, in image order) */
    uint64_t dbeta_off;
    int N, C;
} cu_pgrad_item;
/* items: DEVICE array of n_items entries; max_c >= every item's C.  grad_base: the buffer the offsets refer to (a training
 * step writes every gradient into ONE flat buffer that is new each step: the table then stays valid from step to step and
 * only this pointer changes); NULL = the offsets are absolute addresses.  A destination may appear in ONE item only: the
 * work items of a launch read-modify-write their destinations without atomics. */
int cu_norm_param_grads_batch(const cu_pgrad_item* items, int n_items, int max_c, float* grad_base, void* stream);
/* out == NULL (cu_instnorm_fwd_fused with the two-pass kernels, cu_instnorm_fwd_given): statistics only -- the layer's
 * consumers then normalise + activate while they stage the raw tensor (scale / shift of cu_conv_gemm / cu_conv_wgrad). */
size_t cu_instnorm_resident_ws_floats(int N, int C);
int cu_instnorm_fwd_fused(int dtype, int N, int HW, int C, const void* z, const float* gamma, const float* beta,
                          float eps, float slope, float* stats, void* out, float* ws, int mode, void* stream);
/* forward from statistics gathered elsewhere (cu_conv_gemm_stats): sums [N][C][2] = {sum, sum of squares} of
 * (z - shift[c]) (shift [C] or NULL) -> stats (planes as above) and out = LeakyReLU(z*scale + shift): the statistics
 * pass over z is gone, the apply pass remains. */
int cu_instnorm_fwd_given(int dtype, int N, int HW, int C, const void* z, const float* gamma, const float* beta,
                          float eps, float slope, const float* sums, const float* shift, float* stats, void* out,
                          void* stream);
/* backward from the two sums a convolution epilogue gathered (cu_conv_gemm_ex mode 2): the apply pass only. */
int cu_instnorm_bwd_given(int dtype, int N, int HW, int C, void* g, const void* z, const float* stats,
                          const float* gamma, float slope, float* dgamma, float* dbeta, const float* sums, void* stream);
int cu_instnorm_bwd_fused(int dtype, int N, int HW, int C, void* g, const void* z, const float* stats,
                          const float* gamma, float slope, float* dgamma, float* dbeta, float* ws, int mode,
                          void* stream);
/* x[n][p][c] *= mask[n][c] in place: nn.Dropout2d(p) between conv and norm (layers.py:154-164,199-202), forward and
 * backward (mask entries are 0 or 1/(1-p)) */
int cu_channel_scale(int dtype, int N, int HW, int C, void* x, const float* mask, void* stream);
/* nn.MaxPool2d(2, 2) of the `vital` U-Net (vital/vital/models/segmentation/unet.py:137-139) on NHWC tensors:
 * x [N][2 OH][2 OW][C] -> y [N][OH][OW][C], idx (same shape, bytes) = window position dy*2+dx of the maximum (first one
 * on ties, NaN propagates: ATen's rule); backward routes dy to that position and zeroes the other three.  C % 4 == 0. */
int cu_maxpool2_fwd(int dtype, int N, int OH, int OW, int C, const void* x, void* y, unsigned char* idx, void* stream);
int cu_maxpool2_bwd(int dtype, int N, int OH, int OW, int C, const void* dy, const unsigned char* idx, void* dx, void* stream);
/* plain activation backward for layers without norm (ConfidenceNet ReLU): g *= (z > 0 ? 1 : slope); dbias[c] += sum g */
int cu_act_bwd(int dtype, int N, int HW, int C, void* g, const void* z, float slope, float* dbias, void* stream);
/* The same by ONE workgroup over the whole batch (N * HW <= 2^20 pixels: the ConfidenceNet head): dbias has a fixed
 * summation order and a single adder -- deterministic mode. */
int cu_act_bwd_det(int dtype, int N, int HW, int C, void* g, const void* z, float slope, float* dbias, void* stream);
/* materialise act(z*scale+shift) as NCHW f32 (the bottleneck clone handed to the skew head, unet2.py:186) and back */
int cu_act_to_nchw_f32(int dtype, int N, int HW, int C, const void* z, const float* stats, float slope,
                       float* out, void* stream);
/* NCHW f32 [N][C][HW] -> NHWC (dtype) [N][HW][CP], channels >= C zero-filled (CP % 8 == 0) */
int cu_nchw_f32_to_nhwc(int dtype, int N, int HW, int C, int CP, const float* in, void* out, void* stream);
int cu_nhwc_to_nchw_f32(int dtype, int N, int HW, int C, const void* in, float* out, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * DSNT head: flat_softmax + dsnt + pixel rescale + get_cov_matrix
 * (dsnt/utils.py:7-47,71-77,95-105; dsnt_al.py:52-60; aleatoric.py:138-144)
 * ---------------------------------------------------------------------------------------------------------------- */
/* logits [N*K][H][W] f32 (H == W) -> mu [N*K][2] (pixel x,y), sigma [N*K][3] = {xx, yy, xy} (pixel^2),
 * aux [N*K][8] = {max, 1/sumexp, xbar, ybar, varx, vary, covar (normalised units), 0} kept for backward. */
int cu_dsnt_head_fwd(int NK, int H, int W, const float* logits, int use_covar, float* mu, float* sigma, float* aux,
                     void* stream);
/* gmu [N*K][2], gsigma [N*K][3] = dL/d(mu), dL/d{xx,yy,xy} (xy = the single covariance scalar feeding both
 * off-diagonal entries) -> dlogits [N*K][H][W] f32. */
int cu_dsnt_head_bwd(int NK, int H, int W, const float* logits, const float* aux, const float* gmu,
                     const float* gsigma, int use_covar, float* dlogits, void* stream);
/* The same gradient in the layout its consumers read (the Z operand of the 1x1 output convolution's weight / input
 * gradients, layers.py:456-463): dl [N][H][W][32] of dtype, channels K..31 zero; K <= 32.  Saves the NCHW f32 round trip. */
int cu_dsnt_head_bwd_nhwc(int dtype, int N, int K, int H, int W, const float* logits, const float* aux,
                          const float* gmu, const float* gsigma, int use_covar, void* dl, void* stream);

/* Fused head of the bf16 production path (head_fused.hip): the LAST ConvLayer's InstanceNorm + LeakyReLU (layers.py:192-205),
 * the 1x1 OutputBlock (layers.py:441-463) and the DSNT moments above in one pass over that layer's RAW output -- the
 * activation, the logits and dL/dlogits never exist in HBM.
 *   z [N][H][W][32] bf16 (raw conv output), stats [4][N][32] f32 (the planes of cu_instnorm_stats), slope;
 *   w_cls [32][32] bf16 = the 1x1 weight, class-major, rows K..31 zero (cu_weight_prep's forward copy with COP = 32);
 *   w_ch  [32][32] bf16 = the same weight channel-major (cu_weight_prep's input-gradient copy).
 * Square maps, W % 32 == 0, H % 16 == 0, K <= 32.
 * forward: ws = scratch of cu_head_fused_ws_floats(N, H, W) floats (per-tile partial moments); mu / sigma / aux as
 *   cu_dsnt_head_fwd (aux feeds either backward).
 * backward: g [N][H][W][32] bf16 = dL/d(activation of the last ConvLayer) (written); sums [N][32][2] += the two sums of that
 *   layer's InstanceNorm backward (consumer: cu_instnorm_bwd_given); parts = scratch of >= 1025 * 1024 floats that receives
 *   *nparts partial 1x1 weight gradients in the plain [32 classes][32 channels] layout (consumer: cu_grad_unprep_parts with
 *   layout CU_PARTS_PLAIN = 0xffff, T = 1, COP = 32).  gmu / gsigma as cu_dsnt_head_bwd. */
size_t cu_head_fused_ws_floats(int N, int H, int W);
int cu_head_fused_fwd(int N, int H, int W, int K, const void* z, const float* stats, float slope, const void* w_cls,
                      int use_covar, float* ws, size_t ws_floats, float* mu, float* sigma, float* aux, void* stream);
int cu_head_fused_bwd(int N, int H, int W, int K, const void* z, const float* stats, float slope, const void* w_cls,
                      const void* w_ch, const float* aux, const float* gmu, const float* gsigma, int use_covar, void* g,
                      float* sums, float* parts, size_t parts_floats, int* nparts, void* stream);

/* Gaussian NLL of dsnt_al.py:64-74 and skew-normal NLL of bivariateskewnormal.py:36-61 (closed-form 2x2 algebra,
 * Sigma^-1/2 = ((Sigma + sqrt(det) I)/sqrt(tr + 2 sqrt(det)))^-1 instead of distributions/utils.py:100-129's eig).
 * y [M][2]; alpha [M][2] or NULL (gauss).  logs[8] = {loss, distance_loss, term1, term2, term3, alpha_norm, 0, 0}
 * (means over M).  Gradients of `loss` (already divided by M): gmu [M][2], gsigma [M][3], galpha [M][2].
 * terms [M][4] (optional) = per-point {nll, term1, term2, term3} as BivariateSkewNormal.nll returns them. */
int cu_nll_fwd_bwd(int M, int skew, float w_mse, float w_log, const float* mu, const float* sigma, const float* y,
                   const float* alpha, float* logs, float* gmu, float* gsigma, float* galpha, float* terms,
                   void* stream);

/* ConfidenceNet Linear (unet2.py:28-29): x [N][IN] f32, w [OUT][IN], out [N][OUT]; and its backward. */
int cu_linear_fwd(int N, int IN, int OUT, const float* x, const float* w, const float* b, float* out, void* stream);
int cu_linear_bwd(int N, int IN, int OUT, const float* x, const float* w, const float* gout, float* gx, float* gw,
                  float* gb, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Parameters: f32 master weights live in one flat buffer; kernels read per-step bf16/f32 "operand copies".
 * ---------------------------------------------------------------------------------------------------------------- */
/* master weight in the reference's logical layout, element (t, co, ci) at co*s_co + ci*s_ci + t (Conv2d OIHW: s_co =
 * CI*T, s_ci = T; ConvTranspose2d IOHW: s_co = T, s_ci = CO*T) -> fwd operand [T][COP][CI] and dgrad operand
 * [T][CI][COP] (dtype), rows co >= CO zero-filled; either output may be NULL. */
int cu_weight_prep(int dtype, int T, int CO, int CI, int COP, long s_co, long s_ci, const float* master, void* w_fwd,
                   void* w_dgrad, void* stream);
/* kernel-layout gradient dWk [T][COP][CI] f32 (what cu_conv_wgrad accumulates) -> logical-layout gradient.
 * accumulate: bit 0 = add to grad instead of overwriting it; bit 1 = zero every dWk element read (the accumulator is left
 * clean for the next layer: one persistent workspace, no per-layer memset). */
int cu_grad_unprep(int T, int CO, int CI, int COP, long s_co, long s_ci, float* dwk, float* grad, int accumulate,
                   void* stream);

/* Sum of the nparts slabs cu_conv_wgrad_parts wrote into `parts` (same parts_floats, nparts and layout; T = its weight-tap
 * count, COP / CI = its CO and C0+C1, CO <= COP = rows of the logical gradient; `parts` is scratch: more than 16 slabs are first summed in groups in place, the plain
 * [T][CO][CI] sum is formed at its end) -> logical-layout gradient as cu_grad_unprep; accumulate bit 0 as there. */
int cu_grad_unprep_parts(int T, int CO, int CI, int COP, long s_co, long s_ci, float* parts, size_t parts_floats,
                         int nparts, int layout, float* grad, int accumulate, void* stream);

/* Batched form: one launch for every conv layer of the network.  `items` is a DEVICE array (blk0 ascending, blk0 of
 * item i = number of 256-thread blocks of items 0..i-1; an item has tiles_co * tiles_ci blocks of 32 x 32 x T weights,
 * T <= 9 and the taps innermost in the logical layout).
 * cu_weight_prep_batch: master -> w_fwd / w_dgrad of element type `dtype` (either may be NULL). */
typedef struct {
    const float* master;
    void* w_fwd;
    void* w_dgrad;
    int T, CO, CI, COP;
    long long s_co, s_ci;       /* element strides of the logical layout */
    int blk0, tiles_ci, tiles_co, pad;
} cu_prep_item;
int cu_weight_prep_batch(int dtype, int n_items, const cu_prep_item* items, int total_blocks, void* stream);

/* torch.optim.Adam(lr, betas, eps, weight_decay) semantics (vital/vital/config/task/optim/adam.yaml:1-4):
 * g += wd*p; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps). */
int cu_adam_step(size_t n, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int step, float grad_scale, void* stream);
/* The same update with the step count on the device (hipGraph-capturable: a replayed graph has no host side to count):
 * bias correction uses *steps_done + 1; cu_step_advance increments the counter once all parameter runs of the step have
 * been launched. */
int cu_adam_step_dev(size_t n, float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2, float eps,
                     float weight_decay, const int* steps_done, float grad_scale, void* stream);
int cu_step_advance(int* steps_done, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Monte-Carlo contour sampler, Gaussian posterior shape model
 * (sampler/posterior_shape_model/psm.py:73-93,199-440; posteriorshapemodel.py:9-81), one workgroup per frame.
 *   mu_pred [F][K][2] pixel (x,y); cov_pred [F][K][3] = {xx, yy, xy}; cov0 [2K][2K] = covariance of the PSM training
 *   shapes about their own mean xbar [2K] (transformed units); smean / sscale [2K] = the PSM scaler.
 *   init_pts (HOST array, n_init <= 8): anchor points drawn from their own predicted distribution.
 *   tables (DEVICE ints, n_levels rows of 2 + 48 + 32): {ng, nt, g_flat[48], t_pts[32]} = flat indices already known
 *   and points produced at that level; sigma2 / sample_level (HOST arrays): PSM slack and 1 = draw / 0 = fill with the
 *   conditional mean.  eps [F][S][K][2] standard-normal draws or NULL (then a counter-based generator keyed by seed).
 *   out [F][S][K][2].
 * ---------------------------------------------------------------------------------------------------------------- */
int cu_psm_sample_gauss(int F, int S, int K, const float* mu_pred, const float* cov_pred, const float* cov0,
                        const float* xbar, const float* smean, const float* sscale, int n_init, const int* init_pts,
                        int n_levels, const int* tables, const float* sigma2, const int* sample_level,
                        const float* eps, uint64_t seed, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Skew-normal and ED/ES sequence samplers (sampler/posterior_shape_model/psm_skew.py:45-158,162-503,
 * sequence_sampler.py:13-160, psm_skew_sequence.py:21-166).
 *
 * cu_psm_setup: per-frame record of the PSM algebra (PCA re-centred on mu_pred [F][P], P = 2K <= 96): for every level
 *   row of `tables` the gains C[t,g](C[g,g]+sigma2 I)^-1 and the scaled 2x2 conditional covariance of every target
 *   point.  rec [F][rec_stride] floats, rec_stride >= cu_psm_record_floats(n_levels, ng[], nt[]) (HOST arrays).
 *   Record layout: m[96] | covc[48][4] {xx,xy,yx,yy} | gains (level-major, rows 2*nt, cols ng).
 * cu_psm_sample_skew: one workgroup per (frame, sample).  Anchors: BivariateSkewNormal.rvs_fast (or, when
 *   use_initial_pdf, a draw from skew-pdf x prior on the grid).  Points with their bit set in skew_bits:
 *   `numerical_sampling` = inverse-CDF draw from skew-pdf(mu_pred, cov_pred, alpha) x N(mu_c, cov_c) [x prior] on the
 *   grid x grid lattice linspace(0,255,grid)^2 in torch.meshgrid(indexing='ij') order; other points: product-of-
 *   Gaussians merge + draw.  alpha_y_sign = -1 applies psm_skew.py:232.  prior_mu [F][S][K][2] / prior_cov [F][S][K][3]
 *   {xx,yy,xy} or NULL.  eps [F][S][K][3] standard normals or NULL; u [F][S][K] uniforms in [0,1) or NULL (then a
 *   counter-based generator keyed by seed).  A table without mass falls back to mu_c (psm_skew.py:135-154).
 * cu_psm_condition: conditional mean of a one-level record given sampled contours known [N][P] (sample i uses record
 *   i / per_rec): mu_c [N][nt][2], cov_c [R][nt][4]; with mu_p [R][P] / cov_p [R][P/2][3] also the product-of-Gaussians
 *   merge mu_f [N][nt][2], cov_f [R][nt][4] (sequence_sampler.py:83-91).
 * ---------------------------------------------------------------------------------------------------------------- */
int cu_psm_record_floats(int n_levels, const int* ng, const int* nt);
int cu_psm_setup(int F, int P, const float* mu_pred, const float* cov0, const float* xbar, const float* smean,
                 const float* sscale, int n_levels, const int* tables, const float* sigma2, float* rec, int rec_stride,
                 void* stream);
int cu_psm_sample_skew(int F, int S, int K, const float* mu_pred, const float* cov_pred, const float* alpha,
                       float alpha_y_sign, uint64_t skew_bits, const float* rec, int rec_stride, const float* smean,
                       const float* sscale, int n_init, const int* init_pts, int n_levels, const int* tables,
                       const int* sample_level, const float* prior_mu, const float* prior_cov, int use_initial_pdf,
                       int grid, const float* eps, const float* u, uint64_t seed, float* out, void* stream);
int cu_psm_condition(int N, int P, int per_rec, const float* rec, int rec_stride, const int* table, int nt,
                     const float* known, const float* smean, const float* sscale, const float* mu_p, const float* cov_p,
                     float* mu_c, float* cov_c, float* mu_f, float* cov_f, void* stream);

/* log-density of M bivariate normal (alpha == NULL) or skew-normal distributions at P points [P][2]
 * (distributions/bivariatenormal.py:15-36, bivariateskewnormal.py:19-49): out [M][P], or [P] when pairwise (M == P). */
int cu_logpdf_grid(int M, int P, int pairwise, const float* pts, const float* mu, const float* sigma,
                   const float* alpha, float* out, void* stream);
/* BivariateSkewNormal.rvs_fast (bivariateskewnormal.py:159-191): S draws of each of M distributions, out [M][S][2];
 * eps [M][S][3] standard-normal draws or NULL (counter-based generator keyed by seed). */
int cu_skew_rvs(int M, int S, const float* mu, const float* sigma, const float* alpha, const float* eps, uint64_t seed,
                float* out, void* stream);

/* ----------------------------------------------------------------------------------------------------------------
 * Sampled contours -> filled masks -> entropy map (csrc/masks.hip).
 * cu_contour_masks: `reconstruction` (reference contour_uncertainty/utils/contour.py:28-40: interpolating cubic spline at
 *   1000 parameters, round, upper clip, closing line last -> first landmark, binary_fill_holes) of M contours
 *   [M][K][2] (x, y) in pixels, K <= 32, H, W <= 256.  round_landmarks != 0 rounds the landmarks first, as
 *   USContourToMask does (reference data/camus/utils.py:31-45).  Contours with duplicate consecutive landmarks or
 *   K < 4 use the raw landmarks, like the reference's bare `except`.  mode 0 = the filled mask described above;
 *   mode 1 = only the curve as `uncertainty_map` draws it (reference contour_uncertainty/utils/umap.py:24-31: 1001 spline
 *   points clipped to the image on both sides + the line between the TRUNCATED end landmarks, no fill); mode 2 = mode 1
 *   without that line.  Outputs (either may be NULL): packed
 *   [M][H][8] uint32 (bit x%32 of word x/32 = pixel (y, x)), bytes [M][H][W] of 0/1.
 * cu_mask_entropy: UncertaintyTask.sample_entropy (reference task/uncertainty.py:107-133) over the S packed masks of
 *   each of F frames (packed [F][S][H][8]): mean [F][H][W] and/or its base-2 binary entropy [F][H][W] (0 where the
 *   mean is 0 or 1).
 * ---------------------------------------------------------------------------------------------------------------- */
int cu_contour_masks(int M, int K, int H, int W, const float* contours, int round_landmarks, int mode, uint32_t* packed,
                     uint8_t* bytes, void* stream);
/* Clinical measures of M contours in ONE launch (reference contour_uncertainty/utils/clinical.py:11-89):
 *   area [M] int32   = pixels of the filled mask cu_contour_masks(mode 0) draws = EchoMeasure.structure_area(mask, LV)
 *                      (`lv_area`, reference vital/vital/utils/image/measure.py:21-40), the operand of lv_FAC / compute_FAC;
 *   length [M] float = length of the open polyline through contour_spline(contour, n = 1001) (reference utils/contour.py:9-25),
 *                      the operand of `perimeter` / `global_longitudinal_strain` / `compute_gls`; the raw landmark polyline
 *                      when the spline fit fails (duplicate consecutive landmarks, K < 4), like the reference's `except`.
 * Either output may be NULL.  Same limits as cu_contour_masks (K <= 32, H, W <= 256). */
int cu_contour_measures(int M, int K, int H, int W, const float* contours, int round_landmarks, int32_t* area, float* length,
                        void* stream);
int cu_mask_entropy(int F, int S, int H, int W, const uint32_t* packed, float* mean, float* entropy, void* stream);
/* Weighted form for the skew-normal uncertainty map (reference contour_uncertainty/utils/skew_umap.py:74-79): mean =
 * sum_s weights[s] * mask_s (weights [S], normalised by the caller), entropy = natural-log binary entropy of the mean. */
int cu_mask_weighted_entropy(int F, int S, int H, int W, const uint32_t* packed, const float* weights, float* mean,
                             float* entropy, void* stream);
/* out [H][W] = values[s] of the last of the S packed masks that covers the pixel, 0 where none does: the sequential
 * overwrites of `uncertainty_map` (reference contour_uncertainty/utils/umap.py:22-31). */
int cu_mask_last_value(int S, int H, int W, const uint32_t* packed, const float* values, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * On-device training augmentation: the CAMUS data module's Compose([RandomRotation(3),
 * RandomBrightnessContrast(0.2, 0.2), RandomGamma((0.8, 1.2)), RandomTranslation(5, 5)]) (reference
 * contour_uncertainty/data/camus/datamodule.py:46-55, augmentations/{affine,brightnesscontrast,gamma}.py), which the
 * reference runs per item on CPU workers through torchvision.transforms.functional, for a whole batch in two launches.
 *   img / out [N][H][W] f32 in [0, 1] (distinct buffers); params [N][8] f32 = {angle (degrees), tx, ty (whole pixels),
 *   brightness factor, contrast factor, gamma, 0, 0} per image; mean_ws [N] f32 scratch.
 *   Order and semantics of the reference: rotate (nearest, zero fill, about the image centre) -> adjust_brightness ->
 *   adjust_contrast (blend with the mean of its input, clamp to [0, 1]) -> adjust_gamma -> translate (zero fill).
 * cu_augment_labels: the geometric half (rotate, translate; nearest) for the int64 label maps [N][H][W].
 * The key-point half (21 x 2 numbers per item) is host-side tensor arithmetic (contour_uncertainty/augmentations/affine.py).
 * ---------------------------------------------------------------------------------------------------------------- */
int cu_augment_image(int N, int H, int W, const float* img, const float* params, float* mean_ws, float* out, void* stream);
int cu_augment_labels(int N, int H, int W, const long long* labels, const float* params, long long* out, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange: one communicator per process (= per GPU), RCCL over xGMI.
 * The reference is single-device; these are the N-GPU form of the training path, used by cu_hip/comm.py when
 * CONTOUR_COMM=native (default: torch.distributed's "nccl" backend, which is the same RCCL).
 *   cu_comm_unique_id   rank 0 fills 128 bytes, the host side hands them to every rank (any side channel)
 *   cu_comm_init        collective over all `world` ranks; *out owns the communicator until cu_comm_destroy
 *   *_bucket            in-place f32 SUM all-reduce of one gradient bucket / reduce-scatter / all-gather of equal shards,
 *                       stream-ordered on `stream` (order it behind the kernels with an event; never the null stream)
 * Return 0, -22 (bad argument), -38 (RCCL not available), -5 (RCCL error; cu_last_error() has the text).
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct cu_comm cu_comm_t;
int cu_comm_unique_id(void* id128);
int cu_comm_init(int rank, int world, const void* id128, cu_comm_t** out);
int cu_comm_allreduce_bucket(cu_comm_t* comm, float* buf, size_t n, void* stream);
int cu_comm_reduce_scatter_bucket(cu_comm_t* comm, const float* send, float* recv, size_t n_per_rank, void* stream);
int cu_comm_allgather_bucket(cu_comm_t* comm, const float* send, float* recv, size_t n_per_rank, void* stream);
int cu_comm_destroy(cu_comm_t* comm);

#ifdef __cplusplus
}
#endif
#endif /* CONTOUR_HIP_H */
