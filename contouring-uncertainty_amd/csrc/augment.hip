// On-device training augmentation of the CAMUS contour data module (reference
// contour_uncertainty/data/camus/datamodule.py:46-55: Compose([RandomRotation(3), RandomBrightnessContrast(0.2, 0.2),
// RandomGamma((0.8, 1.2)), RandomTranslation(5, 5)]), augmentations/{affine,brightnesscontrast,gamma}.py).
// The reference applies the four transforms per item on the CPU workers through torchvision.transforms.functional
// (torchvision is the third-party dependency whose algorithm is restated here: rotate / affine = inverse affine matrix ->
// affine grid -> grid_sample(nearest, zeros, align_corners=False); adjust_brightness / adjust_contrast = blend + clamp to
// [0, 1]; adjust_gamma = pow + clamp).  Here a whole batch is transformed by two launches, each image with its own parameters:
//   pass 1  per image: mean of clamp(brightness * rotate(img)) -- adjust_contrast blends with the mean of ITS input
//   pass 2  per output pixel: un-translate -> un-rotate -> nearest sample -> brightness -> contrast -> gamma -> store
// HBM-bound streaming (4 B in + 4 B out per pixel, the gather stays within a few rows): no LDS, no MFMA.
#include "common.h"

namespace {

struct AugP { float angle, tx, ty, bright, contrast, gamma; };

// source pixel of rotate(img, angle) at output pixel (x, y): torchvision's grid in float32, then grid_sample's
// un-normalisation and round-half-to-even; returns false outside the image (zero fill)
__device__ __forceinline__ bool rot_src(int x, int y, int W, int H, float c, float s, int& sx, int& sy) {
    const float xc = (float)x - 0.5f * (float)W + 0.5f, yc = (float)y - 0.5f * (float)H + 0.5f;      // base grid
    const float hw = 0.5f * (float)W, hh = 0.5f * (float)H;
    const float gx = xc * (c / hw) + yc * (-s / hw);          // theta = [cos, -sin, 0; sin, cos, 0] (inverse of +angle)
    const float gy = xc * (s / hh) + yc * (c / hh);
    const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    const float rx = nearbyintf(ix), ry = nearbyintf(iy);
    sx = (int)rx; sy = (int)ry;
    return rx >= 0.f && rx <= (float)(W - 1) && ry >= 0.f && ry <= (float)(H - 1);
}

__global__ __launch_bounds__(1024) void augment_mean_kernel(const float* __restrict__ img, const float* __restrict__ params,
                                                            float* __restrict__ mean, int H, int W) {
    __shared__ float red[16];
    const int n = blockIdx.x;
    const float* pp = params + (size_t)n * 8;
    const float rad = pp[0] * 0.017453292519943295f;
    const float c = cosf(rad), s = sinf(rad), bright = pp[3];
    const float* src = img + (size_t)n * H * W;
    float acc = 0.f;
    for (int i = threadIdx.x; i < H * W; i += 1024) {
        const int y = i / W, x = i - y * W;
        int sx, sy;
        float v = rot_src(x, y, W, H, c, s, sx, sy) ? src[(size_t)sy * W + sx] : 0.f;
        v = fminf(fmaxf(bright * v, 0.f), 1.f);
        acc += v;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        mean[n] = t / (float)(H * W);
    }
}

template <typename T, bool COLOR>
__global__ __launch_bounds__(256) void augment_apply_kernel(const T* __restrict__ img, const float* __restrict__ params,
                                                            const float* __restrict__ mean, T* __restrict__ out, int H, int W) {
    const int n = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const float* pp = params + (size_t)n * 8;
    const float rad = pp[0] * 0.017453292519943295f;
    const float c = cosf(rad), s = sinf(rad);
    const int tx = (int)pp[1], ty = (int)pp[2];
    const int y = i / W, x = i - y * W;
    const int xr = x - tx, yr = y - ty;                       // translate: F.affine with matrix [1, 0, -tx; 0, 1, -ty]
    T v = (T)0;
    if (xr >= 0 && xr < W && yr >= 0 && yr < H) {
        int sx, sy;
        const T* src = img + (size_t)n * H * W;
        v = rot_src(xr, yr, W, H, c, s, sx, sy) ? src[(size_t)sy * W + sx] : (T)0;
        if constexpr (COLOR) {
            float f = (float)v;
            f = fminf(fmaxf(pp[3] * f, 0.f), 1.f);                                        // adjust_brightness
            f = fminf(fmaxf(pp[4] * f + (1.f - pp[4]) * mean[n], 0.f), 1.f);               // adjust_contrast
            if (pp[5] != 1.f) f = fminf(fmaxf(powf(f, pp[5]), 0.f), 1.f);                   // adjust_gamma (gain 1; x ** 1 = x exactly)
            v = (T)f;
        }
    }
    out[(size_t)n * H * W + i] = v;
}

}  // namespace

extern "C" int cu_augment_image(int N, int H, int W, const float* img, const float* params, float* mean_ws, float* out,
                                void* stream) {
    CU_CHECK_ARG(N > 0 && H > 0 && W > 0 && img && params && mean_ws && out && img != out, "cu_augment_image: bad argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(augment_mean_kernel, dim3(N), dim3(1024), 0, st, img, params, mean_ws, H, W);
    CU_LAUNCH_CHECK();
    hipLaunchKernelGGL((augment_apply_kernel<float, true>), dim3(cdiv(H * W, 256), N), dim3(256), 0, st, img, params,
                       (const float*)mean_ws, out, H, W);
    CU_LAUNCH_CHECK();
    return 0;
}

extern "C" int cu_augment_labels(int N, int H, int W, const long long* labels, const float* params, long long* out,
                                 void* stream) {
    CU_CHECK_ARG(N > 0 && H > 0 && W > 0 && labels && params && out && labels != out, "cu_augment_labels: bad argument");
    hipLaunchKernelGGL((augment_apply_kernel<long long, false>), dim3(cdiv(H * W, 256), N), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), labels, params, (const float*)nullptr, out, H, W);
    CU_LAUNCH_CHECK();
    return 0;
}
