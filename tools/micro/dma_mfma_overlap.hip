// Micro-experiment (not part of the product): can LDS-DMA issue of one wave overlap with the MFMAs of the other wave
// on the same SIMD?  8 waves per workgroup; waves 0-3 stream 1-KiB LDS-DMA pieces (64-byte rows gathered from a
// per-CU 8 MiB window), waves 4-7 run dependent-free 32x32x16 bf16 MFMAs.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/dma_mfma_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ void dma16(const i32x4 rsrc, unsigned voff, unsigned lds_base) {
    unsigned keep;
    lds_base = __builtin_amdgcn_readfirstlane(lds_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_base) : "memory");
}

template <int MODE, int SPREAD>   // MODE bit0: DMA waves active, bit1: MFMA waves active; SPREAD: DMA waves also do MFMAs between pieces
__global__ __launch_bounds__(512) void k(const unsigned char* src, float* out, int iters, int pieces) {
    extern __shared__ unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned long long a = reinterpret_cast<unsigned long long>(src + (size_t)blockIdx.x * (8u << 20));
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r[2] = 8 << 20; r[3] = 0x00020000;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    bf16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(lane * 0.01f); y[j] = (__bf16)(j * 0.1f); }
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)smem;
    for (int it = 0; it < iters; ++it) {
        if (wave < 4) {
            if (MODE & 1) {
                for (int p = 0; p < pieces; ++p) {
                    // 16 rows of 64 B per piece, rows 512 B apart (a 256-channel bf16 tensor), windows rotate
                    const unsigned row = ((unsigned)(it * pieces + p) * 64u + wave * 16u + (lane >> 2)) & 16383u;
                    dma16(r, row * 512u + (lane & 3) * 16u, lds0 + (wave * 16 + (p & 15)) * 1024);
                    if (SPREAD)
                        for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[m], 0, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else if (MODE & 2) {
            for (int p = 0; p < pieces; ++p)
                for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[m], 0, 0, 0);
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 12345.678f) out[tid] = s + smem[tid];
}

template <int MODE, int SPREAD>
float run(const unsigned char* src, float* out, int iters, int pieces) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, SPREAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, SPREAD>), dim3(256), dim3(512), 64 * 1024, 0, src, out, 2, pieces);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, SPREAD>), dim3(256), dim3(512), 64 * 1024, 0, src, out, iters, pieces);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

int main() {
    unsigned char* src; float* out;
    hipMalloc(&src, (size_t)256 * (8u << 20) + 4096); hipMalloc(&out, 4096);
    hipMemset(src, 1, (size_t)256 * (8u << 20));
    const int iters = 64, pieces = 16;
    const float a = run<1, 0>(src, out, iters, pieces), b = run<2, 0>(src, out, iters, pieces), c = run<3, 0>(src, out, iters, pieces);
    const float d = run<1, 1>(src, out, iters, pieces), e = run<3, 1>(src, out, iters, pieces);
    const double bytes = 256.0 * 4 * iters * pieces * 1024, flops = 256.0 * 4 * iters * pieces * 8 * 32768.0;
    printf("DMA only        %8.1f us  (%.2f TB/s)\n", a, bytes / a / 1e6);
    printf("MFMA only       %8.1f us  (%.0f TFLOP/s on 4 of 8 waves)\n", b, flops / b / 1e6);
    printf("both (2 roles)  %8.1f us  -> sum %.1f, max %.1f\n", c, a + b, a > b ? a : b);
    printf("DMA waves interleave 8 MFMAs per piece, others idle: %8.1f us\n", d);
    printf("same + MFMA waves busy:                              %8.1f us\n", e);
    return 0;
}
