// Micro-experiment (not part of the product): LDS-DMA gather throughput when the source is L2-resident, as in the
// weight-gradient kernel (every tile byte is fetched by four workgroups of the same XCD).  NW issuing waves per
// workgroup (one workgroup per CU) stream 1-KiB pieces; a piece = 16-byte lanes forming rows of ROW bytes, rows STRIDE
// bytes apart, inside a WINDOW shared by all workgroups.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/dma_l2_gather.hip -o /tmp/dmal2 && /tmp/dmal2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) int i32x4;

__device__ __forceinline__ void dma16(const i32x4 rsrc, unsigned voff, unsigned lds_base) {
    unsigned keep;
    lds_base = __builtin_amdgcn_readfirstlane(lds_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_base) : "memory");
}

__global__ __launch_bounds__(512) void k(const unsigned char* src, float* out, int iters, int pieces, int nw, unsigned row,
                                         unsigned stride, unsigned window) {
    extern __shared__ unsigned char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const unsigned long long a = reinterpret_cast<unsigned long long>(src);
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r[2] = (int)window; r[3] = 0x00020000;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)smem;
    const unsigned lanes_per_row = row / 16u, rows_per_piece = 1024u / row;
    for (int it = 0; it < iters; ++it) {
        if (wave < nw) {
            for (int p = 0; p < pieces; ++p) {
                const unsigned piece = (unsigned)((it * pieces + p) * nw + wave) + blockIdx.x * 977u;
                const unsigned rw = piece * rows_per_piece + lane / lanes_per_row;
                const unsigned off = (rw * stride + (lane % lanes_per_row) * 16u) % window;
                dma16(r, off & ~15u, lds0 + (wave * 16 + (p & 15)) * 1024);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    if (out && smem[tid] == 123 && iters < 0) out[tid] = 1.f;
}

int main() {
    unsigned char* src; float* out;
    const size_t cap = 512u << 20;
    hipMalloc(&src, cap); hipMalloc(&out, 4096);
    hipMemset(src, 1, cap);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    const int iters = 64, pieces = 16;
    const unsigned windows[] = {2u << 20, 32u << 20, 256u << 20};
    const unsigned rows[] = {64, 128, 256, 1024};
    for (unsigned window : windows)
        for (unsigned row : rows)
            for (int nw : {4, 8}) {
                const unsigned stride = row == 1024 ? 1024 : 512;
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipLaunchKernelGGL(k, dim3(256), dim3(512), 128 * 1024, 0, src, out, 2, pieces, nw, row, stride, window);
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(256), dim3(512), 128 * 1024, 0, src, out, iters, pieces, nw, row, stride, window);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double bytes = 256.0 * nw * iters * pieces * 1024;
                printf("window %4u MiB  row %4u B (stride %4u)  %d waves: %8.1f us  %6.2f TB/s  %6.1f GB/s per CU\n", window >> 20, row,
                       stride, nw, ms * 1e3, bytes / ms / 1e9, bytes / 256 / ms / 1e6);
            }
    return 0;
}
