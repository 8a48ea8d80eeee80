// Shared helpers for the gfx950 kernels of libcontour_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/contour_hip.h"

typedef unsigned short bf16_t;   // raw bf16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

void cu_set_error(const char* fmt, ...);

#define CU_CHECK_ARG(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            cu_set_error(__VA_ARGS__);     \
            return -22; /* -EINVAL */      \
        }                                  \
    } while (0)

#define CU_LAUNCH_CHECK()                                                       \
    do {                                                                        \
        hipError_t e__ = hipGetLastError();                                     \
        if (e__ != hipSuccess) {                                                \
            cu_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,         \
                         hipGetErrorString(e__));                               \
            return -5; /* -EIO */                                               \
        }                                                                       \
    } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned int)v) << 16); }
// round-to-nearest-even; NaN stays NaN through the plain cast (MI355X_MICROARCH "Correctness boundaries")
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return *reinterpret_cast<bf16_t*>(&b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int PIECE = 4;   // elements per 16-byte piece
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int PIECE = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// load / store one 16-byte piece as floats
template <typename T> __device__ __forceinline__ void load_piece(const T* p, float (&v)[Elem<T>::PIECE]);
template <> __device__ __forceinline__ void load_piece<float>(const float* p, float (&v)[4]) {
    f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
}
template <> __device__ __forceinline__ void load_piece<bf16_t>(const bf16_t* p, float (&v)[8]) {
    u32x4 t = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(t[i] << 16);
        v[2 * i + 1] = __uint_as_float(t[i] & 0xffff0000u);
    }
}
template <typename T> __device__ __forceinline__ void store_piece(T* p, const float (&v)[Elem<T>::PIECE]);
template <> __device__ __forceinline__ void store_piece<float>(float* p, const float (&v)[4]) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
}
template <> __device__ __forceinline__ void store_piece<bf16_t>(bf16_t* p, const float (&v)[8]) {
    u32x4 t;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        t[i] = (unsigned int)f32_to_bf16(v[2 * i]) | ((unsigned int)f32_to_bf16(v[2 * i + 1]) << 16);
    *reinterpret_cast<u32x4*>(p) = t;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- halving butterfly over the 32 pixel lanes of a half wave (MFMA 32x32 output layout: a lane owns one pixel, register i
// of a column block is channel (i & 3) + 8 (i >> 2) + 4 h).  8 + 4 + 2 + 1 exchanges bring 16 registers down to one per
// lane in DPP / permlane-swap form; lane l then holds the total of register
// i = (l & 1) << 3 | (l & 2) << 1 | (l & 8) >> 2 | (l & 16) >> 4 (bit 2 of l: replicated).  See igemm_conv.hip (tile_stats).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_reduce16(float (&v)[16], int lane) {
    {   // lanes l, l ^ 1 (quad_perm [1,0,3,2]): odd lanes keep registers 8..15
        const bool up = lane & 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float lo = v[j] + dpp_mov<0xB1>(v[j]), hi = v[j + 8] + dpp_mov<0xB1>(v[j + 8]);
            v[j] = up ? hi : lo;
        }
    }
    {   // l, l ^ 2 (quad_perm [2,3,0,1])
        const bool up = lane & 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = v[j] + dpp_mov<0x4E>(v[j]), hi = v[j + 4] + dpp_mov<0x4E>(v[j + 4]);
            v[j] = up ? hi : lo;
        }
    }
    {   // l, l ^ 8 (row_ror:8)
        const bool up = lane & 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float lo = v[j] + dpp_mov<0x128>(v[j]), hi = v[j + 2] + dpp_mov<0x128>(v[j + 2]);
            v[j] = up ? hi : lo;
        }
    }
    // l, l ^ 16: v_permlane16_swap exchanges the odd rows of its first operand with the even rows of its second: the two
    // results are (v0 of rows 0,0,2,2 | v1 of rows ... ) such that their sum is v0 + v0' in even rows, v1 + v1' in odd rows
    const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[0]), __float_as_uint(v[1]), false, false);
    const float t = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    return t + __shfl_xor(t, 4, 64);
}


// ---- LDS-DMA helpers (gfx950: buffer_load_dwordx4 ... lds) ----------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) int i32x4;

// buffer resource (V#) of a raw byte buffer: out-of-range offsets read zeros
__device__ __forceinline__ i32x4 make_rsrc(const void* ptr, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
// One LDS-DMA wave instruction: lane l copies 16 bytes from rsrc + voff(l) to LDS byte address lds_base + 16 l.
// Issued from inline asm on purpose: hipcc orders every later LDS read behind an LDS-DMA it knows about
// (s_waitcnt vmcnt(0) in front of the first ds_read_tr of the k-loop), which would serialise the copy of tile i+1 with
// the MFMAs of tile i.  The copies are retired by hand: dma_wait() before the barrier that hands the image over.
__device__ __forceinline__ void dma16(const i32x4 rsrc, unsigned voff, unsigned lds_base) {
    unsigned keep;
    lds_base = __builtin_amdgcn_readfirstlane(lds_base);      // wave-uniform by construction: make it an SGPR
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p);
}

// ---- timing experiments (tools/): compiled OUT of the product library.  `make TUNING=1` builds with -DCU_TUNING, which
// lets CU_CONV_DBG bits skip stores / MFMAs / loads and lets CU_* environment variables override the kernel selection;
// without it no environment variable can change what a launch computes or which kernel runs.
#ifdef CU_TUNING
#include <stdlib.h>
#define CU_DBG(p, bits) ((p).dbg & (bits))
static inline int cu_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static inline bool cu_env_set(const char* name) { return getenv(name) != nullptr; }
#else
#define CU_DBG(p, bits) 0
static inline int cu_env_int(const char*, int dflt) { return dflt; }
static inline bool cu_env_set(const char*) { return false; }
#endif

// layout code of cu_conv_wgrad_parts for slabs in the plain [wtaps][CO][CI] layout (else (NBLK << 8) | CBLK: native)
#define CU_PARTS_PLAIN 0xffff

static inline int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return ((1 << l) == v) ? l : -1;
}
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
