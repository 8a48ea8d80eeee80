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

static inline int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return ((1 << l) == v) ? l : -1;
}
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
