// Shared device helpers for the layout2img gfx950 kernels.
// gfx950 only: 64-wide wavefronts, MFMA, 160 KiB LDS. No portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define L2I_OK 0
#define L2I_ERR_ARG (-1)
#define L2I_ERR_LAUNCH (-2)

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment, 4 VGPRs
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// round-to-nearest-even f32 -> bf16 (NaN preserved as quiet NaN)
__device__ __forceinline__ bf16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

template <typename T> struct OpT;
template <> struct OpT<bf16_t> {
    static constexpr int EPG = 8;  // elements per 16-byte group
    __device__ static __forceinline__ bf16_t from(float f) { return f2bf(f); }
    __device__ static __forceinline__ float to(bf16_t v) { return bf2f(v); }
};
template <> struct OpT<float> {
    static constexpr int EPG = 4;
    __device__ static __forceinline__ float from(float f) { return f; }
    __device__ static __forceinline__ float to(float v) { return v; }
};

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

// block-wide sum for blockDim.x <= 1024 (multiple of 64); red must hold 16 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += red[i];
    __syncthreads();  // red[] may be reused by the caller's next reduction
    return r;
}

static inline int l2i_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? L2I_OK : L2I_ERR_LAUNCH;
}
