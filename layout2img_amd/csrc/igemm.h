// Implicit-GEMM building blocks shared by the conv forward/dgrad kernel and the
// wgrad kernel: LDS tile geometry, MFMA fragment reads and the K-step compute.
//
// LDS image of both operands is [rows][BK] with the reduction index contiguous,
// rows padded to 144 bytes (128 B of data + 16 B) so that ds_read_b128 fragment
// reads of 32 consecutive rows spread over all 16-byte slots of the bank row.
#pragma once
#include "common.h"

#define IG_ROWB 144  // bytes per LDS tile row (128 data + 16 pad)

template <typename T> struct Mma;

// bf16: v_mfma_f32_32x32x16_bf16, BK = 64 per K-step (4 MFMA k-slices)
template <> struct Mma<bf16_t> {
    static constexpr int BK = 64;
    template <int TM, int TN>
    __device__ static __forceinline__ void step(const char* As, const char* Bs, int wrow, int wcol, int lane,
                                                f32x16_t (&acc)[TM][TN]) {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8_t a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const bf16x8_t*>(As + (wrow + i * 32 + r) * IG_ROWB + kk * 32 + h * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const bf16x8_t*>(Bs + (wcol + j * 32 + r) * IG_ROWB + kk * 32 + h * 16);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a[i]),
                        __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b[j]), acc[i][j], 0, 0, 0);
        }
    }
};

// f32: v_mfma_f32_32x32x2_f32 (exact f32 fma chain), BK = 32 per K-step.
// Lane half h owns k in [16h, 16h+16): MFMA slice t contracts k = t and k = 16+t.
template <> struct Mma<float> {
    static constexpr int BK = 32;
    template <int TM, int TN>
    __device__ static __forceinline__ void step(const char* As, const char* Bs, int wrow, int wcol, int lane,
                                                f32x16_t (&acc)[TM][TN]) {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4_t a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const f32x4_t*>(As + (wrow + i * 32 + r) * IG_ROWB + h * 64 + q * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const f32x4_t*>(Bs + (wcol + j * 32 + r) * IG_ROWB + h * 64 + q * 16);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
        }
    }
};

// ---- LDS-DMA image: unpadded 128-byte rows, 16-byte chunks XOR-swizzled by ((row >> 1) & 7).
// ds_read_b128 is served in 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... over a 256-byte bank row
// (= two 128-byte tile rows): the row's parity picks the half, so the 8 same-parity rows of a group must get 8
// different chunk positions; their (row >> 1) & 7 are all different, their row & 7 are not (measured: 50 % of the
// LDS cycles were bank conflicts with the row & 7 form).
// Half K-step (HK = 1): 64-byte rows, four rows per 256-byte bank row; the 4 rows of a 16-lane group that share a
// quarter have distinct (row >> 2) & 3, which is the swizzle there.
#define IG2_ROWB 128
__device__ __forceinline__ int ig2_swz(int row) { return (row >> 1) & 7; }
template <int HK> __device__ __forceinline__ int ig2_swz_t(int row) { return HK ? ((row >> 2) & 3) : ((row >> 1) & 7); }
template <int HK> __device__ __forceinline__ int ig2_off(int row, int chunk) {
    return row * (HK ? 64 : 128) + ((chunk ^ ig2_swz_t<HK>(row)) << 4);
}

template <typename T> struct Mma2;
template <> struct Mma2<bf16_t> {
    template <int TM, int TN, int HK>
    __device__ static __forceinline__ void step(const char* As, const char* Bs, int wrow, int wcol, int lane,
                                                f32x16_t (&acc)[TM][TN]) {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int kk = 0; kk < (HK ? 2 : 4); ++kk) {
            bf16x8_t a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(As + ig2_off<HK>(wrow + i * 32 + r, kk * 2 + h));
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(Bs + ig2_off<HK>(wcol + j * 32 + r, kk * 2 + h));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a[i]),
                        __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b[j]), acc[i][j], 0, 0, 0);
        }
    }
};
template <> struct Mma2<float> {
    template <int TM, int TN, int HK>
    __device__ static __forceinline__ void step(const char* As, const char* Bs, int wrow, int wcol, int lane,
                                                f32x16_t (&acc)[TM][TN]) {
        const int r = lane & 31, h = lane >> 5;
        constexpr int NQ = HK ? 2 : 4;  // float4 chunks per lane half
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            f32x4_t a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4_t*>(As + ig2_off<HK>(wrow + i * 32 + r, h * NQ + q));
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4_t*>(Bs + ig2_off<HK>(wcol + j * 32 + r, h * NQ + q));
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
        }
    }
};

// XCD-aware bijective block remap: the dispatcher places block b on XCD b % 8;
// give each XCD a contiguous run of tiles so neighbouring tiles share its L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}
