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
        // All fragment reads of a batch of k16 sub-steps are issued BEFORE its MFMAs (sched_barrier keeps hipcc from
        // sinking them back): left alone it keeps 16 fragment registers and alternates "4 reads, wait, 4 MFMAs", which
        // exposes the ~150-cycle LDS latency four times per K-step (measured: 1500 cycles per 512 cycles of MFMA).
        constexpr int NKK = HK ? 2 : 4;
        constexpr int KB = (TM + TN) * NKK <= 16 ? NKK : 2;   // sub-steps per batch: <= 64 fragment VGPRs
#pragma unroll
        for (int k0 = 0; k0 < NKK; k0 += KB) {
            bf16x8_t a[KB][TM], b[KB][TN];
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) a[kk][i] = *reinterpret_cast<const bf16x8_t*>(As + ig2_off<HK>(wrow + i * 32 + r, (k0 + kk) * 2 + h));
#pragma unroll
                for (int j = 0; j < TN; ++j) b[kk][j] = *reinterpret_cast<const bf16x8_t*>(Bs + ig2_off<HK>(wcol + j * 32 + r, (k0 + kk) * 2 + h));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < KB; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b[kk][j]),   // weights first: the accumulator is
                            __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a[kk][i]), acc[i][j], 0, 0, 0);   // the transposed tile (conv_epilogue)
            __builtin_amdgcn_sched_barrier(0);
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
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j][t], a[i][t], acc[i][j], 0, 0, 0);   // transposed tile
        }
    }
};

// Exact n / d by one v_mul_hi_u32 for n * d < 2^32 (tile and halo-row indices: both far below 65536): the integer divisions
// of the kernels' prologues (each ~35 instructions, 25-37 of them per workgroup) were a third of a 3.5-us prologue that a
// workgroup of a short-reduction layer pays in front of a 6-us main loop (tools/perf/conv_trace.py).
// magic = ceil(2^32 / d); 0 encodes d = 1.
static inline unsigned fastdiv_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }
__device__ __forceinline__ int fastdiv(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }

// XCD-aware bijective block remap: the dispatcher places block b on XCD b % 8;
// give each XCD a contiguous run of tiles so neighbouring tiles share its L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// 16-byte LDS-DMA through a buffer descriptor: lane i's data lands at lds_dst + 16 i; an out-of-range byte offset
// makes the hardware write zeros. (Kept in a __device__ helper: the descriptor type does not exist in the host pass.)
//
// Issued through inline asm on purpose: hipcc tracks a builtin LDS-DMA as a pending LDS write and puts
// `s_waitcnt vmcnt(0)` in front of the next ds_read of the same __shared__ array -- i.e. right after the tile for a
// LATER K-step was issued -- which serialises "issue, wait for everything, compute" and leaves no DMA in flight under
// the MFMAs. An asm load is invisible to that bookkeeping; its completion is counted by hand (s_waitcnt vmcnt(N)
// before the barrier of the K-step that consumes it). M0 (LDS base of the wave-instruction) is written in the same
// statement that uses it.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t make_rsrc(const void* base, unsigned nbytes) {
    const unsigned long long a = (unsigned long long)base;
    u32x4_t r;
    r[0] = (unsigned)a;
    r[1] = (unsigned)(a >> 32) & 0xffffu;  // stride 0
    r[2] = nbytes;
    r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ void buf_load_lds16(u32x4_t rsrc, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
// the same with an SGPR byte offset added to the lane offset (no VALU for the per-step part of an address)
#define L2I_DMA16_S(rsrc, voff, soff, ldsaddr)                                                                        \
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), \
                 "s"(soff), "s"(ldsaddr)                                                                              \
                 : "memory")
__device__ __forceinline__ unsigned lds_addr_of(const char* p) {
    return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p;
}

