// Weight gradient of the implicit-GEMM convolution (gfx950 MFMA), NHWC operands.
//
//   dW[co, k] += alpha * sum_m dYfull[m, co] * A[m, k]
//   m = (b, y, x) over the conv-output grid (Ho, Wo), k = (ky, kx, ci)
//   dYfull[m]  = dY[b, y >> pool2, x >> pool2]      (alpha carries the 1/4 of avg_pool2d)
//   A[m, k]    = X[b, (y+ky-pad) >> up2, (x+kx-pad) >> up2, ci]
//
// This is the autograd of nn.Conv2d / nn.Linear weights on the reference path
// (model/resnet_generator_app_v2.py:633-639, model/rcnn_discriminator_app.py:297-326)
// and is a "TN" GEMM: both operands are stored with the reduction index (pixels)
// strided and the channel contiguous, while MFMA fragments want the reduction
// index contiguous per lane. Each thread therefore loads an EPG-pixel x
// EPG-channel block (EPG 16-byte loads), transposes it in registers and writes
// EPG 16-byte rows [channel][pixel0..] into LDS; after that the LDS image and
// the MFMA loop are the same as the forward kernel's. The pixel reduction is
// split over `splits` workgroups per output tile, combined by f32 atomics.
#include <stdlib.h>
#include <mutex>
#include "igemm.h"

L2I_TRACE_DEFINE(wgrad)

struct WgradArgs {
    const void* x;   // T [B, Hi, Wi, Ci]
    const void* dy;  // T [B, Hd, Wd, Co]
    float* dw;       // f32 [Co, ldw], accumulated atomically
    int B, Hi, Wi, Ci, Ho, Wo, Co, KH, up2, pool2;
    int K, ldw, M, Mper, tiles_co, tiles_k, splits;
    float alpha;
    unsigned x_bytes, dy_bytes;   // buffer-descriptor ranges of x and dy
    const int* nimg;              // device int (optional): reduce over the first *nimg images only
    float* dbias;                 // optional [Co]: += alpha * sum_m dYfull[m, co] (the bias gradient), by the tile_k == 0 workgroups
    int no_epi;                   // ablation builds only (-DL2I_ABLATIONS + L2I_WGRAD_NOEPI=1, results are wrong): skip the atomic epilogue to measure what it costs
    int lgbk;                     // log2 of the kernel's pixel step (6; 7 for the eight-wave kernel)
    float* part;                  // optional scratch: every workgroup STORES its partial tile there ([tile][split][BMO][128] f32) and
                                  // wgrad_reduce_kernel adds the splits into dw -- instead of one f32 atomic per element and split
    // Folded 1x1 shortcut (l2i_conv2d_wgrad_sc): a residual block's conv2 and its shortcut receive the SAME dY, so the
    // shortcut's weight gradient dW_sc = dY^T x_sc is tiles_k - tiles_k_main more column tiles of this launch: they stage the
    // same dY steps and read x_sc (centre tap; optionally nearest-upsampled, sc_up2) instead of im2col(x). null sc_x: none.
    const void* sc_x;
    float* sc_dw;                 // f32 [Co, sc_ldw]
    float* dbias2;                // optional: the shortcut's bias gradient (= this layer's: the same sum over dY)
    int sc_Ci, sc_ldw, sc_up2, tiles_k_main;   // sc_up2: the shortcut reads x_sc [B, Ho/2, Wo/2, sc_Ci] at (y >> 1, x >> 1) (generator blocks)
    unsigned sc_x_bytes;
    // DUAL launch (l2i_conv2d_wgrad_dual): the B images are two passes of B/2 images (D(real) and D(fake) of the discriminator step,
    // reference train_context_app_v2.py:158,167) whose weight gradients go to different accumulators -- each pass has its own
    // W / sigma and therefore its own sigma-correction in the spectral-norm backward. Splits [0, splits/2) reduce the pixels of the
    // first half into dw (sc_dw), splits [splits/2, splits) those of the second half into dw_b (sc_dw_b); `nimg` counts the live
    // images of EACH half. null dw_b: single.
    float* dw_b;
    float* sc_dw_b;
    // In-kernel reduction of the splits (round 4, L2I_WGRAD_FUSE=1): the workgroup that stores the LAST partial tile of an output
    // tile (a device-scope counter per tile and accumulator) adds all of the tile's partial tiles into dw itself -- no
    // wgrad_reduce_kernel launch behind the main kernel. null: the separate reduce kernel.
    unsigned* fuse_cnt;
    int overwrite;    // caller's promise: this launch is the ONLY writer of the dW slices it touches since they were zeroed (one weight-
                      // gradient launch per layer application and pass) -- the result is then STORED: no f32 atomics where a tile has one
                      // split, no read-modify-write in the reduce kernel. 0: dW += (the ABI's default semantics)
    int nw2_layout;   // (the register-order decode of the partial tiles: as wgrad_reduce_kernel's nw2)
    int zero_targets; // overwrite + a reduce that combines split GROUPS with atomics: the main kernel (which writes only partial tiles to
                      // scratch) clears the dW slices on its way in, so the reduce kernel behind it adds into zeros -- no memset launches
    // Round 6, no float atomics when the caller's scratch has room:
    float* bpart;     // bias-gradient shares: row (split x nb_bias + tile_k) of [splits * nb_bias][bpart_ld], alpha applied, STORED by the workgroups
    int bpart_ld;     // that sum the bias (tiles_co * BMO columns); rows_fold adds the rows in order behind the launch. null: one atomic per channel
    float* part_rm;   // generic kernel (f32 operands, odd shapes): partial tiles stored ROW-major [tile][split][BMO][128], summed by wgrad_reduce_rm_kernel
    int nb_bias;      // LDS-DMA kernel: column tiles that share a (channel tile, split)'s bias sum (1, 2 or 4: set by the launcher)
};

// grid-wide clear of one dW slice (n floats) by the threads of the weight-gradient kernel
__device__ __forceinline__ void wg_zero_slice(float* __restrict__ d, long long n, long long gtid, long long gthreads) {
    if (!d) return;
    if ((((unsigned long long)d & 15ull) | ((unsigned long long)n & 3ull)) == 0) {
        float4* d4 = reinterpret_cast<float4*>(d);
        for (long long i = gtid; i < (n >> 2); i += gthreads) d4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        for (long long i = gtid; i < n; i += gthreads) d[i] = 0.f;
    }
}

// Device-scope ("sc1") accesses for the partial tiles of the fused reduction: on gfx950 every XCD has its own L2, so data one
// workgroup hands to a workgroup on another XCD must be written through and read around the L2s. With sc1 on the stores and on
// the last arriver's loads, `s_waitcnt vmcnt(0)` + a relaxed device-scope counter is all the ordering needed -- the compiler's
// agent-scope release / acquire FENCES (buffer_wbl2 / buffer_inv: write back / invalidate the whole L2, once per workgroup) made
// the iteration 4.6 ms slower (measured: 19.7 -> 24.3 ms).
__device__ __forceinline__ void wg_store_sc1(float4* ptr, f32x4_t v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4_t wg_load_sc1(const float4* ptr) {   // (the caller waits: s_waitcnt vmcnt(0))
    f32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(ptr) : "memory");
    return v;
}

#define L2I_WGRAD_CNT (1 << 20)
__device__ unsigned g_wgrad_cnt[L2I_WGRAD_CNT];   // zero at module load; every counter is reset by its last arriver

// The reduction range [m_begin, m_end) of one split. With a device-side image count the live pixels are divided over
// the launch's `splits` again on the device (whole 64-pixel steps), so every split still does an equal share.
__device__ __forceinline__ void wgrad_range(const WgradArgs& p, int split, int& m_begin, int& m_end) {
    int M = p.M, Mper = p.Mper, nsp = p.splits, base = 0;
    if (p.dw_b) {   // dual: this split's half of the images (p.M, p.Mper are per half then)
        nsp >>= 1;
        if (split >= nsp) { split -= nsp; base = p.M; }
    }
    if (p.nimg) {
        M = min(M, *p.nimg * p.Ho * p.Wo);
        const int steps = (M + (1 << p.lgbk) - 1) >> p.lgbk;
        Mper = ((steps + nsp - 1) / nsp) << p.lgbk;
    }
    m_begin = base + split * Mper;
    m_end = min(base + M, m_begin + Mper);
}
__device__ __forceinline__ bool wgrad_second(const WgradArgs& p, int split) { return p.dw_b && split >= (p.splits >> 1); }

template <typename T> struct Tr;
template <> struct Tr<bf16_t> {
    // r[j] = 8 channels of pixel j  ->  o[c] = 8 pixels of channel c
    __device__ static __forceinline__ void run(const uint4 (&r)[8], uint4 (&o)[8]) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = d == 0 ? r[j].x : d == 1 ? r[j].y : d == 2 ? r[j].z : r[j].w;
            uint4 lo, hi;
            lo.x = (v[0] & 0xffffu) | (v[1] << 16); hi.x = (v[0] >> 16) | (v[1] & 0xffff0000u);
            lo.y = (v[2] & 0xffffu) | (v[3] << 16); hi.y = (v[2] >> 16) | (v[3] & 0xffff0000u);
            lo.z = (v[4] & 0xffffu) | (v[5] << 16); hi.z = (v[4] >> 16) | (v[5] & 0xffff0000u);
            lo.w = (v[6] & 0xffffu) | (v[7] << 16); hi.w = (v[6] >> 16) | (v[7] & 0xffff0000u);
            o[2 * d] = lo;
            o[2 * d + 1] = hi;
        }
    }
};
template <> struct Tr<float> {
    __device__ static __forceinline__ void run(const uint4 (&r)[4], uint4 (&o)[4]) {
        o[0] = make_uint4(r[0].x, r[1].x, r[2].x, r[3].x);
        o[1] = make_uint4(r[0].y, r[1].y, r[2].y, r[3].y);
        o[2] = make_uint4(r[0].z, r[1].z, r[2].z, r[3].z);
        o[3] = make_uint4(r[0].w, r[1].w, r[2].w, r[3].w);
    }
};

template <typename T, int BMO>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs p) {
    constexpr int BNK = 128;  // k-columns per tile
    constexpr int BK = Mma<T>::BK;
    constexpr int EPG = OpT<T>::EPG;
    constexpr int TM = BMO / 64, TN = 2;
    constexpr int N1 = 8 * (BMO / EPG);  // staging tasks for the dY operand
    constexpr int N2 = 8 * (BNK / EPG);  // staging tasks for the im2col operand
    constexpr int NIT = (N1 + N2 + 255) / 256;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                  // [BMO][pixels]
    char* Bs = smem + BMO * IG_ROWB;  // [BNK][pixels]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    const int split = bid % p.splits; bid /= p.splits;
    const int tile_k = bid % p.tiles_k, tile_co = bid / p.tiles_k;
    const int co0 = tile_co * BMO, kc0 = tile_k * BNK;
    const int pad = p.KH >> 1;
    const int Hd = p.Ho >> p.pool2, Wd = p.Wo >> p.pool2;
    const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ DY = reinterpret_cast<const T*>(p.dy);

    int m_begin, m_end;
    wgrad_range(p, split, m_begin, m_end);

    // per-task constants
    int t_pg[NIT], t_row[NIT], t_chan[NIT], t_ky[NIT], t_kx[NIT];
    bool t_is1[NIT], t_on[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int task = tid + it * 256;
        t_is1[it] = task < N1;
        const int u = t_is1[it] ? task : task - N1;
        t_pg[it] = u & 7;
        const int cg = u >> 3;
        t_row[it] = cg * EPG;
        t_on[it] = task < N1 + N2;
        t_ky[it] = 0; t_kx[it] = 0;
        if (t_is1[it]) {
            t_chan[it] = co0 + cg * EPG;
            t_on[it] = t_on[it] && t_chan[it] < p.Co;
        } else {
            const int kc = kc0 + cg * EPG;
            const int tap = kc / p.Ci;
            t_chan[it] = kc - tap * p.Ci;
            t_ky[it] = tap / p.KH;
            t_kx[it] = tap - t_ky[it] * p.KH;
            t_on[it] = t_on[it] && kc < p.K;
        }
    }

    uint4 stage[NIT][EPG];
    auto load_step = [&](int mstep) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int m = mstep + t_pg[it] * EPG;
            int b = m / (p.Ho * p.Wo);
            int rem = m - b * p.Ho * p.Wo;
            int y = rem / p.Wo, x = rem - y * p.Wo;
#pragma unroll
            for (int j = 0; j < EPG; ++j) {
                uint4 v = make_uint4(0, 0, 0, 0);
                if (t_on[it] && m + j < m_end) {
                    if (t_is1[it]) {
                        const size_t off = ((size_t)(b * Hd + (y >> p.pool2)) * Wd + (x >> p.pool2)) * p.Co + t_chan[it];
                        v = *reinterpret_cast<const uint4*>(DY + off);
                    } else {
                        const int yy = y + t_ky[it] - pad, xx = x + t_kx[it] - pad;
                        if (yy >= 0 && yy < p.Ho && xx >= 0 && xx < p.Wo) {
                            const size_t off = ((size_t)(b * p.Hi + (yy >> p.up2)) * p.Wi + (xx >> p.up2)) * p.Ci + t_chan[it];
                            v = *reinterpret_cast<const uint4*>(X + off);
                        }
                    }
                }
                stage[it][j] = v;
                if (++x == p.Wo) { x = 0; if (++y == p.Ho) { y = 0; ++b; } }
            }
        }
    };
    auto store_step = [&]() {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (tid + it * 256 >= N1 + N2) continue;
            uint4 o[EPG];
            Tr<T>::run(stage[it], o);
            char* tile = t_is1[it] ? As : Bs;
#pragma unroll
            for (int c = 0; c < EPG; ++c)
                *reinterpret_cast<uint4*>(tile + (t_row[it] + c) * IG_ROWB + t_pg[it] * 16) = o[c];
        }
    };

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float bsum = 0.f;

    const int wrow = (wave >> 1) * (BMO / 2), wcol = (wave & 1) * 64;
    if (m_begin < m_end) {
        load_step(m_begin);
        for (int ms = m_begin; ms < m_end; ms += BK) {
            store_step();
            __syncthreads();
            if (ms + BK < m_end) load_step(ms + BK);
            if (p.dbias && tile_k == 0 && tid < 2 * BMO) {   // bias gradient: row (= channel) sums of the dY tile
                const T* rowp = reinterpret_cast<const T*>(As + (tid >> 1) * IG_ROWB) + (tid & 1) * (BK / 2);
#pragma unroll 8
                for (int j = 0; j < BK / 2; ++j) bsum += OpT<T>::to(rowp[j]);
            }
            Mma<T>::template step<TM, TN>(As, Bs, wrow, wcol, lane, acc);
            __syncthreads();
        }
    }
    if (p.dbias && tile_k == 0 && tid < 2 * BMO) {
        bsum += __shfl_xor(bsum, 1, 64);
        const int co = co0 + (tid >> 1);
        if (!(tid & 1)) {
            if (p.bpart) p.bpart[(size_t)split * p.bpart_ld + co] = p.alpha * bsum;   // (every split's row is written, zeros included)
            else if (co < p.Co && bsum != 0.f) atomicAdd(p.dbias + co, p.alpha * bsum);
        }
    }

    const int c = lane & 31, h = lane >> 5;
    if (p.part_rm) {   // the split's partial tile, row-major and unscaled (wgrad_reduce_rm_kernel adds the splits in order)
        float* t = p.part_rm + ((size_t)(tile_co * p.tiles_k + tile_k) * p.splits + split) * (size_t)(BMO * 128);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    t[(size_t)(wrow + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * 128 + wcol + j * 32 + c] = acc[i][j][e];
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = kc0 + wcol + j * 32 + c;
            if (col >= p.K) continue;
            float* dwp = wgrad_second(p, split) ? p.dw_b : p.dw;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = co0 + wrow + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row >= p.Co) continue;
                if (p.overwrite && p.splits == 1) dwp[(size_t)row * p.ldw + col] = p.alpha * acc[i][j][e];
                else atomicAdd(dwp + (size_t)row * p.ldw + col, p.alpha * acc[i][j][e]);
            }
        }
}

// dw[row][col] (+)= alpha * sum over the splits of the generic kernel's row-major partial tiles (WgradArgs::part_rm), in order.
// One thread = four consecutive columns of one row of one tile; gridDim.z = 2: dual launch (second half of the splits -> dw_b).
__global__ __launch_bounds__(256) void wgrad_reduce_rm_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ dw_b, int BMO,
                                                              int tiles_k, int ntiles, int splits, int Co, int K, int ldw, float alpha, int overwrite) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int per_tile = BMO * 32;
    const int tile = (int)(gid / per_tile), f = (int)(gid - (long long)tile * per_tile);
    if (tile >= ntiles) return;
    const int tile_k = tile % tiles_k, tile_co = tile / tiles_k;
    const int row = tile_co * BMO + (f >> 5), col = tile_k * 128 + 4 * (f & 31);
    if (row >= Co || col >= K) return;
    const int hs = splits / (int)gridDim.z, s_off = (int)blockIdx.z * hs;
    if (blockIdx.z) dw = dw_b;
    const float4* src = reinterpret_cast<const float4*>(part + ((size_t)tile * splits) * (size_t)(BMO * 128)) + f;
    const size_t tsz4 = (size_t)BMO * 32;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = s_off;
    for (; s + 4 <= s_off + hs; s += 4) {
        const float4 v0 = src[(size_t)s * tsz4], v1 = src[(size_t)(s + 1) * tsz4], v2 = src[(size_t)(s + 2) * tsz4], v3 = src[(size_t)(s + 3) * tsz4];
        a.x += (v0.x + v1.x) + (v2.x + v3.x); a.y += (v0.y + v1.y) + (v2.y + v3.y);
        a.z += (v0.z + v1.z) + (v2.z + v3.z); a.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; s < s_off + hs; ++s) {
        const float4 v0 = src[(size_t)s * tsz4];
        a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
    }
    float* d = dw + (size_t)row * ldw + col;
    const float v[4] = {alpha * a.x, alpha * a.y, alpha * a.z, alpha * a.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (col + e < K) d[e] = overwrite ? v[e] : d[e] + v[e];
}

// ---------------------------------------------------------------- bf16 fast path: LDS-DMA + transposing LDS reads
// Both operands are staged UNTRANSPOSED ([pixel][channel], the layout they have in HBM) by LDS-DMA
// (global_load_lds_dwordx4, no VGPR round trip, no ds_write), double-buffered with one barrier per 64-pixel step.
// The MFMA fragments (8 consecutive pixels of one channel per lane) are produced by ds_read_b64_tr_b16, gfx950's
// transposing LDS read: within a 16-lane group lane t addresses the 8-byte piece (row t>>2, 4-channel block t&3)
// of a 4-pixel x 16-channel block and lane c receives channel c's 4 pixels. Rows that a half-wave reads together
// are spread over the 64 banks by XOR-ing the 16-byte chunk index with a function of the pixel row, applied on
// the DMA source side and on the read. Requires power-of-two Ho, Wo (all layers of this path).

typedef __attribute__((ext_vector_type(4))) short s16x4_t;

template <int RS>  // RS = bytes per LDS pixel row (256: 128 channels, 128: 64 channels)
__device__ __forceinline__ int wg_swz(int p) { return RS == 256 ? ((p & 3) << 2) : (((p >> 1) & 1) << 2); }

template <int RS>
__device__ __forceinline__ bf16x8_t wg_frag(const char* tile, int pixbase, int chbase, int lane) {
    const int t = lane & 15, cb = ((lane >> 4) & 1) * 16, h = lane >> 5;
    const int ch = chbase + cb + (t & 3) * 4;
    const int L = ch >> 3, half = (ch >> 2) & 1;
    bf16x8_t out;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int p = pixbase + 8 * h + 4 * r + (t >> 2);
        const int addr = p * RS + ((L ^ wg_swz<RS>(p)) << 4) + half * 8;
        const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)((__attribute__((address_space(3))) char*)tile + addr));
        out[4 * r + 0] = v[0]; out[4 * r + 1] = v[1]; out[4 * r + 2] = v[2]; out[4 * r + 3] = v[3];
    }
    return out;
}

// MODE 1 (FAST): no pool / upsample, M % 64 == 0: all addresses are lane constants + an SGPR step base.
// MODE 2 (SEMI): pool and/or upsample, M % 64 == 0 and Ho*Wo % 64 == 0: a 64-pixel step lies inside one image and starts
//   on a row (or half-row) boundary, so image / row / column of the step are SCALARS and a lane adds its constant (dx, dy).
// MODE 0: general (any size), per-lane decode of the pixel index.
// NW = 4: four waves as 2 x 2, each a (BMO/2) x 64 piece of the tile, 64-pixel steps.
// NW = 2: two waves side by side, each BMO x 64 (a wave then reads 12 KB of LDS per 16 MFMAs instead of 16 KB: the
//   transposing reads, not the MFMAs, bound the 2 x 2 form), 32-pixel steps so that twice as many workgroups fit a CU
//   and the waves per SIMD stay the same. Either way a wave stages 16 pixel rows per step.
// NW = 8: TWO groups of four waves in one workgroup (one workgroup per CU, the same two waves per SIMD). A step is 128
//   pixels; group g multiplies pixels [64 g, 64 g + 64) of it into its own accumulators, and at the end group 1 hands its
//   tile to group 0 through LDS. Half as many partial tiles leave the chip for the same occupancy: combining the
//   splits (stores + wgrad_reduce_kernel, or atomics) is what a weight-gradient launch pays per resident wave.
// DEEP (NW = 4 only): 32-pixel steps in a FOUR-stage ring instead of 64-pixel steps in two stages -- the same 64 KB of LDS
//   (two workgroups per CU), but three stages (48 KB) in flight instead of one (32 KB); one barrier per 32 pixels. An
//   experiment that answered "is the 64-pixel step (~2500 cycles against 2 x 512 cycles of MFMA work per SIMD,
//   tools/perf/wgrad_trace.py) waiting for its DMA?" with NO: it is 7-19 % slower on every layer (see launch_wgrad). Off by default.
template <int BMO, int MODE, int NW = 4, bool DEEP = false>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void conv_wgrad_dma_kernel(WgradArgs p, int lgW, int lgH) {
    static_assert(!DEEP || NW == 4, "deep pipeline: four-wave kernel only");
    constexpr bool FAST = MODE == 1, SEMI = MODE == 2;
    constexpr int BNK = 128, BK = DEEP ? 32 : 16 * NW, NT = 64 * NW, LGBK = DEEP ? 5 : (NW == 8 ? 7 : (NW == 4 ? 6 : 5));
    constexpr int NST = DEEP ? 4 : 2;            // ring stages
    constexpr int RPW = BK / NW;                 // pixel rows a wave stages per step (16 | 8)
    constexpr int RSA = BMO * 2, RSB = BNK * 2;
    constexpr int WM = NW >= 4 ? 2 : 1;          // waves along the channel (row) dimension of the tile (of a group)
    constexpr int TM = BMO / (32 * WM), TN = 2;
    constexpr int STAGE = BK * (RSA + RSB);
    constexpr int KS16 = (NW == 8 ? 64 : BK) / 16;   // k16 sub-steps a wave multiplies per step (its group's 64 pixels)
    constexpr int A_ROWS = 1024 / RSA;           // pixel rows per wave-instruction (4 | 8)
    constexpr int A_Q = RPW / A_ROWS;            // A instructions per wave per step (4 | 2; deep: 2 | 1)
    constexpr int B_Q = RPW / 4;                 // im2col instructions per wave per step (4 rows each)
    constexpr int A_CH = RSA / 16;               // 16-byte chunks per A row (16 | 8)

    L2I_TR(0);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    int bid = blockIdx.x;
    if (p.zero_targets) {   // (WgradArgs::zero_targets: fire-and-forget stores, nothing in this kernel reads or writes dW)
        const long long gtid = (long long)blockIdx.x * NT + tid, gthreads = (long long)gridDim.x * NT;
        wg_zero_slice(p.dw, (long long)p.Co * p.ldw, gtid, gthreads);
        wg_zero_slice(p.dw_b, (long long)p.Co * p.ldw, gtid, gthreads);
        if (p.sc_x) {
            wg_zero_slice(p.sc_dw, (long long)p.Co * p.sc_ldw, gtid, gthreads);
            wg_zero_slice(p.sc_dw_b, (long long)p.Co * p.sc_ldw, gtid, gthreads);
        }
    }
    const int split = bid % p.splits; bid /= p.splits;
    const int tile_k = bid % p.tiles_k, tile_co = bid / p.tiles_k;
    // column tiles past tiles_k_main belong to the folded 1x1 shortcut: another source tensor, one (centre) tap, no upsampling
    const bool is_sc = tile_k >= p.tiles_k_main;   // (workgroup-uniform)
    const int co0 = tile_co * BMO, kc0 = (is_sc ? tile_k - p.tiles_k_main : tile_k) * BNK;
    const int pad = p.KH >> 1;
    const int Ci_ = is_sc ? p.sc_Ci : p.Ci, K_ = is_sc ? p.sc_Ci : p.K, up2_ = is_sc ? p.sc_up2 : p.up2;
    const int Hi_ = is_sc ? (p.Ho >> p.sc_up2) : p.Hi, Wi_ = is_sc ? (p.Wo >> p.sc_up2) : p.Wi;
    const bool sc_gen = is_sc && p.sc_up2;   // an upsampled shortcut inside the constant-offset kernel: its tiles decode their pixels per lane
    const int Hd = p.Ho >> p.pool2, Wd = p.Wo >> p.pool2;
    int m_begin, m_end;
    wgrad_range(p, split, m_begin, m_end);

    // per-lane constants: A (dY) chunk
    const int a_row = lane / A_CH;                                     // row within the wave-instruction
    const int a_lchunk = (lane % A_CH) ^ wg_swz<RSA>(a_row);           // logical chunk (rows bases are multiples of 8)
    const int a_chan = co0 + a_lchunk * 8;
    const bool a_on = a_chan < p.Co;
    // B (im2col) chunk: 16 chunks per row, 4 rows per instruction
    const int b_row = lane >> 4;
    const int b_lchunk = (lane & 15) ^ wg_swz<RSB>(b_row);
    const int kc = kc0 + b_lchunk * 8;
    const int tap = is_sc ? 0 : kc / p.Ci, b_ci = kc - tap * Ci_;
    const int b_ky = is_sc ? pad : tap / p.KH, b_kx = is_sc ? pad : tap - b_ky * p.KH;
    const bool b_on = kc < K_;

    // LDS-DMA through buffer descriptors (inline asm, see igemm.h): masked lanes use an out-of-range offset -> zeros.
    // The loop below is written to ISSUE few instructions (a wave issues one per ~4 cycles; the first version spent
    // 16.6 VALU/SALU/LDS/VMEM instructions per MFMA against the ~5 that fit under a 32-cycle MFMA):
    //  * dY rows are contiguous in the pixel index (no pool): the per-lane offset is a constant and the step's base goes
    //    in the instruction's SGPR offset; rows past the end of the tensor are out of the descriptor's range = zeros,
    //    which also zeroes the products of whatever the im2col side fetches there;
    //  * the im2col offset is linear in the pixel index, only the border test needs the pixel's (x, y);
    //  * two steps are unrolled so the LDS stage is an immediate; fragment addresses are loop constants; all 32
    //    transposing reads of a step are issued before its 16 MFMAs.
    const void* xsrc = is_sc ? p.sc_x : p.x;
    const unsigned xsrc_bytes = is_sc ? p.sc_x_bytes : p.x_bytes;
    const u32x4_t rs_dy = make_rsrc(p.dy, p.dy_bytes), rs_x = make_rsrc(xsrc, xsrc_bytes);
    // The range check of a buffer access looks at the LANE offset only (not at the SGPR offset), so fast-path lane
    // offsets must be non-negative and may not rely on the check for the step base: the im2col descriptor starts
    // (Wo + 1) pixels BEFORE x (lanes that would read there are masked by the border test), and steps that cross the
    // end of the tensor take the general path.
    const unsigned x_shift = (unsigned)((p.Wo + 1) * Ci_) * 2u;
    const u32x4_t rs_xs = make_rsrc(reinterpret_cast<const char*>(xsrc) - x_shift, xsrc_bytes + x_shift);
    const unsigned smem_addr = lds_addr_of(smem);
    constexpr unsigned OOB = 0x80000000u;
    unsigned a_base[A_Q];      // fast path: byte offset of this lane's chunk for pixel (prow + a_row), without the step base
    int b_mlane[B_Q];
    unsigned b_base[B_Q];      // fast path: ((ky-1)*Wo + (kx-1) + prow + b_row) * Ci + b_ci, in bytes (may wrap below 0)
#pragma unroll
    for (int q = 0; q < A_Q; ++q)
        a_base[q] = a_on ? (unsigned)((wv * RPW + q * A_ROWS + a_row) * p.Co + a_chan) * 2u : OOB;
#pragma unroll
    for (int q = 0; q < B_Q; ++q) {
        b_mlane[q] = wv * RPW + q * 4 + b_row;
        b_base[q] = (unsigned)(((b_ky - pad) * p.Wo + (b_kx - pad) + b_mlane[q]) * Ci_ + b_ci) * 2u + x_shift;
    }
    const unsigned co2 = (unsigned)p.Co * 2u, ci2 = (unsigned)Ci_ * 2u;
    // SEMI: lane constants -- position (xl, yl) of the lane's pixel within a step, folded into offsets
    unsigned a_semi[A_Q];
    int b_cy[B_Q], b_cx[B_Q];
#pragma unroll
    for (int q = 0; q < A_Q; ++q) {
        const int pl = wv * RPW + q * A_ROWS + a_row;
        const int xl = pl & (p.Wo - 1), yl = pl >> lgW;
        a_semi[q] = a_on ? (unsigned)(((yl >> p.pool2) * Wd + (xl >> p.pool2)) * p.Co + a_chan) * 2u : OOB;
    }
#pragma unroll
    for (int q = 0; q < B_Q; ++q) {
        const int pl = b_mlane[q];
        b_cx[q] = (pl & (p.Wo - 1)) + b_kx - pad;
        b_cy[q] = (pl >> lgW) + b_ky - pad;
    }
    auto issue = [&](int mstep, unsigned stage) {   // stage: LDS byte address of the stage
        // SEMI: scalar image / row / column of the step
        const int sb = mstep >> (lgW + lgH), sy = (mstep >> lgW) & (p.Ho - 1), sx = mstep & (p.Wo - 1);
        const unsigned soff_a = (unsigned)(((sb * Hd + (sy >> p.pool2)) * Wd + (sx >> p.pool2)) * p.Co) * 2u;
        const unsigned soff_b = (unsigned)(sb * Hi_ * Wi_) * ci2;
#pragma unroll
        for (int q = 0; q < A_Q; ++q) {
            const unsigned dst = stage + (unsigned)(wv * RPW + q * A_ROWS) * RSA;
            if constexpr (FAST) {
                L2I_DMA16_S(rs_dy, a_base[q], (unsigned)mstep * co2, dst);
            } else if constexpr (SEMI) {
                L2I_DMA16_S(rs_dy, a_semi[q], soff_a, dst);
            } else {
                const int m = mstep + wv * RPW + q * A_ROWS + a_row;
                const int x = m & (p.Wo - 1), y = (m >> lgW) & (p.Ho - 1), b = m >> (lgW + lgH);
                const unsigned off = (unsigned)(((b * Hd + (y >> p.pool2)) * Wd + (x >> p.pool2)) * p.Co + a_chan) * 2u;
                L2I_DMA16_S(rs_dy, (a_on && m < m_end) ? off : OOB, 0u, dst);   // (select, not a branch: keeps the loop straight-line)
            }
        }
#pragma unroll
        for (int q = 0; q < B_Q; ++q) {
            const unsigned dst = stage + BK * RSA + (unsigned)(wv * RPW + q * 4) * RSB;
            if constexpr (SEMI) {
                const int yy = sy + b_cy[q], xx = sx + b_cx[q];
                const bool in = b_on && (unsigned)yy < (unsigned)p.Ho && (unsigned)xx < (unsigned)p.Wo;
                const unsigned off = (unsigned)(((yy >> up2_) * Wi_ + (xx >> up2_)) * Ci_ + b_ci) * 2u;
                L2I_DMA16_S(rs_x, in ? off : OOB, soff_b, dst);
                continue;
            }
            const int m = mstep + b_mlane[q];
            const int x = m & (p.Wo - 1), y = (m >> lgW) & (p.Ho - 1);
            const int yy = y + b_ky - pad, xx = x + b_kx - pad;
            const bool in = b_on && (unsigned)yy < (unsigned)p.Ho && (unsigned)xx < (unsigned)p.Wo;
            if (FAST && !sc_gen) {
                L2I_DMA16_S(rs_xs, in ? b_base[q] : OOB, (unsigned)mstep * ci2, dst);
            } else {
                const int b = m >> (lgW + lgH);
                const unsigned off = (unsigned)(((b * Hi_ + (yy >> up2_)) * Wi_ + (xx >> up2_)) * Ci_ + b_ci) * 2u;
                L2I_DMA16_S(rs_x, (in && m < m_end) ? off : OOB, 0u, dst);
            }
        }
    };

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int grp = NW == 8 ? (wave >> 2) : 0, wq = NW == 8 ? (wave & 3) : wave;   // pixel group, position within the group
    const int wrow = NW >= 4 ? (wq >> 1) * (BMO / 2) : 0, wcol = (NW >= 4 ? (wq & 1) : wq) * 64;
    // fragment read addresses (stage-relative): piece (row 8h + (t>>2), 4-channel block) of the k16 sub-step 0, r = 0;
    // sub-step kk adds 16 rows and r adds 4 rows -- neither changes the row's swizzle, so they are immediates
    unsigned fa_addr[TM], fb_addr[TN];
    {
        const int t = lane & 15, cb = ((lane >> 4) & 1) * 16, h = lane >> 5;
        const int prow0 = 64 * grp + 8 * h + (t >> 2);   // (64 rows: the swizzle of a row is unchanged)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int ch = wrow + i * 32 + cb + (t & 3) * 4;
            fa_addr[i] = (unsigned)(prow0 * RSA + (((ch >> 3) ^ wg_swz<RSA>(prow0)) << 4) + ((ch >> 2) & 1) * 8);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ch = wcol + j * 32 + cb + (t & 3) * 4;
            fb_addr[j] = (unsigned)(BK * RSA + prow0 * RSB + (((ch >> 3) ^ wg_swz<RSB>(prow0)) << 4) + ((ch >> 2) & 1) * 8);
        }
    }
    // bias gradient: the tiles_k workgroups that share a (channel tile, split) stage the same dY steps; workgroup tile_k
    // sums the steps with index % tiles_k == tile_k (every pixel once, the extra work spread evenly -- giving it all to
    // the tile_k == 0 workgroups made them the tail of the launch: measured +25 % on the whole weight-gradient time).
    // Per such step every thread adds up one 8-channel chunk of the dY stage over 64 / (256 / A_CH) pixel rows; the
    // 256 / A_CH partial sums per channel are combined at the end.
    // (Shared by only min(tiles_k, 4) of them, and combined across the four waves in LDS: every atomic on a bias
    // address costs ~0.09 us of serialised tail, and 500+ workgroups adding to the same 128 addresses tripled the launch.)
    const int nb_bias = p.nb_bias;   // (a power of two: 4 / 2 / 1 by the number of column tiles; 1 on unsplit launches, see launch_wgrad)
    const bool do_bias = (p.dbias != nullptr || p.dbias2 != nullptr) && tile_k < nb_bias;
    float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    constexpr int BS_PG = NT / A_CH, BS_PP = BK / BS_PG;   // pixel groups, pixels per thread per step
#define WG_BIAS(STG, MS)                                                                                              \
    if (do_bias && ((((MS) - m_begin) >> LGBK) & (nb_bias - 1)) == tile_k) {                                               \
        const int c_ = tid % A_CH, pg_ = tid / A_CH;                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < BS_PP; ++i_) {                                                        \
            const int pr_ = pg_ + BS_PG * i_;                                                                         \
            const uint4 v_ = *reinterpret_cast<const uint4*>(smem + (STG) * STAGE + pr_ * RSA + ((c_ ^ wg_swz<RSA>(pr_)) << 4)); \
            bs[0] += __uint_as_float(v_.x << 16); bs[1] += __uint_as_float(v_.x & 0xffff0000u);                       \
            bs[2] += __uint_as_float(v_.y << 16); bs[3] += __uint_as_float(v_.y & 0xffff0000u);                       \
            bs[4] += __uint_as_float(v_.z << 16); bs[5] += __uint_as_float(v_.z & 0xffff0000u);                       \
            bs[6] += __uint_as_float(v_.w << 16); bs[7] += __uint_as_float(v_.w & 0xffff0000u);                       \
        }                                                                                                             \
    }
#define WG_TR(ADDR) __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)((__attribute__((address_space(3))) char*)smem + (ADDR)))
#define WG_STEP(STG)                                                                                                  \
    {                                                                                                                 \
        s16x4_t ra[KS16][TM][2], rb[KS16][TN][2];                                                               \
        _Pragma("unroll") for (int kk = 0; kk < KS16; ++kk) {                                                            \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int r = 0; r < 2; ++r)              \
                ra[kk][i][r] = WG_TR(fa_addr[i] + (STG) * STAGE + (kk * 16 + 4 * r) * RSA);                           \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) _Pragma("unroll") for (int r = 0; r < 2; ++r)              \
                rb[kk][j][r] = WG_TR(fb_addr[j] + (STG) * STAGE + (kk * 16 + 4 * r) * RSB);                           \
        }                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
        _Pragma("unroll") for (int kk = 0; kk < KS16; ++kk)                                                        \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) {           \
                bf16x8_t fa, fb;                                                                                      \
                fa[0] = ra[kk][i][0][0]; fa[1] = ra[kk][i][0][1]; fa[2] = ra[kk][i][0][2]; fa[3] = ra[kk][i][0][3];   \
                fa[4] = ra[kk][i][1][0]; fa[5] = ra[kk][i][1][1]; fa[6] = ra[kk][i][1][2]; fa[7] = ra[kk][i][1][3];   \
                fb[0] = rb[kk][j][0][0]; fb[1] = rb[kk][j][0][1]; fb[2] = rb[kk][j][0][2]; fb[3] = rb[kk][j][0][3];   \
                fb[4] = rb[kk][j][1][0]; fb[5] = rb[kk][j][1][1]; fb[6] = rb[kk][j][1][2]; fb[7] = rb[kk][j][1][3];   \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                                  \
                    __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa),                               \
                    __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb), acc[i][j], 0, 0, 0);          \
            }                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }
    L2I_TR(1);
    if constexpr (DEEP) {
        if (m_begin < m_end) {
            constexpr int IPS = A_Q + B_Q;   // DMA instructions a wave issues per step (vmcnt counts them in order)
            const int nsteps = (m_end - m_begin + BK - 1) >> LGBK;
#pragma unroll
            for (int s_ = 0; s_ < NST - 1; ++s_)
                if (s_ < nsteps) issue(m_begin + s_ * BK, smem_addr + (unsigned)(s_ * STAGE));
            for (int it = 0; it < nsteps; ++it) {
                // step `it` must have landed; the (up to two) steps issued after it may stay in flight
                const int rem = nsteps - 1 - it;
                if (rem >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPS) : "memory");
                else if (rem == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPS) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();   // everyone's pieces landed, and stage (it - 1) % NST is free
                asm volatile("" ::: "memory");
                if (it + NST - 1 < nsteps) issue(m_begin + (it + NST - 1) * BK, smem_addr + (unsigned)(((it + NST - 1) & (NST - 1)) * STAGE));
                const int stg = it & (NST - 1);
                const int ms = m_begin + it * BK;
                WG_BIAS(stg, ms)
                WG_STEP(stg)
            }
        }
    } else
    if (m_begin < m_end) {
        // (whole pairs of steps in the loop, an odd last step after it: a conditional second step inside the loop makes
        // the compiler merge two accumulator register sets with 64 AGPR<->VGPR copies per iteration)
        issue(m_begin, smem_addr);
        int ms = m_begin;
        for (; ms + 2 * BK <= m_end; ms += 2 * BK) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the tile landed ...
            __builtin_amdgcn_s_barrier();                       // ... everyone's did, and the other stage is free
            asm volatile("" ::: "memory");
            issue(ms + BK, smem_addr + STAGE);                  // in flight under the MFMAs below
            WG_BIAS(0, ms)
            WG_STEP(0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (ms + 2 * BK < m_end) issue(ms + 2 * BK, smem_addr);
            WG_BIAS(1, ms + BK)
            WG_STEP(1)
        }
        if (ms < m_end) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            WG_BIAS(0, ms)
            WG_STEP(0)
        }
    }
#undef WG_STEP
#undef WG_TR
#undef WG_BIAS
    L2I_TR(2);
    if (do_bias) {   // lanes that share a chunk within a wave are A_CH apart; waves combine through LDS: one atomic per channel
        float* red = reinterpret_cast<float*>(smem);   // [NW waves][BMO]
        __syncthreads();                                // every wave is done with the stages
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = bs[e];
            for (int o = A_CH; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
            if (lane < A_CH) red[wave * BMO + lane * 8 + e] = v;
        }
        __syncthreads();
        if (tid < BMO) {
            float v = 0.f;
#pragma unroll
            for (int w_ = 0; w_ < NW; ++w_) v += red[w_ * BMO + tid];
            if (p.bpart) {   // this workgroup's share (its steps of this split): one row of the partial matrix, zeros included
                p.bpart[(size_t)(split * nb_bias + tile_k) * p.bpart_ld + co0 + tid] = p.alpha * v;
            } else if (co0 + tid < p.Co && v != 0.f) {
                if (p.dbias) atomicAdd(p.dbias + co0 + tid, p.alpha * v);
                if (p.dbias2) atomicAdd(p.dbias2 + co0 + tid, p.alpha * v);
            }
        }
    }
    if (NW == 8) {   // group 1 hands its tile to group 0: [wave of the group][register][lane] f32, 64 KB
        float* xch = reinterpret_cast<float*>(smem);
        __syncthreads();   // every wave is done with the stages (and with the bias scratch)
        if (grp == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) xch[((wq * TM * TN + i * TN + j) * 16 + e) * 64 + lane] = acc[i][j][e];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] += xch[((wq * TM * TN + i * TN + j) * 16 + e) * 64 + lane];
    }

#ifdef L2I_ABLATIONS
    if (p.no_epi) {   // (keeps the accumulators live)
        float s_ = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) s_ += acc[i][j][e];
        if (s_ == 1.2345e30f) p.dw[0] = s_;
        return;
    }
#endif
    const int c = lane & 31, h = lane >> 5;
    if (p.part) {   // plain stores of the whole tile (dead splits store their zeros): measured 5.8 TB/s against 1.3 TB/s for the
                    // same bytes as f32 atomics (tools/perf/t_atomic.hip). The scratch tile is kept in REGISTER order --
                    // [wave][i][j][register group g][lane] float4 -- so that one store instruction writes 64 x 16 contiguous
                    // bytes (round 2 stored row-major: 64 four-byte store instructions per wave, a quarter of the bytes per
                    // instruction); wgrad_reduce_kernel reads it in the same order and un-permutes on its (one per tile, not
                    // per split) write to dw.
        float4* t = reinterpret_cast<float4*>(p.part + ((size_t)(tile_co * p.tiles_k + tile_k) * p.splits + split) * (size_t)(BMO * BNK)) +
                    wq * (TM * TN * 4 * 64) + lane;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    if (p.fuse_cnt) {
                        f32x4_t v_;
                        v_[0] = acc[i][j][4 * g]; v_[1] = acc[i][j][4 * g + 1]; v_[2] = acc[i][j][4 * g + 2]; v_[3] = acc[i][j][4 * g + 3];
                        wg_store_sc1(t + ((i * TN + j) * 4 + g) * 64, v_);
                    } else {
                        t[((i * TN + j) * 4 + g) * 64] = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                    }
                }
        if (p.fuse_cnt) {
            // last arriver: release this workgroup's stores at device scope, count, and -- if every other split of this tile
            // (and accumulator) has counted already -- acquire and reduce them all
            const int hs = p.dw_b ? p.splits >> 1 : p.splits;
            const bool sec = wgrad_second(p, split);
            const int tile = tile_co * p.tiles_k + tile_k;
            __shared__ int s_last;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this lane's sc1 stores are visible device-wide
            __syncthreads();
            if (tid == 0) {
                unsigned* c = p.fuse_cnt + (p.dw_b ? 2 * tile + (sec ? 1 : 0) : tile);
                const unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last = old == (unsigned)(hs - 1);
                if (s_last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch that gets this slot
            }
            __syncthreads();
            if (s_last) {
                float* dw_ = is_sc ? (sec ? p.sc_dw_b : p.sc_dw) : (sec ? p.dw_b : p.dw);
                const int ldw_ = is_sc ? p.sc_ldw : p.ldw;
                const size_t tsz4 = (size_t)BMO * 32;
                const float4* src0 = reinterpret_cast<const float4*>(p.part) + ((size_t)tile * p.splits + (sec ? hs : 0)) * tsz4;
                constexpr int TMr = BMO / 64;
                for (int f = tid; f < BMO * 32; f += NT) {
                    const int ln = f & 63, g_ = (f >> 6) & 3, j_ = (f >> 8) & 1;
                    const int wi = f >> 9, i_ = wi % TMr, wq_ = wi / TMr;
                    const int wrow_ = (wq_ >> 1) * (BMO / 2), wcol_ = (wq_ & 1) * 64;
                    const int row0 = co0 + wrow_ + i_ * 32 + 8 * g_ + 4 * (ln >> 5), col = kc0 + wcol_ + j_ * 32 + (ln & 31);
                    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4* src = src0 + f;
                    int s_ = 0;
                    for (; s_ + 4 <= hs; s_ += 4) {
                        f32x4_t v0 = wg_load_sc1(src + (size_t)s_ * tsz4), v1 = wg_load_sc1(src + (size_t)(s_ + 1) * tsz4);
                        f32x4_t v2 = wg_load_sc1(src + (size_t)(s_ + 2) * tsz4), v3 = wg_load_sc1(src + (size_t)(s_ + 3) * tsz4);
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)::"memory");   // (the asm loads are invisible to the compiler's waitcnt insertion)
                        a.x += (v0[0] + v1[0]) + (v2[0] + v3[0]); a.y += (v0[1] + v1[1]) + (v2[1] + v3[1]);
                        a.z += (v0[2] + v1[2]) + (v2[2] + v3[2]); a.w += (v0[3] + v1[3]) + (v2[3] + v3[3]);
                    }
                    for (; s_ < hs; ++s_) {
                        f32x4_t v0 = wg_load_sc1(src + (size_t)s_ * tsz4);
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v0)::"memory");
                        a.x += v0[0]; a.y += v0[1]; a.z += v0[2]; a.w += v0[3];
                    }
                    if (col >= K_) continue;
                    float* d = dw_ + (size_t)row0 * ldw_ + col;
                    const bool ow = p.overwrite != 0;   // (overwrite: the slice's previous contents do not count)
                    if (row0 < p.Co) d[0] = (ow ? 0.f : d[0]) + p.alpha * a.x;
                    if (row0 + 1 < p.Co) d[ldw_] = (ow ? 0.f : d[ldw_]) + p.alpha * a.y;
                    if (row0 + 2 < p.Co) d[2 * (size_t)ldw_] = (ow ? 0.f : d[2 * (size_t)ldw_]) + p.alpha * a.z;
                    if (row0 + 3 < p.Co) d[3 * (size_t)ldw_] = (ow ? 0.f : d[3 * (size_t)ldw_]) + p.alpha * a.w;
                }
            }
        }
        L2I_TR(3);
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = kc0 + wcol + j * 32 + c;
            if (col >= K_) continue;
            const bool sec_ = wgrad_second(p, split);
            float* dw_ = is_sc ? (sec_ ? p.sc_dw_b : p.sc_dw) : (sec_ ? p.dw_b : p.dw);
            const int ldw_ = is_sc ? p.sc_ldw : p.ldw;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = co0 + wrow + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (row >= p.Co) continue;
                if (p.overwrite && p.splits == 1) dw_[(size_t)row * ldw_ + col] = p.alpha * acc[i][j][e];   // the tile's only writer: a store (128-byte runs per half-wave)
                else atomicAdd(dw_ + (size_t)row * ldw_ + col, p.alpha * acc[i][j][e]);
            }
        }
}

// dw[row][col] += alpha * sum over splits of the partial tiles stored by conv_wgrad_dma_kernel (WgradArgs::part).
// One thread per float4 of a tile in the REGISTER order the main kernel stores it in (coalesced 16-byte reads over the
// splits): float4 f = ((wave * TM + i) * 2 + j) * 4 + g) * 64 + lane holds accumulator registers 4g .. 4g+3 of lane
// (c = lane & 31, h = lane >> 5) = rows wrow + 32 i + 8 g + 4 h + {0..3}, column wcol + 32 j + c of the tile. A layer's dw
// slice is written by one launch at a time (same stream), so the read-modify-write needs no atomics.
// nw2: the two-wave kernel's geometry (waves side by side, TM = BMO / 32), else four waves as 2 x 2 (TM = BMO / 64).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int BMO, int tiles_k,
                                                           int ntiles, int splits, int Co, int K, int ldw, float alpha, int nw2, int sper,
                                                           int tiles_k_main, float* __restrict__ dw2, int K2, int ldw2,
                                                           float* __restrict__ dw_b, float* __restrict__ dw2_b, int overwrite, float4* __restrict__ part_out,
                                                           RowsFoldArgs bias_fold, int nbx_main, int sl) {
    // workgroups [nbx_main, gridDim.x) of the first (group, pass) plane: the ordered sum of the launch's bias-gradient rows (rows_fold2_body) --
    // carried here instead of one more launch behind every weight-gradient launch (round 6)
    if ((int)blockIdx.x >= nbx_main) {
        __shared__ float4 fold_red[256];
        if (blockIdx.y == 0 && blockIdx.z == 0 && bias_fold.src) rows_fold2_body(bias_fold, (int)blockIdx.x - nbx_main, 0, fold_red);
        return;
    }
    // Layers with few tiles and many splits (the 64-channel layers at 128 x 128: 5 tiles x 153 splits) would run on 40 workgroups, each thread
    // walking 153 partial tiles one dependent load after the other (18 us of a 69-us weight gradient). Round 6: `sl` SPLIT LANES per float4 slot
    // (a power of two <= 16): a workgroup owns 256 / sl slots, lane q of a slot adds splits q, q + sl, ... and the lanes meet in a fixed-order
    // tree in LDS -- one launch, one writer per value, no atomics (rounds 2-5: split GROUPS on blockIdx.y combined by float atomics).
    __shared__ float4 lane_red[256];
    const int nslots = 256 / sl, slot = threadIdx.x % nslots, q = threadIdx.x / nslots;
    const long long gid = (long long)blockIdx.x * nslots + slot;
    const int per_tile = BMO * 32;                      // float4 per tile
    const int tile = (int)(gid / per_tile), f = (int)(gid - (long long)tile * per_tile);
    const bool live = tile < ntiles;
    int tile_k = tile % tiles_k;
    const int tile_co = tile / tiles_k;
    // dual launch (gridDim.z == 2): the second half of the splits belongs to the second pass's accumulators
    const int hs = splits / (int)gridDim.z, s_off = (int)blockIdx.z * hs;
    if (blockIdx.z) { dw = dw_b; dw2 = dw2_b; }
    if (tile_k >= tiles_k_main) {   // column tiles of the folded shortcut: their own gradient buffer
        tile_k -= tiles_k_main;
        dw = dw2; K = K2; ldw = ldw2;
    }
    const int lane = f & 63, g = (f >> 6) & 3, j = (f >> 8) & 1;
    const int TM = nw2 ? BMO / 32 : BMO / 64;
    const int wi = f >> 9, i = wi % TM, wq = wi / TM;
    const int wrow = nw2 ? 0 : (wq >> 1) * (BMO / 2), wcol = (nw2 ? wq : (wq & 1)) * 64;
    const int row0 = tile_co * BMO + wrow + i * 32 + 8 * g + 4 * (lane >> 5), col = tile_k * 128 + wcol + j * 32 + (lane & 31);
    const size_t tsz4 = (size_t)BMO * 32;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        const float4* src = reinterpret_cast<const float4*>(part) + (size_t)tile * splits * tsz4 + f;
        int s = s_off + blockIdx.y * sper + q;
        const int s_end = min(s_off + hs, s_off + (int)blockIdx.y * sper + sper);
        for (; s + 3 * sl < s_end; s += 4 * sl) {
            const float4 v0 = src[(size_t)s * tsz4], v1 = src[(size_t)(s + sl) * tsz4];
            const float4 v2 = src[(size_t)(s + 2 * sl) * tsz4], v3 = src[(size_t)(s + 3 * sl) * tsz4];
            a.x += (v0.x + v1.x) + (v2.x + v3.x); a.y += (v0.y + v1.y) + (v2.y + v3.y);
            a.z += (v0.z + v1.z) + (v2.z + v3.z); a.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; s < s_end; s += sl) {
            const float4 v0 = src[(size_t)s * tsz4];
            a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
        }
    }
    if (sl > 1) {
        lane_red[threadIdx.x] = a;
        __syncthreads();
        for (int st = sl >> 1; st >= 1; st >>= 1) {
            if (q < st) {
                const float4 u = lane_red[threadIdx.x], v = lane_red[threadIdx.x + st * nslots];
                lane_red[threadIdx.x] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
            }
            __syncthreads();
        }
        a = lane_red[threadIdx.x];
    }
    if (!live || q != 0) return;
    if (part_out) {   // first stage of a two-stage reduction (split groups): the group's sum, unscaled, in the partial tiles' own layout
        part_out[((size_t)tile * (gridDim.y * gridDim.z) + blockIdx.z * gridDim.y + blockIdx.y) * tsz4 + f] = a;   // [tile][pass][group]: the second
        return;                                                                                                     // stage reads it as `splits` = groups
    }
    if (col >= K) return;
    float* d = dw + (size_t)row0 * ldw + col;   // (a half-wave writes 32 consecutive columns of one row per statement)
    if (gridDim.y > 1) {
        if (row0 < Co) atomicAdd(d, alpha * a.x);
        if (row0 + 1 < Co) atomicAdd(d + ldw, alpha * a.y);
        if (row0 + 2 < Co) atomicAdd(d + 2 * (size_t)ldw, alpha * a.z);
        if (row0 + 3 < Co) atomicAdd(d + 3 * (size_t)ldw, alpha * a.w);
        return;
    }
    if (overwrite) {   // (WgradArgs::overwrite: nothing else has written this slice since it was zeroed)
        if (row0 < Co) d[0] = alpha * a.x;
        if (row0 + 1 < Co) d[ldw] = alpha * a.y;
        if (row0 + 2 < Co) d[2 * (size_t)ldw] = alpha * a.z;
        if (row0 + 3 < Co) d[3 * (size_t)ldw] = alpha * a.w;
        return;
    }
    if (row0 < Co) d[0] += alpha * a.x;
    if (row0 + 1 < Co) d[ldw] += alpha * a.y;
    if (row0 + 2 < Co) d[2 * (size_t)ldw] += alpha * a.z;
    if (row0 + 3 < Co) d[3 * (size_t)ldw] += alpha * a.w;
}

static int g_wgrad_blocks = 0;   // tuning hook l2i_set_wgrad_blocks: workgroups per wave of the grid (0 = from the tile's occupancy)
static int g_wgrad_force64 = 0;   // tuning: 64-row tiles for every layer (n = -64), back to the default (n = -128)
static int g_wgrad_nw2 = -1;      // tuning: two-wave 128-row kernel on (n = -2) / off (n = -4) / by the heuristic (n = -3)
extern "C" int l2i_set_wgrad_blocks(int n) {
    if (n == -64) { g_wgrad_force64 = 1; return L2I_OK; }
    if (n == -128) { g_wgrad_force64 = 0; return L2I_OK; }
    if (n == -2) { g_wgrad_nw2 = 1; return L2I_OK; }
    if (n == -4) { g_wgrad_nw2 = 0; return L2I_OK; }
    if (n == -3) { g_wgrad_nw2 = -1; return L2I_OK; }
    g_wgrad_blocks = n > 0 ? n : 0;
    return L2I_OK;
}

template <typename T>
static int launch_wgrad(WgradArgs& a, hipStream_t stream, float* scratch, long long scratch_floats) {
    constexpr int BK = Mma<T>::BK;
    constexpr int EPG = OpT<T>::EPG;
    if (a.KH != 1 && a.KH != 3) return L2I_ERR_ARG;
    if (a.Ci % EPG || a.Co % EPG) return L2I_ERR_ARG;
    if (a.up2 && (a.Ho != 2 * a.Hi || a.Wo != 2 * a.Wi)) return L2I_ERR_ARG;
    if (!a.up2 && (a.Ho != a.Hi || a.Wo != a.Wi)) return L2I_ERR_ARG;
    if (a.pool2 && ((a.Ho & 1) || (a.Wo & 1))) return L2I_ERR_ARG;
    a.K = a.KH * a.KH * a.Ci;
    if (a.ldw < a.K) return L2I_ERR_ARG;
    a.M = a.B * a.Ho * a.Wo;
    // few pixels, many tiles (the 1024-channel layers on 4x4 / 8x8 maps: <= 2048 pixels, >= 256 tiles of 128 x 128): the loop is 8-32 steps
    // long and a workgroup's life is its prologue and its 64 KB store -- 64-row tiles (three workgroups per CU, twice the tiles) run
    // these 8-12 % faster (tools/perf: 27.1 -> 23.9, 67.8 -> 62.2, 62.6 -> 57.6 us), while 16 x 16 maps already lose with them
    static const int fewpx_env = getenv("L2I_WGRAD_FEWPX") ? atoi(getenv("L2I_WGRAD_FEWPX")) : 1;
    const bool few_px = fewpx_env && sizeof(T) == 2 && a.M <= 2048 && (long long)((a.Co + 127) / 128) * ((a.K + 127) / 128) >= 256;
    const int BMO = (a.Co <= 64 || g_wgrad_force64 || few_px) ? 64 : 128;
    a.tiles_co = (a.Co + BMO - 1) / BMO;
    a.tiles_k = a.tiles_k_main = (a.K + 127) / 128;
    const bool pow2 = !(a.Ho & (a.Ho - 1)) && !(a.Wo & (a.Wo - 1));
    if (a.sc_x) {
        // The shortcut's columns ride on this launch when the LDS-DMA kernel with four waves runs in one of its scalar-step modes
        // (bf16, whole 64-pixel steps, power-of-two maps); otherwise it is its own 1x1 launch, exactly as the caller would have issued it.
        static const int sc_env = getenv("L2I_SC_WGRAD") ? atoi(getenv("L2I_SC_WGRAD")) : 1;
        const bool whole64 = (a.B * a.Ho * a.Wo) % 64 == 0 && ((!a.pool2 && !a.up2) || (a.Ho * a.Wo) % 64 == 0);
        const bool can = sc_env && sizeof(T) == 2 && pow2 && whole64 && a.KH == 3 && a.sc_Ci % EPG == 0 && g_wgrad_nw2 != 1 &&
                         (!a.sc_up2 || (!(a.Ho & 1) && !(a.Wo & 1))) &&
                         !(getenv("L2I_WGRAD_NW8") && atoi(getenv("L2I_WGRAD_NW8"))) && !(getenv("L2I_WGRAD_DEEP") && atoi(getenv("L2I_WGRAD_DEEP")));
        if (!can) {
            WgradArgs s = a;
            s.x = a.sc_x; s.dw = a.sc_dw; s.Ci = a.sc_Ci; s.Hi = a.Ho >> a.sc_up2; s.Wi = a.Wo >> a.sc_up2; s.KH = 1; s.up2 = a.sc_up2; s.ldw = a.sc_ldw;
            s.dbias = a.dbias2; s.dbias2 = nullptr; s.sc_x = nullptr; s.sc_dw = nullptr;
            s.dw_b = a.sc_dw_b; s.sc_dw_b = nullptr;
            const int rc = launch_wgrad<T>(s, stream, scratch, scratch_floats);
            if (rc != L2I_OK) return rc;
            a.sc_x = nullptr; a.sc_dw = nullptr; a.dbias2 = nullptr; a.sc_dw_b = nullptr;
        } else {
            a.tiles_k += (a.sc_Ci + 127) / 128;
            const size_t sb = (size_t)a.B * (a.Ho >> a.sc_up2) * (a.Wo >> a.sc_up2) * a.sc_Ci * sizeof(T);
            if (sb >= 0x80000000ull) return L2I_ERR_ARG;
            a.sc_x_bytes = (unsigned)sb;
        }
    }
    const int tiles = a.tiles_co * a.tiles_k;
    // eight-wave / two-group kernel (128-pixel steps): bf16, 128-row tiles, whole steps, and -- with pool / upsample -- steps
    // that stay inside one image
    static const int nw8_env = getenv("L2I_WGRAD_NW8") ? atoi(getenv("L2I_WGRAD_NW8")) : 0;   // measured: 25.04 vs 24.86 ms per iteration with it ON (the reduce
    // kernels gain 0.15 ms, the eight-wave main loops lose 0.55 ms): a tuning option, off by default
    const bool nw8 = nw8_env && g_wgrad_nw2 != 1 && BMO == 128 && sizeof(T) == 2 && pow2 && a.M % 128 == 0 &&
                     ((!a.pool2 && !a.up2) || (a.Ho * a.Wo) % 128 == 0) && (!a.nimg || (a.Ho * a.Wo) % 128 == 0);   // (live pixels: whole steps)
    // deep pipeline (32-pixel steps, four-stage ring; conv_wgrad_dma_kernel<.., DEEP>): bf16 DMA kernel, four waves. Measured SLOWER
    // (tools/perf/wgrad_tune.py, MI355X: 32x32x512->512 868 -> 707 TFLOP/s, 256x8x8x512->512 838 -> 725, every shape loses
    // 7-19 %): the main loop is not waiting for its DMA -- halving the MFMA work between two barriers costs more than the two
    // extra stages in flight bring. Kept as a tuning option: L2I_WGRAD_DEEP=1.
    static const int deep_env = getenv("L2I_WGRAD_DEEP") ? atoi(getenv("L2I_WGRAD_DEEP")) : 0;
    const bool nw2_ = BMO == 128 && sizeof(T) == 2 && pow2 && g_wgrad_nw2 == 1;
    const bool deep = deep_env && sizeof(T) == 2 && pow2 && !nw8 && !nw2_;
    const int SBK = nw8 ? 128 : (deep ? 32 : BK);   // pixels per step of the kernel that will run
    a.lgbk = nw8 ? 7 : (deep ? 5 : 6);
    const int steps = (a.M + SBK - 1) / SBK;
    const int steps64 = nw8 ? steps * 2 : (a.M + 63) / 64;   // (the split heuristics count 64-pixel units whatever the kernel's step)
    // Split the pixel (reduction) dimension so that the grid is ONE full wave of co-resident workgroups (2 per CU for
    // the 128-row tile, 3 for the 64-row one: LDS-limited) -- every extra split costs Co*K atomics, and a grid of 1.1-1.9
    // waves leaves half the chip idle in its second round. Tile counts too large for that go to >= 3 waves instead.
    const bool nw2 = BMO == 128 && sizeof(T) == 2 && pow2 && (g_wgrad_nw2 < 0 ? false : g_wgrad_nw2 == 1);
    static const int cap_env = getenv("L2I_WGRAD_CAP") ? atoi(getenv("L2I_WGRAD_CAP")) : 0;   // tuning: workgroups per round, 128-row tiles
    const int cap = g_wgrad_blocks > 0 ? g_wgrad_blocks : (BMO == 64 ? 768 : (nw2 ? 1024 : (nw8 ? 256 : (cap_env > 0 ? cap_env : 512))));
    int splits = cap / tiles;
    if (splits < 1 || (long long)splits * tiles * 5 < (long long)cap * 4) {
        splits = (3 * cap + tiles - 1) / tiles;
        if (splits > steps64 / 32) splits = steps64 / 32;   // short reductions: extra splits are all prologue + atomics
    }
    if (splits > steps64 / 4) splits = steps64 / 4;
    if (splits < 1) splits = 1;
    int per = (steps + splits - 1) / splits;
    a.Mper = per * SBK;
    a.splits = (a.M + a.Mper - 1) / a.Mper;
    const bool dual = a.dw_b != nullptr;
    if (dual) {   // the same number of workgroups, half of them per pass; every split stays inside its half (WgradArgs::dw_b)
        const int Mh = a.M / 2;
        if ((a.B & 1) || Mh % SBK || (a.sc_x && !a.sc_dw_b)) return L2I_ERR_ARG;
        int hs = (a.splits + 1) / 2;
        const int steps_h = Mh / SBK;
        per = (steps_h + hs - 1) / hs;
        a.Mper = per * SBK;
        hs = (Mh + a.Mper - 1) / a.Mper;
        a.splits = 2 * hs;
        a.M = Mh;   // (per half from here on: wgrad_range)
    }
    const int nblk = tiles * a.splits;
    {
        const size_t xb = (size_t)a.B * a.Hi * a.Wi * a.Ci * sizeof(T);
        const size_t yb = (size_t)a.B * (a.Ho >> a.pool2) * (a.Wo >> a.pool2) * a.Co * sizeof(T);
        if (xb >= 0x80000000ull || yb >= 0x80000000ull) return L2I_ERR_ARG;   // 32-bit buffer offsets
        a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)yb;
    }
    a.part = nullptr;
    // overwrite + a path that still ADDS with atomics (several splits without scratch, split groups in the reduce, the f32 kernel's
    // splits): the promise is "nothing else writes these slices", so the library may clear them itself before it accumulates
    auto clear_targets = [&]() -> int {
        if (!a.overwrite) return L2I_OK;
        float* t[4] = {a.dw, a.dw_b, a.sc_x ? a.sc_dw : nullptr, a.sc_x ? a.sc_dw_b : nullptr};
        const size_t n[4] = {(size_t)a.Co * a.ldw, (size_t)a.Co * a.ldw, (size_t)a.Co * a.sc_ldw, (size_t)a.Co * a.sc_ldw};
        for (int i = 0; i < 4; ++i)
            if (t[i] && l2i_zero_async(t[i], sizeof(float) * n[i], stream) != hipSuccess) return L2I_ERR_LAUNCH;
        return L2I_OK;
    };
    if (sizeof(T) == 2 && pow2) {  // bf16: LDS-DMA + transposing-read kernel
        static const int use_part = getenv("L2I_WGRAD_PART") ? atoi(getenv("L2I_WGRAD_PART")) : 1;   // (0: atomics, A/B)
        if (use_part && scratch && (a.splits > 1 || dual) && (long long)nblk * BMO * 128 <= scratch_floats) a.part = scratch;
        // split groups of the reduce kernel (computed here: the groups combine with atomics, which needs cleared slices under `overwrite`)
        int sg = 1, sper = a.splits, sl = 1;   // (sg / sper: the split GROUPS of rounds 2-5, kept in the kernel, no longer planned: sl split lanes instead)
        if (a.part) {
            const unsigned nbx_ = (unsigned)(((long long)tiles * BMO * 32 + 255) / 256);
            const int hsp_ = dual ? a.splits / 2 : a.splits;
            sper = hsp_;
            if (nbx_ * (dual ? 2u : 1u) < 512 && hsp_ > 8) {
                int want = (int)((512 + nbx_ - 1) / nbx_);
                if (want > (hsp_ + 7) / 8) want = (hsp_ + 7) / 8;
                while (sl < want && sl < 16) sl <<= 1;
            }
        }
        if (!a.part && a.splits > 1 && clear_targets() != L2I_OK) return L2I_ERR_LAUNCH;   // (atomics straight from the main kernel)
        // fused reduction (last arriver per tile): the plain four-wave kernel with <= 16 splits per accumulator; its counters are a
        // slice of g_wgrad_cnt handed out round robin (launches that may be in flight together never share a slot: 2^20 slots)
        static const int fuse_env = getenv("L2I_WGRAD_FUSE") ? atoi(getenv("L2I_WGRAD_FUSE")) : 0;
        a.fuse_cnt = nullptr;
        if (fuse_env && a.part && !deep && !nw8 && !nw2 && (dual ? a.splits / 2 : a.splits) <= 16) {
            // (experiment, measured slower, off by default -- DESIGN "dated experiments". The hand-out is serialised across host
            //  threads; slots baked into a captured graph are NOT reserved against later eager launches after the ring wraps, so the
            //  switch must not be combined with graph replay next to eager launches on another stream)
            static unsigned* cnt_base = nullptr;
            static unsigned cnt_next = 0;
            static std::mutex cnt_mu;
            std::lock_guard<std::mutex> lock(cnt_mu);
            if (!cnt_base && hipGetSymbolAddress((void**)&cnt_base, HIP_SYMBOL(g_wgrad_cnt)) != hipSuccess) return L2I_ERR_LAUNCH;
            const unsigned need = (unsigned)tiles * (dual ? 2u : 1u);
            if (cnt_next + need > L2I_WGRAD_CNT) cnt_next = 0;
            a.fuse_cnt = cnt_base + cnt_next;
            cnt_next += need;
        }
        // Round 6: the split groups of the reduce kernel combine through a SECOND region of stored tiles ([tile][pass][group], the main kernel's
        // layout) and a second reduce launch over the groups, and the bias-gradient shares are stored rows that rows_fold adds in order -- both
        // where the caller's scratch has room behind the partial tiles; otherwise the atomics of rounds 2-5.
        const int nz_ = dual ? 2 : 1;
        long long used = a.part ? (long long)nblk * BMO * 128 : 0;
        float* part2 = nullptr;
        if (a.part && sg > 1 && !a.fuse_cnt) {
            const long long need2 = (long long)tiles * sg * nz_ * BMO * 128;
            if (used + need2 <= scratch_floats) { part2 = scratch + used; used += need2; }
        }
        a.bpart = nullptr; a.bpart_ld = 0; a.part_rm = nullptr;
        // an unsplit single-pass launch (the 1024-channel layers on 4- / 8-pixel maps: more tiles than workgroup slots, few pixels): ONE column
        // tile sums a channel tile's bias and adds it with one atomic per channel -- a single contribution per address and launch is already
        // order-independent, and there is no reduce launch for a fold to ride on
        const bool lone = a.splits == 1 && !dual;
        const int nb_bias = lone ? 1 : (a.tiles_k_main >= 4 ? 4 : (a.tiles_k_main >= 2 ? 2 : 1));
        a.nb_bias = nb_bias;
        const int bias_rows = a.splits * nb_bias, bias_ld = a.tiles_co * BMO;
        if ((a.dbias || a.dbias2) && scratch && !lone) {
            const long long needb = (long long)bias_rows * bias_ld + rows_fold_tmp_floats(bias_rows, bias_ld, 1);
            if (used + needb <= scratch_floats) { a.bpart = scratch + used; a.bpart_ld = bias_ld; used += needb; }
        }
        a.zero_targets = (a.overwrite && a.part && !a.fuse_cnt && sg > 1 && !part2) ? 1 : 0;   // (split groups that still add with atomics)
        RowsFoldArgs bf = {};   // the bias fold: rides on the reduce launch when there is one (<= L2I_FOLD_DIRECT rows), else its own launch
        bool bias_ride = false;
        if (a.bpart) {
            float* d0 = a.dbias ? a.dbias : a.dbias2;
            bf = rows_fold_args(a.bpart, bias_rows, bias_ld, 1, d0, nullptr, a.Co, 0, 2, (a.dbias && a.dbias2) ? a.dbias2 : nullptr, bias_ld);
            bias_ride = bias_rows <= L2I_FOLD_DIRECT && a.part && !a.fuse_cnt;
        }
        auto fold_bias = [&]() {
            if (!a.bpart || bias_ride) return;
            float* d0 = a.dbias ? a.dbias : a.dbias2;
            rows_fold(a.bpart, bias_rows, bias_ld, 1, d0, nullptr, a.Co, 0, 2, a.bpart + (size_t)bias_rows * bias_ld, stream,
                      (a.dbias && a.dbias2) ? a.dbias2 : nullptr);
        };
        int lgW = 0, lgH = 0;
        while ((1 << lgW) < a.Wo) ++lgW;
        while ((1 << lgH) < a.Ho) ++lgH;
        const size_t lds2 = (size_t)2 * 64 * (BMO * 2 + 256);   // (= 4 stages of 32 pixels for the deep pipeline)
        const bool whole = a.M % SBK == 0;
        const int mode = !whole ? 0 : (!a.pool2 && !a.up2) ? 1 : ((a.Ho * a.Wo) % SBK == 0 ? 2 : 0);
        if (deep && BMO == 64) {
            if (mode == 1) L2I_LAUNCH(1, (conv_wgrad_dma_kernel<64, 1, 4, true>), dim3(nblk), dim3(256), lds2, stream, a, lgW, lgH);
            else if (mode == 2) L2I_LAUNCH(1, (conv_wgrad_dma_kernel<64, 2, 4, true>), dim3(nblk), dim3(256), lds2, stream, a, lgW, lgH);
            else L2I_LAUNCH(1, (conv_wgrad_dma_kernel<64, 0, 4, true>), dim3(nblk), dim3(256), lds2, stream, a, lgW, lgH);
        } else if (deep) {
            if (mode == 1) L2I_LAUNCH(1, (conv_wgrad_dma_kernel<128, 1, 4, true>), dim3(nblk), dim3(256), lds2, stream, a, lgW, lgH);
            else if (mode == 2) L2I_LAUNCH(1, (conv_wgrad_dma_kernel<128, 2, 4, true>), dim3(nblk), dim3(256), lds2, stream, a, lgW, lgH);
            else L2I_LAUNCH(1, (conv_wgrad_dma_kernel<128, 0, 4, true>), dim3(nblk), dim3(256), lds2, stream, a, lgW, lgH);
        } else if (BMO == 64) {
            if (mode == 1) L2I_LAUNCH(1, (conv_wgrad_dma_kernel<64, 1>), dim3(nblk), dim3(256), lds2, stream, a, lgW, lgH);
            else if (mode == 2) L2I_LAUNCH(1, (conv_wgrad_dma_kernel<64, 2>), dim3(nblk), dim3(256), lds2, stream, a, lgW, lgH);
            else L2I_LAUNCH(1, (conv_wgrad_dma_kernel<64, 0>), dim3(nblk), dim3(256), lds2, stream, a, lgW, lgH);
        } else if (nw8) {
            const size_t lds8 = (size_t)2 * 128 * (BMO * 2 + 256);   // 128 KB: one workgroup per CU
            static bool ready8 = false;
            if (!ready8) {
                (void)hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<128, 1, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);
                (void)hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<128, 2, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8);
                ready8 = true;
            }
            if (mode == 1) L2I_LAUNCH(1, (conv_wgrad_dma_kernel<128, 1, 8>), dim3(nblk), dim3(512), lds8, stream, a, lgW, lgH);
            else L2I_LAUNCH(1, (conv_wgrad_dma_kernel<128, 2, 8>), dim3(nblk), dim3(512), lds8, stream, a, lgW, lgH);
        } else if (nw2) {
            const size_t lds1 = (size_t)2 * 32 * (BMO * 2 + 256);
            if (mode == 1) L2I_LAUNCH(1, (conv_wgrad_dma_kernel<128, 1, 2>), dim3(nblk), dim3(128), lds1, stream, a, lgW, lgH);
            else if (mode == 2) L2I_LAUNCH(1, (conv_wgrad_dma_kernel<128, 2, 2>), dim3(nblk), dim3(128), lds1, stream, a, lgW, lgH);
            else L2I_LAUNCH(1, (conv_wgrad_dma_kernel<128, 0, 2>), dim3(nblk), dim3(128), lds1, stream, a, lgW, lgH);
        } else {
            if (mode == 1) L2I_LAUNCH(1, (conv_wgrad_dma_kernel<128, 1>), dim3(nblk), dim3(256), lds2, stream, a, lgW, lgH);
            else if (mode == 2) L2I_LAUNCH(1, (conv_wgrad_dma_kernel<128, 2>), dim3(nblk), dim3(256), lds2, stream, a, lgW, lgH);
            else L2I_LAUNCH(1, (conv_wgrad_dma_kernel<128, 0>), dim3(nblk), dim3(256), lds2, stream, a, lgW, lgH);
        }
        if (a.part && a.fuse_cnt) { fold_bias(); return l2i_check_launch(); }   // (the last arrivers reduced the splits)
        if (a.part) {
            const long long nthr = (long long)tiles * BMO * 32;
            const unsigned nbx = (unsigned)((nthr + 256 / sl - 1) / (256 / sl));
            const int nw2l = (int)(nw2 && !nw8 && BMO == 128);
            // split groups: enough workgroups to fill the chip (>= ~512), at least 8 splits per group
            // (sg split groups of sper splits each: computed above)
            RowsFoldArgs nobf = {};
            const unsigned nbx_b = nbx + (bias_ride ? (unsigned)bf.nbx : 0u);
            if (part2) {   // two stages, no atomics: groups -> part2, then the groups of a tile in order -> dw
                L2I_LAUNCH(1, wgrad_reduce_kernel, dim3(nbx, (unsigned)sg, (unsigned)nz_), dim3(256), 0, stream, (const float*)a.part, a.dw, BMO,
                           a.tiles_k, tiles, a.splits, a.Co, a.K, a.ldw, a.alpha, nw2l, sper,
                           a.tiles_k_main, a.sc_dw, a.sc_Ci, a.sc_ldw, a.dw_b, a.sc_dw_b, a.overwrite, reinterpret_cast<float4*>(part2), nobf, (int)nbx, sl);
                L2I_LAUNCH(1, wgrad_reduce_kernel, dim3(nbx_b, 1u, (unsigned)nz_), dim3(256), 0, stream, (const float*)part2, a.dw, BMO,
                           a.tiles_k, tiles, sg * nz_, a.Co, a.K, a.ldw, a.alpha, nw2l, sg,
                           a.tiles_k_main, a.sc_dw, a.sc_Ci, a.sc_ldw, a.dw_b, a.sc_dw_b, a.overwrite, (float4*)nullptr, bias_ride ? bf : nobf, (int)nbx, sl);
            } else {
                L2I_LAUNCH(1, wgrad_reduce_kernel, dim3(nbx_b, (unsigned)sg, (unsigned)nz_), dim3(256), 0, stream, (const float*)a.part, a.dw, BMO,
                           a.tiles_k, tiles, a.splits, a.Co, a.K, a.ldw, a.alpha, nw2l, sper,
                           a.tiles_k_main, a.sc_dw, a.sc_Ci, a.sc_ldw, a.dw_b, a.sc_dw_b, a.overwrite, (float4*)nullptr, bias_ride ? bf : nobf, (int)nbx, sl);
            }
        }
        fold_bias();
        return l2i_check_launch();
    }
    // generic kernel (f32 operands, maps that are not powers of two): the splits' tiles row-major in the scratch + an ordered reduce when they fit
    a.bpart = nullptr; a.bpart_ld = 0; a.part_rm = nullptr;
    long long used = 0;
    const int nz_ = dual ? 2 : 1;
    if (scratch && (a.splits > 1 || dual) && (long long)nblk * BMO * 128 <= scratch_floats) { a.part_rm = scratch; used = (long long)nblk * BMO * 128; }
    const int bias_ld = a.tiles_co * BMO;
    if (a.dbias && scratch) {
        const long long needb = (long long)a.splits * bias_ld + rows_fold_tmp_floats(a.splits, bias_ld, 1);
        if (used + needb <= scratch_floats) { a.bpart = scratch + used; a.bpart_ld = bias_ld; }
    }
    if (!a.part_rm && a.splits > 1 && clear_targets() != L2I_OK) return L2I_ERR_LAUNCH;
    const size_t lds = (size_t)(BMO + 128) * IG_ROWB;
    if (BMO == 64)
        L2I_LAUNCH(1, (conv_wgrad_kernel<T, 64>), dim3(nblk), dim3(256), lds, stream, a);
    else
        L2I_LAUNCH(1, (conv_wgrad_kernel<T, 128>), dim3(nblk), dim3(256), lds, stream, a);
    if (a.part_rm) {
        const unsigned nbx = (unsigned)(((long long)tiles * BMO * 32 + 255) / 256);
        L2I_LAUNCH(1, wgrad_reduce_rm_kernel, dim3(nbx, 1u, (unsigned)nz_), dim3(256), 0, stream, (const float*)a.part_rm, a.dw, a.dw_b, BMO, a.tiles_k, tiles,
                   a.splits, a.Co, a.K, a.ldw, a.alpha, a.overwrite);
    }
    if (a.bpart) rows_fold(a.bpart, a.splits, bias_ld, 1, a.dbias, nullptr, a.Co, 0, 2, a.bpart + (size_t)a.splits * bias_ld, stream);
    return l2i_check_launch();
}

extern "C" int l2i_conv2d_wgrad_dual(const void* x, const void* dy, float* dw, int dtype, int B, int Hi, int Wi, int Ci,
                                     int Ho, int Wo, int Co, int KH, int up2, int pool2, int ldw, float alpha,
                                     const int* nimg, float* dbias, float* scratch, long long scratch_floats,
                                     const void* sc_x, float* sc_dw, int sc_Ci, int sc_up2, int sc_ldw, float* sc_dbias,
                                     float* dw_b, float* sc_dw_b, int overwrite, void* stream);

extern "C" int l2i_conv2d_wgrad(const void* x, const void* dy, float* dw, int dtype, int B, int Hi, int Wi, int Ci,
                                int Ho, int Wo, int Co, int KH, int up2, int pool2, int ldw, float alpha,
                                const int* nimg, float* dbias, float* scratch, long long scratch_floats, void* stream) {
    return l2i_conv2d_wgrad_dual(x, dy, dw, dtype, B, Hi, Wi, Ci, Ho, Wo, Co, KH, up2, pool2, ldw, alpha, nimg, dbias, scratch, scratch_floats,
                                 nullptr, nullptr, 0, 0, 0, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int l2i_conv2d_wgrad_sc(const void* x, const void* dy, float* dw, int dtype, int B, int Hi, int Wi, int Ci,
                                   int Ho, int Wo, int Co, int KH, int up2, int pool2, int ldw, float alpha,
                                   const int* nimg, float* dbias, float* scratch, long long scratch_floats,
                                   const void* sc_x, float* sc_dw, int sc_Ci, int sc_up2, int sc_ldw, float* sc_dbias, void* stream) {
    return l2i_conv2d_wgrad_dual(x, dy, dw, dtype, B, Hi, Wi, Ci, Ho, Wo, Co, KH, up2, pool2, ldw, alpha, nimg, dbias, scratch, scratch_floats,
                                 sc_x, sc_dw, sc_Ci, sc_up2, sc_ldw, sc_dbias, nullptr, nullptr, 0, stream);
}

// Dual launch: dw_b non-null -> the B images are two passes of B/2 images; the first half's gradient is added to dw (sc_dw), the
// second half's to dw_b (sc_dw_b); both bias gradients go to dbias (a bias has no spectral norm: one sum). `nimg` counts the live
// leading images of EACH half. (B/2) * Ho * Wo must be a whole number of the kernel's pixel steps (64), else L2I_ERR_ARG: the
// caller then issues two launches.
extern "C" int l2i_conv2d_wgrad_dual(const void* x, const void* dy, float* dw, int dtype, int B, int Hi, int Wi, int Ci,
                                     int Ho, int Wo, int Co, int KH, int up2, int pool2, int ldw, float alpha,
                                     const int* nimg, float* dbias, float* scratch, long long scratch_floats,
                                     const void* sc_x, float* sc_dw, int sc_Ci, int sc_up2, int sc_ldw, float* sc_dbias,
                                     float* dw_b, float* sc_dw_b, int overwrite, void* stream) {
    if (!x || !dy || !dw) return L2I_ERR_ARG;
    if (!dw_b && sc_dw_b) return L2I_ERR_ARG;
    if (sc_x && (!sc_dw || sc_Ci <= 0 || sc_ldw < sc_Ci)) return L2I_ERR_ARG;
    if (nimg && (Ho * Wo) % 64) return L2I_ERR_ARG;
    WgradArgs a;
#ifdef L2I_ABLATIONS   // wrong-result switch: ablation builds only (L2I_EXTRA_FLAGS=-DL2I_ABLATIONS)
    static const int no_epi = getenv("L2I_WGRAD_NOEPI") ? atoi(getenv("L2I_WGRAD_NOEPI")) : 0;
    a.no_epi = no_epi;
#else
    a.no_epi = 0;
#endif
    a.nimg = nimg;
    a.dbias = dbias;
    a.sc_x = sc_x; a.sc_dw = sc_x ? sc_dw : nullptr; a.dbias2 = sc_x ? sc_dbias : nullptr; a.sc_Ci = sc_Ci; a.sc_ldw = sc_ldw; a.sc_up2 = sc_up2 ? 1 : 0;
    a.sc_x_bytes = 0; a.tiles_k_main = 0;
    a.dw_b = dw_b; a.sc_dw_b = (sc_x && dw_b) ? sc_dw_b : nullptr;
    a.fuse_cnt = nullptr; a.nw2_layout = 0; a.overwrite = overwrite ? 1 : 0; a.zero_targets = 0; a.nb_bias = 1; a.bpart = nullptr; a.bpart_ld = 0; a.part_rm = nullptr;
    a.x = x; a.dy = dy; a.dw = dw;
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ci = Ci; a.Ho = Ho; a.Wo = Wo; a.Co = Co; a.KH = KH;
    a.up2 = up2 ? 1 : 0; a.pool2 = pool2 ? 1 : 0; a.ldw = ldw; a.alpha = alpha;
    if (dtype == 0) return launch_wgrad<float>(a, (hipStream_t)stream, scratch, scratch_floats);
    if (dtype == 1) return launch_wgrad<bf16_t>(a, (hipStream_t)stream, scratch, scratch_floats);
    return L2I_ERR_ARG;
}
