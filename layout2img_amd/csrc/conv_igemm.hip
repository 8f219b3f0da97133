// Implicit-GEMM convolution for gfx950 (MFMA), NHWC operands.
//
// Replaces on the hot path: nn.Conv2d 3x3/1x1 forward as used by the generator
// ResBlock (reference model/resnet_generator_app_v2.py:653-670, nearest-2x
// upsample fused into the A gather), by the discriminator blocks
// (model/rcnn_discriminator_app.py:303-344, avg_pool2d(2) fused into the
// epilogue), nn.Linear (H=W=1), and the data-gradient of all of those (same
// kernel, tap-flipped/transposed weight pack; upsample<->pool swap roles).
//
//   Y[m, n] = alpha * pool?( sum_k A[m, k] * Wp[n, k] ) + bias[n] + res[m, n]
//   m = (b, y, x) over the conv-output grid (Ho, Wo), k = (ky, kx, ci), n = co
//   A[m, k] = X[b, (y+ky-pad)>>up2, (x+kx-pad)>>up2, ci]   (zero outside the grid)
//
// Tile: BM=128 output pixels x BN (64|128) channels per 256-thread workgroup,
// 4 waves as 2x2, each wave (64 x BN/2) of 32x32 MFMA tiles. The 128 pixels of
// a tile are a PHxPW patch enumerated quad-major (4 consecutive rows of the
// GEMM = one 2x2 pixel quad), so a 32x32 accumulator's registers 4g..4g+3 are
// one quad and the 2x2 pool is an in-register sum.
#include <stdlib.h>
#include "igemm.h"

L2I_TRACE_DEFINE(conv)   // wave-level timestamps of the last launch (-DL2I_TRACE builds only; common.h)

// Folded 1x1 shortcut of a residual block (l2i_conv2d_fwd_sc): result += conv1x1(x) + bias, accumulated as extra K-chunks
// of the launch (conv_sc_tail) -- the shortcut's own result is never written and never read back as `res`.
// The kernels read these fields through late_sc() only, AFTER their 3x3 reduction: loaded with the other kernel arguments
// at kernel entry they stay live in SGPRs across the main loop, and the 128x64 tile (capped at 168 VGPRs for three
// workgroups per CU) then spills -- every launch, folded or not, ran 3 % slower.
struct ScArgs {
    const void* x;       // T [B, Hi, Wi, Ci], read at (y >> up2, x >> up2) of the launch's pre-pool output grid; null = none
    const void* w;       // T [Npad, Kpad] forward pack of the 1x1 weight
    const void* w_b;     // dual launch (ConvArgs::w_b): the shortcut's pack of the second half of the images
    const float* bias;   // [Co] or null
    float* out;          // f32, shape of out: where the shortcut goes when it cannot be folded (split-K, generic kernel): it then runs as its own launch and is read back as `res`
    int Ci, Hi, Wi, up2, Kpad, stages;
    unsigned x_bytes, w_bytes;
    // DATA-GRADIENT fold (l2i_conv2d_dgrad_sc, round 5): the launch is conv1's data gradient of a pre-activation block and the tail is
    // the data gradient of the block's 1x1 shortcut: dx = relu'(x) . (alpha W1^T * dh) + sc_alpha Wsc^T . dy(y >> up2, x >> up2).
    // The ReLU mask belongs to the 3x3 part only, so it is applied to the ACCUMULATORS (with the ratio of the two alphas) before
    // the tail's K-steps (conv_mask_first); the epilogue then scales by sc_alpha and adds bias-free residuals as usual.
    const void* mask_first;   // T, shape of out; null: the forward fold
    float pre_scale;          // alpha / sc_alpha
};

struct ConvArgs {
    const void* x;      // T  [B, Hi, Wi, Ci]
    const void* w;      // T  [Npad, Kpad], K order (ky, kx, ci)
    const void* w_b;    // DUAL launch (l2i_conv2d_fwd_dual), or null: images [B/2, B) are multiplied with THIS pack. The two passes of
                        // the discriminator step (D(real), D(fake): reference train_context_app_v2.py:158,167) run as one batch of 2b
                        // images, but each pass has its own power iteration, i.e. its own W / sigma (model/rcnn_discriminator_app.py
                        // spectral_norm hooks) -- one launch, twice the tiles, two packs.
    int half_rows;      // dual: (B/2) * Ho, the first GEMM row-of-pixels of the second half (a multiple of every tile's PH); with
                        // `nimg` the live-image count then applies to EACH half (rows [0, n Ho) and [half_rows, half_rows + n Ho)). 0: single
    const float* bias;  // [Co] or null
    const float* res;   // f32, shape of out, or null
    float* out;         // f32 [B, Hout, Wout, Co] or null
    void* out_op;       // T   same shape, optional operand copy (relu'd if relu_op)
    void* out_op_raw;   // T   same shape, optional un-activated operand copy
    const void* relu_mask;  // T, shape of out: result is zeroed where mask <= 0 (ReLU backward), before res is added
    const int* nimg;        // device int (optional): only the first *nimg images are live; rows of the others are written as zeros
    int B, Hi, Wi, Ci, Ho, Wo, Co, KH;
    int up2, pool2, relu_op;
    int Kpad, K;
    int PW, PH, hw_shift, lin, tiles_c, tiles_m, tiles_n;
    int chunk_major;     // K order (see issue_tiles)
    int nks;             // K-steps in total
    unsigned x_bytes, w_bytes;
    int splits, ks_per;  // split-K: `out` pre-zeroed, partials combined with f32 atomics
    // halo kernel geometry (conv_halo_kernel): sub-patches of PHs x PW pixels, halo rows h = sp*SUBH + hy*P + hx
    int PHs, sub_shift, P, SUBH, HR, HWd, halo_pieces;
    int compact;         // conv_halo3_kernel: sub-patches are whole images, no border rows are stored
    unsigned mg_tn, mg_tc, mg_ho, mg_subh, mg_p;   // fastdiv magics of tiles_n, tiles_c, Ho, SUBH, P (igemm.h)
    float alpha;
    float* stat_part;    // optional: partial rows [tiles_m * stat_wm][2 Co] in the caller's scratch -- every wave of the epilogue STORES the per-channel
                         // sum and sum of squares of its pixels of the f32 result there (what the batch-norm layer reading this result needs: no
                         // separate pass over it), rows_fold (common.h) adds the rows in a fixed order: deterministic, no atomics (round 6)
    int stat_wm;         // waves along M of the tile (set by the launcher)
    long long stat_cap;  // floats available at stat_part
    int* stat_rows_out;  // HOST side only: the launcher reports tiles_m * stat_wm here
    int roi_remap;       // tuning (L2I_ROI_REMAP=1): keep the XCD remap on launches with a live-image count (A/B)
    int no_epi;          // ablation builds only (-DL2I_ABLATIONS + L2I_CONV_NOEPI=1, results are wrong): skip the epilogue to measure what it costs
    int epi_lds;         // 1: coalesced epilogue through LDS (conv_epilogue_lds; default), 0: direct stores from the accumulator layout (L2I_EPI=0, A/B)
    ScArgs sc;           // folded 1x1 shortcut (see ScArgs); sc.x == null: none
    float* scratch;      // optional caller-owned f32 scratch (l2i_conv2d_fwd_dual): partial tiles of the weight-stationary small-map kernel
    long long scratch_floats;
    float* part;         // split-K with STORED partial tiles (conv_store_partial -> conv_split_reduce_kernel, which carries the epilogue); null: atomics
};

typedef const ScArgs __attribute__((address_space(4)))* ScArgsPtr;
__device__ __forceinline__ ScArgsPtr late_sc() {   // the shortcut arguments, from the kernel-argument segment, not before this point
    const char __attribute__((address_space(4)))* ka = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    return (ScArgsPtr)(ka + offsetof(ConvArgs, sc));
}


__device__ __forceinline__ void idx2pix(int idx, int hw_shift, int lin, int& py, int& px) {
    if (lin) { py = idx; px = 0; return; }
    const int q = idx >> 2, s = idx & 3;
    const int qy = q >> hw_shift, qx = q & ((1 << hw_shift) - 1);
    py = 2 * qy + (s >> 1);
    px = 2 * qx + (s & 1);
}

// Epilogue shared by the convolution kernels: alpha, bias, ReLU-backward mask, residual, optional operand copies,
// optional 2x2 average pool, split-K atomics.
// The MFMAs are issued with the operands swapped (weights as "A", pixels as "B"), so an accumulator tile is the
// TRANSPOSE of the output tile: lane l holds pixel (l & 31) of the 32-row tile and its 16 registers are output channels
// 8g + 4(l >> 5) + {0..3}, g = 0..3 -- four CONSECUTIVE channels per register group. Every global access of the epilogue
// is therefore 16 bytes per lane (8 for bf16 operands) instead of 4: a quarter of the instructions of the row-per-
// register layout. The 2x2 pool sums the 4 pixels of a quad = 4 adjacent lanes (quad-major pixel enumeration) with two
// DPP quad permutes.
__device__ __forceinline__ float quad_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    return v;
}
// Split-K partial tiles are combined with f32 atomics, which only run at speed when a wave's 64 addresses are
// contiguous runs; the transposed accumulator would scatter them over 32 rows. Each wave therefore turns its 32x32
// tiles back through a private 32 x 36-float LDS patch (the ring is idle by then) and adds 128-byte row segments.
template <typename T, int TM, int TN>
__device__ __forceinline__ void conv_epilogue_splitk(const ConvArgs& p, f32x16_t (&acc)[TM][TN], int wrow, int wcol, int lane,
                                                     int wave, int tile_r, int tile_c, int n0, int split, int rows_total,
                                                     char* smem) {
    constexpr int LD = 36;
    float* patch = reinterpret_cast<float*>(smem) + wave * 32 * LD;
    const T* __restrict__ Mask = reinterpret_cast<const T*>(p.relu_mask);
    const int m = lane & 31, h = lane >> 5;
    __syncthreads();   // every wave is done reading the ring
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(patch + m * LD + 8 * g + 4 * h) =
                    make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
            // (same wave wrote and reads: LDS operations of a wave complete in order)
            const int n = n0 + wcol + j * 32 + m;
#pragma unroll 4
            for (int rr = 0; rr < 16; ++rr) {
                const int row = 2 * rr + h;               // pixel within the 32-row tile
                float v = patch[row * LD + m] * p.alpha;
                const int idx = wrow + i * 32 + row;
                size_t rowoff;
                bool live;
                if (p.pool2) {   // the 4 pixels of a quad all add (x alpha = 1/4) into the pooled pixel
                    const int q = idx >> 2;
                    const int qy = q >> p.hw_shift, qx = q & ((1 << p.hw_shift) - 1);
                    const int r2 = tile_r * (p.PH >> 1) + qy;
                    const int Hq = p.Ho >> 1, Wq = p.Wo >> 1;
                    const int b = r2 / Hq, y2 = r2 - b * Hq, x2 = tile_c * (p.PW >> 1) + qx;
                    rowoff = ((size_t)(b * Hq + y2) * Wq + x2) * p.Co;
                    live = 2 * r2 < rows_total;
                } else {
                    int py, px;
                    idx2pix(idx, p.hw_shift, p.lin, py, px);
                    const int r = tile_r * p.PH + py;
                    rowoff = ((size_t)r * p.Wo + tile_c * p.PW + px) * p.Co;
                    live = r < rows_total;
                }
                if (!live || n >= p.Co) continue;
                // out = mask * (sum of partials + bias) + res: bias and res enter once (first split, first pixel of a quad)
                const bool once = split == 0 && (!p.pool2 || (idx & 3) == 0);
                if (p.bias && once) v += p.bias[n];   // (launches with a folded shortcut never run split-K)
                if (Mask && !(OpT<T>::to(Mask[rowoff + n]) > 0.f)) v = 0.f;
                if (p.res && once) v += p.res[rowoff + n];
                atomicAdd(p.out + rowoff + n, v);
            }
        }
}

// Split-K by STORES: a workgroup's partial tile goes to the caller's scratch in REGISTER order -- float4 index
// ((((tile * splits + split) * NW + wave) * TM + i) * TN + j) * 4 + g) * 64 + lane = acc[i][j][4g .. 4g + 3] -- every store
// instruction a contiguous 1 KB, no transposition, no atomics (plain stores run at 4.5x the rate of f32 atomics on this chip,
// DESIGN 4.1b); conv_split_reduce_kernel sums the splits in the same order and applies the WHOLE epilogue (alpha, biases, ReLU
// mask, residual, 2x2 pool, operand copies, statistics, live-row count), so a split launch is no longer restricted to plain f32
// results. Tiles of dead images (ROI heads) store nothing: the reduce kernel does not read them either.
template <int TM, int TN>
__device__ __forceinline__ void conv_store_partial(const ConvArgs& p, f32x16_t (&acc)[TM][TN], int tile, int split, int nw, int wave, int lane) {
    float4* t = reinterpret_cast<float4*>(p.part) + ((size_t)(tile * p.splits + split) * nw + wave) * (TM * TN * 4 * 64) + lane;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                t[((i * TN + j) * 4 + g) * 64] = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
}

template <typename T, int TM, int TN, bool SC = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16_t (&acc)[TM][TN], int wrow, int wcol, int lane,
                                              int tile_r, int tile_c, int n0, int split, int rows_total, int rows_live) {
#ifdef L2I_ABLATIONS
    if (p.no_epi) {   // (keeps the accumulators live)
        float s_ = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) s_ += acc[i][j][e];
        if (s_ == 1.2345e30f && p.out) p.out[0] = s_;
        return;
    }
#endif
    const int m = lane & 31, h = lane >> 5;
    T* __restrict__ OutOp = reinterpret_cast<T*>(p.out_op);
    T* __restrict__ OutRaw = reinterpret_cast<T*>(p.out_op_raw);
    const T* __restrict__ Mask = reinterpret_cast<const T*>(p.relu_mask);
    const bool lead = split == 0;   // bias and residual are added by the first split only
    const float* sc_bias = SC ? late_sc()->bias : nullptr;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int idx = wrow + i * 32 + m;   // this lane's pixel (GEMM row)
        size_t rowoff;
        bool live, dead;   // dead: a row of an image past *nimg -- written as zeros
        if (p.pool2) {
            const int q = idx >> 2;
            const int qy = q >> p.hw_shift, qx = q & ((1 << p.hw_shift) - 1);
            const int r2 = tile_r * (p.PH >> 1) + qy;
            const int Hq = p.Ho >> 1, Wq = p.Wo >> 1;
            const int b = r2 / Hq, y2 = r2 - b * Hq, x2 = tile_c * (p.PW >> 1) + qx;
            rowoff = ((size_t)(b * Hq + y2) * Wq + x2) * p.Co;
            live = r2 < p.B * Hq && (lane & 3) == 0;   // one lane of the quad writes the pooled pixel
            dead = 2 * r2 >= rows_live;
        } else {
            int py, px;
            idx2pix(idx, p.hw_shift, p.lin, py, px);
            const int r = tile_r * p.PH + py;
            rowoff = ((size_t)r * p.Wo + tile_c * p.PW + px) * p.Co;
            live = r < rows_total;
            dead = r >= rows_live;
        }
        if (p.splits > 1 && dead) live = false;   // split-K: dead rows keep the zeros of the memset
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wcol + j * 32 + 8 * g + 4 * h;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[i][j][4 * g + e];
                    if (p.pool2) v[e] = quad_sum(v[e]);
                    v[e] *= p.alpha;
                }
                if (!live || n >= p.Co) continue;
                const size_t off = rowoff + n;
                if (p.bias && lead) {
                    const float4 bb = *reinterpret_cast<const float4*>(p.bias + n);
                    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                }
                if (SC && sc_bias) {   // folded shortcut: its bias (never on a split-K launch)
                    const float4 bb = *reinterpret_cast<const float4*>(sc_bias + n);
                    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                }
                if (Mask) {
                    float mk[4];
                    Op4<T>::load(Mask + off, mk);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (!(mk[e] > 0.f)) v[e] = 0.f;
                }
                if (p.res && lead) {
                    const float4 rr = *reinterpret_cast<const float4*>(p.res + off);
                    v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                }
                if (dead) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }   // (a select: whatever was read for a dead row is dropped)
                if (p.splits > 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(p.out + off + e, v[e]);
                    continue;
                }
                if (p.out) *reinterpret_cast<float4*>(p.out + off) = make_float4(v[0], v[1], v[2], v[3]);
                if (OutRaw) Op4<T>::store(OutRaw + off, v);
                if (OutOp) {
                    if (p.relu_op) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    Op4<T>::store(OutOp + off, v);
                }
            }
        }
    }
}

// The same epilogue with COALESCED global accesses. In conv_epilogue a wave's store instruction touches 32 pixel rows
// with 32 bytes each (the transposed accumulator gives a lane 4 consecutive channels of ONE pixel, and the 32 lanes of a
// half-wave are 32 different pixels): in-situ ablation put that epilogue at 29 % of all conv time, a third of the HBM
// rate. Here each wave turns its 32-pixel x (32 TN)-channel slab through a private LDS patch (the ring and the halo are
// idle by then) and reads it back pixel-major: one instruction then covers 64 / (8 TN) pixels with 128 TN contiguous
// bytes each (f32; half of that for the bf16 copies / the ReLU mask), i.e. whole cache lines.
// The 2x2 average pool sums the 4 LDS rows of a quad (quad-major pixel order) instead of DPP permutes.
//
// Round 3: batched loads, no per-element branches. The round-2 form did, per 16-byte group, "load mask -> s_waitcnt
// vmcnt(0) -> load residual -> s_waitcnt vmcnt(0) -> store", each step behind a branch (the disassembly showed one full
// memory round trip per load, and vmcnt(0) also waits for the stores before it). Now every global access goes through a
// buffer descriptor with a 32-bit byte offset, so a lane that must not touch memory (row past the tensor, channel past
// Co, image past *nimg) simply carries an out-of-range offset (loads return zero, stores are dropped); the ReLU-mask and
// residual loads of a batch of 4 stages are issued together with the stages' LDS reads, then the 4 stages are finished
// and stored.
template <typename T> struct EpiT;
template <> struct EpiT<bf16_t> {
    typedef unsigned __attribute__((ext_vector_type(2))) raw_t;
    __device__ static __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0); }
    __device__ static __forceinline__ void unpack(raw_t u, float (&v)[4]) {
        v[0] = __uint_as_float(u[0] << 16); v[1] = __uint_as_float(u[0] & 0xffff0000u);
        v[2] = __uint_as_float(u[1] << 16); v[3] = __uint_as_float(u[1] & 0xffff0000u);
    }
    __device__ static __forceinline__ void store(__amdgpu_buffer_rsrc_t r, unsigned off, const float (&v)[4]) {
        raw_t u;
        u[0] = f2bf2(v[0], v[1]);
        u[1] = f2bf2(v[2], v[3]);
        __builtin_amdgcn_raw_buffer_store_b64(u, r, off, 0, 0);
    }
};
template <> struct EpiT<float> {
    typedef unsigned __attribute__((ext_vector_type(4))) raw_t;
    __device__ static __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); }
    __device__ static __forceinline__ void unpack(raw_t u, float (&v)[4]) {
        v[0] = __uint_as_float(u[0]); v[1] = __uint_as_float(u[1]); v[2] = __uint_as_float(u[2]); v[3] = __uint_as_float(u[3]);
    }
    __device__ static __forceinline__ void store(__amdgpu_buffer_rsrc_t r, unsigned off, const float (&v)[4]) {
        raw_t u;
        u[0] = __float_as_uint(v[0]); u[1] = __float_as_uint(v[1]); u[2] = __float_as_uint(v[2]); u[3] = __float_as_uint(v[3]);
        __builtin_amdgcn_raw_buffer_store_b128(u, r, off, 0, 0);
    }
};
__device__ __forceinline__ __amdgpu_buffer_rsrc_t epi_rsrc(const void* base, unsigned nbytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, base ? (int)nbytes : 0, 0x00020000);
}

template <typename T, int TM, int TN, int BMAX = 8, bool SC = false>   // BMAX: cap on the stages per batch (registers: 12-14 per stage in flight)
__device__ __forceinline__ void conv_epilogue_lds(const ConvArgs& p, f32x16_t (&acc)[TM][TN], int wrow, int wcol, int lane, int wave,
                                                  int tile_r, int tile_c, int n0, int rows_total, int rows_live, char* smem) {
#ifdef L2I_ABLATIONS
    if (p.no_epi) {   // (keeps the accumulators live)
        float s_ = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) s_ += acc[i][j][e];
        if (s_ == 1.2345e30f && p.out) p.out[0] = s_;
        return;
    }
#endif
    typedef unsigned __attribute__((ext_vector_type(4))) u4_t;
    constexpr int NCH = TN * 32, LD = NCH + 4;   // floats per LDS row: +4 keeps 16-byte alignment and spreads the banks
    constexpr int L4 = NCH / 4, PPI = 64 / L4;   // float4 groups per pixel; pixels (or quads) per wave-instruction
    constexpr int NIT = 32 / PPI;                // stages per 32-pixel slab (a quarter of them with the 2x2 pool)
    constexpr unsigned OOB = 0x80000000u;
    constexpr int SZT = (int)sizeof(T);
    float* patch = reinterpret_cast<float*>(smem) + wave * ((TN == 1 && TM % 2 == 0) ? 2 : 1) * 32 * LD;
    const int m = lane & 31, h = lane >> 5;
    const int cq = lane % L4, pp0 = lane / L4;
    const int n = n0 + wcol + cq * 4;
    const bool nv = n < p.Co;
    float bb[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && nv) {
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
        bb[0] = b4.x; bb[1] = b4.y; bb[2] = b4.z; bb[3] = b4.w;
    }
    const float* sc_bias = SC ? late_sc()->bias : nullptr;
    if (SC && sc_bias && nv) {   // folded shortcut: its bias
        const float4 b4 = *reinterpret_cast<const float4*>(sc_bias + n);
        bb[0] += b4.x; bb[1] += b4.y; bb[2] += b4.z; bb[3] += b4.w;
    }
    // descriptors over the result-shaped tensors (null pointer -> zero records: every access out of range)
    const unsigned out_elems = (unsigned)p.B * (unsigned)(p.Ho >> p.pool2) * (unsigned)(p.Wo >> p.pool2) * (unsigned)p.Co;
    const __amdgpu_buffer_rsrc_t rs_out = epi_rsrc(p.out, out_elems * 4u), rs_res = epi_rsrc(p.res, out_elems * 4u);
    const __amdgpu_buffer_rsrc_t rs_mask = epi_rsrc(p.relu_mask, out_elems * SZT), rs_op = epi_rsrc(p.out_op, out_elems * SZT);
    const __amdgpu_buffer_rsrc_t rs_raw = epi_rsrc(p.out_op_raw, out_elems * SZT);
    const bool has_mask = p.relu_mask != nullptr, has_res = p.res != nullptr, has_out = p.out != nullptr;
    const bool has_raw = p.out_op_raw != nullptr, has_op = p.out_op != nullptr, has_stats = p.stat_part != nullptr;
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};   // per-channel statistics of this lane's pixels (stat_part)
    const int wrow_s = __builtin_amdgcn_readfirstlane(wrow);
    const unsigned tile_base = (unsigned)((tile_r * p.PH * p.Wo + tile_c * p.PW) * p.Co);
    const int nstage = p.pool2 ? NIT / 4 : NIT;   // live stages per slab

    // Stages are taken in BATCHES of B inside a real (not unrolled) loop: the loads + LDS reads of B stages are issued, then
    // the B stages are finished and stored. The loop is deliberately NOT unrolled: this code runs once per workgroup, and a
    // fully unrolled epilogue is 25-30 KB of straight-line code that every CU has to pull through its instruction cache
    // cold at the end of every launch (a launch is one round of workgroups) -- the fetch, not the memory traffic, was the
    // bulk of the "epilogue time" (the round-2 ablation priced the epilogue at 80-120 us on a layer whose 100 MB of
    // results take 20 us to write).
    // B stages per batch = loads in flight per wave: one memory round trip (~2 us under load) per batch is what the epilogue
    // costs, so narrow tiles (TN = 1: 4 stages per slab) take TWO slabs per pass (SP) -- one round trip for the wave's whole tile.
    constexpr int SP = (TN == 1 && TM % 2 == 0) ? 2 : 1;   // slabs per pass (the wave's LDS patch holds SP x 32 pixel rows)
    constexpr int B = (TN >= 4 || sizeof(T) == 4 || BMAX < 8) ? 4 : 8;
    constexpr int NST = SP * NIT;                          // stage slots per pass
    __syncthreads();   // every wave is done reading the ring / the halo
#pragma unroll
    for (int i0 = 0; i0 < TM; i0 += SP) {
        // this wave's next SP 32-pixel slabs -> its LDS patch (same wave wrote and reads: LDS operations of a wave complete in order)
#pragma unroll
        for (int sl = 0; sl < SP; ++sl)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(patch + (sl * 32 + m) * LD + j * 32 + 8 * g + 4 * h) =
                        make_float4(acc[i0 + sl][j][4 * g], acc[i0 + sl][j][4 * g + 1], acc[i0 + sl][j][4 * g + 2], acc[i0 + sl][j][4 * g + 3]);
#pragma unroll 1
        for (int st0 = 0; st0 < NST; st0 += B) {
            float sv[B][4];
            u4_t srr[B];
            typename EpiT<T>::raw_t smk[B];
            unsigned so_ld[B], so_st[B];   // f32 byte offsets: loads (OOB for rows that are not read), stores (OOB for rows that do not exist)
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const int st = st0 + k;
                const int sl = SP == 1 ? 0 : st / NIT, it = SP == 1 ? st : st % NIT;
                const int i = i0 + sl;
                const float* pslab = patch + sl * 32 * LD;
                unsigned rowoff;
                bool live, dead;
                if (p.pool2) {
                    const int ql = min(it * PPI + pp0, 7);             // quad within the slab (clamped: stages past the slab are masked below)
                    const float* r0 = pslab + (4 * ql) * LD + cq * 4;
                    const float4 a0 = *reinterpret_cast<const float4*>(r0), a1 = *reinterpret_cast<const float4*>(r0 + LD);
                    const float4 a2 = *reinterpret_cast<const float4*>(r0 + 2 * LD), a3 = *reinterpret_cast<const float4*>(r0 + 3 * LD);
                    sv[k][0] = (a0.x + a1.x) + (a2.x + a3.x); sv[k][1] = (a0.y + a1.y) + (a2.y + a3.y);
                    sv[k][2] = (a0.z + a1.z) + (a2.z + a3.z); sv[k][3] = (a0.w + a1.w) + (a2.w + a3.w);
                    const int q = (wrow + i * 32) / 4 + ql;
                    const int qy = q >> p.hw_shift, qx = q & ((1 << p.hw_shift) - 1);
                    const int r2 = tile_r * (p.PH >> 1) + qy;
                    const int Hq = p.Ho >> 1, Wq = p.Wo >> 1;
                    const int b = r2 / Hq, y2 = r2 - b * Hq, x2 = tile_c * (p.PW >> 1) + qx;
                    rowoff = (unsigned)(((b * Hq + y2) * Wq + x2) * p.Co);
                    live = r2 < p.B * Hq && it < nstage;
                    dead = 2 * r2 >= rows_live;
                } else {
                    const int pix = min(it * PPI + pp0, 31);
                    const float4 a0 = *reinterpret_cast<const float4*>(pslab + pix * LD + cq * 4);
                    sv[k][0] = a0.x; sv[k][1] = a0.y; sv[k][2] = a0.z; sv[k][3] = a0.w;
                    // pixel index = (wave-uniform base, a multiple of PPI) + pp0: the quad-major decode splits into a SCALAR part
                    // and lane constants
                    const int pbase = wrow_s + i * 32 + it * PPI;
                    int py, px;
                    if (p.lin) { py = pbase + pp0; px = 0; }
                    else {
                        const int q = (pbase >> 2) + (PPI >= 4 ? (pp0 >> 2) : 0);
                        const int sq = PPI >= 4 ? (pp0 & 3) : ((pbase & 3) + pp0);
                        const int qy = q >> p.hw_shift, qx = q & ((1 << p.hw_shift) - 1);
                        py = 2 * qy + (sq >> 1);
                        px = 2 * qx + (sq & 1);
                    }
                    const int r = tile_r * p.PH + py;
                    rowoff = tile_base + (unsigned)((py * p.Wo + px) * p.Co);
                    live = r < rows_total && it < nstage;
                    dead = r >= rows_live;
                }
                const unsigned off = (rowoff + (unsigned)n) * 4u;
                so_st[k] = (live && nv) ? off : OOB;
                so_ld[k] = (live && nv && !dead) ? off : OOB;
                const unsigned off_t = SZT == 4 ? so_ld[k] : (unsigned)((int)so_ld[k] >> 1);   // (arithmetic shift: OOB stays out of range)
                if (has_mask) smk[k] = EpiT<T>::load(rs_mask, off_t);
                if (has_res) srr[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, so_ld[k], 0, 0);
            }
#pragma unroll
            for (int k = 0; k < B; ++k) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = sv[k][e] * p.alpha + bb[e];
                if (has_mask) {
                    float mk[4];
                    EpiT<T>::unpack(smk[k], mk);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (!(mk[e] > 0.f)) v[e] = 0.f;
                }
                if (has_res) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += __uint_as_float(srr[k][e]);
                }
                if (so_ld[k] != so_st[k]) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }   // a row of an image past *nimg: zeros (a select)
                if (has_stats) {
                    const bool cnt = so_st[k] != OOB;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float w_ = cnt ? v[e] : 0.f; ssum[e] += w_; ssq[e] = fmaf(w_, w_, ssq[e]); }
                }
                const unsigned st_t = SZT == 4 ? so_st[k] : (unsigned)((int)so_st[k] >> 1);
                if (has_out) {
                    u4_t o_;
                    o_[0] = __float_as_uint(v[0]); o_[1] = __float_as_uint(v[1]); o_[2] = __float_as_uint(v[2]); o_[3] = __float_as_uint(v[3]);
                    __builtin_amdgcn_raw_buffer_store_b128(o_, rs_out, so_st[k], 0, 0);
                }
                if (has_raw) EpiT<T>::store(rs_raw, st_t, v);
                if (has_op) {
                    if (p.relu_op) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    EpiT<T>::store(rs_op, st_t, v);
                }
            }
        }
    }
    if (has_stats) {   // lanes that share a channel group are L4 apart: combine them (xor butterfly: a fixed order), then ONE row per wave of
                       // the launch -- row = tile x stat_wm + the wave's position along M, columns = this wave's channels -- stored, not added
#pragma unroll
        for (int o = L4; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) { ssum[e] += __shfl_xor(ssum[e], o, 64); ssq[e] += __shfl_xor(ssq[e], o, 64); }
        if (lane < L4 && nv) {
            const int row = (tile_r * p.tiles_c + tile_c) * p.stat_wm + wrow_s / (TM * 32);
            float* d = p.stat_part + (size_t)row * (2 * p.Co) + n;
            *reinterpret_cast<float4*>(d) = make_float4(ssum[0], ssum[1], ssum[2], ssum[3]);
            *reinterpret_cast<float4*>(d + p.Co) = make_float4(ssq[0], ssq[1], ssq[2], ssq[3]);
        }
    }
}

// Tile geometry: BM x BN outputs per workgroup of WM x WN waves (each wave a (BM/WM) x (BN/WN) block of 32x32
// MFMA tiles), NS-stage LDS ring.
// HK = 1 halves the K-step (64-byte LDS rows): twice as many, half as large ring stages, so that two workgroups per
// CU can each keep three tiles in flight within the 160 KB of LDS (the DMA round trip under load is ~1.5 us).
// (second launch bound = waves per SIMD the LDS ring allows: keeps the register allocation of the -- fully unrolled,
//  load-batching -- epilogue within the occupancy the main loop was tuned for)
template <typename T, int BM, int BN, int WM, int WN, int NS, int HK>
__global__ __launch_bounds__(WM* WN * 64, ((160 * 1024) / (NS * (BM + BN) * (HK ? 64 : 128))) * (WM * WN) / 4) void conv_igemm_kernel(ConvArgs p) {
    constexpr int THREADS = WM * WN * 64;
    constexpr int ROWB = HK ? 64 : 128;     // bytes per LDS tile row
    constexpr int CPR = ROWB / 16;        // 16-byte chunks per LDS row
    constexpr int RPP = THREADS / CPR;        // tile rows filled per DMA pass
    constexpr int BK = Mma<T>::BK >> HK;
    constexpr int EPG = OpT<T>::EPG;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int AP = BM / RPP, BP = BN / RPP;  // LDS-DMA instructions per thread per K-step
    static_assert(RPP % 16 == 0 && BM % RPP == 0 && BN % RPP == 0, "tile geometry");

    // LDS: two stages of [A 128 rows | B BN rows], 128-byte rows, filled by LDS-DMA (global_load_lds_dwordx4:
    // lane i of a wave-instruction lands at base + 16 i, i.e. 8 rows x 8 chunks), no VGPR staging, no ds_write.
    // (RPP rows per pass: the +RPP*q rows of a thread share (row >> 1) & 7, so one swizzle per thread.)
    // Bank conflicts of the ds_read_b128 fragment reads are avoided by an XOR swizzle applied on the SOURCE
    // side (the lane that fills physical chunk c of row r fetches logical chunk c ^ ig2_swz(r)) and on the read.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = (BM + BN) * ROWB;

    L2I_TR(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = p.tiles_m * p.tiles_n;
    const int split = blockIdx.x / nblk;
    // (ROI heads: the live rows are compacted to the FRONT, so the contiguous-run-per-XCD remap would give all live tiles
    //  to the first XCDs and leave the others idle; the dispatcher's own round robin spreads them evenly)
    const int bid = (p.nimg && !p.roi_remap) ? (int)blockIdx.x - split * nblk : xcd_remap(blockIdx.x - split * nblk, nblk);
    const int tile_m = fastdiv(bid, p.mg_tn), tile_n = bid - tile_m * p.tiles_n;
    const int tile_r = fastdiv(tile_m, p.mg_tc), tile_c = tile_m - tile_r * p.tiles_c;
    const int n0 = tile_n * BN;
    const int rows_total = p.B * p.Ho;
    const bool second = p.half_rows > 0 && tile_r * p.PH >= p.half_rows;   // dual launch: this tile's images use the second pack
    // live rows of this tile's half end at rows_live (the count applies to each half of a dual launch)
    const int rows_live = p.nimg ? min(p.half_rows > 0 ? p.half_rows : rows_total, *p.nimg * p.Ho) + (second ? p.half_rows : 0) : rows_total;   // (scalar load)
    const bool tile_dead = !p.lin && tile_r * p.PH >= rows_live;   // every row belongs to a dead image: no reduction, zeros out
    const int pad = p.KH >> 1;

    // ---- per-thread A rows (fixed for the whole K loop). Loads are buffer_load ... lds: a 32-bit byte offset per
    // lane, and an out-of-range offset makes the hardware write zeros -- that is how padding taps, rows past the
    // end and channels past Ci are zero-filled without a branch or a 64-bit address select.
    constexpr unsigned OOB = 0x80000000u;
    constexpr int SZ = (int)sizeof(T);
    const int lrow = tid / CPR;
    const int lchunk = (tid % CPR) ^ ig2_swz_t<HK>(lrow);  // logical 16-byte chunk this lane fetches (rows +RPP*q: same swizzle)
    int a_y[AP], a_x[AP];
    unsigned a_off[AP], a_mask[AP];                 // byte offset of the row's pixel (+ lane chunk); 9-bit tap validity
#pragma unroll
    for (int q = 0; q < AP; ++q) {
        int py, px;
        idx2pix(lrow + RPP * q, p.hw_shift, p.lin, py, px);
        const int r = tile_r * p.PH + py;
        const int b = fastdiv(r, p.mg_ho);
        const bool rv = r < rows_total;
        const int y = r - b * p.Ho, x = tile_c * p.PW + px;
        a_y[q] = y;
        a_x[q] = x;
        a_off[q] = (unsigned)(((b * p.Hi + (y >> p.up2)) * p.Wi + (x >> p.up2)) * p.Ci) * SZ;
        // 9-bit tap validity without per-tap divisions (the loop over t with t / KH, t % KH cost ~3000 instructions of
        // prologue per workgroup: 8 us of an 13-us workgroup on the 3-channel image conv, tools/perf/conv_trace.py)
        unsigned m = 0;
        if (p.KH == 1) m = rv ? 1u : 0u;
        else {
            const unsigned ry = (y >= 1 ? 1u : 0u) | 2u | (y + 1 < p.Ho ? 4u : 0u);   // rows ky = 0, 1, 2 inside the map
            const unsigned rx = (x >= 1 ? 1u : 0u) | 2u | (x + 1 < p.Wo ? 4u : 0u);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
                if ((ry >> ky) & 1u) m |= rx << (3 * ky);
            if (!rv) m = 0;
        }
        a_mask[q] = m;
    }
    unsigned b_off[BP];
#pragma unroll
    for (int q = 0; q < BP; ++q) b_off[q] = (unsigned)((n0 + lrow + RPP * q) * p.Kpad) * SZ;
    const int wbase = __builtin_amdgcn_readfirstlane(wave) * 1024;  // one wave-instruction fills 1 KB = this wave's rows of a pass
    const int lane_c = lchunk * EPG;                                          // first channel of the lane's chunk

    // K traversal. chunk_major (3x3, Ci >= BK): steps run channel-chunk-major, tap-minor -- the 9 taps of one
    // 64-channel chunk are consecutive, a workgroup re-reads nearly the same input lines 9 steps in a row (L1/L2
    // hits), and all per-step index arithmetic is scalar (tap counters) plus ~4 VALU per row. Otherwise
    // (1x1 / linear / Ci < BK) the step covers k = ks*BK.. linearly and a lane derives its own tap.
    int nx_tap, nx_cb;  // running (tap, channel-chunk base) of the NEXT tile to issue (chunk_major)
    {
        const int taps = p.KH * p.KH;
        const int ksa = split * p.ks_per;
        nx_cb = (ksa / taps) * BK;
        nx_tap = ksa - (ksa / taps) * taps;
    }
    int nx_ks = split * p.ks_per;
    auto advance = [](int& tap, int& cb, int& ks, int bk) {
        ++ks;
        if (++tap == 9) { tap = 0; cb += bk; }
    };
    const u32x4_t rsrc_x = make_rsrc(p.x, p.x_bytes), rsrc_w = make_rsrc(second ? p.w_b : p.w, p.w_bytes);
    const unsigned smem_addr = lds_addr_of(smem);
    // (iterator state is passed by value: captured-by-reference counters ended up in scratch memory, and scratch
    // loads count on vmcnt just like the DMA)
    auto issue_tiles = [&](unsigned stage, const int it_tap, const int it_cb, const int it_ks) {  // stage: LDS byte address
        if (p.chunk_major) {
            const int ky = it_tap / 3, kx = it_tap - ky * 3;
            const unsigned tapbit = 1u << it_tap;
            const bool cv = it_cb + lane_c < p.Ci;
            const unsigned cadd = (unsigned)(it_cb + lane_c) * SZ;
            const int s_delta = ((ky - 1) * p.Wi + (kx - 1)) * p.Ci * SZ;   // non-up2 tap displacement (scalar)
#pragma unroll
            for (int q = 0; q < AP; ++q) {
                unsigned off;
                if (p.up2) {
                    const int dy = (((a_y[q] & 1) + ky - 1) >> 1), dx = (((a_x[q] & 1) + kx - 1) >> 1);
                    off = a_off[q] + (unsigned)((dy * p.Wi + dx) * p.Ci * SZ) + cadd;
                } else {
                    off = a_off[q] + (unsigned)s_delta + cadd;
                }
                const unsigned voff = ((a_mask[q] & tapbit) && cv) ? off : OOB;
                buf_load_lds16(rsrc_x, voff, stage + q * RPP * ROWB + wbase);
            }
            const unsigned kadd = (unsigned)(it_tap * p.Ci) * SZ + cadd;
#pragma unroll
            for (int q = 0; q < BP; ++q)
                buf_load_lds16(rsrc_w, b_off[q] + kadd, stage + (BM + q * RPP) * ROWB + wbase);
        } else {
            const int k0 = it_ks * BK + lane_c;
            const bool kvalid = k0 < p.K;
            if (p.KH == 1) {   // 1x1 / Linear (wave-uniform branch): no tap arithmetic, no integer divisions in the K loop
#pragma unroll
                for (int q = 0; q < AP; ++q) {
                    const unsigned voff = (kvalid && (a_mask[q] & 1u)) ? a_off[q] + (unsigned)(k0 * SZ) : OOB;
                    buf_load_lds16(rsrc_x, voff, stage + q * RPP * ROWB + wbase);
                }
            } else {
                const int tap = k0 / p.Ci, ci = k0 - tap * p.Ci;
                const int ky = tap / p.KH, kx = tap - ky * p.KH;
#pragma unroll
                for (int q = 0; q < AP; ++q) {
                    const int ys = (a_y[q] + ky - pad) >> p.up2, xs = (a_x[q] + kx - pad) >> p.up2;
                    const int y0 = a_y[q] >> p.up2, x0 = a_x[q] >> p.up2;
                    const unsigned off = a_off[q] + (unsigned)((((ys - y0) * p.Wi + (xs - x0)) * p.Ci + ci) * SZ);
                    const unsigned voff = (kvalid && ((a_mask[q] >> tap) & 1u)) ? off : OOB;
                    buf_load_lds16(rsrc_x, voff, stage + q * RPP * ROWB + wbase);
                }
            }
#pragma unroll
            for (int q = 0; q < BP; ++q)
                buf_load_lds16(rsrc_w, b_off[q] + (unsigned)k0 * SZ, stage + (BM + q * RPP) * ROWB + wbase);
        }
    };

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int wrow = (wave / WN) * (BM / WM), wcol = (wave % WN) * (BN / WN);
    const int ks0 = split * p.ks_per;
    const int nks = tile_dead ? ks0 : min(p.nks, ks0 + p.ks_per);
    // NS-stage LDS ring. Tiles ks .. ks+NS-2 are in flight while tile ks is consumed: each wave waits with a COUNTED
    // s_waitcnt vmcnt for its own DMA of tile ks (its LPT newest-but-... loads may stay outstanding), then a raw
    // s_barrier makes every wave's part of the tile visible and proves the stage about to be refilled is no longer
    // read. (__syncthreads() would drain vmcnt to 0 and serialise the ring.) One K-step of a single workgroup is
    // otherwise bound by the ~1.5 us DMA round trip, not by its 0.2 us of MFMA work.
    constexpr int LPT = AP + BP;  // LDS-DMA instructions per thread per tile
    L2I_TR(1);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (ks0 + s < nks) {
            issue_tiles(smem_addr + s * STAGE, nx_tap, nx_cb, nx_ks);
            advance(nx_tap, nx_cb, nx_ks, BK);
        }
    for (int ks = ks0; ks < nks; ++ks) {
        const int it = ks - ks0;
        const int ahead = min(NS - 2, nks - 1 - ks);  // tiles allowed to stay in flight behind tile ks
        if (NS >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPT) : "memory");
        else if (NS >= 3 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (ks + NS - 1 < nks) {
            issue_tiles(smem_addr + ((it + NS - 1) % NS) * STAGE, nx_tap, nx_cb, nx_ks);
            advance(nx_tap, nx_cb, nx_ks, BK);
        }
        char* cur = smem + (it % NS) * STAGE;
        Mma2<T>::template step<TM, TN, HK>(cur, cur + BM * ROWB, wrow, wcol, lane, acc);
    }

    L2I_TR(2);
    if (p.splits > 1 && p.part) conv_store_partial<TM, TN>(p, acc, bid, split, WM * WN, wave, lane);   // (round 6: stored partial tiles + conv_split_reduce_kernel, as the halo kernels)
    else if (p.splits > 1) conv_epilogue_splitk<T, TM, TN>(p, acc, wrow, wcol, lane, wave, tile_r, tile_c, n0, split, rows_live, smem);
    else if (p.epi_lds) conv_epilogue_lds<T, TM, TN>(p, acc, wrow, wcol, lane, wave, tile_r, tile_c, n0, rows_total, rows_live, smem);
    else conv_epilogue<T, TM, TN>(p, acc, wrow, wcol, lane, tile_r, tile_c, n0, split, rows_total, rows_live);
    L2I_TR(3);
}

// ---------------------------------------------------------------- 3x3 convolution with an LDS-resident input halo
// The implicit-GEMM kernel above re-fetches the A tile for each of the 9 taps, and with 128x128 tiles its LDS-DMA traffic
// (32 KB per 64-MFMA K-step = 64 B/clk/CU) sits exactly at the CU's vector-memory peak: measured 30-38 % of the MFMA
// peak, L1-bound. Here a workgroup's output pixels are spatial patches (sub-patches of PHs x PW pixels, never crossing an
// image), so the inputs of ALL 9 taps of one 64-channel chunk are the patch plus a one-pixel border: that halo is
// DMA'd into LDS ONCE per chunk (double-buffered, one 1 KB piece per wave per tap step while the previous chunk
// computes), and a tap's A fragment is a ds_read_b128 at a shifted halo row. Per chunk a 128x128 tile moves
// 23 KB (halo) + 9 x 16 KB (weights) instead of 9 x 32 KB. With the nearest-2x upsample fused (up2) the halo is held at
// INPUT resolution (PHs/2+2 x PW/2+2) and the shift is ((y + ky - 1) >> 1).
//
// Halo row h = sp*SUBH + hy*P + hx (P, SUBH even); its eight 16-byte channel chunks are stored XOR-swizzled with
// ((hx >> 1) + 4 hy + 2 sp) & 7, which makes every 16-lane phase of the fragment reads hit 16 distinct bank groups
// for all taps and tile shapes used (exhaustive check: tools/perf/halo_check.py). B tiles: as in the kernel above.
// Main loop. A first version of this kernel kept tap, ring stage and addresses as run-time state: per 16 MFMAs a
// wave also issued ~90 SALU,
// ~90 VALU, 16 LDS and 5 VMEM instructions -- 12.7 other instructions per MFMA. A wave issues at most one instruction
// per 4 cycles, so the K-step cost ~900 issue cycles against 512 cycles of MFMA work: the loop was ISSUE-bound (and the
// LDS, L2 and DMA counters all sat below 25 %). This form removes the bookkeeping instead of hiding it:
//   * the 9 taps of two consecutive chunks (18 K-steps) are fully unrolled, so tap, halo buffer, ring stage (9 % 3 == 0)
//     and fragment-register set are compile-time constants;
//   * every LDS address is precomputed: a_addr[i][tap] holds the swizzled byte address of the lane's A row for each tap
//     (the k16 sub-step only XORs bits 5-6: ((2kk + h) ^ swz) << 4 == ((h ^ swz) << 4) ^ (kk << 5)), b_addr[j][kk] the
//     B row, with the ring stage as the instruction's immediate offset; the halo buffer flip is one add per address
//     per chunk;
//   * DMA instructions take the per-tap part of their source offset as the SGPR soffset operand (no VALU), are spread
//     between the MFMAs, and the ring is 3 stages with a constant s_waitcnt vmcnt(BP) (each step issues its optional
//     halo piece BEFORE its weight pieces, so "all but the newest BP" always covers the tile about to be read); tiles
//     past the end of the reduction are issued anyway (clamped to the last chunk) to keep the count constant;
//   * PIPE (one wave per SIMD, 128x128 tile): the fragment reads of step s are issued before the MFMAs of step s-1
//     (two register sets), so LDS latency never stalls the MFMA pipe.
#define H2_DMA(rsrc, voff, soff, ldsaddr)                                                                             \
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), \
                 "s"(soff), "s"(ldsaddr)                                                                              \
                 : "memory")

// ---------------------------------------------------------------- folded 1x1 shortcut: extra K-chunks behind the 3x3 reduction
// A residual block's result is conv2(h) + c_sc(x) (reference model/resnet_generator_app_v2.py:664-678,
// model/rcnn_discriminator_app.py:317-344). As its own launch the 1x1 shortcut is bound by the f32 result it writes and
// that conv2's epilogue reads back (a 32x32, 256 -> 512 shortcut moves 84 MB in 27 us for 1 GFLOP); here it is sc_Ci / 64
// more K-steps of conv2's tile: after the 3x3 reduction the workgroup stages, per 64 input channels, the BM centre pixels of
// the shortcut's input ([BM][64] operand rows, read at (y >> sc_up2, x >> sc_up2): nearest upsampling of the generator
// blocks) and BN rows of the 1x1 pack in the LDS the halo and the ring no longer need, and multiplies them into the same
// accumulators. Two stages when the kernel's LDS allocation holds them (p.sc_stages), else one.
// ReLU mask of the 3x3 part applied in ACCUMULATOR order (see ScArgs::mask_first): lane l holds pixel (l & 31) of each 32-row
// tile and 4 consecutive channels per register group, so a lane reads 8 bytes of the mask per group -- 32 rows apart per
// half-wave, i.e. uncoalesced per instruction, but every byte of the tile's mask rows is used across the groups (L2 hits after
// the first touch), once per workgroup. No pool / upsampling on these launches (conv1 of a discriminator block has neither).
template <int TM, int TN>
__device__ __forceinline__ void conv_mask_first(const ConvArgs& p, ScArgsPtr sc, f32x16_t (&acc)[TM][TN], int wrow, int wcol, int lane,
                                                int tile_r, int tile_c, int n0, int rows_total) {
    const bf16_t* __restrict__ M = reinterpret_cast<const bf16_t*>(sc->mask_first);
    const float s = sc->pre_scale;
    const int m = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int py, px;
        idx2pix(wrow + i * 32 + m, p.hw_shift, 0, py, px);
        const int r = tile_r * p.PH + py;
        const size_t rowoff = ((size_t)r * p.Wo + tile_c * p.PW + px) * p.Co;
        const bool live = r < rows_total;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wcol + j * 32 + 8 * g + 4 * h;
                float mk[4] = {0.f, 0.f, 0.f, 0.f};
                if (live && n < p.Co) Op4<bf16_t>::load(M + rowoff + n, mk);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = mk[e] > 0.f ? acc[i][j][4 * g + e] * s : 0.f;
            }
    }
}

template <int BM, int BN, int TM, int TN, int THREADS = 256>
__device__ __forceinline__ void conv_sc_tail(const ConvArgs& p, ScArgsPtr sc, f32x16_t (&acc)[TM][TN], char* smem, unsigned smem_addr, int tid, int lane,
                                             int wv, int wrow, int wcol, int tile_r, int tile_c, int n0, int rows_total, bool second) {
    constexpr int RPP = THREADS / 8, AP = BM / RPP, BPS = BN / RPP;   // the workgroup stages RPP rows x 128 bytes per pass
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile geometry");
    constexpr unsigned ASZ = BM * 128u, STG = (BM + BN) * 128u, OOB = 0x80000000u;
    const u32x4_t rx = make_rsrc(sc->x, sc->x_bytes), rw = make_rsrc(second ? sc->w_b : sc->w, sc->w_bytes);
    const int sc_Hi = sc->Hi, sc_Wi = sc->Wi, sc_Ci = sc->Ci, sc_up2 = sc->up2, sc_Kpad = sc->Kpad;
    const int lrow = tid >> 3, hh = lane >> 5;
    unsigned aoff[AP], boff[BPS];
#pragma unroll
    for (int q = 0; q < AP; ++q) {
        const int row = lrow + RPP * q;
        int py, px;
        idx2pix(row, p.hw_shift, 0, py, px);
        const int gr = tile_r * p.PH + py, x = tile_c * p.PW + px;
        const int b = fastdiv(gr, p.mg_ho), y = gr - b * p.Ho;
        const int ch = (tid & 7) ^ ig2_swz(row);
        aoff[q] = (gr < rows_total && x < p.Wo)
                      ? (unsigned)((((b * sc_Hi + (y >> sc_up2)) * sc_Wi + (x >> sc_up2)) * sc_Ci + ch * 8) * 2)
                      : OOB;
    }
#pragma unroll
    for (int q = 0; q < BPS; ++q) {
        const int row = lrow + RPP * q;
        boff[q] = (unsigned)(((n0 + row) * sc_Kpad + ((tid & 7) ^ ig2_swz(row)) * 8) * 2);
    }
    unsigned a_rd[TM], b_rd[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wrow + i * 32 + (lane & 31);
        a_rd[i] = (unsigned)(row * 128 + ((hh ^ ig2_swz(row)) << 4));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wcol + j * 32 + (lane & 31);
        b_rd[j] = ASZ + (unsigned)(row * 128 + ((hh ^ ig2_swz(row)) << 4));
    }
    const int nch = sc_Ci >> 6;
    const bool two = sc->stages >= 2;
#define SC_ISSUE(C, S)                                                                                                 \
    {                                                                                                                  \
        const unsigned so_ = (unsigned)(C) * 128u, base_ = smem_addr + (unsigned)(S) * STG + (unsigned)wv * 1024u;     \
        _Pragma("unroll") for (int q_ = 0; q_ < AP; ++q_) H2_DMA(rx, aoff[q_], so_, base_ + (unsigned)q_ * (RPP * 128u));     \
        _Pragma("unroll") for (int q_ = 0; q_ < BPS; ++q_) H2_DMA(rw, boff[q_], so_, base_ + ASZ + (unsigned)q_ * (RPP * 128u)); \
    }
    __syncthreads();   // every wave is past its last fragment read of the 3x3 reduction: halo and ring are free
    SC_ISSUE(0, 0)
    for (int c = 0; c < nch; ++c) {
        const unsigned st = two ? (unsigned)(c & 1) * STG : 0u;
        if (two && c + 1 < nch) {
            SC_ISSUE(c + 1, (c + 1) & 1)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP + BPS) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            bf16x8_t fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(smem + st + (a_rd[i] ^ (unsigned)(kk << 5)));
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(smem + st + (b_rd[j] ^ (unsigned)(kk << 5)));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb[j]),   // weights first: transposed tile
                        __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa[i]), acc[i][j], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // the stage is free again
        asm volatile("" ::: "memory");
        if (!two && c + 1 < nch) SC_ISSUE(c + 1, 0)
    }
#undef SC_ISSUE
}

// SC: the instantiation that folds a block's 1x1 shortcut behind the 3x3 reduction (conv_sc_tail). A separate instantiation:
// with the tail compiled into every kernel the 128x64 tile (168-VGPR cap) spilled 37 registers to scratch in its epilogue,
// and a kernel that uses scratch at all pays for it at dispatch -- every launch, folded or not, took 2.6 us longer.
template <int BM, int BN, int WM, int WN, int NSB, bool PIPE, bool H1 = false, int ABL = 0, bool SC = false>   // H1: ONE halo buffer, refilled at each chunk boundary
__global__ __launch_bounds__(WM* WN * 64, (BM == 128 && BN == 64) ? 3 : 2) void conv_halo2_kernel(ConvArgs p) {   // ABL: ablations for tools/perf (results are wrong): 1 no weight DMA, 2 no DMA, 3 no DMA + no fragment reads, 4 no barrier
    typedef bf16_t T;
    constexpr int THREADS = WM * WN * 64, NW = WM * WN;
    constexpr int BK = 64, SZ = 2;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int RPP = THREADS / 8, BP = BN / RPP;
    constexpr int HPMAX = 7;
    constexpr unsigned OOB = 0x80000000u;
    constexpr unsigned BSTAGE = BN * 128u;
    static_assert(BN % RPP == 0 && BP >= 1 && BP <= 4, "tile geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    L2I_TR(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int nblk = p.tiles_m * p.tiles_n;
    const int split = blockIdx.x / nblk;
    // (ROI heads: the live rows are compacted to the FRONT, so the contiguous-run-per-XCD remap would give all live tiles
    //  to the first XCDs and leave the others idle; the dispatcher's own round robin spreads them evenly)
    const int bid = (p.nimg && !p.roi_remap) ? (int)blockIdx.x - split * nblk : xcd_remap(blockIdx.x - split * nblk, nblk);
    const int tile_m = fastdiv(bid, p.mg_tn), tile_n = bid - tile_m * p.tiles_n;
    const int tile_r = fastdiv(tile_m, p.mg_tc), tile_c = tile_m - tile_r * p.tiles_c;
    const int n0 = tile_n * BN;
    const int rows_total = p.B * p.Ho;
    const bool second = p.half_rows > 0 && tile_r * p.PH >= p.half_rows;   // dual launch (ConvArgs::w_b): the second half's pack
    const int rows_live = p.nimg ? min(p.half_rows > 0 ? p.half_rows : rows_total, *p.nimg * p.Ho) + (second ? p.half_rows : 0) : rows_total;   // (scalar load)
    const bool tile_dead = tile_r * p.PH >= rows_live;   // every row belongs to a dead image: no reduction, zeros out
    const u32x4_t rsrc_x = make_rsrc(p.x, p.x_bytes), rsrc_w = make_rsrc(second ? p.w_b : p.w, p.w_bytes);
    const unsigned smem_addr = lds_addr_of(smem);
    const unsigned halo_bytes = (unsigned)p.halo_pieces * 1024u;
    const unsigned ring_off = (H1 ? 1u : 2u) * halo_bytes;
    const unsigned halo_flip = H1 ? 0u : halo_bytes;

    // ---- halo pieces of this wave (piece q = wv + NW q; lane -> halo row / chunk), as in conv_halo_kernel
    unsigned hoff[HPMAX];
    int hlim[HPMAX];
#pragma unroll
    for (int q = 0; q < HPMAX; ++q) {
        const int piece = wv + NW * q;
        const int h = piece * 8 + (lane >> 3), pch = lane & 7;
        const int sp = fastdiv(h, p.mg_subh), rem = h - sp * p.SUBH;
        const int hy = fastdiv(rem, p.mg_p), hx = rem - hy * p.P;
        const int gr0 = tile_r * p.PH + sp * p.PHs;
        const int b = fastdiv(gr0, p.mg_ho), y0 = gr0 - b * p.Ho, x0 = tile_c * p.PW;
        const int iy = (y0 >> p.up2) + hy - 1, ix = (x0 >> p.up2) + hx - 1;
        const bool ok = h < p.HR && hx < p.HWd && gr0 < rows_total && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        const int lch = pch ^ (((hx >> 1) + 4 * hy + 2 * sp) & 7);
        hoff[q] = ok ? (unsigned)(((b * p.Hi + iy) * p.Wi + ix) * p.Ci + lch * 8) * SZ : OOB;
        hlim[q] = p.Ci - lch * 8;
    }
    const int nq = (p.halo_pieces - wv + NW - 1) / NW;   // pieces this wave owns (wave-uniform)
    // ---- B rows
    const int lrow = tid >> 3;
    const int lchunk = (tid & 7) ^ ig2_swz(lrow);
    unsigned b_off[BP];
#pragma unroll
    for (int q = 0; q < BP; ++q) b_off[q] = (unsigned)((n0 + lrow + RPP * q) * p.Kpad + lchunk * 8) * SZ;
    const unsigned ring_w = smem_addr + ring_off + (unsigned)wv * 1024u;   // this wave's 1 KB slot of a pass
    const unsigned halo_w = smem_addr + (unsigned)wv * 1024u;

    // ---- LDS addresses of the fragments
    const int wrow = (wave / WN) * (BM / WM), wcol = (wave % WN) * (BN / WN);
    const int hh = lane >> 5;
    unsigned a_addr[TM][9];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int py, px;
        idx2pix(wrow + i * 32 + (lane & 31), p.hw_shift, 0, py, px);
        const int sp = py >> p.sub_shift, pyl = py & (p.PHs - 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            int hy, hx;
            if (p.up2) { hy = ((pyl + ky - 1) >> 1) + 1; hx = ((px + kx - 1) >> 1) + 1; }
            else { hy = pyl + ky; hx = px + kx; }
            const int swz = ((hx >> 1) + 4 * hy + 2 * sp) & 7;
            a_addr[i][tap] = (unsigned)((sp * p.SUBH + hy * p.P + hx) * 128 + ((hh ^ swz) << 4));
        }
    }
    unsigned b_addr[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wcol + j * 32 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) b_addr[j][kk] = ring_off + (unsigned)(row * 128 + (((kk * 2 + hh) ^ ig2_swz(row)) << 4));
    }

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- reduction range of this split: chunks [c_begin, c_end)
    const int nchunks_all = (p.Ci + BK - 1) / BK, cper = p.ks_per / 9;
    const int c_begin = split * cper, c_end = tile_dead ? c_begin : min(nchunks_all, c_begin + cper);
    const int last_cb = (c_end - 1) * BK;
    const int ci2 = p.Ci * SZ;

    bf16x8_t fa[2][4][TM], fb[2][4][TN];   // fragment register sets (only set 0 without PIPE)

#define H2_READS(SET, TAP, STG)                                                                                        \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                                 \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                 \
            fa[SET][kk][i] = *reinterpret_cast<const bf16x8_t*>(smem + (a_addr[i][TAP] ^ (unsigned)(kk << 5)));        \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                                 \
            fb[SET][kk][j] = *reinterpret_cast<const bf16x8_t*>(smem + b_addr[j][kk] + (STG) * BSTAGE);                \
    }
#define H2_MFMA(SET, KK, I, J)                                                                                         \
    acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                                               \
        __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb[SET][KK][J]),   /* weights first: transposed tile, see conv_epilogue */ \
        __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa[SET][KK][I]), acc[I][J], 0, 0, 0)
#define H2_MFMA4(SET, KK)                                                                                              \
    {                                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) H2_MFMA(SET, KK, i_, j_); \
    }
    // One K-step. S = step within the unrolled pair of chunks (0..17): tap = S % 9, halo buffer = S / 9, stage = tap % 3.
    // cb_cur: channel base of the current chunk; cb_nxt: of the chunk after it (clamped).
#define H2_STEP(S, cb_cur, cb_nxt, DO_MFMA_PREV)                                                                       \
    {                                                                                                                  \
        constexpr int TAP = (S) % 9, HB = (S) / 9, STG = (S) % NSB;            /* 18 % NSB == 0 for NSB 2, 3 */        \
        constexpr int AHEAD = NSB - 1;                                                                                 \
        constexpr int TAP2 = (TAP + AHEAD) % 9, STG2 = ((S) + AHEAD) % NSB;                                            \
        constexpr int SET = PIPE ? ((S)&1) : 0, PSET = PIPE ? (((S) + 1) & 1) : 0;                                     \
        if (PIPE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSB == 3 ? BP : 0) : "memory");                                       \
        if (!PIPE) { _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) asm volatile("" : "+v"(a_addr[i_][TAP])); }   /* keep the 72 XOR-ed copies out of registers */ \
        if (ABL != 4) __builtin_amdgcn_s_barrier();                                                                    \
        asm volatile("" ::: "memory");                                                                                 \
        if (H1 && TAP == 0 && (DO_MFMA_PREV) && ABL < 2) {   /* single halo buffer: everyone is past tap 8, refill and wait */ \
            _Pragma("unroll") for (int q_ = 0; q_ < HPMAX; ++q_) if (q_ < nq) {                                        \
                const unsigned vo_ = (cb_cur) < hlim[q_] ? hoff[q_] : OOB;                                             \
                H2_DMA(rsrc_x, vo_, (unsigned)((cb_cur)*SZ), halo_w + (unsigned)(NW * q_) * 1024u);                     \
            }                                                                                                          \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                           \
            __builtin_amdgcn_s_barrier();                                                                              \
            asm volatile("" ::: "memory");                                                                             \
        }                                                                                                              \
        if (ABL != 3) { H2_READS(SET, TAP, STG) }                                                                     \
        else { _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) { _Pragma("unroll") for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(fa[SET][kk][i])); _Pragma("unroll") for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(fb[SET][kk][j])); } } \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        const int cb2 = TAP + AHEAD >= 9 ? (cb_nxt) : (cb_cur);                                                        \
        const unsigned kadd = (unsigned)(TAP2 * ci2 + cb2 * SZ);                                                       \
        if (!PIPE || (DO_MFMA_PREV)) { H2_MFMA4(PSET, 0); }                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if (!H1 && TAP < HPMAX && TAP < nq && (ABL < 2 || ABL == 4)) {                                                 \
            const unsigned vo = (cb_nxt) < hlim[TAP] ? hoff[TAP] : OOB;                                                \
            H2_DMA(rsrc_x, vo, (unsigned)((cb_nxt)*SZ), halo_w + (unsigned)(1 - HB) * halo_bytes + (unsigned)(NW * TAP) * 1024u); \
        }                                                                                                              \
        if (ABL == 0 || ABL == 4) H2_DMA(rsrc_w, b_off[0], kadd, ring_w + STG2 * BSTAGE);                               \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if (!PIPE || (DO_MFMA_PREV)) { H2_MFMA4(PSET, 1); }                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if (BP > 1 && (ABL == 0 || ABL == 4)) H2_DMA(rsrc_w, b_off[BP > 1 ? 1 : 0], kadd, ring_w + STG2 * BSTAGE + 1u * RPP * 128u); \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if (!PIPE || (DO_MFMA_PREV)) { H2_MFMA4(PSET, 2); }                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if (BP > 2 && (ABL == 0 || ABL == 4)) H2_DMA(rsrc_w, b_off[BP > 2 ? 2 : 0], kadd, ring_w + STG2 * BSTAGE + 2u * RPP * 128u); \
        if (BP > 3 && (ABL == 0 || ABL == 4)) H2_DMA(rsrc_w, b_off[BP > 3 ? 3 : 0], kadd, ring_w + STG2 * BSTAGE + 3u * RPP * 128u); \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if (!PIPE || (DO_MFMA_PREV)) { H2_MFMA4(PSET, 3); }                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    }

    L2I_TR(1);
    if (c_begin < c_end) {
        // prologue: the whole first halo, then the weight tiles of taps 0 and 1
        {
            const int cb0 = c_begin * BK;
#pragma unroll
            for (int q = 0; q < HPMAX; ++q)
                if (q < nq) {
                    const unsigned vo = cb0 < hlim[q] ? hoff[q] : OOB;
                    H2_DMA(rsrc_x, vo, (unsigned)(cb0 * SZ), halo_w + (unsigned)(NW * q) * 1024u);
                }
#pragma unroll
            for (int t = 0; t < NSB - 1; ++t) {
                const unsigned kadd = (unsigned)(t * ci2 + cb0 * SZ);
#pragma unroll
                for (int q = 0; q < BP; ++q) H2_DMA(rsrc_w, b_off[q], kadd, ring_w + (unsigned)t * BSTAGE + (unsigned)(q * RPP) * 128u);
            }
        }
        bool first = true;
        int pending = 0;   // PIPE: fragment set holding the not-yet-multiplied last step
        for (int c = c_begin; c < c_end; c += 2) {
            const int cbA = c * BK;
            const int cbB = min(cbA + BK, last_cb), cbC = min(cbA + 2 * BK, last_cb);
            H2_STEP(0, cbA, cbB, !first)
            H2_STEP(1, cbA, cbB, true)
            H2_STEP(2, cbA, cbB, true)
            H2_STEP(3, cbA, cbB, true)
            H2_STEP(4, cbA, cbB, true)
            H2_STEP(5, cbA, cbB, true)
            H2_STEP(6, cbA, cbB, true)
            H2_STEP(7, cbA, cbB, true)
            H2_STEP(8, cbA, cbB, true)
            first = false;
            pending = 0;
            // halo buffer flip for the second chunk of the pair (and back afterwards)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) a_addr[i][tap] += halo_flip;
            if (c + 1 < c_end) {
                H2_STEP(9, cbB, cbC, true)
                H2_STEP(10, cbB, cbC, true)
                H2_STEP(11, cbB, cbC, true)
                H2_STEP(12, cbB, cbC, true)
                H2_STEP(13, cbB, cbC, true)
                H2_STEP(14, cbB, cbC, true)
                H2_STEP(15, cbB, cbC, true)
                H2_STEP(16, cbB, cbC, true)
                H2_STEP(17, cbB, cbC, true)
                pending = 1;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) a_addr[i][tap] -= halo_flip;
        }
        if (PIPE) {   // the last step's fragments are still waiting for their MFMAs
            if (pending) { H2_MFMA4(1, 0); H2_MFMA4(1, 1); H2_MFMA4(1, 2); H2_MFMA4(1, 3); }
            else { H2_MFMA4(0, 0); H2_MFMA4(0, 1); H2_MFMA4(0, 2); H2_MFMA4(0, 3); }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped extra tiles must not land after the LDS is reused / the wave ends
    }
#undef H2_STEP
#undef H2_MFMA4
#undef H2_MFMA
#undef H2_READS
    if constexpr (SC) {
        const ScArgsPtr sc = late_sc();
        if (sc->x && !tile_dead && sc->mask_first) conv_mask_first<TM, TN>(p, sc, acc, wrow, wcol, lane, tile_r, tile_c, n0, rows_total);   // (every split: the mask is linear)
        if (sc->x && !tile_dead && split == p.splits - 1)   // (a split launch: the shortcut's K-steps belong to the last split)
            conv_sc_tail<BM, BN, TM, TN, THREADS>(p, sc, acc, smem, smem_addr, tid, lane, wv, wrow, wcol, tile_r, tile_c, n0, rows_total, second);
    }
    L2I_TR(2);
    if (p.splits > 1 && p.part) { if (!tile_dead) conv_store_partial<TM, TN>(p, acc, bid, split, NW, wave, lane); }
    else if (p.splits > 1) conv_epilogue_splitk<T, TM, TN>(p, acc, wrow, wcol, lane, wave, tile_r, tile_c, n0, split, rows_live, smem);
    else if (p.epi_lds) conv_epilogue_lds<T, TM, TN, (SC && TN == 1) ? 4 : 8, SC>(p, acc, wrow, wcol, lane, wave, tile_r, tile_c, n0, rows_total, rows_live, smem);   // (the folding 128x64 tile: 4 stages in flight keep it under its 168-VGPR cap without scratch)
    else conv_epilogue<T, TM, TN, SC>(p, acc, wrow, wcol, lane, tile_r, tile_c, n0, split, rows_total, rows_live);
    L2I_TR(3);
}

// ---------------------------------------------------------------- 256-pixel tiles: conv_halo3_kernel
// Measured on the kernel above (tools/perf/conv_abl.py, 32x32x512->512, back to back): 1830 TFLOP/s with neither DMA nor
// fragment reads, 1351 with the fragment reads, 1105 with everything -- the LDS fragment traffic and the weight stream,
// not the MFMA issue, are what a 128x128 tile of four 64x64 waves pays for. This kernel doubles the work per byte on
// both: a workgroup is still 4 waves (two workgroups per CU, one wave of each per SIMD), but its tile is 256 pixels x BN
// channels and a wave owns 64 pixels x ALL BN channels (TM = 2, TN = BN/32): per k16 sub-step 2 A + TN B fragment reads
// feed 2 TN MFMAs (0.75 reads per MFMA at BN = 128 instead of 1), the weight tile of a K-step is shared by 256 pixels
// instead of 128 (half the L2 -> LDS bytes per MFMA), and a K-step between two barriers is 32 MFMAs per wave, not 16.
//   * fragment reads are software-pipelined inside the K-step (two register sets, sub-step kk+2 is read under the MFMAs
//     of kk): a quarter of the fragment registers of "read everything first", which is what lets 128 accumulators + two
//     waves per SIMD fit;
//   * ONE halo buffer (refilled at the chunk boundary) + a 2-stage weight ring: 41.5 + 32 KB = two workgroups per CU;
//   * COMPACT halo when a sub-patch is a whole image (8x8 ROI maps, 16x16 maps, 8->16 upsampling): the one-pixel border
//     is all padding, so only the image's own rows are staged (4 ROI maps: 32 KB instead of 50) and an out-of-image tap
//     reads a 256-byte zero region at the bank slot the border row would have had (conflict-free by the same argument
//     as the bordered layout: tools/perf/halo_check.py);
//   * needs Ci % 64 == 0 (every layer this is used for); everything else as above (transposed accumulator, epilogue).
// (launch bounds: the 256 x 64 tile fits three workgroups per CU -- 161 VGPRs in round 2; without the bound the batched epilogue
//  took 207 and the ROI heads, which run on this tile, lost a quarter of their speed to the missing third workgroup)
template <int BN, int ABL = 0, bool PF = false, bool SC = false>   // SC: see conv_halo2_kernel
__global__ __launch_bounds__(256, BN == 64 ? 3 : 2) void conv_halo3_kernel(ConvArgs p) {
    typedef bf16_t T;
    constexpr int NW = 4, TM = 2, TN = BN / 32, BP = BN / 32, SZ = 2;
    constexpr int HPMAX = 11;                       // 18 x 18 halo rows = 41 KB-pieces over 4 waves
    constexpr unsigned OOB = 0x80000000u, BSTAGE = BN * 128u;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    L2I_TR(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int nblk = p.tiles_m * p.tiles_n;
    const int split = blockIdx.x / nblk;
    // (ROI heads: the live rows are compacted to the FRONT, so the contiguous-run-per-XCD remap would give all live tiles
    //  to the first XCDs and leave the others idle; the dispatcher's own round robin spreads them evenly)
    const int bid = (p.nimg && !p.roi_remap) ? (int)blockIdx.x - split * nblk : xcd_remap(blockIdx.x - split * nblk, nblk);
    const int tile_m = fastdiv(bid, p.mg_tn), tile_n = bid - tile_m * p.tiles_n;
    const int tile_r = fastdiv(tile_m, p.mg_tc), tile_c = tile_m - tile_r * p.tiles_c;
    const int n0 = tile_n * BN;
    const int rows_total = p.B * p.Ho;
    const bool second = p.half_rows > 0 && tile_r * p.PH >= p.half_rows;   // dual launch (ConvArgs::w_b): the second half's pack
    const int rows_live = p.nimg ? min(p.half_rows > 0 ? p.half_rows : rows_total, *p.nimg * p.Ho) + (second ? p.half_rows : 0) : rows_total;
    const bool tile_dead = tile_r * p.PH >= rows_live;
    const u32x4_t rsrc_x = make_rsrc(p.x, p.x_bytes), rsrc_w = make_rsrc(second ? p.w_b : p.w, p.w_bytes);
    const unsigned smem_addr = lds_addr_of(smem);
    const unsigned halo_bytes = (unsigned)p.halo_pieces * 1024u;
    const unsigned ring_off = halo_bytes, zero_off = halo_bytes + 2u * BSTAGE;   // [halo][ring x 2][256 zero bytes]
    const int border = p.compact ? 0 : 1;          // rows of border stored around a sub-patch
    const int Hh = (p.up2 ? p.PHs >> 1 : p.PHs), Wh = (p.up2 ? p.PW >> 1 : p.PW);   // sub-patch extent at input resolution

    if (tid < 16) *reinterpret_cast<uint4*>(smem + zero_off + tid * 16) = make_uint4(0, 0, 0, 0);

    // ---- halo pieces of this wave: piece = wv + 4 q, lane -> (row, 16-byte chunk). Stored row h = sp*SUBH + hy*P + hx;
    // (hyp, hxp) are the coordinates in the bordered frame (border row / column = 0), which the swizzle is written in.
    unsigned hoff[HPMAX];
#pragma unroll
    for (int q = 0; q < HPMAX; ++q) {
        const int piece = wv + NW * q;
        const int h = piece * 8 + (lane >> 3), pch = lane & 7;
        const int sp = fastdiv(h, p.mg_subh), rem = h - sp * p.SUBH;
        const int hy = fastdiv(rem, p.mg_p), hx = rem - hy * p.P;
        const int hyp = hy + 1 - border, hxp = hx + 1 - border;
        const int gr0 = tile_r * p.PH + sp * p.PHs;
        const int b = fastdiv(gr0, p.mg_ho), y0 = gr0 - b * p.Ho, x0 = tile_c * p.PW;
        const int iy = (y0 >> p.up2) + hyp - 1, ix = (x0 >> p.up2) + hxp - 1;
        const bool ok = h < p.HR && hx < p.HWd && gr0 < rows_total && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        const int lch = pch ^ (((hxp >> 1) + 4 * hyp + 2 * sp) & 7);
        hoff[q] = ok ? (unsigned)(((b * p.Hi + iy) * p.Wi + ix) * p.Ci + lch * 8) * SZ : OOB;
    }
    const int nq = (p.halo_pieces - wv + NW - 1) / NW;
    // ---- weight rows: 256 threads fill 32 rows x 128 bytes per pass
    const int lrow = tid >> 3;
    const int lchunk = (tid & 7) ^ ig2_swz(lrow);
    unsigned b_off[BP];
#pragma unroll
    for (int q = 0; q < BP; ++q) b_off[q] = (unsigned)((n0 + lrow + 32 * q) * p.Kpad + lchunk * 8) * SZ;
    const unsigned ring_w = smem_addr + ring_off + (unsigned)wv * 1024u;
    const unsigned halo_w = smem_addr + (unsigned)wv * 1024u;

    // ---- fragment addresses. A: this lane's pixel of each 32-row MFMA tile, for each tap; B: row (lane & 31) of column
    // tile 0 for each k16 sub-step (column tile j adds 4096 bytes, ring stage BSTAGE: immediates)
    const int wrow = wave * 64;
    const int hh = lane >> 5;
    unsigned a_addr[TM][9];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int py, px;
        idx2pix(wrow + i * 32 + (lane & 31), p.hw_shift, 0, py, px);
        const int sp = py >> p.sub_shift, pyl = py & (p.PHs - 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            int hyp, hxp;   // bordered-frame coordinates of the input pixel this tap reads
            if (p.up2) { hyp = ((pyl + ky - 1) >> 1) + 1; hxp = ((px + kx - 1) >> 1) + 1; }
            else { hyp = pyl + ky; hxp = px + kx; }
            const unsigned sl = (unsigned)((hh ^ (((hxp >> 1) + 4 * hyp + 2 * sp) & 7)) << 4);
            const bool inside = hyp >= 1 && hyp <= Hh && hxp >= 1 && hxp <= Wh;
            if (border || inside) a_addr[i][tap] = (unsigned)((sp * p.SUBH + (hyp - 1 + border) * p.P + (hxp - 1 + border)) * 128) + sl;
            else a_addr[i][tap] = zero_off + (unsigned)(((hxp - 1) & 1) * 128) + sl;   // (parity of the row the border pixel would have had)
        }
    }
    unsigned b_addr[4];
    {
        const int row = lane & 31;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) b_addr[kk] = ring_off + (unsigned)(row * 128 + (((kk * 2 + hh) ^ ig2_swz(row)) << 4));
    }

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nchunks_all = p.Ci >> 6, cper = p.ks_per / 9;
    const int c_begin = split * cper, c_end = tile_dead ? c_begin : min(nchunks_all, c_begin + cper);
    const int last_cb = (c_end - 1) * 64;
    const int ci2 = p.Ci * SZ;

    bf16x8_t fa[2][TM], fb[2][TN];   // two fragment register sets: sub-step kk lives in set kk & 1

#define H3_RD(KK, TAP, STG)                                                                                            \
    {                                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_)                                                              \
            fa[(KK)&1][i_] = *reinterpret_cast<const bf16x8_t*>(smem + (a_addr[i_][TAP] ^ (unsigned)((KK) << 5)));     \
        _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_)                                                              \
            fb[(KK)&1][j_] = *reinterpret_cast<const bf16x8_t*>(smem + b_addr[KK] + (STG) * BSTAGE + j_ * 4096u);      \
    }
#define H3_MM(KK)                                                                                                      \
    {                                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_)            \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                                     \
                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb[(KK)&1][j_]),   /* weights first: transposed tile */ \
                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa[(KK)&1][i_]), acc[i_][j_], 0, 0, 0); \
    }
#define H3_HALO(cb)                                                                                                    \
    _Pragma("unroll") for (int q_ = 0; q_ < HPMAX; ++q_) if (q_ < nq) {                                                \
        H2_DMA(rsrc_x, hoff[q_], (unsigned)((cb)*SZ), halo_w + (unsigned)(NW * q_) * 1024u);                           \
    }
    // One K-step. S = step within the unrolled pair of chunks (0..17): tap = S % 9, ring stage = S & 1.
#define H3_STEP(S, cb_cur, cb_nxt, REFILL)                                                                             \
    {                                                                                                                  \
        constexpr int TAP = (S) % 9, STG = (S)&1, TAP2 = (TAP + 1) % 9;                                                \
        if (TAP == 0 && (REFILL)) {   /* every wave is past tap 8 of the previous chunk: refill the halo */              \
            __builtin_amdgcn_s_barrier();                                                                              \
            asm volatile("" ::: "memory");                                                                             \
            if (ABL < 2) { H3_HALO(cb_cur) }                                                                           \
        }                                                                                                              \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   /* this step's weight tile (issued a step ago) and the halo */ \
        if (ABL != 4) __builtin_amdgcn_s_barrier();                                                                    \
        asm volatile("" ::: "memory");                                                                                 \
        const unsigned kadd = (unsigned)(TAP2 * ci2 + (TAP == 8 ? (cb_nxt) : (cb_cur)) * SZ);                          \
        { _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) asm volatile("" : "+v"(a_addr[i_][TAP])); }   /* keep the 72 XOR-ed copies out of registers */ \
        if (ABL != 3) { H3_RD(0, TAP, STG) H3_RD(1, TAP, STG) }                                                        \
        else { _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) { _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) asm volatile("" : "+v"(fa[s_][i_])); _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) asm volatile("" : "+v"(fb[s_][j_])); } } \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H3_MM(0)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if (ABL == 0 || ABL == 4) {                                                                                    \
            H2_DMA(rsrc_w, b_off[0], kadd, ring_w + (1 - STG) * BSTAGE);                                               \
            if (BP > 1) H2_DMA(rsrc_w, b_off[BP > 1 ? 1 : 0], kadd, ring_w + (1 - STG) * BSTAGE + 4096u);              \
        }                                                                                                              \
        if (ABL != 3) { H3_RD(2, TAP, STG) }                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H3_MM(1)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if (ABL == 0 || ABL == 4) {                                                                                    \
            if (BP > 2) H2_DMA(rsrc_w, b_off[BP > 2 ? 2 : 0], kadd, ring_w + (1 - STG) * BSTAGE + 2 * 4096u);          \
            if (BP > 3) H2_DMA(rsrc_w, b_off[BP > 3 ? 3 : 0], kadd, ring_w + (1 - STG) * BSTAGE + 3 * 4096u);          \
        }                                                                                                              \
        if (ABL != 3) { H3_RD(3, TAP, STG) }                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H3_MM(2)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H3_MM(3)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    }

    // ---- PF: the K-step's barrier sits BETWEEN its third and fourth k16 sub-step, and the first sub-step's fragments of
    // the NEXT K-step are read right behind it (under the last 8 MFMAs): a wave comes out of the barrier with MFMAs to
    // issue and its next fragments already in flight, instead of "barrier -> reads -> wait ~200 cycles -> first MFMA".
    // Invariant at the top of a step: set 0 holds (or is receiving) sub-step 0's fragments, this step's weight tile is
    // visible, the next one is in flight. Mid-step: this wave's reads of the current stage are complete (lgkmcnt(0)),
    // its pieces of the next tile landed (vmcnt(0)), barrier -> the current stage is free for the tile after next.
    // A chunk boundary (tap 8 -> tap 0) refills the halo behind the mid-step barrier of tap 8 and pays one ordinary
    // "wait, barrier, read" at tap 0.
#define H3P_STEP(S, cb_cur, cb_nxt, cb_nn, HEAD)                                                                       \
    {                                                                                                                  \
        constexpr int TAP = (S) % 9, STG = (S)&1, TAP1 = (TAP + 1) % 9, TAP2 = (TAP + 2) % 9;                          \
        if (HEAD) {   /* first step of a chunk: the halo (and this step's tile) must have landed; no prefetched set */ \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                           \
            __builtin_amdgcn_s_barrier();                                                                              \
            asm volatile("" ::: "memory");                                                                             \
            { _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) asm volatile("" : "+v"(a_addr[i_][TAP])); }          \
            H3_RD(0, TAP, STG)                                                                                         \
        }                                                                                                              \
        { _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) asm volatile("" : "+v"(a_addr[i_][TAP])); }              \
        H3_RD(1, TAP, STG)                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H3_MM(0)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H3_RD(2, TAP, STG)                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H3_MM(1)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H3_RD(3, TAP, STG)                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H3_MM(2)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                    \
        __builtin_amdgcn_s_barrier();                                                                                  \
        asm volatile("" ::: "memory");                                                                                 \
        {   /* the tile after next -> the stage this step has just finished reading */                                 \
            const unsigned kadd2 = (unsigned)(TAP2 * ci2 + (TAP >= 7 ? (TAP == 7 ? (cb_nxt) : (cb_nxt)) : (cb_cur)) * SZ); \
            _Pragma("unroll") for (int q_ = 0; q_ < BP; ++q_) H2_DMA(rsrc_w, b_off[q_], kadd2, ring_w + STG * BSTAGE + (unsigned)q_ * 4096u); \
        }                                                                                                              \
        if (TAP == 8) { H3_HALO(cb_nxt) }   /* nobody reads the halo any more: refill it for the next chunk */           \
        else {                                                                                                         \
            { _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) asm volatile("" : "+v"(a_addr[i_][TAP1])); }         \
            H3_RD(0, TAP1, 1 - STG)                                                                                    \
        }                                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H3_MM(3)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    }
    L2I_TR(1);
    if (PF) {
        if (c_begin < c_end) {
            {   // prologue: the first halo and the weight tiles of taps 0 and 1
                const int cb0 = c_begin * 64;
                H3_HALO(cb0)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const unsigned kadd0 = (unsigned)(t * ci2 + cb0 * SZ);
#pragma unroll
                    for (int q = 0; q < BP; ++q) H2_DMA(rsrc_w, b_off[q], kadd0, ring_w + (unsigned)t * BSTAGE + (unsigned)q * 4096u);
                }
            }
            for (int c = c_begin; c < c_end; c += 2) {
                const int cbA = c * 64;
                const int cbB = min(cbA + 64, last_cb), cbC = min(cbA + 128, last_cb);
                H3P_STEP(0, cbA, cbB, cbC, true)
                H3P_STEP(1, cbA, cbB, cbC, false)
                H3P_STEP(2, cbA, cbB, cbC, false)
                H3P_STEP(3, cbA, cbB, cbC, false)
                H3P_STEP(4, cbA, cbB, cbC, false)
                H3P_STEP(5, cbA, cbB, cbC, false)
                H3P_STEP(6, cbA, cbB, cbC, false)
                H3P_STEP(7, cbA, cbB, cbC, false)
                H3P_STEP(8, cbA, cbB, cbC, false)
                if (c + 1 < c_end) {
                    H3P_STEP(9, cbB, cbC, cbC, true)
                    H3P_STEP(10, cbB, cbC, cbC, false)
                    H3P_STEP(11, cbB, cbC, cbC, false)
                    H3P_STEP(12, cbB, cbC, cbC, false)
                    H3P_STEP(13, cbB, cbC, cbC, false)
                    H3P_STEP(14, cbB, cbC, cbC, false)
                    H3P_STEP(15, cbB, cbC, cbC, false)
                    H3P_STEP(16, cbB, cbC, cbC, false)
                    H3P_STEP(17, cbB, cbC, cbC, false)
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    } else
    if (c_begin < c_end) {
        {   // prologue: the first halo and the weight tile of tap 0
            const int cb0 = c_begin * 64;
            H3_HALO(cb0)
            const unsigned kadd0 = (unsigned)(cb0 * SZ);
#pragma unroll
            for (int q = 0; q < BP; ++q) H2_DMA(rsrc_w, b_off[q], kadd0, ring_w + (unsigned)q * 4096u);
        }
        bool first = true;
        for (int c = c_begin; c < c_end; c += 2) {
            const int cbA = c * 64;
            const int cbB = min(cbA + 64, last_cb), cbC = min(cbA + 128, last_cb);
            H3_STEP(0, cbA, cbB, !first)
            H3_STEP(1, cbA, cbB, false)
            H3_STEP(2, cbA, cbB, false)
            H3_STEP(3, cbA, cbB, false)
            H3_STEP(4, cbA, cbB, false)
            H3_STEP(5, cbA, cbB, false)
            H3_STEP(6, cbA, cbB, false)
            H3_STEP(7, cbA, cbB, false)
            H3_STEP(8, cbA, cbB, false)
            first = false;
            if (c + 1 < c_end) {
                H3_STEP(9, cbB, cbC, true)
                H3_STEP(10, cbB, cbC, false)
                H3_STEP(11, cbB, cbC, false)
                H3_STEP(12, cbB, cbC, false)
                H3_STEP(13, cbB, cbC, false)
                H3_STEP(14, cbB, cbC, false)
                H3_STEP(15, cbB, cbC, false)
                H3_STEP(16, cbB, cbC, false)
                H3_STEP(17, cbB, cbC, false)
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped extra tile must not land after the LDS is reused / the wave ends
    }
#undef H3_STEP
#undef H3P_STEP
#undef H3_HALO
#undef H3_MM
#undef H3_RD
    if constexpr (SC) {
        const ScArgsPtr sc = late_sc();
        if (sc->x && !tile_dead && split == p.splits - 1) conv_sc_tail<256, BN, TM, TN, 256>(p, sc, acc, smem, smem_addr, tid, lane, wv, wrow, 0, tile_r, tile_c, n0, rows_total, second);
    }
    L2I_TR(2);
    if (p.splits > 1 && p.part) { if (!tile_dead) conv_store_partial<TM, TN>(p, acc, bid, split, NW, wave, lane); }
    else if (p.splits > 1) conv_epilogue_splitk<T, TM, TN>(p, acc, wrow, 0, lane, wave, tile_r, tile_c, n0, split, rows_live, smem);
    else if (p.epi_lds) conv_epilogue_lds<T, TM, TN, 4, SC>(p, acc, wrow, 0, lane, wave, tile_r, tile_c, n0, rows_total, rows_live, smem);   // (4: the 256 x 64 tile runs three workgroups per CU on 168 VGPRs)
    else conv_epilogue<T, TM, TN, SC>(p, acc, wrow, 0, lane, tile_r, tile_c, n0, split, rows_total, rows_live);
    L2I_TR(3);
}

// ---------------------------------------------------------------- 256 x 256 tiles, ONE workgroup of 8 waves per CU: conv_halo8_kernel
// Round 5 experiment: the "one workgroup per CU, eight waves, deep weight pipeline" structure of the GEMM
// guide on the convolution: tile = 256 pixels (a 16 x 16 patch, or whole images of smaller maps) x 256 output channels, waves
// as 4 (pixels) x 2 (channels), a wave owns 64 x 128 (TM = 2, TN = 4: the per-wave step of conv_halo3_kernel<128>). Against two
// 4-wave workgroups of 256 x 128 per CU: the halo of a chunk is staged ONCE per CU instead of twice, DOUBLE-buffered (the next
// chunk's pieces ride on taps 0..5 of the current one: no refill stall at the chunk boundary, which the single-buffer 256-pixel
// kernel pays once per 9 K-steps), and the weight tile of a K-step (256 rows x 128 B = 32 KB) streams through a ring whose
// depth is a template parameter in HALF K-steps (NH half-tiles of 16 KB in flight or being read):
//   NH = 4: [k 0..31 | k 32..63] of step s being read, both halves of step s + 1 in flight (counted vmcnt: never 0 in the loop)
// LDS: 2 x 42 KB halo + NH x 16 KB ring + 256 zero bytes = 148.25 KB at NH = 4 -- one workgroup per CU, 256 VGPRs per wave.
// bf16, 3x3, Ci % 64 == 0, Co % 256 == 0 for full tiles (other Co: rows past Co read zeros through the descriptor).
template <bool SC = false, bool DEEP = false>   // DEEP: two barriers per K-step, three half-tiles of weights in flight, counted vmcnt (never 0 in the loop)
__global__ __launch_bounds__(512, 1) void conv_halo8_kernel(ConvArgs p) {
    typedef bf16_t T;
    constexpr int NW = 8, TM = 2, TN = 4, BN = 256, SZ = 2;
    constexpr int HPMAX = 6;                        // 42 halo pieces of 1 KB over 8 waves
    constexpr unsigned OOB = 0x80000000u, HSTAGE = BN * 64u;   // a half K-step of the weight tile: 256 rows x 64 bytes
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int nblk = p.tiles_m * p.tiles_n;
    const int bid = (p.nimg && !p.roi_remap) ? (int)blockIdx.x : xcd_remap(blockIdx.x, nblk);
    const int tile_m = fastdiv(bid, p.mg_tn), tile_n = bid - tile_m * p.tiles_n;
    const int tile_r = fastdiv(tile_m, p.mg_tc), tile_c = tile_m - tile_r * p.tiles_c;
    const int n0 = tile_n * BN;
    const int rows_total = p.B * p.Ho;
    const int rows_live = p.nimg ? min(rows_total, *p.nimg * p.Ho) : rows_total;
    const bool tile_dead = tile_r * p.PH >= rows_live;
    const u32x4_t rsrc_x = make_rsrc(p.x, p.x_bytes), rsrc_w = make_rsrc(p.w, p.w_bytes);
    const unsigned smem_addr = lds_addr_of(smem);
    const unsigned halo_bytes = (unsigned)p.halo_pieces * 1024u;
    const unsigned ring_off = 2u * halo_bytes, zero_off = ring_off + 4u * HSTAGE;   // [halo x 2][ring: 4 half-tiles][256 zero bytes]
    const int border = p.compact ? 0 : 1;
    const int Hh = (p.up2 ? p.PHs >> 1 : p.PHs), Wh = (p.up2 ? p.PW >> 1 : p.PW);

    if (tid < 16) *reinterpret_cast<uint4*>(smem + zero_off + tid * 16) = make_uint4(0, 0, 0, 0);

    // ---- halo pieces of this wave (as conv_halo3_kernel, eight waves)
    unsigned hoff[HPMAX];
#pragma unroll
    for (int q = 0; q < HPMAX; ++q) {
        const int piece = wv + NW * q;
        const int h = piece * 8 + (lane >> 3), pch = lane & 7;
        const int sp = fastdiv(h, p.mg_subh), rem = h - sp * p.SUBH;
        const int hy = fastdiv(rem, p.mg_p), hx = rem - hy * p.P;
        const int hyp = hy + 1 - border, hxp = hx + 1 - border;
        const int gr0 = tile_r * p.PH + sp * p.PHs;
        const int b = fastdiv(gr0, p.mg_ho), y0 = gr0 - b * p.Ho, x0 = tile_c * p.PW;
        const int iy = (y0 >> p.up2) + hyp - 1, ix = (x0 >> p.up2) + hxp - 1;
        const bool ok = h < p.HR && hx < p.HWd && gr0 < rows_total && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        const int lch = pch ^ (((hxp >> 1) + 4 * hyp + 2 * sp) & 7);
        hoff[q] = ok ? (unsigned)(((b * p.Hi + iy) * p.Wi + ix) * p.Ci + lch * 8) * SZ : OOB;
    }
    const int nq = (p.halo_pieces - wv + NW - 1) / NW;
    // ---- weight rows of a HALF K-step: 256 rows x 64 bytes = 4 chunks of 16 bytes per row; 512 threads x 16 bytes = 128 rows per
    // pass, two passes. Row r's chunk c is stored at chunk (c ^ ((r >> 2) & 3)) (64-byte rows: four rows per 256-byte bank row).
    const int lrow = tid >> 2;
    const int lchunk = (tid & 3) ^ ((lrow >> 2) & 3);
    unsigned b_off[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) b_off[q] = (unsigned)((n0 + lrow + 128 * q) * p.Kpad + lchunk * 8) * SZ;   // (+ 64 bytes for the second half of a K-step)
    const unsigned ring_w = smem_addr + ring_off + (unsigned)wv * 1024u;   // this wave's 16 rows x 64 bytes of a pass
    const unsigned halo_w = smem_addr + (unsigned)wv * 1024u;

    // ---- fragment addresses
    const int wrow = (wave >> 1) * 64, wcol = (wave & 1) * 128;
    const int hh = lane >> 5;
    unsigned a_addr[TM][9];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int py, px;
        idx2pix(wrow + i * 32 + (lane & 31), p.hw_shift, 0, py, px);
        const int sp = py >> p.sub_shift, pyl = py & (p.PHs - 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            int hyp, hxp;
            if (p.up2) { hyp = ((pyl + ky - 1) >> 1) + 1; hxp = ((px + kx - 1) >> 1) + 1; }
            else { hyp = pyl + ky; hxp = px + kx; }
            const unsigned sl = (unsigned)((hh ^ (((hxp >> 1) + 4 * hyp + 2 * sp) & 7)) << 4);
            const bool inside = hyp >= 1 && hyp <= Hh && hxp >= 1 && hxp <= Wh;
            if (border || inside) a_addr[i][tap] = (unsigned)((sp * p.SUBH + (hyp - 1 + border) * p.P + (hxp - 1 + border)) * 128) + sl;
            else a_addr[i][tap] = zero_off + (unsigned)(((hxp - 1) & 1) * 128) + sl;
        }
    }
    // B fragment of k16 sub-step kk (0..3): half = kk >> 1, within the half's 64-byte rows chunk (kk & 1) * 2 + hh, swizzled
    unsigned b_addr[2];
    {
        const int row = wcol + (lane & 31);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) b_addr[k2] = ring_off + (unsigned)(row * 64 + (((k2 * 2 + hh) ^ ((row >> 2) & 3)) << 4));
    }

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nchunks = tile_dead ? 0 : (p.Ci >> 6);
    const int ci2 = p.Ci * SZ;
    bf16x8_t fa[2][TM], fb[2][TN];

    // half-tile h of K-step (tap, cb) -> ring slot SLOT (0..3): two passes of 128 rows
#define H8_W(TAPX, CBX, HALF, SLOT)                                                                                    \
    {                                                                                                                  \
        const unsigned kadd_ = (unsigned)((TAPX) * ci2 + (CBX) * SZ + (HALF) * 64);                                    \
        H2_DMA(rsrc_w, b_off[0], kadd_, ring_w + (unsigned)(SLOT) * HSTAGE);                                            \
        H2_DMA(rsrc_w, b_off[1], kadd_, ring_w + (unsigned)(SLOT) * HSTAGE + 8192u);                                    \
    }
#define H8_RD(KK, TAP, SLOT0)                                                                                          \
    {                                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_)                                                              \
            fa[(KK)&1][i_] = *reinterpret_cast<const bf16x8_t*>(smem + (a_addr[i_][TAP] ^ (unsigned)((KK) << 5)));     \
        _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_)                                                              \
            fb[(KK)&1][j_] = *reinterpret_cast<const bf16x8_t*>(smem + b_addr[(KK)&1] + (unsigned)((SLOT0) + ((KK) >> 1)) * HSTAGE + j_ * 2048u); \
    }
#define H8_MM(KK)                                                                                                      \
    {                                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_)            \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                                     \
                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb[(KK)&1][j_]),                        \
                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa[(KK)&1][i_]), acc[i_][j_], 0, 0, 0); \
    }
    // One K-step S (0..17 over a pair of chunks): tap = S % 9, halo buffer = S / 9, ring slots 2 (S & 1), 2 (S & 1) + 1.
    // Top of the step: this step's two half-tiles (issued one step ago) have landed for every wave [vmcnt + barrier]; the
    // barrier also says every wave has finished reading the other two slots (step S - 1), so the next step's halves go there.
#define H8_STEP(S, cb_cur, cb_nxt)                                                                                     \
    {                                                                                                                  \
        constexpr int TAP = (S) % 9, HB = (S) / 9, SL = 2 * ((S)&1), TAP2 = (TAP + 1) % 9;                             \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                               \
        __builtin_amdgcn_s_barrier();                                                                                  \
        asm volatile("" ::: "memory");                                                                                 \
        { _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) asm volatile("" : "+v"(a_addr[i_][TAP])); }              \
        H8_RD(0, TAP, SL) H8_RD(1, TAP, SL)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H8_MM(0)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H8_W(TAP2, (TAP == 8 ? (cb_nxt) : (cb_cur)), 0, 2 - SL)                                                        \
        H8_RD(2, TAP, SL)                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H8_MM(1)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H8_W(TAP2, (TAP == 8 ? (cb_nxt) : (cb_cur)), 1, 3 - SL)                                                        \
        if (TAP < HPMAX && TAP < nq) {   /* one piece of the NEXT chunk's halo into the other buffer */                \
            H2_DMA(rsrc_x, hoff[TAP], (unsigned)((cb_nxt)*SZ), halo_w + (unsigned)(1 - HB) * halo_bytes + (unsigned)(NW * TAP) * 1024u); \
        }                                                                                                              \
        H8_RD(3, TAP, SL)                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H8_MM(2)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H8_MM(3)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    }
    // DEEP. Half-tiles are numbered h = 2 S + {0, 1} and live in ring slot h & 3; half h + 3 is issued the moment half h - 1 has been
    // read by every wave (the barrier of half-phase h), i.e. THREE half-tiles are in flight while one is read -- 1.5 K-steps of
    // latency cover instead of one -- and a wave never waits for more than it needs: in-order completion lets `s_waitcnt vmcnt(N)`
    // with N = the DMAs issued AFTER the half about to be read (two per later half-tile, one per halo piece: all compile-time).
    //   phase A (half 2S):     wait N_A = 4 + [tap(S-2) < 6] + [tap(S-1) < 6]; barrier; issue half 2S + 3; sub-steps 0, 1
    //   phase B (half 2S + 1): wait N_B = 4 + [tap(S-1) < 6];                  barrier; issue half 2S + 4 + this tap's halo piece; sub-steps 2, 3
#define H8D_STEP(S, cb_cur, cb_nxt)                                                                                    \
    {                                                                                                                  \
        constexpr int TAP = (S) % 9, HB = (S) / 9, SL = 2 * ((S)&1), TAP1 = (TAP + 1) % 9, TAP2 = (TAP + 2) % 9;       \
        constexpr int HP1 = ((TAP + 8) % 9) < HPMAX ? 1 : 0, HP2 = ((TAP + 7) % 9) < HPMAX ? 1 : 0;                    \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + HP1 + HP2) : "memory");                                           \
        __builtin_amdgcn_s_barrier();                                                                                  \
        asm volatile("" ::: "memory");                                                                                 \
        H8_W(TAP1, (TAP == 8 ? (cb_nxt) : (cb_cur)), 1, 3 - SL)                                                        \
        { _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) asm volatile("" : "+v"(a_addr[i_][TAP])); }              \
        H8_RD(0, TAP, SL) H8_RD(1, TAP, SL)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H8_MM(0)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H8_MM(1)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + HP1) : "memory");                                                 \
        __builtin_amdgcn_s_barrier();                                                                                  \
        asm volatile("" ::: "memory");                                                                                 \
        H8_W(TAP2, (TAP >= 7 ? (cb_nxt) : (cb_cur)), 0, SL)                                                            \
        if (TAP < HPMAX) {   /* one piece of the NEXT chunk's halo (a wave with fewer pieces re-loads its first one: the count stays static) */ \
            const bool own_ = TAP < nq;                                                                                \
            H2_DMA(rsrc_x, own_ ? hoff[TAP] : hoff[0], (unsigned)((cb_nxt)*SZ),                                        \
                   halo_w + (unsigned)(1 - HB) * halo_bytes + (unsigned)(NW * (own_ ? TAP : 0)) * 1024u);              \
        }                                                                                                              \
        H8_RD(2, TAP, SL) H8_RD(3, TAP, SL)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H8_MM(2)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        H8_MM(3)                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    }
    if (DEEP && nchunks > 0) {
        {   // prologue: the first halo, then half-tiles 0, 1, 2 (tap 0 both halves, tap 1 first half)
#pragma unroll
            for (int q = 0; q < HPMAX; ++q)
                if (q < nq) H2_DMA(rsrc_x, hoff[q], 0u, halo_w + (unsigned)(NW * q) * 1024u);
            H8_W(0, 0, 0, 0)
            H8_W(0, 0, 1, 1)
            H8_W(1, 0, 0, 2)
        }
        const int last_cb = (nchunks - 1) * 64;
        for (int c = 0; c < nchunks; c += 2) {
            const int cbA = c * 64;
            const int cbB = min(cbA + 64, last_cb), cbC = min(cbA + 128, last_cb);
            H8D_STEP(0, cbA, cbB) H8D_STEP(1, cbA, cbB) H8D_STEP(2, cbA, cbB) H8D_STEP(3, cbA, cbB) H8D_STEP(4, cbA, cbB)
            H8D_STEP(5, cbA, cbB) H8D_STEP(6, cbA, cbB) H8D_STEP(7, cbA, cbB) H8D_STEP(8, cbA, cbB)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
                    if (a_addr[i][tap] < zero_off) a_addr[i][tap] += halo_bytes;
            if (c + 1 < nchunks) {
                H8D_STEP(9, cbB, cbC) H8D_STEP(10, cbB, cbC) H8D_STEP(11, cbB, cbC) H8D_STEP(12, cbB, cbC) H8D_STEP(13, cbB, cbC)
                H8D_STEP(14, cbB, cbC) H8D_STEP(15, cbB, cbC) H8D_STEP(16, cbB, cbC) H8D_STEP(17, cbB, cbC)
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
                    if (a_addr[i][tap] < zero_off) a_addr[i][tap] -= halo_bytes;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#undef H8D_STEP
    if (!DEEP && nchunks > 0) {
        {   // prologue: the first halo, the two half-tiles of tap 0
#pragma unroll
            for (int q = 0; q < HPMAX; ++q)
                if (q < nq) H2_DMA(rsrc_x, hoff[q], 0u, halo_w + (unsigned)(NW * q) * 1024u);
            H8_W(0, 0, 0, 0)
            H8_W(0, 0, 1, 1)
        }
        const int last_cb = (nchunks - 1) * 64;
        for (int c = 0; c < nchunks; c += 2) {
            const int cbA = c * 64;
            const int cbB = min(cbA + 64, last_cb), cbC = min(cbA + 128, last_cb);
            H8_STEP(0, cbA, cbB) H8_STEP(1, cbA, cbB) H8_STEP(2, cbA, cbB) H8_STEP(3, cbA, cbB) H8_STEP(4, cbA, cbB)
            H8_STEP(5, cbA, cbB) H8_STEP(6, cbA, cbB) H8_STEP(7, cbA, cbB) H8_STEP(8, cbA, cbB)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
                    if (a_addr[i][tap] < zero_off) a_addr[i][tap] += halo_bytes;
            if (c + 1 < nchunks) {
                H8_STEP(9, cbB, cbC) H8_STEP(10, cbB, cbC) H8_STEP(11, cbB, cbC) H8_STEP(12, cbB, cbC) H8_STEP(13, cbB, cbC)
                H8_STEP(14, cbB, cbC) H8_STEP(15, cbB, cbC) H8_STEP(16, cbB, cbC) H8_STEP(17, cbB, cbC)
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap)
                    if (a_addr[i][tap] < zero_off) a_addr[i][tap] -= halo_bytes;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#undef H8_STEP
#undef H8_MM
#undef H8_RD
#undef H8_W
    if constexpr (SC) {
        const ScArgsPtr sc = late_sc();
        if (sc->x && !tile_dead) conv_sc_tail<256, BN, TM, TN, 512>(p, sc, acc, smem, smem_addr, tid, lane, wv, wrow, wcol, tile_r, tile_c, n0, rows_total, false);
    }
    if (p.epi_lds) conv_epilogue_lds<T, TM, TN, 4, SC>(p, acc, wrow, wcol, lane, wave, tile_r, tile_c, n0, rows_total, rows_live, smem);
    else conv_epilogue<T, TM, TN, SC>(p, acc, wrow, wcol, lane, tile_r, tile_c, n0, 0, rows_total, rows_live);
}

// ---------------------------------------------------------------- split-K reduce that carries the epilogue
// One workgroup per output tile, the SAME wave / lane -> element mapping as the kernel that stored the partial tiles
// (conv_store_partial), so the sum of the splits lands in the accumulator layout and conv_epilogue_lds / conv_epilogue run
// unchanged on it: alpha, both biases, ReLU mask, residual, 2x2 pool, operand copies, batch statistics, live-row count.
// ConvArgs must be the FIRST kernel argument (late_sc() reads the shortcut's bias from the kernel-argument segment).
template <int BM, int BN, int WM, int WN, bool SC>
__global__ __launch_bounds__(WM* WN * 64) void conv_split_reduce_kernel(ConvArgs p, int nsplit) {
    typedef bf16_t T;
    constexpr int NW = WM * WN, TM = BM / (WM * 32), TN = BN / (WN * 32);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = p.tiles_m * p.tiles_n;
    const int bid = (p.nimg && !p.roi_remap) ? (int)blockIdx.x : xcd_remap(blockIdx.x, nblk);   // (the producers' block -> tile map: same XCD)
    const int tile_m = fastdiv(bid, p.mg_tn), tile_n = bid - tile_m * p.tiles_n;
    const int tile_r = fastdiv(tile_m, p.mg_tc), tile_c = tile_m - tile_r * p.tiles_c;
    const int n0 = tile_n * BN;
    const int rows_total = p.B * p.Ho;
    const bool second = p.half_rows > 0 && tile_r * p.PH >= p.half_rows;
    const int rows_live = p.nimg ? min(p.half_rows > 0 ? p.half_rows : rows_total, *p.nimg * p.Ho) + (second ? p.half_rows : 0) : rows_total;
    const bool tile_dead = tile_r * p.PH >= rows_live;
    const int wrow = (wave / WN) * (BM / WM), wcol = (wave % WN) * (BN / WN);
    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    if (!tile_dead) {
        const float4* src = reinterpret_cast<const float4*>(p.part) + ((size_t)bid * nsplit * NW + wave) * (TM * TN * 4 * 64) + lane;
        for (int s_ = 0; s_ < nsplit; ++s_) {
            float4 v[TM * TN * 4];
#pragma unroll
            for (int q = 0; q < TM * TN * 4; ++q) v[q] = src[q * 64];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 w_ = v[(i * TN + j) * 4 + g];
                        acc[i][j][4 * g] += w_.x; acc[i][j][4 * g + 1] += w_.y; acc[i][j][4 * g + 2] += w_.z; acc[i][j][4 * g + 3] += w_.w;
                    }
            src += (size_t)NW * (TM * TN * 4 * 64);
        }
    }
    if (p.epi_lds) conv_epilogue_lds<T, TM, TN, 8, SC>(p, acc, wrow, wcol, lane, wave, tile_r, tile_c, n0, rows_total, rows_live, smem);
    else conv_epilogue<T, TM, TN, SC>(p, acc, wrow, wcol, lane, tile_r, tile_c, n0, 0, rows_total, rows_live);
}

// Batch statistics from the epilogue (ConvArgs::stat_part): the launcher fixes the row count of the partial matrix -- one row per wave along M
// of every pixel tile -- and reports it to conv2d_impl, which runs rows_fold behind the launch.
static int stat_rows(ConvArgs& a, int wm) {
    if (!a.stat_part) return L2I_OK;
    a.stat_wm = wm;
    const long long rows = (long long)a.tiles_m * wm;
    if (rows * 2 * a.Co > a.stat_cap) return L2I_ERR_ARG;
    if (a.stat_rows_out) *a.stat_rows_out = (int)rows;
    return L2I_OK;
}

// Split planning for the stored-partials path: a launch whose tiles fill less than 3/4 of the resident workgroup slots
// (`per_cu` workgroups x 256 CUs) is split along K so that tiles x splits is at most ONE full round, each split keeping at least
// two 64-channel chunks (18 K-steps); the partial tiles must fit the caller's scratch. Returns 1 when the launch stays whole.
static int g_last_splits = 1;   // debug aid (l2i_debug_occupancy(100, 0)): splits of the last halo launch, negative when combined by atomics
static int g_part_mode = -1;   // L2I_CONV_PART: 0 off (atomics where the old rule splits), 1 on (default)
static int plan_part_splits(const ConvArgs& a, int nblk, int nchunks, int per_cu, long long tile_floats) {
    if (g_part_mode < 0) g_part_mode = getenv("L2I_CONV_PART") ? atoi(getenv("L2I_CONV_PART")) : 1;
    if (!g_part_mode || !a.scratch || nchunks < 4) return 1;
    if (!a.out && !a.out_op && !a.out_op_raw) return 1;
    const int slots = per_cu * 256;
    if (4 * nblk > 3 * slots) return 1;
    int splits = slots / nblk;
    if (splits > nchunks / 2) splits = nchunks / 2;
    while (splits > 1 && (long long)nblk * splits * tile_floats > a.scratch_floats) --splits;
    return splits < 1 ? 1 : splits;
}

template <int BM, int BN, int WM, int WN, bool SC>
static int launch_split_reduce(const ConvArgs& a, int nblk, size_t lds, hipStream_t stream) {
    ConvArgs r = a;
    const int nsplit = a.splits;
    r.splits = 1;
    static bool ready = false;
    if (!ready) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)conv_split_reduce_kernel<BM, BN, WM, WN, SC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        ready = true;
    }
    L2I_LAUNCH(0, (conv_split_reduce_kernel<BM, BN, WM, WN, SC>), dim3(nblk), dim3(WM * WN * 64), lds, stream, r, nsplit);
    return l2i_check_launch();
}

static int ilog2(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

static int g_split_target = 512;   // tuning hook: workgroups a split-K launch aims for
static int g_no_sc_fold = 0;       // A/B switch (L2I_SC_FOLD=0): a block's 1x1 shortcut always runs as its own launch
static int g_force_splits = 0;     // tuning hook (l2i_set_conv_config(2000 + n)): split count of the 256-pixel-tile kernel
static int g_generic_cfg = -1;     // tuning hook (l2i_set_conv_config(3000 + n)): tile configuration of the generic kernel only
// Launch one instantiation; LDS rings above 64 KB need the opt-in attribute (set once per instantiation).
template <typename T, int BM, int BN, int WM, int WN, int NS, int HK = 0>
static int launch_cfg(ConvArgs a, hipStream_t stream) {
    constexpr int BK = Mma<T>::BK >> HK;
    constexpr size_t ring = (size_t)NS * (BM + BN) * (HK ? 64 : 128);
    constexpr int TMc = BM / (WM * 32), TNc = BN / (WN * 32);
    constexpr size_t epi = (size_t)WM * WN * ((TNc == 1 && TMc % 2 == 0) ? 2 : 1) * 32 * (BN / WN + 4) * 4;   // conv_epilogue_lds: one (narrow tiles: two) 32-pixel slabs per wave
    constexpr size_t lds = ring > epi ? ring : epi;
    a.chunk_major = (a.KH == 3 && a.Ci >= BK) ? 1 : 0;
    a.nks = a.chunk_major ? 9 * ((a.Ci + BK - 1) / BK) : a.Kpad / BK;
    a.PH = BM / a.PW;
    if (!a.lin && (a.PH & 1)) return L2I_ERR_ARG;
    if (a.half_rows && (a.lin || a.half_rows % a.PH)) return L2I_ERR_ARG;   // dual launch: no tile may straddle the two halves
    const int rows = a.B * a.Ho;
    a.tiles_m = ((rows + a.PH - 1) / a.PH) * a.tiles_c;
    a.tiles_n = (a.Co + BN - 1) / BN;
    if (stat_rows(a, WM) != L2I_OK) return L2I_ERR_ARG;
    a.mg_tn = fastdiv_magic(a.tiles_n); a.mg_tc = fastdiv_magic(a.tiles_c); a.mg_ho = fastdiv_magic(a.Ho);
    a.mg_subh = fastdiv_magic(a.SUBH); a.mg_p = fastdiv_magic(a.P);
    const int nblk = a.tiles_m * a.tiles_n;
    const int nks = a.nks;
    // split-K for small grids with a long reduction (D block5/6, G res1/res2, ROI heads): fill the 256 CUs
    int splits = 1;
    // (bf16 operands only: the splits of this kernel are combined by float atomics, whose order changes from run to run -- the exact-f32
    //  mode stays bit-reproducible, as the reference's CPU convolutions are, and pays for it on its small grids; round 6)
    if (sizeof(T) == 2 && a.out && !a.out_op && !a.out_op_raw && !a.stat_part && nblk < 192 && nks >= 16) {
        splits = (g_split_target + nblk - 1) / nblk;
        const int min_steps = nblk < 32 ? 4 : 16;   // >= 16 K-steps per split (shorter ones are all prologue + atomic epilogue), except for the
        if (splits > nks / min_steps) splits = nks / min_steps;   // handful-of-tiles Linear layers, which otherwise run on 6 CUs
        if (splits < 1) splits = 1;
    }
    a.ks_per = (nks + splits - 1) / splits;
    a.splits = (nks + a.ks_per - 1) / a.ks_per;
    a.part = nullptr;
    // Round 6: the splits of this kernel too are STORED partial tiles summed (in order, with the whole epilogue) by conv_split_reduce_kernel when
    // the caller's scratch has room -- the tile configurations a split launch can take (128 x 128 and 128 x 64, four waves) have their reduce
    // kernels from the halo path. Without scratch: float atomics into the cleared result, as before.
    constexpr bool can_part = sizeof(T) == 2 && WM == 2 && WN == 2 && BM == 128 && (BN == 128 || BN == 64) && !HK;
    if (a.splits > 1 && can_part && a.scratch && (long long)nblk * a.splits * BM * BN <= a.scratch_floats && !a.nimg && !a.half_rows) a.part = a.scratch;
    if (a.splits > 1 && !a.part) {
        const size_t bytes = sizeof(float) * (size_t)a.B * (a.Ho >> a.pool2) * (a.Wo >> a.pool2) * a.Co;
        if (l2i_zero_async(a.out, bytes, stream) != hipSuccess) return L2I_ERR_LAUNCH;
    }
    static bool ready = false;
    if (!ready) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<T, BM, BN, WM, WN, NS, HK>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        ready = true;
    }
    (void)BK;
    L2I_LAUNCH(0, (conv_igemm_kernel<T, BM, BN, WM, WN, NS, HK>), dim3(nblk * a.splits), dim3(WM * WN * 64), lds, stream, a);
    if constexpr (can_part) {
        if (a.part) return launch_split_reduce<BM, BN, WM, WN, false>(a, nblk, epi, stream);
    }
    return l2i_check_launch();
}

// Halo kernel launch (bf16, 3x3, Ci >= 64, Wo >= 8, no upsample into 8-wide maps). Returns -100 when the shape is
// not covered so that the caller falls through to the generic kernel.
template <typename T> static int launch_conv(ConvArgs& a, hipStream_t stream);

// A shortcut that cannot ride on this launch (split-K, the generic kernel, channel counts that are not multiples of 64)
// runs as its own 1x1 launch into `sc_out` and comes back as the residual -- what the caller would have done itself.
template <typename T>
static int sc_unfold(ConvArgs& a, hipStream_t stream) {
    if (!a.sc.x) return L2I_OK;
    const bool mf = a.sc.mask_first != nullptr;   // data-gradient fold: the 1x1 launch carries the caller's residual, the 3x3 launch its own alpha + mask
    if (!a.sc.out || (a.res && !mf)) return L2I_ERR_ARG;
    ConvArgs s = a;
    s.x = a.sc.x; s.w = a.sc.w; s.w_b = a.sc.w_b; s.bias = a.sc.bias; s.res = mf ? a.res : nullptr; s.relu_mask = nullptr;
    if (mf) { a.relu_mask = a.sc.mask_first; a.alpha = a.alpha * a.sc.pre_scale; a.sc.mask_first = nullptr; }   // (a.alpha was the tail's: s keeps it)
    s.sc.mask_first = nullptr;
    s.out = a.sc.out; s.out_op = nullptr; s.out_op_raw = nullptr; s.stat_part = nullptr;
    s.Hi = a.sc.Hi; s.Wi = a.sc.Wi; s.Ci = a.sc.Ci; s.KH = 1; s.up2 = a.sc.up2; s.Kpad = a.sc.Kpad; s.relu_op = 0;
    s.sc.x = nullptr; s.sc.w = nullptr; s.sc.w_b = nullptr; s.sc.bias = nullptr; s.sc.out = nullptr;
    s.SUBH = 1; s.P = 1;
    const int rc = launch_conv<T>(s, stream);
    a.res = a.sc.out;
    a.sc.x = nullptr; a.sc.w = nullptr; a.sc.w_b = nullptr; a.sc.bias = nullptr;
    return rc;
}

// LDS a folded shortcut needs behind a BM x BN tile: one stage of (BM + BN) 128-byte rows, two when the allocation has room
static void sc_plan(ConvArgs& a, size_t& lds, int BM, int BN) {
    if (!a.sc.x) return;
    const size_t stg = (size_t)(BM + BN) * 128;
    if (lds < stg) lds = stg;
    a.sc.stages = lds >= 2 * stg ? 2 : 1;
}

template <int BM, int BN, int WM, int WN, int NSB, bool PIPE, bool H1 = false, int ABL = 0, bool CAN_SC = false>   // CAN_SC: the folding twin of this tile is compiled
static int launch_halo2(ConvArgs a, hipStream_t stream) {
    a.PH = BM / a.PW;
    if (a.half_rows % a.PH) return L2I_ERR_ARG;   // dual launch: no tile may straddle the two halves
    a.PHs = a.PH < a.Ho ? a.PH : a.Ho;
    a.sub_shift = ilog2(a.PHs);
    const int nsp = a.PH / a.PHs;
    a.HWd = (a.up2 ? a.PW / 2 : a.PW) + 2;
    const int hh = (a.up2 ? a.PHs / 2 : a.PHs) + 2;
    a.P = (a.HWd + 1) & ~1;
    a.SUBH = hh * a.P;
    a.HR = nsp * a.SUBH;
    a.halo_pieces = (a.HR + 7) / 8;
    if (a.halo_pieces > 7 * WM * WN) return -100;
    size_t lds = (size_t)(H1 ? 1 : 2) * a.halo_pieces * 1024 + (size_t)NSB * BN * 128;
    if (lds > 160 * 1024) return -100;
    constexpr int TMc = BM / (WM * 32), TNc = BN / (WN * 32);
    constexpr size_t epi = (size_t)WM * WN * ((TNc == 1 && TMc % 2 == 0) ? 2 : 1) * 32 * (BN / WN + 4) * 4;   // conv_epilogue_lds: one (narrow tiles: two) 32-pixel slabs per wave
    if (lds < epi) lds = epi;
    const int nchunks = (a.Ci + 63) / 64;
    a.nks = 9 * nchunks;
    const int rows = a.B * a.Ho;
    a.tiles_m = ((rows + a.PH - 1) / a.PH) * a.tiles_c;
    a.tiles_n = (a.Co + BN - 1) / BN;
    if (stat_rows(a, WM) != L2I_OK) return L2I_ERR_ARG;
    a.mg_tn = fastdiv_magic(a.tiles_n); a.mg_tc = fastdiv_magic(a.tiles_c); a.mg_ho = fastdiv_magic(a.Ho);
    a.mg_subh = fastdiv_magic(a.SUBH); a.mg_p = fastdiv_magic(a.P);
    const int nblk = a.tiles_m * a.tiles_n;
    int splits = 1;
    a.part = nullptr;
    // stored partial tiles + a reduce kernel that carries the epilogue (any epilogue option, folded shortcut included) ...
    const int psplits = ABL ? 1 : plan_part_splits(a, nblk, nchunks, (BM == 128 && BN == 64) ? 3 : 2, (long long)BM * BN);
    if (psplits > 1) { splits = psplits; a.part = a.scratch; }
    // ... else the round-1 rule: atomics into a zeroed plain f32 result
    else if (a.out && !a.out_op && !a.out_op_raw && !a.stat_part && nblk < 192 && nchunks >= 4) {
        splits = (g_split_target + nblk - 1) / nblk;
        if (splits > nchunks / 2) splits = nchunks / 2;   // >= 18 K-steps per split
        if (splits < 1) splits = 1;
    }
    const int cper = (nchunks + splits - 1) / splits;
    a.ks_per = 9 * cper;
    a.splits = (nchunks + cper - 1) / cper;
    if (a.splits == 1) a.part = nullptr;
    if (a.sc.x && (!CAN_SC || (a.splits > 1 && !a.part) || a.sc.Ci % 64 || g_no_sc_fold)) {
        const int rc = sc_unfold<bf16_t>(a, stream);
        if (rc != L2I_OK) return rc;
    }
    g_last_splits = a.part ? a.splits : -a.splits;
    sc_plan(a, lds, BM, BN);
    if (a.splits > 1 && !a.part) {
        const size_t bytes = sizeof(float) * (size_t)a.B * (a.Ho >> a.pool2) * (a.Wo >> a.pool2) * a.Co;
        if (l2i_zero_async(a.out, bytes, stream) != hipSuccess) return L2I_ERR_LAUNCH;
    }
    static bool ready = false;
    if (!ready) {
        (void)hipFuncSetAttribute((const void*)conv_halo2_kernel<BM, BN, WM, WN, NSB, PIPE, H1, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if constexpr (CAN_SC)
            (void)hipFuncSetAttribute((const void*)conv_halo2_kernel<BM, BN, WM, WN, NSB, PIPE, H1, ABL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        ready = true;
    }
    if constexpr (CAN_SC) {
        if (a.sc.x) {
            L2I_LAUNCH(0, (conv_halo2_kernel<BM, BN, WM, WN, NSB, PIPE, H1, ABL, true>), dim3(nblk * a.splits), dim3(WM * WN * 64), lds, stream, a);
            if (a.part) return launch_split_reduce<BM, BN, WM, WN, true>(a, nblk, epi, stream);
            return l2i_check_launch();
        }
    }
    L2I_LAUNCH(0, (conv_halo2_kernel<BM, BN, WM, WN, NSB, PIPE, H1, ABL>), dim3(nblk * a.splits), dim3(WM * WN * 64), lds, stream, a);
    if (a.part) return launch_split_reduce<BM, BN, WM, WN, false>(a, nblk, epi, stream);
    return l2i_check_launch();
}

// conv_halo3_kernel launch (bf16, 3x3, Ci % 64 == 0, Wo >= 8). Returns -100 when the shape is not covered.
template <int BN, int ABL = 0, bool PF = false, bool CAN_SC = false>
static int launch_halo3(ConvArgs a, hipStream_t stream, int force_splits = 0) {
    if (a.Ci % 64 || a.Wo < 4 || (a.up2 && a.Wo < 8)) return -100;
    if (force_splits == 0) force_splits = g_force_splits;
    a.PH = 256 / a.PW;
    if (a.half_rows % a.PH) return L2I_ERR_ARG;   // dual launch: no tile may straddle the two halves
    a.PHs = a.PH < a.Ho ? a.PH : a.Ho;
    a.sub_shift = ilog2(a.PHs);
    const int nsp = a.PH / a.PHs;
    const int Wh = a.up2 ? a.PW / 2 : a.PW, Hh = a.up2 ? a.PHs / 2 : a.PHs;
    a.compact = (a.PW == a.Wo && a.PHs == a.Ho) ? 1 : 0;
    if (a.compact) { a.HWd = Wh; a.P = Wh; a.SUBH = Hh * Wh; }
    else { a.HWd = Wh + 2; a.P = (a.HWd + 1) & ~1; a.SUBH = (Hh + 2) * a.P; }
    if ((a.P & 1) || (a.SUBH & 1)) return -100;
    a.HR = nsp * a.SUBH;
    a.halo_pieces = (a.HR + 7) / 8;
    if (a.halo_pieces > 44) return -100;
    size_t lds = (size_t)a.halo_pieces * 1024 + (size_t)2 * BN * 128 + 256;
    constexpr size_t epi = (size_t)4 * 32 * (BN + 4) * 4;   // conv_epilogue_lds: a 32-pixel slab of all BN channels per wave
    if (lds < epi) lds = epi;
    const int nchunks = a.Ci / 64;
    a.nks = 9 * nchunks;
    const int rows = a.B * a.Ho;
    a.tiles_m = ((rows + a.PH - 1) / a.PH) * a.tiles_c;
    a.tiles_n = (a.Co + BN - 1) / BN;
    if (stat_rows(a, 4) != L2I_OK) return L2I_ERR_ARG;
    a.mg_tn = fastdiv_magic(a.tiles_n); a.mg_tc = fastdiv_magic(a.tiles_c); a.mg_ho = fastdiv_magic(a.Ho);
    a.mg_subh = fastdiv_magic(a.SUBH); a.mg_p = fastdiv_magic(a.P);
    const int nblk = a.tiles_m * a.tiles_n;
    int splits = 1;
    a.part = nullptr;
    const int psplits = ABL ? 1 : plan_part_splits(a, nblk, nchunks, BN == 64 ? 3 : 2, 256LL * BN);   // (see launch_halo2)
    if (psplits > 1) { splits = psplits; a.part = a.scratch; }
    else if (a.out && !a.out_op && !a.out_op_raw && !a.stat_part && nblk < 256 && nchunks >= 4) {   // fill the 512 workgroup slots (two per CU); more than 8
        splits = (g_split_target + nblk / 2) / nblk;                           // splits lose to their atomics (tools/perf/conv_small.py)
        if (splits > nchunks / 2) splits = nchunks / 2;   // >= 18 K-steps per split
        if (splits > 8) splits = 8;
        if (splits < 1) splits = 1;
    }
    if (force_splits > 0 && (a.part || (a.out && !a.out_op && !a.out_op_raw && !a.stat_part))) splits = force_splits < nchunks ? force_splits : nchunks;
    const int cper = (nchunks + splits - 1) / splits;
    a.ks_per = 9 * cper;
    a.splits = (nchunks + cper - 1) / cper;
    if (a.splits == 1 || (long long)nblk * a.splits * 256 * BN > a.scratch_floats) a.part = nullptr;
    if (!a.part && a.splits > 1 && !(a.out && !a.out_op && !a.out_op_raw && !a.stat_part)) { a.splits = 1; a.ks_per = 9 * nchunks; }
    if (a.sc.x && (!CAN_SC || (a.splits > 1 && !a.part) || a.sc.Ci % 64 || g_no_sc_fold || a.sc.mask_first)) {   // (mask-first folds: 128-pixel tiles only)
        const int rc = sc_unfold<bf16_t>(a, stream);
        if (rc != L2I_OK) return rc;
    }
    g_last_splits = a.part ? a.splits : -a.splits;
    sc_plan(a, lds, 256, BN);
    if (a.splits > 1 && !a.part) {
        const size_t bytes = sizeof(float) * (size_t)a.B * (a.Ho >> a.pool2) * (a.Wo >> a.pool2) * a.Co;
        if (l2i_zero_async(a.out, bytes, stream) != hipSuccess) return L2I_ERR_LAUNCH;
    }
    static bool ready = false;
    if (!ready) {
        (void)hipFuncSetAttribute((const void*)conv_halo3_kernel<BN, ABL, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if constexpr (CAN_SC)
            (void)hipFuncSetAttribute((const void*)conv_halo3_kernel<BN, ABL, PF, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        ready = true;
    }
    if constexpr (CAN_SC) {
        if (a.sc.x) {
            L2I_LAUNCH(0, (conv_halo3_kernel<BN, ABL, PF, true>), dim3(nblk * a.splits), dim3(256), lds, stream, a);
            if (a.part) return launch_split_reduce<256, BN, 4, 1, true>(a, nblk, epi, stream);
            return l2i_check_launch();
        }
    }
    L2I_LAUNCH(0, (conv_halo3_kernel<BN, ABL, PF>), dim3(nblk * a.splits), dim3(256), lds, stream, a);
    if (a.part) return launch_split_reduce<256, BN, 4, 1, false>(a, nblk, epi, stream);
    return l2i_check_launch();
}

// conv_halo8_kernel launch (round-5 experiment; tuning configuration 40 / 41 only). Returns -100 when the shape is not covered.
template <bool CAN_SC = false, bool DEEP = false>
static int launch_halo8(ConvArgs a, hipStream_t stream) {
    if (a.Ci % 64 || a.Wo < 4 || (a.up2 && a.Wo < 8) || a.half_rows) return -100;
    a.PH = 256 / a.PW;
    a.PHs = a.PH < a.Ho ? a.PH : a.Ho;
    a.sub_shift = ilog2(a.PHs);
    const int nsp = a.PH / a.PHs;
    const int Wh = a.up2 ? a.PW / 2 : a.PW, Hh = a.up2 ? a.PHs / 2 : a.PHs;
    a.compact = (a.PW == a.Wo && a.PHs == a.Ho) ? 1 : 0;
    if (a.compact) { a.HWd = Wh; a.P = Wh; a.SUBH = Hh * Wh; }
    else { a.HWd = Wh + 2; a.P = (a.HWd + 1) & ~1; a.SUBH = (Hh + 2) * a.P; }
    if ((a.P & 1) || (a.SUBH & 1)) return -100;
    a.HR = nsp * a.SUBH;
    a.halo_pieces = (a.HR + 7) / 8;
    if (a.halo_pieces > 48 || a.halo_pieces < 8) return -100;
    size_t lds = (size_t)2 * a.halo_pieces * 1024 + (size_t)4 * 256 * 64 + 256;
    constexpr size_t epi = (size_t)8 * 32 * (128 + 4) * 4;
    if (lds < epi) lds = epi;
    a.nks = 9 * (a.Ci / 64);
    const int rows = a.B * a.Ho;
    a.tiles_m = ((rows + a.PH - 1) / a.PH) * a.tiles_c;
    a.tiles_n = (a.Co + 255) / 256;
    if (stat_rows(a, 4) != L2I_OK) return L2I_ERR_ARG;
    a.mg_tn = fastdiv_magic(a.tiles_n); a.mg_tc = fastdiv_magic(a.tiles_c); a.mg_ho = fastdiv_magic(a.Ho);
    a.mg_subh = fastdiv_magic(a.SUBH); a.mg_p = fastdiv_magic(a.P);
    const int nblk = a.tiles_m * a.tiles_n;
    a.splits = 1; a.ks_per = a.nks; a.part = nullptr;
    g_last_splits = -1;
    if (a.sc.x && (!CAN_SC || a.sc.Ci % 64 || g_no_sc_fold || a.sc.mask_first)) {
        const int rc = sc_unfold<bf16_t>(a, stream);
        if (rc != L2I_OK) return rc;
    }
    if (a.sc.x) {
        const size_t stg = (size_t)(256 + 256) * 128;
        if (lds < stg) lds = stg;
        a.sc.stages = lds >= 2 * stg ? 2 : 1;
    }
    static bool ready = false;
    if (!ready) {
        (void)hipFuncSetAttribute((const void*)conv_halo8_kernel<false, DEEP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if constexpr (CAN_SC) (void)hipFuncSetAttribute((const void*)conv_halo8_kernel<true, DEEP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        ready = true;
    }
    if constexpr (CAN_SC) {
        if (a.sc.x) {
            L2I_LAUNCH(0, (conv_halo8_kernel<true, DEEP>), dim3(nblk), dim3(512), lds, stream, a);
            return l2i_check_launch();
        }
    }
    L2I_LAUNCH(0, (conv_halo8_kernel<false, DEEP>), dim3(nblk), dim3(512), lds, stream, a);
    return l2i_check_launch();
}

// ---------------------------------------------------------------- 4x4 maps: weight-stationary split-K (conv_wstat_kernel)
// 3x3 convolutions on 4x4 maps (D block6, reference model/rcnn_discriminator_app.py:94-96: 1024 -> 1024 on 32 x 16 = 512 pixels)
// are all weights: 18.9 MB of pack for 9.7 GFLOP. The generic kernel streams every weight tile through a 2-stage ring per
// 128-pixel tile (4 tiles of pixels x 16 of channels x 8 K-splits: 226 MB through the L2s, a latency-bound K loop of 18
// steps) and combines the splits with 16.8 MB of f32 atomics: 40-45 us per launch at 0.09 of the MFMA peak. Here a workgroup
// owns ONE 64-channel chunk of the reduction for 64 output channels and ALL (up to 512) pixels: its 72 KB slice of the pack
// (9 taps x 64 x 64) and the compact 64 KB input halo of its chunk (32 images x 16 pixels x 64 channels; taps outside an
// image read a zero row) are DMA'd into LDS once, 288 MFMAs per wave run from LDS without a barrier, and the 512 x 64
// partial tile is STORED (register order, 16 bytes per lane) to the caller's scratch; conv_wstat_reduce_kernel adds the
// Ci / 64 partial tiles and applies the epilogue (alpha, bias, ReLU mask, residual). Every weight element is read by exactly
// one workgroup per 512 pixels. bf16, KH = 3, Ho = Wo = 4, no up / pool, Ci % 64 == 0.
struct WstatArgs {
    const void* x; const void* w;
    float* part;
    int B, Ci, Co, Kpad, tiles_n, nsplit;
    unsigned x_bytes, w_bytes;
};
#define WS_HALO (512 * 128)
#define WS_ZERO 256
#define WS_LDS (WS_HALO + WS_ZERO + 9 * 64 * 128)
__global__ __launch_bounds__(256, 1) void conv_wstat_kernel(WstatArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr unsigned OOB = 0x80000000u;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    int bid = blockIdx.x;
    const int tn = bid % p.tiles_n; bid /= p.tiles_n;
    const int split = bid % p.nsplit;
    const int tm = bid / p.nsplit;
    const int n0 = tn * 64, c0 = split * 64, img0 = tm * 32;
    const u32x4_t rsrc_x = make_rsrc(p.x, p.x_bytes), rsrc_w = make_rsrc(p.w, p.w_bytes);
    const unsigned smem_addr = lds_addr_of(smem);
    if (tid < WS_ZERO / 16) *reinterpret_cast<uint4*>(smem + WS_HALO + tid * 16) = make_uint4(0, 0, 0, 0);
    // ---- DMA: the halo (64 wave-instructions of 8 rows x 128 bytes) and the pack slice (72), source-side swizzled
    {
        const int r8 = lane >> 3, pch = lane & 7;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = (wv * 16 + q) * 8 + r8;                 // pixel of the tile: image row >> 4, pixel row & 15
            const int img = img0 + (row >> 4);
            const int lch = pch ^ ig2_swz(row);
            const unsigned off = img < p.B ? (unsigned)(((img * 16 + (row & 15)) * p.Ci + c0 + lch * 8) * 2) : OOB;
            buf_load_lds16(rsrc_x, off, smem_addr + (unsigned)((wv * 16 + q) * 1024));
        }
#pragma unroll
        for (int q = 0; q < 18; ++q) {   // tap-major: instructions 2t, 2t + 1 of every wave are tap t's rows, so tap t can start early
            const int tap = q >> 1, n = (wv * 2 + (q & 1)) * 8 + r8;
            const int lch = pch ^ ig2_swz(n);
            const unsigned off = (unsigned)((((n0 + n) * p.Kpad) + tap * p.Ci + c0 + lch * 8) * 2);
            buf_load_lds16(rsrc_w, off, smem_addr + (unsigned)(WS_HALO + WS_ZERO + (tap * 64 + (wv * 2 + (q & 1)) * 8) * 128));
        }
    }
    // ---- fragment addresses: A = this lane's pixel of each of the wave's four 32-pixel tiles, for each tap
    constexpr int TM = 4, TN = 2;
    const int hh = lane >> 5;
    unsigned a_addr[TM][9];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pl = wave * 128 + i * 32 + (lane & 31);
        const int y = (pl >> 2) & 3, x = pl & 3;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            const int row = (pl & ~15) + yy * 4 + xx;
            const bool in = yy >= 0 && yy < 4 && xx >= 0 && xx < 4;
            a_addr[i][tap] = in ? (unsigned)(row * 128 + ((hh ^ ig2_swz(row)) << 4)) : (unsigned)(WS_HALO + (hh << 4));
        }
    }
    unsigned b_addr[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = j * 32 + (lane & 31);
        b_addr[j] = (unsigned)(WS_HALO + WS_ZERO + n * 128 + ((hh ^ ig2_swz(n)) << 4));
    }
    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // (waiting per tap -- the pack rows are issued tap-major -- and starting the MFMAs under the rest of the DMA was measured:
    //  17.5 against 17.0 us per launch; the kernel's time is its 128 KB partial-tile store and the HBM read of its pack slice)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
        for (int k0 = 0; k0 < 4; k0 += 2) {
            bf16x8_t fa[2][TM], fb[2][TN];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[kk][i] = *reinterpret_cast<const bf16x8_t*>(smem + (a_addr[i][tap] ^ (unsigned)((k0 + kk) << 5)));
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[kk][j] = *reinterpret_cast<const bf16x8_t*>(smem + ((b_addr[j] ^ (unsigned)((k0 + kk) << 5)) + (unsigned)(tap * 8192)));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fb[kk][j]),   // weights first: transposed tile
                            __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, fa[kk][i]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- partial tile in register order: float4 f = (((wave * TM + i) * TN + j) * 4 + g) * 64 + lane
    float4* t = reinterpret_cast<float4*>(p.part) + ((size_t)(tm * p.tiles_n + tn) * p.nsplit + split) * 8192 + wave * (TM * TN * 4 * 64) + lane;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                t[((i * TN + j) * 4 + g) * 64] = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
}

// out = mask(alpha * sum over the splits + bias) + res for the partial tiles above; one thread per float4 of a tile.
__global__ __launch_bounds__(256) void conv_wstat_reduce_kernel(const float* __restrict__ part, ConvArgs p, int tiles_n, int nsplit) {
    const int tile = blockIdx.x >> 5, f = ((blockIdx.x & 31) << 8) + threadIdx.x;
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const float4* src = reinterpret_cast<const float4*>(part) + (size_t)tile * nsplit * 8192 + f;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 4 <= nsplit; s += 4) {
        const float4 v0 = src[(size_t)s * 8192], v1 = src[(size_t)(s + 1) * 8192], v2 = src[(size_t)(s + 2) * 8192], v3 = src[(size_t)(s + 3) * 8192];
        a.x += (v0.x + v1.x) + (v2.x + v3.x); a.y += (v0.y + v1.y) + (v2.y + v3.y);
        a.z += (v0.z + v1.z) + (v2.z + v3.z); a.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; s < nsplit; ++s) {
        const float4 v0 = src[(size_t)s * 8192];
        a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
    }
    const int lane = f & 63, g = (f >> 6) & 3, j = (f >> 8) & 1, wi = f >> 9, i = wi & 3, w = wi >> 2;
    const int pix = tm * 512 + w * 128 + i * 32 + (lane & 31);
    const int ch = tn * 64 + j * 32 + 8 * g + 4 * (lane >> 5);
    if (pix >= p.B * 16 || ch >= p.Co) return;
    float v[4] = {a.x * p.alpha, a.y * p.alpha, a.z * p.alpha, a.w * p.alpha};
    if (p.bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + ch);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
    }
    const size_t off = (size_t)pix * p.Co + ch;
    if (p.relu_mask) {
        float mk[4];
        Op4<bf16_t>::load(reinterpret_cast<const bf16_t*>(p.relu_mask) + off, mk);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (!(mk[e] > 0.f)) v[e] = 0.f;
    }
    if (p.res) {
        const float4 r4 = *reinterpret_cast<const float4*>(p.res + off);
        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
    }
    *reinterpret_cast<float4*>(p.out + off) = make_float4(v[0], v[1], v[2], v[3]);
}

// Returns -100 when the launch is not covered (the caller falls through to the other kernels).
static int launch_wstat(ConvArgs& a, hipStream_t stream) {
    static const int on = getenv("L2I_WSTAT") ? atoi(getenv("L2I_WSTAT")) : 1;
    if (!on || a.KH != 3 || a.Ho != 4 || a.Wo != 4 || a.up2 || a.pool2 || a.Ci % 64 || a.Ci < 256 || a.Co % 4 || a.nimg || a.half_rows || a.sc.x ||
        !a.out || a.out_op || a.out_op_raw || a.stat_part || !a.scratch)
        return -100;
    WstatArgs w;
    w.x = a.x; w.w = a.w; w.part = a.scratch;
    w.B = a.B; w.Ci = a.Ci; w.Co = a.Co; w.Kpad = a.Kpad;
    w.tiles_n = (a.Co + 63) / 64; w.nsplit = a.Ci / 64;
    w.x_bytes = a.x_bytes; w.w_bytes = a.w_bytes;
    const int tiles_m = (a.B * 16 + 511) / 512, tiles = tiles_m * w.tiles_n;
    if ((long long)tiles * w.nsplit * 32768 > a.scratch_floats) return -100;
    static bool ready = false;
    if (!ready) {
        (void)hipFuncSetAttribute((const void*)conv_wstat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS);
        ready = true;
    }
    L2I_LAUNCH(0, conv_wstat_kernel, dim3(tiles * w.nsplit), dim3(256), WS_LDS, stream, w);
    L2I_LAUNCH(0, conv_wstat_reduce_kernel, dim3(tiles * 32), dim3(256), 0, stream, (const float*)a.scratch, a, w.tiles_n, w.nsplit);
    return l2i_check_launch();
}

static int g_conv_cfg_override = -1;  // tuning hook (l2i_set_conv_config): -1 = heuristic
static int g_epi_mode = -1;           // forced epilogue form (tests / tuning), -1 = default
extern "C" int l2i_set_conv_config(int cfg) {
    if (cfg >= 4000) { g_epi_mode = cfg == 4009 ? -1 : cfg - 4000; return L2I_OK; }   // 4000 + m: epilogue form m (see l2i_conv2d_fwd); 4009: default
    if (cfg >= 3000) { g_generic_cfg = cfg - 3000 - 1; return L2I_OK; }   // 3000 = heuristic, 3001 + n = configuration n
    if (cfg >= 2000) { g_force_splits = cfg - 2000; return L2I_OK; }   // 2000 + n: forced split count, conv_halo3 (tuning only)
    if (cfg >= 1000) { g_split_target = cfg - 1000; return L2I_OK; }   // 1000 + n: split-K target (tuning only)
    g_conv_cfg_override = cfg;
    return L2I_OK;
}

template <typename T>
static int launch_conv(ConvArgs& a, hipStream_t stream) {
    constexpr int BK = Mma<T>::BK;
    constexpr int EPG = OpT<T>::EPG;
    if (a.KH != 1 && a.KH != 3) return L2I_ERR_ARG;
    if (a.Ci % EPG) return L2I_ERR_ARG;
    if (a.up2 && (a.Ho != 2 * a.Hi || a.Wo != 2 * a.Wi)) return L2I_ERR_ARG;
    if (!a.up2 && (a.Ho != a.Hi || a.Wo != a.Wi)) return L2I_ERR_ARG;
    if (a.pool2 && ((a.Ho & 1) || (a.Wo & 1))) return L2I_ERR_ARG;
    a.K = a.KH * a.KH * a.Ci;
    if (a.Kpad < a.K || a.Kpad % BK) return L2I_ERR_ARG;
    a.chunk_major = (a.KH == 3 && a.Ci >= BK) ? 1 : 0;
    a.nks = a.chunk_major ? 9 * ((a.Ci + BK - 1) / BK) : a.Kpad / BK;
    {   // buffer descriptors and lane offsets are 32-bit (and 0x80000000 is the "out of range" marker)
        const size_t xb = (size_t)a.B * a.Hi * a.Wi * a.Ci * sizeof(T);
        const size_t wb = (size_t)((a.Co + 127) / 128 * 128) * a.Kpad * sizeof(T);
        if (xb >= 0x80000000ull || wb >= 0x80000000ull) return L2I_ERR_ARG;
        a.x_bytes = (unsigned)xb;
        a.w_bytes = (unsigned)wb;
        // conv_epilogue_lds addresses the result-shaped tensors through buffer descriptors with 32-bit byte offsets
        const size_t ob = (size_t)a.B * (a.Ho >> a.pool2) * (a.Wo >> a.pool2) * a.Co * 4;
        if (ob >= 0x80000000ull) {
            if (a.stat_part) return L2I_ERR_ARG;
            a.epi_lds = 0;   // (64-bit addressing in the direct form)
        }
    }
    a.lin = a.Wo < 2;
    if (a.lin) {
        if (a.Ho != 1 || a.pool2) return L2I_ERR_ARG;
        a.PW = 1;
        a.hw_shift = 0;
    } else {
        if (a.Wo & (a.Wo - 1)) return L2I_ERR_ARG;  // power-of-two widths (4..128 on this path)
        if (a.Ho & 1) return L2I_ERR_ARG;
        a.PW = a.Wo < 16 ? a.Wo : 16;
        a.hw_shift = ilog2(a.PW >> 1);
    }
    a.tiles_c = a.Wo / a.PW;
    // Configurations (see DESIGN.md for the measurements behind the choice):
    //  0: 128x128 tile, 4 waves, 2 stages (two workgroups per CU)      1: 128x64, 4 waves, 2 stages (Co <= 64)
    //  2: 128x128, 8 waves (64x32 each), 4 stages                       3: 256x128, 8 waves (64x64 each), 3 stages
    //  4: 256x256, 8 waves (64x128 each), 2 stages                      5/6: as 0/1 with half K-steps and a 4-stage ring
    // Heuristic from tools/perf/conv_tune.py on MI355X (TFLOP/s, bf16): big-tile configs pay only when their grid still
    // fills the 256 CUs; a single wave of 128x128 tiles (one workgroup per CU) prefers the 8-wave deep ring.
    const long long M = (long long)a.B * a.Ho * a.Wo;
    // tuning: bit 0 = 4-wide maps, bit 1 = 4 -> 8 upsampling stay on the generic kernel. Same-box A/B inside the iteration
    // (profiles/r02_small_map_ab.txt): 4 -> 8 layers 870 -> 657 us and 240 -> 193 us on conv_halo3; the 4x4 maps of D
    // block6 are FASTER on the generic split-K kernel there (570 vs 666 us) although slower back to back -> default 1.
    static const int no_small = getenv("L2I_NO_SMALL_HALO") ? atoi(getenv("L2I_NO_SMALL_HALO")) : 1;
    if (sizeof(T) == 2 && g_conv_cfg_override < 0 && g_generic_cfg < 0) {   // 4x4 maps: the weight-stationary split-K kernel (needs the caller's scratch)
        const int rc = launch_wstat(a, stream);
        if (rc != -100) return rc;
    }
    const bool small_map = a.Wo < 8 || (a.up2 && a.Wo < 16);   // 4-wide maps, 4 -> 8 upsampling: only conv_halo3's compact halo covers them
    if (sizeof(T) == 2 && a.KH == 3 && a.Ci >= 64 && !a.lin && a.Wo >= 4 && !(a.up2 && a.Wo < 8) && a.Ho >= 2 &&
        (!small_map || (a.Ci % 64 == 0 && a.Ci >= 256 && !((no_small & 1) && a.Wo < 8) && !((no_small & 2) && a.up2))) &&
        (g_conv_cfg_override < 0 || g_conv_cfg_override >= 10)) {   // (-2: tuning, single halo buffer for every 128x128 launch)
        // 128x128 tiles (two workgroups per CU, 2-stage weight ring) when they make at least one full wave of
        // workgroups; otherwise 128x64 tiles (twice the workgroups, 3-stage ring). Measured: tools/perf/conv_tune.py.
        // ROI-head launches (device-side live-image count): only the live rows make workgroups that do work; the grid-size
        // thresholds below are applied to an ESTIMATE of the live pixels (L2I_ROI_LIVE percent of the rows; tuning, 100 = off)
        static const int roi_live = getenv("L2I_ROI_LIVE") ? atoi(getenv("L2I_ROI_LIVE")) : 100;
        const long long Ml = a.nimg ? M * roi_live / 100 : M;
        const long long t128h = ((Ml + 127) / 128) * ((a.Co + 127) / 128);
        // 128x128 tiles need >= 512 workgroups (one full wave at two per CU); with them, the double-buffered halo + 2-stage
        // ring wins on long reductions, the single halo buffer + 3-stage ring on short ones and on 8-wide maps (whose two
        // 8x8 sub-patch halos only fit twice per CU single-buffered). Measured: tools/perf/conv_tune.py + in-iteration profile.
        int hc = 5;   // 128x64 tiles, single halo buffer: 48 KB, three workgroups per CU
        // (in-situ sweep, profiles/r02_insitu_cfg_sweep.txt: the single-halo 128x128 tile also beats the double-buffered one
        //  on the long reductions now -- 210 vs 270 us on 32x32x512->256 -- so 0 is only a tuning option)
        if (a.Co > 64 && t128h >= 512) hc = 4;
        // Round 5 experiment, OFF (L2I_PART_BIG=1 turns it on): with split-K by stored partial tiles an under-filled grid could buy its
        // workgroups with K splits of the 128x128 tile (1.0 fragment read per MFMA against 1.5 on 128x64) instead of smaller tiles.
        // Measured in situ (profiles/r05_ab_big_tiles.txt): every affected layer got SLOWER -- (32,16,16,512->512) 43 -> 55 us,
        // (32,16,16,256->512) 26 -> 38, (32,32,32,256->128) 29 -> 41 -- the iteration 19.68 -> 20.03 ms: at 4-8 chunks per tile the
        // partial-tile round trip + the reduce launch cost more than the better tile returns; splits pay only where the
        // reduction is long AND the tiles are few (the <= 8-px maps: plan_part_splits' 3/4 rule on the tile the heuristic picked).
        static const int part_big = getenv("L2I_PART_BIG") ? atoi(getenv("L2I_PART_BIG")) : 0;
        if (part_big && hc == 5 && a.Co > 64 && a.scratch && !a.nimg && !small_map && a.Ci % 64 == 0) {
            const int nch = a.Ci / 64;
            long long sp = t128h > 0 ? 512 / t128h : 1;
            if (sp > nch / 2) sp = nch / 2;
            if (nch >= 4 && sp >= 2 && t128h * sp >= 384 && t128h * sp * 16384LL <= a.scratch_floats) hc = 4;
        }
        // (64-wide maps with <= 128 input channels -- 9-18 K-steps per tile -- and the 104 -> 528 PSP bottleneck: 128x64 tiles,
        //  5-10 % in the sweep of profiles/r02b_insitu_cfg_sweep.txt)
        if (hc == 4 && a.Ho == 64 && a.Ci <= 128 && (a.Co <= 128 || a.Ci == 104)) hc = 5;
        // 256-pixel tiles (conv_halo3_kernel; tools/perf/conv_sweep.py, profiles/r02_conv_sweep.txt): +4..11 % on the long
        // reductions whose 256x128 grid still fills both workgroup slots of every CU (obj4 conv2 at 32x32, the
        // 1024-channel ROI heads), and -- as 256x64 tiles -- on the upsampling layers from 32x32 outputs up.
        // (9 / 19: their variants with the barrier inside the K-step, +2..4 % on these shapes)
        // (256x64 tiles looked 5-7 % better on the upsampling layers back to back, but lose 15-25 % to the 128x64 tiles
        //  inside the iteration: not used there)
        if (a.Ci % 64 == 0 && a.Ci >= 512 && a.Co >= 512 && ((Ml + 255) / 256) * ((a.Co + 127) / 128) >= 512) hc = 9;
        // (256x128 / 8-wave tiles are ~10 % faster on the 1024-channel ROI-head layers in isolation but not inside the
        //  iteration -- rocprofv3: 1.90 vs 1.77 ms for those 9 launches -- so they stay a tuning option: cfg 12)
        // ROI heads (device-side live-row count; ~60 % of the rows live, spread over the XCDs by the dispatcher's round robin):
        // smaller tiles balance the live tiles over the CUs. In-situ sweep after the remap change
        // (profiles/r02b_insitu_cfg_sweep.txt): 1024 output channels 256x64 tiles (559 -> 448, 592 -> 455, 338 -> 259 us),
        // 512 output channels 128x64 tiles (447 -> 356, 394 -> 321, 368 -> 300 us).
        static const int roi_tiles = getenv("L2I_ROI_TILES") ? atoi(getenv("L2I_ROI_TILES")) : 1;
        if (a.nimg && roi_tiles && a.Ci % 64 == 0 && a.Ci >= 256) hc = a.Co >= 1024 ? 19 : 5;
        if (g_conv_cfg_override >= 10) hc = g_conv_cfg_override - 10;
        if (small_map && hc != 7 && hc != 8 && hc != 9 && hc != 19)   // 256x128 tiles when they still make a full grid, else 256x64 + split-K
            hc = ((M + 255) / 256) * ((a.Co + 127) / 128) >= 256 ? 9 : 19;
        if (small_map && a.nimg && roi_tiles && g_conv_cfg_override < 10) hc = 19;
        int rc;
        switch (hc) {
            case 1: rc = launch_halo2<128, 64, 2, 2, 3, false>(a, stream); break;
            case 2: rc = launch_halo2<256, 128, 4, 2, 2, false>(a, stream); break;
            case 3: rc = launch_halo2<128, 128, 2, 2, 3, false>(a, stream); break;   // one workgroup per CU
            case 4: rc = launch_halo2<128, 128, 2, 2, 3, false, true, 0, true>(a, stream); break;   // single halo buffer: 72 KB, two per CU, 2 tiles ahead
            case 5: rc = launch_halo2<128, 64, 2, 2, 3, false, true, 0, true>(a, stream); break;    // 48 KB: three per CU
            case 6: rc = launch_halo2<128, 64, 2, 2, 2, false, true>(a, stream); break;    // 40 KB: four per CU
            case 7: rc = launch_halo3<128>(a, stream); break;    // 256 x 128 tiles, 4 waves of 64 x 128
            case 8: rc = launch_halo3<64>(a, stream); break;     // 256 x 64 tiles
            case 9: rc = launch_halo3<128, 0, true, true>(a, stream); break;   // the same with the barrier moved inside the K-step (PF)
            case 19: rc = launch_halo3<64, 0, true, true>(a, stream); break;
            case 30: rc = launch_halo8<true, false>(a, stream); break;   // 256 x 256 tiles, eight waves, one workgroup per CU (round-5 experiment)
            case 31: rc = launch_halo8<true, true>(a, stream); break;    // ... with three half-tiles of weights in flight and counted vmcnt
#ifdef L2I_ABLATIONS
            case 41: rc = launch_halo3<128, 1>(a, stream); break;
            case 42: rc = launch_halo3<128, 2>(a, stream); break;
            case 43: rc = launch_halo3<128, 3>(a, stream); break;
            case 44: rc = launch_halo3<128, 4>(a, stream); break;
            case 21: rc = launch_halo2<128, 128, 2, 2, 3, false, true, 1>(a, stream); break;
            case 22: rc = launch_halo2<128, 128, 2, 2, 3, false, true, 2>(a, stream); break;
            case 23: rc = launch_halo2<128, 128, 2, 2, 3, false, true, 3>(a, stream); break;
            case 24: rc = launch_halo2<128, 128, 2, 2, 3, false, true, 4>(a, stream); break;
            case 31: rc = launch_halo2<256, 128, 4, 2, 2, false, false, 1>(a, stream); break;
            case 32: rc = launch_halo2<256, 128, 4, 2, 2, false, false, 2>(a, stream); break;
            case 33: rc = launch_halo2<256, 128, 4, 2, 2, false, false, 3>(a, stream); break;
#endif
            default: rc = launch_halo2<128, 128, 2, 2, 2, false>(a, stream); break;
        }
        if (rc != -100) return rc;
    }
    if (a.sc.x) {   // the generic kernel does not fold shortcuts
        const int rc = sc_unfold<T>(a, stream);
        if (rc != L2I_OK) return rc;
    }
    const long long t128 = ((M + 127) / 128) * ((a.Co + 127) / 128);
    const long long t256x128 = ((M + 255) / 256) * ((a.Co + 127) / 128);
    const long long t256x256 = ((M + 255) / 256) * ((a.Co + 255) / 256);
    int cfg;
    if (a.Co <= 64) cfg = 1;
    else if (!a.lin && a.Co % 256 == 0 && t256x256 >= 240 && a.nks >= 144) cfg = 4;   // 1024-channel 3x3 ROI heads
    else if (!a.lin && a.Co % 128 == 0 && t256x128 >= 384 && a.nks >= 72) cfg = 3;
    else if (t128 >= 192 && t128 <= 288) cfg = 2;
    else cfg = (t128 < 512 && !(t128 >= 64 && t128 < 192 && a.nks >= 16 && a.out && !a.out_op && !a.out_op_raw)) ? 1 : 0;   // small grids: 128x64 tiles (48 KB, three
                                                          // workgroups per CU) -- unless split-K applies: 128x128 + split-K
    if (g_generic_cfg >= 0 && a.Co > 64 && !a.lin) {
        cfg = g_generic_cfg;
        if ((cfg == 3 || cfg == 4) && t256x128 < 128) cfg = 0;
    }
    if (g_conv_cfg_override == 5 && a.Co <= 64) cfg = 6;
    if (g_conv_cfg_override >= 0 && a.Co > 64) {
        cfg = g_conv_cfg_override;
        if ((cfg == 3 || cfg == 4) && (a.lin || t256x128 < 128)) cfg = 0;  // too few tiles to be meaningful
    }
    switch (cfg) {
        case 1: return launch_cfg<T, 128, 64, 2, 2, 2>(a, stream);
        case 2: return launch_cfg<T, 128, 128, 2, 4, 4>(a, stream);
        case 3: return launch_cfg<T, 256, 128, 4, 2, 3>(a, stream);
        case 4: return launch_cfg<T, 256, 256, 4, 2, 2>(a, stream);
        case 5: return launch_cfg<T, 128, 128, 2, 2, 4, 1>(a, stream);   // half K-steps, 4-stage ring, two workgroups per CU
        case 6: return launch_cfg<T, 128, 64, 2, 2, 4, 1>(a, stream);
        default: return launch_cfg<T, 128, 128, 2, 2, 2>(a, stream);
    }
}

// C ABI -- see include/l2i.h
extern "C" int l2i_timing(int on) {
    for (int c = 0; c < 2; ++c) {
        for (hipEvent_t e : g_l2i_timer.start[c]) (void)hipEventDestroy(e);
        for (hipEvent_t e : g_l2i_timer.stop[c]) (void)hipEventDestroy(e);
        g_l2i_timer.start[c].clear();
        g_l2i_timer.stop[c].clear();
    }
    g_l2i_timer.on = on != 0;
    return L2I_OK;
}
extern "C" int l2i_timing_read(int cls, double* total_ms, int* launches) {
    if (cls < 0 || cls > 1 || !total_ms || !launches) return L2I_ERR_ARG;
    double t = 0.0;
    const size_t n = g_l2i_timer.start[cls].size();
    for (size_t i = 0; i < n; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(g_l2i_timer.stop[cls][i]) != hipSuccess) return L2I_ERR_LAUNCH;
        if (hipEventElapsedTime(&ms, g_l2i_timer.start[cls][i], g_l2i_timer.stop[cls][i]) != hipSuccess) return L2I_ERR_LAUNCH;
        t += ms;
    }
    *total_ms = t;
    *launches = (int)n;
    return L2I_OK;
}

extern "C" int l2i_conv2d_fwd_dual(const void* x, const void* w, const float* bias, const float* res,
                                   const void* relu_mask, float* out, void* out_op, void* out_op_raw, int dtype, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH,
                                   int up2, int pool2, int relu_op, int Kpad, float alpha, const int* nimg, float* stats, float* ws,
                                   const void* sc_x, const void* sc_w, const float* sc_bias, float* sc_out, int sc_Hi, int sc_Wi, int sc_Ci,
                                   int sc_up2, int sc_Kpad, const void* w_b, const void* sc_w_b, float* scratch, long long scratch_floats, void* stream);

extern "C" int l2i_conv2d_fwd(const void* x, const void* w, const float* bias, const float* res,
                              const void* relu_mask, float* out, void* out_op, void* out_op_raw, int dtype, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH,
                              int up2, int pool2, int relu_op, int Kpad, float alpha, const int* nimg, float* stats, float* ws,
                              void* stream) {
    return l2i_conv2d_fwd_dual(x, w, bias, res, relu_mask, out, out_op, out_op_raw, dtype, B, Hi, Wi, Ci, Ho, Wo, Co, KH, up2, pool2, relu_op,
                               Kpad, alpha, nimg, stats, ws, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, 0, stream);
}

extern "C" int l2i_conv2d_fwd_sc(const void* x, const void* w, const float* bias, const float* res,
                                 const void* relu_mask, float* out, void* out_op, void* out_op_raw, int dtype, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH,
                                 int up2, int pool2, int relu_op, int Kpad, float alpha, const int* nimg, float* stats, float* ws,
                                 const void* sc_x, const void* sc_w, const float* sc_bias, float* sc_out, int sc_Hi, int sc_Wi, int sc_Ci,
                                 int sc_up2, int sc_Kpad, void* stream) {
    return l2i_conv2d_fwd_dual(x, w, bias, res, relu_mask, out, out_op, out_op_raw, dtype, B, Hi, Wi, Ci, Ho, Wo, Co, KH, up2, pool2, relu_op,
                               Kpad, alpha, nimg, stats, ws, sc_x, sc_w, sc_bias, sc_out, sc_Hi, sc_Wi, sc_Ci, sc_up2, sc_Kpad, nullptr, nullptr, nullptr, 0, stream);
}

// Dual launch: w_b (and sc_w_b with a folded shortcut) non-null -> the B images are two passes of B/2 images each that share
// every tensor argument but the weight packs: images [0, B/2) use w (sc_w), images [B/2, B) use w_b (sc_w_b); `nimg` then counts
// the live leading images of EACH half. B must be even and (B/2) * Ho a multiple of the tile's pixel rows (L2I_ERR_ARG otherwise:
// the caller then issues the two halves as two launches).
static int conv2d_impl(const void* x, const void* w, const float* bias, const float* res,
                                   const void* relu_mask, float* out, void* out_op, void* out_op_raw, int dtype, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH,
                                   int up2, int pool2, int relu_op, int Kpad, float alpha, const int* nimg, float* stats, float* ws,
                                   const void* sc_x, const void* sc_w, const float* sc_bias, float* sc_out, int sc_Hi, int sc_Wi, int sc_Ci,
                                   int sc_up2, int sc_Kpad, const void* w_b, const void* sc_w_b, float* scratch, long long scratch_floats, void* stream,
                                   float sc_alpha, int mask_first) {
    if (!x || !w || (!out && !out_op && !out_op_raw)) return L2I_ERR_ARG;
    if (mask_first && (!sc_x || !relu_mask || sc_bias || bias || up2 || pool2 || stats || w_b || dtype != 1 || !(sc_alpha > 0.f))) return L2I_ERR_ARG;
    if (w_b && ((B & 1) || stats || (sc_x && !sc_w_b))) return L2I_ERR_ARG;
    if (!w_b && sc_w_b) return L2I_ERR_ARG;
    if (sc_x) {   // folded shortcut: 1x1 on the (optionally nearest-upsampled) pre-pool grid of this launch; its result stands in for `res`
        if (!sc_w || !sc_out || ((res || relu_mask) && !mask_first) || sc_Ci <= 0 || sc_Ci % 8 || sc_Kpad < sc_Ci) return L2I_ERR_ARG;
        if (sc_Hi << (sc_up2 ? 1 : 0) != Ho || sc_Wi << (sc_up2 ? 1 : 0) != Wo) return L2I_ERR_ARG;
    }
    (void)ws;   // (round 5: the statistics went through the replicated atomic workspace; round 6: stored partial rows in `scratch`, summed in a fixed order)
    if (stats && (!scratch || !out || Co % 4 || nimg)) return L2I_ERR_ARG;
    ConvArgs a;
    a.SUBH = 1; a.P = 1;   // (halo geometry: set by the halo launchers)
#ifdef L2I_ABLATIONS   // wrong-result switches exist only in ablation builds (L2I_EXTRA_FLAGS=-DL2I_ABLATIONS), never in the shipped library
    static const int no_epi = getenv("L2I_CONV_NOEPI") ? atoi(getenv("L2I_CONV_NOEPI")) : 0;
    a.no_epi = no_epi;
#else
    a.no_epi = 0;
#endif
    static const int roi_remap = getenv("L2I_ROI_REMAP") ? atoi(getenv("L2I_ROI_REMAP")) : 0;
    a.roi_remap = roi_remap;
    // Epilogue form. 2 (default): conv_epilogue_lds for every launch; 1: only when the epilogue touches operand-dtype
    // tensors (ReLU mask, operand copies); 0: direct stores from the accumulator layout. Same-box A/B of the training
    // iteration: 25.4 ms (0), 25.0-25.2 (1), 24.3-24.9 (2). (Delaying every second workgroup of a CU so that one stores while
    // the other multiplies was tried as well, with s_sleep at kernel start: it only adds the delay to the launch.)
    static const int epi_env = getenv("L2I_EPI") ? atoi(getenv("L2I_EPI")) : 2;
    const int epi_mode = g_epi_mode >= 0 ? g_epi_mode : epi_env;
    a.epi_lds = epi_mode == 2 || (epi_mode == 1 && (relu_mask != nullptr || out_op != nullptr || out_op_raw != nullptr));
    a.nimg = nimg;
    a.w_b = w_b; a.half_rows = w_b ? (B / 2) * Ho : 0;
    a.scratch = scratch; a.scratch_floats = scratch ? scratch_floats : 0; a.part = nullptr;
    a.x = x; a.w = w; a.bias = bias; a.res = res; a.out = out; a.out_op = out_op; a.out_op_raw = out_op_raw; a.relu_mask = relu_mask;
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ci = Ci; a.Ho = Ho; a.Wo = Wo; a.Co = Co; a.KH = KH;
    a.up2 = up2 ? 1 : 0; a.pool2 = pool2 ? 1 : 0; a.relu_op = relu_op ? 1 : 0;
    a.Kpad = Kpad; a.alpha = alpha;
    a.stat_part = nullptr; a.stat_wm = 1; a.stat_cap = 0; a.stat_rows_out = nullptr;
    int stat_nrows = 0;
    float* stat_tmp = nullptr;
    if (stats) {   // the tail of the scratch: [partial rows: at most one per 32 pixels + a tile row of padding][fold chunks]; split-K partial tiles keep the front
        const long long L = 2LL * Co, rows_max = ((long long)B * Ho * Wo + 31) / 32 + 8LL * 1024;
        const long long need = rows_max * L + (long long)L2I_FOLD_CHUNKS * L;
        if (need > a.scratch_floats) return L2I_ERR_ARG;
        a.scratch_floats = (a.scratch_floats - need) & ~3LL;
        a.stat_part = scratch + a.scratch_floats;
        a.stat_cap = rows_max * L;
        stat_tmp = a.stat_part + a.stat_cap;
        a.stat_rows_out = &stat_nrows;
        if ((size_t)a.stat_part & 15) return L2I_ERR_ARG;
    }
    static const int sc_fold_env = getenv("L2I_SC_FOLD") ? atoi(getenv("L2I_SC_FOLD")) : 1;
    g_no_sc_fold = !sc_fold_env;
    const size_t esz = dtype == 0 ? 4 : 2;
    a.sc.w_b = (sc_x && w_b) ? sc_w_b : nullptr;
    a.sc.x = sc_x; a.sc.w = sc_x ? sc_w : nullptr; a.sc.bias = sc_x ? sc_bias : nullptr; a.sc.out = sc_out;
    a.sc.Hi = sc_Hi; a.sc.Wi = sc_Wi; a.sc.Ci = sc_Ci; a.sc.up2 = sc_up2 ? 1 : 0; a.sc.Kpad = sc_Kpad; a.sc.stages = 1;
    a.sc.mask_first = nullptr; a.sc.pre_scale = 1.f;
    if (mask_first) {   // data-gradient fold: the epilogue scales by the TAIL's alpha, the 3x3 accumulators are masked and pre-scaled in front of the tail
        a.sc.mask_first = relu_mask; a.relu_mask = nullptr;
        a.sc.pre_scale = alpha / sc_alpha; a.alpha = sc_alpha;
    }
    a.sc.x_bytes = sc_x ? (unsigned)((size_t)B * sc_Hi * sc_Wi * sc_Ci * esz) : 0;
    a.sc.w_bytes = sc_x ? (unsigned)((size_t)((Co + 127) / 128 * 128) * sc_Kpad * esz) : 0;
    if (sc_x && ((size_t)B * sc_Hi * sc_Wi * sc_Ci * esz >= (1ull << 31))) return L2I_ERR_ARG;
    if (stats) a.epi_lds = 1;   // (the statistics are gathered by the LDS form of the epilogue; no split-K on such a launch)
    int rc;
    if (dtype == 0) rc = launch_conv<float>(a, (hipStream_t)stream);
    else if (dtype == 1) rc = launch_conv<bf16_t>(a, (hipStream_t)stream);
    else return L2I_ERR_ARG;
    if (rc == L2I_OK && stats) {
        if (stat_nrows <= 0) return L2I_ERR_ARG;   // (a kernel without the LDS epilogue took the launch: not a statistics launch)
        rows_fold(a.stat_part, stat_nrows, 2 * Co, 1, stats, stats + Co, Co, 0, 1, stat_tmp, (hipStream_t)stream);
        rc = l2i_check_launch();
    }
    return rc;
}

extern "C" int l2i_conv2d_fwd_dual(const void* x, const void* w, const float* bias, const float* res,
                                   const void* relu_mask, float* out, void* out_op, void* out_op_raw, int dtype, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH,
                                   int up2, int pool2, int relu_op, int Kpad, float alpha, const int* nimg, float* stats, float* ws,
                                   const void* sc_x, const void* sc_w, const float* sc_bias, float* sc_out, int sc_Hi, int sc_Wi, int sc_Ci,
                                   int sc_up2, int sc_Kpad, const void* w_b, const void* sc_w_b, float* scratch, long long scratch_floats, void* stream) {
    return conv2d_impl(x, w, bias, res, relu_mask, out, out_op, out_op_raw, dtype, B, Hi, Wi, Ci, Ho, Wo, Co, KH, up2, pool2, relu_op, Kpad, alpha, nimg, stats, ws,
                       sc_x, sc_w, sc_bias, sc_out, sc_Hi, sc_Wi, sc_Ci, sc_up2, sc_Kpad, w_b, sc_w_b, scratch, scratch_floats, stream, alpha, 0);
}

// Data gradient of a pre-activation residual block's first convolution WITH the data gradient of the block's 1x1 shortcut folded in
// (reference model/rcnn_discriminator_app.py:326,336-341: x -> relu -> conv1 ... and x -> c_sc -> avg_pool): see include/l2i.h.
extern "C" int l2i_conv2d_dgrad_sc(const void* dh, const void* w, const float* res, const void* relu_mask, float* out, void* out_op_raw,
                                   int B, int H, int W, int Ci, int Co, int Kpad, float alpha, const int* nimg,
                                   const void* sc_dy, const void* sc_w, float* sc_out, int sc_Hi, int sc_Wi, int sc_Ci, int sc_up2, int sc_Kpad,
                                   float sc_alpha, float* scratch, long long scratch_floats, void* stream) {
    return conv2d_impl(dh, w, nullptr, res, relu_mask, out, nullptr, out_op_raw, 1, B, H, W, Ci, H, W, Co, 3, 0, 0, 0, Kpad, alpha, nimg, nullptr, nullptr,
                       sc_dy, sc_w, nullptr, sc_out, sc_Hi, sc_Wi, sc_Ci, sc_up2, sc_Kpad, nullptr, nullptr, scratch, scratch_floats, stream, sc_alpha, 1);
}

// Debug aid: co-resident workgroups per CU the runtime computes for a few instantiations (tools/perf/occupancy.py).
extern "C" int l2i_debug_occupancy(int which, int lds_bytes) {
    int n = -1;
    hipError_t e = hipSuccess;
    if (which == 100) return g_last_splits;   // splits of the last halo-kernel launch: > 1 stored partial tiles + reduce, < -1 atomics
    switch (which) {
        case 0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)conv_igemm_kernel<bf16_t, 128, 128, 2, 2, 2, 0>, 256, lds_bytes); break;
        case 10: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)conv_halo2_kernel<128, 128, 2, 2, 2, false>, 256, lds_bytes); break;
        case 11: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)conv_halo2_kernel<128, 64, 2, 2, 3, false>, 256, lds_bytes); break;
        default: break;
    }
    return e == hipSuccess ? n : -(int)e;
}
