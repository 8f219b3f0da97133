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
#include "igemm.h"

__device__ uint4 g_zero16[1] = {{0u, 0u, 0u, 0u}};  // source of out-of-image im2col taps

struct ConvArgs {
    const void* x;      // T  [B, Hi, Wi, Ci]
    const void* w;      // T  [Npad, Kpad], K order (ky, kx, ci)
    const float* bias;  // [Co] or null
    const float* res;   // f32, shape of out, or null
    float* out;         // f32 [B, Hout, Wout, Co] or null
    void* out_op;       // T   same shape, optional operand copy (relu'd if relu_op)
    void* out_op_raw;   // T   same shape, optional un-activated operand copy
    const void* relu_mask;  // T, shape of out: result is zeroed where mask <= 0 (ReLU backward), before res is added
    int B, Hi, Wi, Ci, Ho, Wo, Co, KH;
    int up2, pool2, relu_op;
    int Kpad, K;
    int PW, PH, hw_shift, lin, tiles_c, tiles_m, tiles_n;
    int splits, ks_per;  // split-K: `out` pre-zeroed, partials combined with f32 atomics
    float alpha;
};

__device__ __forceinline__ void idx2pix(int idx, int hw_shift, int lin, int& py, int& px) {
    if (lin) { py = idx; px = 0; return; }
    const int q = idx >> 2, s = idx & 3;
    const int qy = q >> hw_shift, qx = q & ((1 << hw_shift) - 1);
    py = 2 * qy + (s >> 1);
    px = 2 * qx + (s & 1);
}

template <typename T, int BN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p) {
    constexpr int BM = 128;
    constexpr int BK = Mma<T>::BK;
    constexpr int EPG = OpT<T>::EPG;
    constexpr int TM = 2, TN = BN / 64;
    constexpr int AP = BM / 32, BP = BN / 32;  // 16-byte loads per thread per K-step

    // LDS: two stages of [A 128 rows | B BN rows], 128-byte rows, filled by LDS-DMA (global_load_lds_dwordx4:
    // lane i of a wave-instruction lands at base + 16 i, i.e. 8 rows x 8 chunks), no VGPR staging, no ds_write.
    // Bank conflicts of the ds_read_b128 fragment reads are avoided by an XOR swizzle applied on the SOURCE
    // side (the lane that fills physical chunk c of row r fetches logical chunk c ^ (r & 7)) and on the read.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = (BM + BN) * IG2_ROWB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = p.tiles_m * p.tiles_n;
    const int split = blockIdx.x / nblk;
    const int bid = xcd_remap(blockIdx.x - split * nblk, nblk);
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int tile_c = tile_m % p.tiles_c, tile_r = tile_m / p.tiles_c;
    const int n0 = tile_n * BN;
    const int rows_total = p.B * p.Ho;
    const int pad = p.KH >> 1;

    // ---- per-thread A rows (fixed for the whole K loop)
    const int lrow = tid >> 3;
    const int lchunk = (tid & 7) ^ (lrow & 7);  // logical 16-byte chunk this lane fetches
    int a_y[AP], a_x[AP], a_boff[AP];
#pragma unroll
    for (int q = 0; q < AP; ++q) {
        int py, px;
        idx2pix(lrow + 32 * q, p.hw_shift, p.lin, py, px);
        const int r = tile_r * p.PH + py;
        const int b = r / p.Ho;
        a_y[q] = r < rows_total ? r - b * p.Ho : -(1 << 20);  // invalid rows fail the bounds test
        a_x[q] = tile_c * p.PW + px;
        a_boff[q] = b * p.Hi * p.Wi * p.Ci;
    }
    const T* __restrict__ X = reinterpret_cast<const T*>(p.x);
    const T* __restrict__ W = reinterpret_cast<const T*>(p.w);
    const T* zsrc = reinterpret_cast<const T*>(g_zero16);
    const int wbase = __builtin_amdgcn_readfirstlane(wave) * 8 * IG2_ROWB;  // this wave's 8 rows of each 32-row slab

    auto issue_tiles = [&](int ks, char* stage) {
        const int k0 = ks * BK + lchunk * EPG;
        const int tap = k0 / p.Ci, ci = k0 - tap * p.Ci;
        const int ky = tap / p.KH, kx = tap - ky * p.KH;
        const bool kvalid = k0 < p.K;
#pragma unroll
        for (int q = 0; q < AP; ++q) {
            const int yy = a_y[q] + ky - pad, xx = a_x[q] + kx - pad;
            const bool inb = kvalid && yy >= 0 && yy < p.Ho && xx >= 0 && xx < p.Wo;
            const int ys = yy >> p.up2, xs = xx >> p.up2;
            const int off = a_boff[q] + (ys * p.Wi + xs) * p.Ci + ci;
            const T* src = inb ? X + off : zsrc;  // out-of-image taps read a 16-byte zero block
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(stage + q * 32 * IG2_ROWB + wbase), 16, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < BP; ++q) {
            const T* src = W + (size_t)(n0 + lrow + 32 * q) * p.Kpad + k0;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(stage + (BM + q * 32) * IG2_ROWB + wbase), 16, 0, 0);
        }
    };

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int wrow = (wave >> 1) * 64, wcol = (wave & 1) * (BN / 2);
    const int ks0 = split * p.ks_per;
    const int nks = min(p.Kpad / BK, ks0 + p.ks_per);
    issue_tiles(ks0, smem);
    for (int ks = ks0; ks < nks; ++ks) {
        char* cur = smem + ((ks - ks0) & 1) * STAGE;
        // one barrier per K-step: (i) tile ks has landed (hipcc drains vmcnt before the barrier because an LDS-DMA
        // is in flight), (ii) every wave is done reading the other stage, which the next DMA overwrites.
        __syncthreads();
        if (ks + 1 < nks) issue_tiles(ks + 1, smem + ((ks + 1 - ks0) & 1) * STAGE);  // lands under the MFMAs
        Mma2<T>::template step<TM, TN>(cur, cur + BM * IG2_ROWB, wrow, wcol, lane, acc);
    }

    // ---- epilogue
    const int c = lane & 31, h = lane >> 5;
    T* __restrict__ OutOp = reinterpret_cast<T*>(p.out_op);
    T* __restrict__ OutRaw = reinterpret_cast<T*>(p.out_op_raw);
    const T* __restrict__ Mask = reinterpret_cast<const T*>(p.relu_mask);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int idx0 = wrow + i * 32 + 8 * g + 4 * h;  // first GEMM row of this lane's quad
            if (p.pool2) {
                const int q = idx0 >> 2;
                const int qy = q >> p.hw_shift, qx = q & ((1 << p.hw_shift) - 1);
                const int r2 = tile_r * (p.PH >> 1) + qy;
                const int Hq = p.Ho >> 1, Wq = p.Wo >> 1;
                if (r2 >= p.B * Hq) continue;
                const int b = r2 / Hq, y2 = r2 - b * Hq, x2 = tile_c * (p.PW >> 1) + qx;
                const size_t rowoff = ((size_t)(b * Hq + y2) * Wq + x2) * p.Co;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wcol + j * 32 + c;
                    if (n >= p.Co) continue;
                    float v = acc[i][j][4 * g] + acc[i][j][4 * g + 1] + acc[i][j][4 * g + 2] + acc[i][j][4 * g + 3];
                    v *= p.alpha;
                    if (p.bias && split == 0) v += p.bias[n];
                    if (Mask && !(OpT<T>::to(Mask[rowoff + n]) > 0.f)) v = 0.f;
                    if (p.res && split == 0) v += p.res[rowoff + n];
                    if (p.splits > 1) { atomicAdd(p.out + rowoff + n, v); continue; }
                    if (p.out) p.out[rowoff + n] = v;
                    if (OutOp) OutOp[rowoff + n] = OpT<T>::from(p.relu_op ? fmaxf(v, 0.f) : v);
                    if (OutRaw) OutRaw[rowoff + n] = OpT<T>::from(v);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int py, px;
                    idx2pix(idx0 + e, p.hw_shift, p.lin, py, px);
                    const int r = tile_r * p.PH + py;
                    if (r >= rows_total) continue;
                    const int x = tile_c * p.PW + px;
                    const size_t rowoff = ((size_t)r * p.Wo + x) * p.Co;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int n = n0 + wcol + j * 32 + c;
                        if (n >= p.Co) continue;
                        float v = acc[i][j][4 * g + e] * p.alpha;
                        if (p.bias && split == 0) v += p.bias[n];
                        if (Mask && !(OpT<T>::to(Mask[rowoff + n]) > 0.f)) v = 0.f;
                        if (p.res && split == 0) v += p.res[rowoff + n];
                        if (p.splits > 1) { atomicAdd(p.out + rowoff + n, v); continue; }
                        if (p.out) p.out[rowoff + n] = v;
                        if (OutOp) OutOp[rowoff + n] = OpT<T>::from(p.relu_op ? fmaxf(v, 0.f) : v);
                        if (OutRaw) OutRaw[rowoff + n] = OpT<T>::from(v);
                    }
                }
            }
        }
    }
}

static int ilog2(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
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
    a.PH = 128 / a.PW;
    a.tiles_c = a.Wo / a.PW;
    const int rows = a.B * a.Ho;
    a.tiles_m = ((rows + a.PH - 1) / a.PH) * a.tiles_c;
    const int BN = a.Co <= 64 ? 64 : 128;
    a.tiles_n = (a.Co + BN - 1) / BN;
    const int nblk = a.tiles_m * a.tiles_n;
    const int nks = a.Kpad / BK;
    // split-K for small grids with a long reduction (D block5/6, G res1/res2, ROI heads): fill the 256 CUs
    int splits = 1;
    if (a.out && !a.out_op && !a.out_op_raw && nblk < 192 && nks >= 16) {
        splits = (512 + nblk - 1) / nblk;
        if (splits > nks / 4) splits = nks / 4;
        if (splits < 1) splits = 1;
    }
    a.ks_per = (nks + splits - 1) / splits;
    a.splits = (nks + a.ks_per - 1) / a.ks_per;
    if (a.splits > 1) {
        const size_t bytes = sizeof(float) * (size_t)a.B * (a.Ho >> a.pool2) * (a.Wo >> a.pool2) * a.Co;
        if (hipMemsetAsync(a.out, 0, bytes, stream) != hipSuccess) return L2I_ERR_LAUNCH;
    }
    const size_t lds = (size_t)2 * (128 + BN) * IG2_ROWB;
    if (BN == 64)
        hipLaunchKernelGGL((conv_igemm_kernel<T, 64>), dim3(nblk * a.splits), dim3(256), lds, stream, a);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<T, 128>), dim3(nblk * a.splits), dim3(256), lds, stream, a);
    return l2i_check_launch();
}

// C ABI -- see include/l2i.h
extern "C" int l2i_conv2d_fwd(const void* x, const void* w, const float* bias, const float* res,
                              const void* relu_mask, float* out, void* out_op, void* out_op_raw, int dtype, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH,
                              int up2, int pool2, int relu_op, int Kpad, float alpha, void* stream) {
    if (!x || !w || (!out && !out_op && !out_op_raw)) return L2I_ERR_ARG;
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.res = res; a.out = out; a.out_op = out_op; a.out_op_raw = out_op_raw; a.relu_mask = relu_mask;
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ci = Ci; a.Ho = Ho; a.Wo = Wo; a.Co = Co; a.KH = KH;
    a.up2 = up2 ? 1 : 0; a.pool2 = pool2 ? 1 : 0; a.relu_op = relu_op ? 1 : 0;
    a.Kpad = Kpad; a.alpha = alpha;
    if (dtype == 0) return launch_conv<float>(a, (hipStream_t)stream);
    if (dtype == 1) return launch_conv<bf16_t>(a, (hipStream_t)stream);
    return L2I_ERR_ARG;
}
