// Weight arena kernels: spectral-norm power iteration, sigma, and the packing of
// every GEMM-shaped weight of a network into the layouts the MFMA kernels read --
// one multi-tensor launch per phase for ALL layers of a network.
//
// Reference semantics (torch.nn.utils.spectral_norm as reached from
// model/resnet_generator_app_v2.py:681-686, model/rcnn_discriminator_app.py:10-15,
// model/norm_module.py:158-159, model/mask_regression.py:64-81):
//   W viewed as [Co, Kt] (Kt = Ci*KH*KW, torch layout).  train mode, one iteration:
//     v <- normalize(W^T u, eps)   u <- normalize(W v, eps)   sigma = u . (W v)   Wbar = W / sigma
//   normalize(x, eps) = x / max(||x||_2, eps).   eval mode: sigma = u . (W v) with the stored u, v.
//   backward: dW = (G - <G, Wbar> u v^T) / sigma,  G = dL/dWbar.
//
// Layer table: L2I_LSTRIDE (20) int64 per layer row (see layout2img_amd/arena.py); a weight that the reference
// applies k times per forward (block_obj4, model/rcnn_discriminator_app.py:137,141 -- each application runs its own
// power iteration) has k rows sharing w/u/v offsets and is processed in k sequential rounds:
//   0 w_off  1 u_off(-1 = no SN)  2 v_off  3 Co  4 Ci  5 KH  6 Co_p  7 Ci_p
//   8 Kpad  9 Npad  10 fwd_off  11 Kpad_d  12 Npad_d  13 dg_off  14 dw_off  15 eps(float bits)
//   16 pass_u_off  17 pass_v_off (offsets into the per-pass u/v snapshot)  18,19 reserved
// norms: f32 [L][4] = {||W^T u||^2, ||W v||^2 (eval: u.Wv), sigma, <G, W>}
#include "common.h"

#define LF(i) (L[(i)])
#define L2I_LSTRIDE 20

__device__ __forceinline__ float layer_eps(const long long* L) { return __uint_as_float((uint32_t)L[15]); }

// phase 1: t = W^T u, partial over a 256-row slab, 256 columns per block. table: (layer, colchunk, rowchunk)
__global__ __launch_bounds__(256) void sn_wtu_kernel(const long long* __restrict__ layers, const int* __restrict__ table,
                                                     const float* __restrict__ params, const float* __restrict__ sn_state,
                                                     float* __restrict__ pass_uv) {
    const int* e = table + 3 * blockIdx.x;
    const long long* L = layers + L2I_LSTRIDE * e[0];
    const int Co = (int)LF(3), Kt = (int)(LF(4) * LF(5) * LF(5));
    const int col = e[1] * 256 + threadIdx.x;
    const int r0 = e[2] * 256, r1 = min(Co, r0 + 256);
    if (col >= Kt) return;
    const float* W = params + LF(0);
    const float* u = sn_state + LF(1);
    float t = 0.f;
    for (int r = r0; r < r1; ++r) t += u[r] * W[(size_t)r * Kt + col];
    atomicAdd(pass_uv + LF(17) + col, t);
}

// phase 2: s = W vhat (train: vhat = t / max(||t||, eps); eval: vhat = stored v). 16 rows per block.
// table: (layer, rowchunk)
__global__ __launch_bounds__(256) void sn_wv_kernel(const long long* __restrict__ layers, const int* __restrict__ table,
                                                    const float* __restrict__ params, const float* __restrict__ sn_state,
                                                    float* __restrict__ pass_uv, float* __restrict__ norms, int training) {
    __shared__ float red[16];
    const int* e = table + 2 * blockIdx.x;
    const int layer = e[0];
    const long long* L = layers + L2I_LSTRIDE * layer;
    const int Co = (int)LF(3), Kt = (int)(LF(4) * LF(5) * LF(5));
    const float* W = params + LF(0);
    const float* vsrc = training ? pass_uv + LF(17) : sn_state + LF(2);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float blk = 0.f;
    float tn2_keep = 0.f;
    for (int rr = 0; rr < 4; ++rr) {
        const int r = e[1] * 16 + wave * 4 + rr;
        if (r >= Co) break;
        float dot = 0.f, tn2 = 0.f;
        for (int k = lane; k < Kt; k += 64) {
            const float t = vsrc[k];
            dot += W[(size_t)r * Kt + k] * t;
            tn2 += t * t;
        }
        dot = wave_sum(dot);
        tn2 = wave_sum(tn2);
        tn2_keep = tn2;
        float s;
        if (training) {
            s = dot / fmaxf(sqrtf(tn2), layer_eps(L));
            if (lane == 0) blk += s * s;
        } else {
            s = dot;
            if (lane == 0) blk += s * sn_state[LF(1) + r];  // sigma = u . (W v)
        }
        if (lane == 0) pass_uv[LF(16) + r] = s;
    }
    // one atomic per block
    __syncthreads();
    if (lane == 0) red[wave] = blk;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(norms + 4 * layer + 1, red[0] + red[1] + red[2] + red[3]);
        if (e[1] == 0) norms[4 * layer + 0] = tn2_keep;
    }
}

// phase 3: sigma; normalise u, v (pass copy + persistent state); write the packed weights.
// table: (layer, kind, chunk) kind 0 = forward pack [Npad][Kpad] (k = ky,kx,ci), kind 1 = dgrad pack
// [Npad_d][Kpad_d] (rows ci, k = ky',kx',co with the taps flipped). 2048 packed elements per block.
template <typename T>
__global__ __launch_bounds__(256) void sn_pack_kernel(const long long* __restrict__ layers, const int* __restrict__ table,
                                                      const float* __restrict__ params, float* __restrict__ sn_state,
                                                      float* __restrict__ pass_uv, float* __restrict__ norms,
                                                      T* __restrict__ packed, int training) {
    const int* e = table + 3 * blockIdx.x;
    const int layer = e[0], kind = e[1], chunk = e[2];
    const long long* L = layers + L2I_LSTRIDE * layer;
    const int Co = (int)LF(3), Ci = (int)LF(4), KH = (int)LF(5), Co_p = (int)LF(6), Ci_p = (int)LF(7);
    const int Kt = Ci * KH * KH;
    const bool sn = LF(1) >= 0;
    float sigma = 1.f;
    if (sn) {
        const float eps = layer_eps(L);
        const float sn2 = norms[4 * layer + 1];
        if (training) {
            const float snorm = sqrtf(sn2), tnorm = sqrtf(norms[4 * layer + 0]);
            sigma = sn2 / fmaxf(snorm, eps);
            if (kind == 0) {
                for (int i = chunk * 2048 + threadIdx.x; i < min(Co, (chunk + 1) * 2048); i += 256) {
                    const float u = pass_uv[LF(16) + i] / fmaxf(snorm, eps);
                    pass_uv[LF(16) + i] = u;
                    sn_state[LF(1) + i] = u;
                }
                for (int i = chunk * 2048 + threadIdx.x; i < min(Kt, (chunk + 1) * 2048); i += 256) {
                    const float v = pass_uv[LF(17) + i] / fmaxf(tnorm, eps);
                    pass_uv[LF(17) + i] = v;
                    sn_state[LF(2) + i] = v;
                }
            }
        } else {
            sigma = sn2;
            if (kind == 0) {
                for (int i = chunk * 2048 + threadIdx.x; i < min(Co, (chunk + 1) * 2048); i += 256)
                    pass_uv[LF(16) + i] = sn_state[LF(1) + i];
                for (int i = chunk * 2048 + threadIdx.x; i < min(Kt, (chunk + 1) * 2048); i += 256)
                    pass_uv[LF(17) + i] = sn_state[LF(2) + i];
            }
        }
        if (kind == 0 && chunk == 0 && threadIdx.x == 0) norms[4 * layer + 2] = sigma;
    } else if (kind == 0 && chunk == 0 && threadIdx.x == 0) {
        norms[4 * layer + 2] = 1.f;
    }
    const float inv = 1.f / sigma;
    const float* W = params + LF(0);
    const int taps = KH * KH;
    // one thread = 8 consecutive packed elements (same tap: Ci_p, Co_p and Kpad are multiples of 8) -> one 16-byte
    // (bf16) or two 16-byte (f32) stores
    const bool fwd = kind == 0;
    const int Kpad = (int)(fwd ? LF(8) : LF(11));
    const long long total = (fwd ? LF(9) : LF(12)) * Kpad;
    T* dst = packed + (fwd ? LF(10) : LF(13));
    const int inner_p = fwd ? Ci_p : Co_p;   // padded channel count of the pack's inner index
    const int inner = fwd ? Ci : Co, rows = fwd ? Co : Ci;
    const long long i = (long long)chunk * 2048 + 8 * threadIdx.x;
    if (i < total) {
        const int n = (int)(i / Kpad), k = (int)(i - (long long)n * Kpad);
        const int tap = k / inner_p, c0 = k - tap * inner_p;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = 0.f;
            if (n < rows && tap < taps && c0 + j < inner)
                v[j] = (fwd ? W[((size_t)n * Ci + c0 + j) * taps + tap] : W[((size_t)(c0 + j) * Ci + n) * taps + (taps - 1 - tap)]) * inv;
        }
        if constexpr (sizeof(T) == 2) {
            uint4 pk;
            pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
            pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
            pk.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
            pk.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(dst) + i) = pk;
        } else {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(dst) + i) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(dst) + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

// backward phase a: <G, W> per SN layer. table: (layer, chunk) over Co*Kt true elements, 4096 per block
__global__ __launch_bounds__(256) void sn_dot_kernel(const long long* __restrict__ layers, const int* __restrict__ table,
                                                     const float* __restrict__ params, const float* __restrict__ dwbar,
                                                     float* __restrict__ norms) {
    __shared__ float red[16];
    const int* e = table + 2 * blockIdx.x;
    const int layer = e[0];
    const long long* L = layers + L2I_LSTRIDE * layer;
    const int Co = (int)LF(3), Ci = (int)LF(4), KH = (int)LF(5), Ci_p = (int)LF(7);
    const int taps = KH * KH, Kt = Ci * taps, Kp = taps * Ci_p;
    const float* W = params + LF(0);
    const float* G = dwbar + LF(14);
    const long long total = (long long)Co * Kt;
    float acc = 0.f;
    for (long long i = (long long)e[1] * 4096 + threadIdx.x; i < min(total, (long long)(e[1] + 1) * 4096); i += 256) {
        const int co = (int)(i / Kt), kt = (int)(i - (long long)co * Kt);
        const int ci = kt / taps, tap = kt - ci * taps;
        acc += W[i] * G[(size_t)co * Kp + tap * Ci_p + ci];
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) atomicAdd(norms + 4 * layer + 3, acc);
}

// backward phase b: grads[w] += (G - <G,Wbar> u v^T) / sigma   (non-SN layers: grads[w] += G)
__global__ __launch_bounds__(256) void sn_apply_kernel(const long long* __restrict__ layers, const int* __restrict__ table,
                                                       const float* __restrict__ dwbar, const float* __restrict__ pass_uv,
                                                       const float* __restrict__ norms, float* __restrict__ grads) {
    const int* e = table + 2 * blockIdx.x;
    const int layer = e[0];
    const long long* L = layers + L2I_LSTRIDE * layer;
    const int Co = (int)LF(3), Ci = (int)LF(4), KH = (int)LF(5), Ci_p = (int)LF(7);
    const int taps = KH * KH, Kt = Ci * taps, Kp = taps * Ci_p;
    const float* G = dwbar + LF(14);
    float* dst = grads + LF(0);
    const bool sn = LF(1) >= 0;
    const float sigma = norms[4 * layer + 2];
    const float inv = 1.f / sigma;
    const float gw = sn ? norms[4 * layer + 3] * inv : 0.f;  // <G, Wbar>
    const long long total = (long long)Co * Kt;
    for (long long i = (long long)e[1] * 4096 + threadIdx.x; i < min(total, (long long)(e[1] + 1) * 4096); i += 256) {
        const int co = (int)(i / Kt), kt = (int)(i - (long long)co * Kt);
        const int ci = kt / taps, tap = kt - ci * taps;
        float g = G[(size_t)co * Kp + tap * Ci_p + ci];
        if (sn) g = (g - gw * pass_uv[LF(16) + co] * pass_uv[LF(17) + kt]) * inv;
        atomicAdd(dst + i, g);  // rows of a multiply-applied weight update the same gradient concurrently
    }
}

extern "C" int l2i_weights_prepare(const long long* layers, int n_layers, const int* tab_wtu, int n_wtu,
                                   const int* tab_wv, int n_wv, const int* tab_pack, int n_pack, const float* params,
                                   float* sn_state, float* pass_uv, long long uv_len, float* norms, void* packed,
                                   int dtype, int training, int clear, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!layers || !params || !packed || !norms) return L2I_ERR_ARG;
    if (clear) {  // first round of a pass
        if (hipMemsetAsync(norms, 0, sizeof(float) * 4 * n_layers, stream) != hipSuccess) return L2I_ERR_LAUNCH;
        if (uv_len > 0 && hipMemsetAsync(pass_uv, 0, sizeof(float) * uv_len, stream) != hipSuccess) return L2I_ERR_LAUNCH;
    }
    if (training && n_wtu > 0)
        hipLaunchKernelGGL(sn_wtu_kernel, dim3(n_wtu), dim3(256), 0, stream, layers, tab_wtu, params, sn_state, pass_uv);
    if (n_wv > 0)
        hipLaunchKernelGGL(sn_wv_kernel, dim3(n_wv), dim3(256), 0, stream, layers, tab_wv, params, sn_state, pass_uv, norms,
                           training);
    if (dtype == 0)
        hipLaunchKernelGGL(sn_pack_kernel<float>, dim3(n_pack), dim3(256), 0, stream, layers, tab_pack, params, sn_state,
                           pass_uv, norms, (float*)packed, training);
    else if (dtype == 1)
        hipLaunchKernelGGL(sn_pack_kernel<bf16_t>, dim3(n_pack), dim3(256), 0, stream, layers, tab_pack, params, sn_state,
                           pass_uv, norms, (bf16_t*)packed, training);
    else
        return L2I_ERR_ARG;
    return l2i_check_launch();
}

extern "C" int l2i_weights_backward(const long long* layers, int n_layers, const int* tab_dot, int n_dot,
                                    const int* tab_apply, int n_apply, const float* params, const float* dwbar,
                                    const float* pass_uv, float* norms, float* grads, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!layers || !params || !dwbar || !norms || !grads) return L2I_ERR_ARG;
    (void)n_layers;
    if (n_dot > 0)
        hipLaunchKernelGGL(sn_dot_kernel, dim3(n_dot), dim3(256), 0, stream, layers, tab_dot, params, dwbar, norms);
    if (n_apply > 0)
        hipLaunchKernelGGL(sn_apply_kernel, dim3(n_apply), dim3(256), 0, stream, layers, tab_apply, dwbar, pass_uv, norms,
                           grads);
    return l2i_check_launch();
}
