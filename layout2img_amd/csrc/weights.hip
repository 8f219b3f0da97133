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
// All of it is HBM-bound streaming over the f32 master weights (D: ~60 M, G: ~35 M parameters): every kernel
// reads W in its natural [Co][Ci][taps] order with coalesced loads and goes through LDS where the destination
// order differs (the two packs, the gradient), so that each phase moves its bytes once.
//
// Layer table: L2I_LSTRIDE (20) int64 per layer row (see layout2img_amd/arena.py); a weight that the reference
// applies k times per forward (block_obj4, model/rcnn_discriminator_app.py:137,141 -- each application runs its own
// power iteration) has k rows sharing w/u/v offsets and is processed in k sequential rounds:
//   0 w_off  1 u_off(-1 = no SN)  2 v_off  3 Co  4 Ci  5 KH  6 Co_p  7 Ci_p
//   8 Kpad  9 Npad  10 fwd_off  11 Kpad_d  12 Npad_d  13 dg_off  14 dw_off  15 eps(float bits)
//   16 pass_u_off  17 pass_v_off (offsets into the per-pass u/v snapshot)  18 multi-use (gradient needs atomics)
//   19 offset of the layer's sn_wv block shares in the scratch of l2i_weights_prepare (one float per block of 4 R rows)
// norms: f32 [L][4] = {||W^T u||^2, ||W v||^2 (eval: u.Wv), sigma, <G, W>}
#include "common.h"

#define LF(i) (L[(i)])
#define L2I_LSTRIDE 20

__device__ __forceinline__ float layer_eps(const long long* L) { return __uint_as_float((uint32_t)L[15]); }

// ---------------------------------------------------------------- phase 1: t = W^T u
// table (layer, col0, row0, rows per wave, partial offset): a block covers 256 columns (one 16-byte load per lane and row) x 4 x `rows
// per wave` rows; its four waves walk different rows of the SAME columns, their partial column sums meet in LDS and are STORED as the
// block's run of 256 columns in row (row0 / rows per block) of the layer's partial matrix tpart[rows blocks][Kt] (caller's scratch);
// sn_tfold_kernel adds the row blocks in order (round 6: t, hence sigma and every packed weight, is bit-identical from run to run --
// rounds 1-5 left one float atomic per column and block, and the 1e-7 run-to-run noise of sigma moved bf16 roundings of the packs). (Round 2's form -- 1024 columns x 64 rows per block, four atomics per thread -- left the
// last column chunk of most layers half empty (Kt = 9 Ci is a multiple of 256, rarely of 1024) and gave the discriminator
// 960 blocks for 256 CUs: 2.5 TB/s.)
__global__ __launch_bounds__(256) void sn_wtu_kernel(const long long* __restrict__ layers, const int* __restrict__ table,
                                                     const float* __restrict__ params, const float* __restrict__ sn_state,
                                                     float* __restrict__ tpart) {
    __shared__ float part[4][256];
    const int* e = table + 5 * blockIdx.x;
    const long long* L = layers + L2I_LSTRIDE * e[0];
    const int Co = (int)LF(3), Kt = (int)(LF(4) * LF(5) * LF(5));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = min(Co, e[2] + wave * e[3]), r1 = min(Co, r0 + e[3]);
    const float* W = params + LF(0);
    const float* u = sn_state + LF(1);
    float* t_out = tpart + e[4];
    if ((Kt & 3) == 0) {
        const int col = e[1] + 4 * lane;
        float4 t = make_float4(0, 0, 0, 0);
        if (col < Kt) {
            int r = r0;
            // eight rows per batch, the eight loads issued back to back (an `unroll 8` of the plain loop keeps its per-row
            // bound check between the loads and waits for each), then single rows
            for (; r + 8 <= r1; r += 8) {
                float4 w[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) w[j] = *reinterpret_cast<const float4*>(W + (size_t)(r + j) * Kt + col);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float ur = u[r + j];
                    t.x = fmaf(ur, w[j].x, t.x); t.y = fmaf(ur, w[j].y, t.y); t.z = fmaf(ur, w[j].z, t.z); t.w = fmaf(ur, w[j].w, t.w);
                }
            }
            for (; r < r1; ++r) {
                const float4 w = *reinterpret_cast<const float4*>(W + (size_t)r * Kt + col);
                const float ur = u[r];
                t.x = fmaf(ur, w.x, t.x); t.y = fmaf(ur, w.y, t.y); t.z = fmaf(ur, w.z, t.z); t.w = fmaf(ur, w.w, t.w);
            }
        }
        *reinterpret_cast<float4*>(&part[wave][4 * lane]) = t;
    } else {   // rows not 16-byte aligned (Ci = 3, odd class counts): scalar columns
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int col = e[1] + c * 64 + lane;
            float t = 0.f;
            if (col < Kt) {
#pragma unroll 8
                for (int r = r0; r < r1; ++r) t = fmaf(u[r], W[(size_t)r * Kt + col], t);
            }
            part[wave][c * 64 + lane] = t;
        }
    }
    __syncthreads();
    const int col = e[1] + threadIdx.x;
    if (col < Kt) t_out[col] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// t[col] = sum over the row blocks, in a fixed order. table: (layer, col0, partial offset of the layer, row blocks, columns per workgroup cw):
// a workgroup = cw columns x (256 / cw) row lanes (cw = 256 for layers with few row blocks, 16 for the 16384-row fc layer: 256 blocks), the lanes
// meet in a fixed-order tree in LDS.
__global__ __launch_bounds__(256) void sn_tfold_kernel(const long long* __restrict__ layers, const int* __restrict__ table,
                                                       const float* __restrict__ tpart, float* __restrict__ pass_uv) {
    __shared__ float red[256];
    const int* e = table + 5 * blockIdx.x;
    const long long* L = layers + L2I_LSTRIDE * e[0];
    const int Kt = (int)(LF(4) * LF(5) * LF(5));
    const int cw = e[4], rl = 256 / cw;
    const int tx = threadIdx.x & (cw - 1), ty = threadIdx.x / cw;
    const int col = e[1] + tx;
    const int nrb = e[3];
    float t = 0.f;
    if (col < Kt) {
        const float* q = tpart + e[2] + col;
        int r = ty;
        for (; r + 3 * rl < nrb; r += 4 * rl) {
            const float a = q[(size_t)r * Kt], b = q[(size_t)(r + rl) * Kt], c = q[(size_t)(r + 2 * rl) * Kt], d = q[(size_t)(r + 3 * rl) * Kt];
            t += (a + b) + (c + d);
        }
        for (; r < nrb; r += rl) t += q[(size_t)r * Kt];
    }
    if (rl > 1) {
        red[threadIdx.x] = t;
        __syncthreads();
        for (int st = rl >> 1; st >= 1; st >>= 1) {
            if (ty < st) red[threadIdx.x] += red[threadIdx.x + st * cw];
            __syncthreads();
        }
        t = red[threadIdx.x];
    }
    if (ty == 0 && col < Kt) pass_uv[LF(17) + col] = t;
}

// ---------------------------------------------------------------- phase 2: s = W vhat
// (train: vhat = t / max(||t||, eps); eval: vhat = stored v). 4 R rows per block, R per wave read together so the
// vector chunk is loaded once per R rows. table: (layer, row0)
template <int R>
__global__ __launch_bounds__(256) void sn_wv_kernel(const long long* __restrict__ layers, const int* __restrict__ table,
                                                    const float* __restrict__ params, const float* __restrict__ sn_state,
                                                    float* __restrict__ pass_uv, float* __restrict__ norms, float* __restrict__ npart, int training) {
    __shared__ float red[16];
    const int* e = table + 2 * blockIdx.x;
    const int layer = e[0];
    const long long* L = layers + L2I_LSTRIDE * layer;
    const int Co = (int)LF(3), Kt = (int)(LF(4) * LF(5) * LF(5));
    const float* W = params + LF(0);
    const float* vsrc = training ? pass_uv + LF(17) : sn_state + LF(2);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rbase = e[1] + wave * R;
    float dot[R], tn2 = 0.f;
    const float* Wr[R];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        dot[rr] = 0.f;
        Wr[rr] = W + (size_t)min(rbase + rr, Co - 1) * Kt;   // clamped rows are discarded below
    }
    if ((Kt & 3) == 0) {
        constexpr int U = 8 / R;   // eight 16-byte loads of W in flight per lane
        int k = 4 * lane;
        for (; k + 256 * (U - 1) < Kt; k += 256 * U) {
            float4 t[U], w[U][R];
#pragma unroll
            for (int q = 0; q < U; ++q) {
                t[q] = *reinterpret_cast<const float4*>(vsrc + k + 256 * q);
#pragma unroll
                for (int rr = 0; rr < R; ++rr) w[q][rr] = *reinterpret_cast<const float4*>(Wr[rr] + k + 256 * q);
            }
#pragma unroll
            for (int q = 0; q < U; ++q) {
                tn2 += t[q].x * t[q].x + t[q].y * t[q].y + t[q].z * t[q].z + t[q].w * t[q].w;
#pragma unroll
                for (int rr = 0; rr < R; ++rr)
                    dot[rr] += w[q][rr].x * t[q].x + w[q][rr].y * t[q].y + w[q][rr].z * t[q].z + w[q][rr].w * t[q].w;
            }
        }
        for (; k < Kt; k += 256) {
            const float4 t = *reinterpret_cast<const float4*>(vsrc + k);
            tn2 += t.x * t.x + t.y * t.y + t.z * t.z + t.w * t.w;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const float4 w = *reinterpret_cast<const float4*>(Wr[rr] + k);
                dot[rr] += w.x * t.x + w.y * t.y + w.z * t.z + w.w * t.w;
            }
        }
    } else {
        for (int k = lane; k < Kt; k += 64) {
            const float t = vsrc[k];
            tn2 += t * t;
#pragma unroll
            for (int rr = 0; rr < R; ++rr) dot[rr] += Wr[rr][k] * t;
        }
    }
    tn2 = wave_sum(tn2);
    float blk = 0.f;
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        const int r = rbase + rr;
        const float d = wave_sum(dot[rr]);
        if (r >= Co) continue;
        float s;
        if (training) {
            s = d / fmaxf(sqrtf(tn2), layer_eps(L));
            blk += s * s;
        } else {
            s = d;
            blk += s * sn_state[LF(1) + r];  // sigma = u . (W v)
        }
        if (lane == 0) pass_uv[LF(16) + r] = s;
    }
    __syncthreads();
    if (lane == 0) red[wave] = blk;
    __syncthreads();
    if (threadIdx.x == 0) {   // the block's share of ||W v||^2 (eval: u . W v): slot (row0 / rows per block) of the layer's run in npart (layer row 19);
        npart[LF(19) + e[1] / (4 * R)] = (red[0] + red[1]) + (red[2] + red[3]);   // layer_sn2 adds the slots in a fixed order (no atomics: round 6)
        if (e[1] == 0) norms[4 * layer + 0] = tn2;
    }
}

// ||W v||^2 (eval: u . W v) of a layer from the stored shares of its sn_wv blocks -- the whole workgroup calls this; every thread gets the
// sum, formed in an order that depends on the layer's shape only. red: 16 floats of LDS.
__device__ __forceinline__ float layer_sn2(const long long* L, const float* npart, int wv_rows, float* red) {
    const int nb = ((int)LF(3) + wv_rows - 1) / wv_rows;
    float a = 0.f;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) a += npart[LF(19) + i];
    return block_sum(a, red);
}

__device__ __forceinline__ float sigma_of(const long long* L, float sn2, int training) {
    return training ? sn2 / fmaxf(sqrtf(sn2), layer_eps(L)) : sn2;
}

// ---------------------------------------------------------------- phase 3a: packs
// One block = a tile of 64 output channels x TCI input channels x all taps of one layer (TCI = 32 for 3x3, 256 for
// 1x1 / linear). The tile is read in W's own order (per co a contiguous run of TCI*taps floats), scaled by 1/sigma
// into LDS, and written twice: forward pack [n = co][k = tap*Ci_p + ci] (runs of TCI elements) and dgrad pack
// [n = ci][k = (taps-1-tap)*Co_p + co] (runs of 64 elements). Only elements with co < Co_p / ci < Ci_p are written:
// the pack buffer's padding rows and K tails are zero from its allocation (arena.py pools the buffers) and stay zero.
// table: (layer, cotile, cichunk)
#define PK_TCO 64
template <typename T>
__device__ __forceinline__ void store8(T* dst, const float (&v)[8]) {
    if constexpr (sizeof(T) == 2) {
        uint4 pk;
        pk.x = f2bf2(v[0], v[1]);
        pk.y = f2bf2(v[2], v[3]);
        pk.z = f2bf2(v[4], v[5]);
        pk.w = f2bf2(v[6], v[7]);
        *reinterpret_cast<uint4*>(dst) = pk;
    } else {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

template <typename T>
__device__ __forceinline__ void store8t(T* dst, const T (&v)[8]) {   // 8 already-converted operand values
    if constexpr (sizeof(T) == 2) {
        uint4 pk;
        pk.x = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
        pk.y = (uint32_t)v[2] | ((uint32_t)v[3] << 16);
        pk.z = (uint32_t)v[4] | ((uint32_t)v[5] << 16);
        pk.w = (uint32_t)v[6] | ((uint32_t)v[7] << 16);
        *reinterpret_cast<uint4*>(dst) = pk;
    } else {
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

// (TAPS is a template parameter: with the tap count a run-time value every index computation of the tile loops was an
//  emulated integer division -- the kernel ran at 35 % of the HBM rate, profiles/r02_hbm_kernels.json)
// SPLIT (bf16 only; dtype code 3 of l2i_weights_prepare, "bf16x3"): the FORWARD pack holds every weight as hi + lo, both bf16
// (hi = bf16(w), lo = bf16(w - hi): w to 16 significant bits), as three Ci_p-wide blocks [hi | hi | lo] per tap. Against an
// operand whose channels are [x_hi | x_lo | x_hi] (l2i_split_cast) the unchanged convolution kernels then accumulate
// x_hi w_hi + x_lo w_hi + x_hi w_lo in f32 -- the product to ~2^-16 relative instead of bf16's 2^-8, at three times the MFMA
// work: the forward-only precision mode that meets the image bar of 1e-3 at MFMA speed. The data-gradient pack is unchanged.
template <typename T, int TAPS, bool SPLIT = false>
__device__ __forceinline__ void sn_pack_body(const long long* __restrict__ L, const int* __restrict__ e, int layer,
                                             const float* __restrict__ params, float sigma,
                                             T* __restrict__ packed, int training, T* tile) {
    const int Co = (int)LF(3), Ci = (int)LF(4), Co_p = (int)LF(6), Ci_p = (int)LF(7);
    constexpr int taps = TAPS;
    constexpr int TCI = TAPS == 1 ? 256 : 32;
    constexpr int RUN = TCI * TAPS, RUNP = RUN + 2;
    T* tile_lo = tile + PK_TCO * RUNP;   // (SPLIT: the low halves, same layout)
    const int co0 = e[1] * PK_TCO, ci0 = e[2] * TCI;
    const int nco = min(PK_TCO, Co - co0), nrun = min(TCI, Ci - ci0) * taps;   // valid rows / valid floats per row (may be <= 0)
    const float inv = 1.f / sigma;
    const float* W = params + LF(0);
    // The tile is kept TAP-major in LDS ([row][tap][ci], row pitch RUNP): the forward pack's 8 consecutive input
    // channels of one tap are then one 16-byte LDS read instead of eight 2-byte gathers (the kernel was bound by its
    // instruction count, not by HBM). W is read in its own order, four floats per lane where the run allows it.
    const bool vec = ((Ci * taps) & 3) == 0 && ((ci0 * taps) & 3) == 0 && (nrun & 3) == 0;
    if (vec && nco > 0 && nrun > 0) {
        // Loads in batches, every one unconditional (clamped address, value dropped afterwards): a load under a condition
        // is waited for before the next one is issued, which made this loop 18 serial round trips to HBM.
        constexpr int NIT = PK_TCO * (RUN / 4) / 256, BATCH = NIT / 2;
        static_assert(PK_TCO * (RUN / 4) % 256 == 0 && NIT % 2 == 0, "tile / thread mapping");
#pragma unroll
        for (int b0 = 0; b0 < NIT; b0 += BATCH) {
            float4 v[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int idx = threadIdx.x + (b0 + u) * 256;
                const int row = idx / (RUN / 4), j = 4 * (idx - row * (RUN / 4));
                v[u] = *reinterpret_cast<const float4*>(W + ((size_t)(co0 + min(row, nco - 1)) * Ci + ci0) * taps + min(j, nrun - 4));
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int idx = threadIdx.x + (b0 + u) * 256;
                const int row = idx / (RUN / 4), j = 4 * (idx - row * (RUN / 4));
                const float sc = (row < nco && j < nrun) ? inv : 0.f;
                const float vv[4] = {v[u].x * sc, v[u].y * sc, v[u].z * sc, v[u].w * sc};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int jj = j + q, cil = jj / taps, tap = jj - cil * taps;
                    const T hi = OpT<T>::from(vv[q]);
                    tile[row * RUNP + tap * TCI + cil] = hi;
                    if constexpr (SPLIT) tile_lo[row * RUNP + tap * TCI + cil] = OpT<T>::from(vv[q] - OpT<T>::to(hi));
                }
            }
        }
    } else {
#pragma unroll 4
        for (int idx = threadIdx.x; idx < PK_TCO * RUN; idx += 256) {
            const int row = idx / RUN, j = idx - row * RUN;
            float v = 0.f;
            if (row < nco && j < nrun) v = W[((size_t)(co0 + row) * Ci + ci0) * taps + j] * inv;
            const int cil = j / taps, tap = j - cil * taps;
            const T hi = OpT<T>::from(v);
            tile[row * RUNP + tap * TCI + cil] = hi;
            if constexpr (SPLIT) tile_lo[row * RUNP + tap * TCI + cil] = OpT<T>::from(v - OpT<T>::to(hi));
        }
    }
    __syncthreads();
    // forward pack
    {
        const int Kpad = (int)LF(8);
        T* dst = packed + LF(10);
        constexpr int c8n = TCI / 8;
        for (int u = threadIdx.x; u < PK_TCO * taps * c8n; u += 256) {
            const int c8 = u % c8n, rest = u / c8n;
            const int tap = rest % taps, row = rest / taps;
            const int co = co0 + row, ci = ci0 + 8 * c8;
            if (co >= Co || ci >= Ci_p) continue;
            T v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[row * RUNP + tap * TCI + 8 * c8 + j];
            if constexpr (SPLIT) {
                T* d3 = dst + (size_t)co * Kpad + tap * 3 * Ci_p + ci;
                store8t<T>(d3, v);
                store8t<T>(d3 + Ci_p, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = tile_lo[row * RUNP + tap * TCI + 8 * c8 + j];
                store8t<T>(d3 + 2 * Ci_p, v);
            } else {
                store8t<T>(dst + (size_t)co * Kpad + tap * Ci_p + ci, v);
            }
        }
    }
    // dgrad pack (taps flipped)
    {
        const int Kpad_d = (int)LF(11);
        T* dst = packed + LF(13);
        for (int u = threadIdx.x; u < TCI * taps * (PK_TCO / 8); u += 256) {
            const int co8 = u % (PK_TCO / 8), rest = u / (PK_TCO / 8);
            const int tap = rest % taps, cil = rest / taps;
            const int ci = ci0 + cil, co = co0 + 8 * co8;
            if (ci >= Ci || co >= Co_p) continue;
            T v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[(8 * co8 + j) * RUNP + tap * TCI + cil];
            store8t<T>(dst + (size_t)ci * Kpad_d + (taps - 1 - tap) * Co_p + co, v);
        }
    }
}

template <typename T, bool SPLIT = false>
__global__ __launch_bounds__(256) void sn_pack_kernel(const long long* __restrict__ layers, const int* __restrict__ table,
                                                      const float* __restrict__ params, const float* __restrict__ npart, int wv_rows,
                                                      T* __restrict__ packed, int training) {
    extern __shared__ __attribute__((aligned(16))) char tile_raw[];
    __shared__ float red[16];
    T* tile = reinterpret_cast<T*>(tile_raw);   // [64][RUN + 2] in the OPERAND type: 37 KB for bf16 -> four workgroups per CU in flight
    const int* e = table + 3 * blockIdx.x;
    const int layer = e[0];
    const long long* L = layers + L2I_LSTRIDE * layer;
    const float sigma = LF(1) < 0 ? 1.f : sigma_of(L, layer_sn2(L, npart, wv_rows, red), training);
    if ((int)LF(5) == 3) sn_pack_body<T, 9, SPLIT>(L, e, layer, params, sigma, packed, training, tile);
    else sn_pack_body<T, 1, SPLIT>(L, e, layer, params, sigma, packed, training, tile);
}

// ---------------------------------------------------------------- phase 3b: sigma, normalised u / v
// One block per layer row of the round: writes sigma, and (train) u <- s/||s||, v <- t/||t|| into the pass snapshot
// and the persistent state; (eval) copies the stored u, v into the snapshot. table: (layer)
__global__ __launch_bounds__(256) void sn_finish_kernel(const long long* __restrict__ layers, const int* __restrict__ table,
                                                        float* __restrict__ sn_state, float* __restrict__ pass_uv,
                                                        float* __restrict__ norms, const float* __restrict__ npart, int wv_rows, int training) {
    __shared__ float red[16];
    const int layer = table[blockIdx.x];
    const long long* L = layers + L2I_LSTRIDE * layer;
    if (LF(1) < 0) {
        if (threadIdx.x == 0 && blockIdx.y == 0) { norms[4 * layer + 0] = 0.f; norms[4 * layer + 1] = 0.f; norms[4 * layer + 2] = 1.f; norms[4 * layer + 3] = 0.f; }
        return;
    }
    const int Co = (int)LF(3), Kt = (int)(LF(4) * LF(5) * LF(5));
    const float eps = layer_eps(L);
    const float sn2 = layer_sn2(L, npart, wv_rows, red), tn2 = norms[4 * layer + 0];
    // (gridDim.y blocks share a layer's vectors: one block per layer took 16 us for the 9216-element v of a 1024-channel 3x3 layer)
    const int i0 = blockIdx.y * 256 + threadIdx.x, istep = 256 * gridDim.y;
    if (training) {
        const float iu = 1.f / fmaxf(sqrtf(sn2), eps), iv = 1.f / fmaxf(sqrtf(tn2), eps);
        for (int i = i0; i < Co; i += istep) {
            const float u = pass_uv[LF(16) + i] * iu;
            pass_uv[LF(16) + i] = u;
            sn_state[LF(1) + i] = u;
        }
        for (int i = i0; i < Kt; i += istep) {
            const float v = pass_uv[LF(17) + i] * iv;
            pass_uv[LF(17) + i] = v;
            sn_state[LF(2) + i] = v;
        }
    } else {
        for (int i = i0; i < Co; i += istep) pass_uv[LF(16) + i] = sn_state[LF(1) + i];
        for (int i = i0; i < Kt; i += istep) pass_uv[LF(17) + i] = sn_state[LF(2) + i];
    }
    if (threadIdx.x == 0 && blockIdx.y == 0) {
        norms[4 * layer + 1] = sn2;
        norms[4 * layer + 2] = sigma_of(L, sn2, training);
        norms[4 * layer + 3] = 0.f;   // (<G, W>: written by the backward; defined from here on -- the buffer is no longer cleared per pass)
    }
}

// ---------------------------------------------------------------- backward
// Both backward kernels walk W's (co, ci) pairs in chunks of 256: the chunk's W / gradient run (256*taps floats) is
// contiguous, and the matching entries of G = dWbar ([co][tap][Ci_p] order) are contiguous in ci per tap; LDS
// converts between the two orders so that every global access is a run. table: (layer, pairchunk)
#define BW_PAIRS 256
// pairs per table entry: 256 for the 3x3 layers (2304 floats of W); 1x1 / linear layers take 2304 pairs per block as well (nine per
// thread) -- at 256 they were 1 KB blocks, as many of them as the 3x3 layers have for a fifteenth of the bytes (arena.py builds the
// tables; tools/perf/sn_bwd_micro.py: the generator's flush 219 -> 150 us)
// 3x3 layers with Ci % 4 == 0: 1024 pairs per block, a thread owns FOUR consecutive input channels of one output channel -- its dWbar
// loads are 16 bytes (a wave reads 1 KB runs instead of 256 B ones) and four times the bytes are in flight per thread
#define BW_QUAD 1024
__device__ __forceinline__ int bw_chunk(int taps, int Ci) { return taps == 1 ? BW_PAIRS * 9 : (taps == 9 && (Ci & 3) == 0) ? BW_QUAD : BW_PAIRS; }   // sn_apply
__device__ __forceinline__ int bw_chunk_dot(int taps) { return taps == 1 ? BW_PAIRS * 9 : BW_PAIRS; }   // sn_dot: the quad form measured no faster there (143 -> 145 us)

// phase a: <G, W> per SN layer: every block STORES its share at ws[block * NP + pass] and sn_dotfold_kernel adds a layer's shares in order into
// ws[n_dot * NP + pass * n_layers + layer] (round 6; rounds 1-5: one float atomic per block into 32 replicas, and the correction term
// <G, Wbar> u v^T of every spectrally normalised weight moved in its last bits from run to run).
// NP = 2: the two passes of one optimiser step that share W (D(real) and D(fake), train_context_app_v2.py:158,167) in one
// launch -- W is read once, each pass's G once.
template <int NP>
__global__ __launch_bounds__(256) void sn_dot_kernel(const long long* __restrict__ layers, const int* __restrict__ table,
                                                     const float* __restrict__ params, const float* __restrict__ dwbar0,
                                                     const float* __restrict__ dwbar1, float* __restrict__ ws, int n_layers) {
    __shared__ float red[16];
    __shared__ float wl[BW_PAIRS * 9];
    const int* e = table + 2 * blockIdx.x;
    const int layer = e[0];
    const long long* L = layers + L2I_LSTRIDE * layer;
    const int Co = (int)LF(3), Ci = (int)LF(4), KH = (int)LF(5), Ci_p = (int)LF(7);
    const int taps = KH * KH, Kp = taps * Ci_p;
    const long long p0 = (long long)e[1] * bw_chunk_dot(taps);
    const int np = (int)min((long long)bw_chunk_dot(taps), (long long)Co * Ci - p0);
    const float* W = params + LF(0) + p0 * taps;
    float acc[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) acc[q] = 0.f;
    if (taps == 1) {   // W and G share the (co, ci) order up to the row pitch: nine pairs per thread, loads issued together
        float wv[9], gv[NP][9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int j = min((int)threadIdx.x + 256 * k, np - 1);
            const long long pr = p0 + j;
            const int co = (int)(pr / Ci), ci = (int)(pr - (long long)co * Ci);
            wv[k] = (int)threadIdx.x + 256 * k < np ? W[j] : 0.f;
#pragma unroll
            for (int q = 0; q < NP; ++q) gv[q][k] = (q == 0 ? dwbar0 : dwbar1)[(size_t)LF(14) + (size_t)co * Kp + ci];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int q = 0; q < NP; ++q) acc[q] = fmaf(wv[k], gv[q][k], acc[q]);
    } else {
    for (int j = threadIdx.x; j < np * taps; j += 256) wl[j] = W[j];
    __syncthreads();
    if ((int)threadIdx.x < np) {
        const long long pr = p0 + threadIdx.x;
        const int co = (int)(pr / Ci), ci = (int)(pr - (long long)co * Ci);
        const size_t goff = (size_t)LF(14) + (size_t)co * Kp + ci;
        if (taps == 9) {   // the strided loads issued together (a run-time trip count waits for each in turn)
            float gv[NP][9];
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) gv[q][tap] = (q == 0 ? dwbar0 : dwbar1)[goff + tap * Ci_p];
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) acc[q] = fmaf(wl[threadIdx.x * 9 + tap], gv[q][tap], acc[q]);
        } else {
            for (int tap = 0; tap < taps; ++tap)
#pragma unroll
                for (int q = 0; q < NP; ++q) acc[q] = fmaf(wl[threadIdx.x * taps + tap], (q == 0 ? dwbar0 : dwbar1)[goff + tap * Ci_p], acc[q]);
        }
    }
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const float t = block_sum(acc[q], red);
        if (threadIdx.x == 0) ws[(size_t)blockIdx.x * NP + q] = t;
        (void)layer; (void)n_layers;
    }
}

// one block per layer row: dsum[pass * n_layers + layer] = the sum of the layer's sn_dot shares, in a fixed order.
// range: (first dot-table entry, number of entries) per layer row, absolute; dot_base: the first entry of THIS launch's sub-table.
template <int NP>
__global__ __launch_bounds__(256) void sn_dotfold_kernel(const int* __restrict__ range, int dot_base, int n_dot, const float* __restrict__ part,
                                                         float* __restrict__ dsum, int n_layers) {
    __shared__ float red[16];
    const int layer = blockIdx.x;
    const int start = range[2 * layer] - dot_base, cnt = range[2 * layer + 1];
    if (cnt <= 0 || start < 0 || start >= n_dot) return;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        float a = 0.f;
        for (int i = threadIdx.x; i < cnt; i += 256) a += part[(size_t)(start + i) * NP + q];
        const float t = block_sum(a, red);
        if (threadIdx.x == 0) dsum[(size_t)q * n_layers + layer] = t;
    }
}

// phase b: grads[w] += sum over the passes of (G - <G,Wbar> u v^T) / sigma   (non-SN layers: grads[w] += sum of G)
template <int NP>
__global__ __launch_bounds__(256) void sn_apply_kernel(const long long* __restrict__ layers, const int* __restrict__ table,
                                                       const float* __restrict__ dwbar0, const float* __restrict__ dwbar1,
                                                       const float* __restrict__ uv0, const float* __restrict__ uv1,
                                                       float* __restrict__ norms0, float* __restrict__ norms1,
                                                       const float* __restrict__ ws, int n_layers, float* __restrict__ grads, int overwrite) {
    __shared__ __attribute__((aligned(16))) float gl[BW_QUAD * 9];
    const int* e = table + 2 * blockIdx.x;
    const int layer = e[0];
    const long long* L = layers + L2I_LSTRIDE * layer;
    const int Co = (int)LF(3), Ci = (int)LF(4), KH = (int)LF(5), Ci_p = (int)LF(7);
    const int taps = KH * KH, Kp = taps * Ci_p;
    const long long p0 = (long long)e[1] * bw_chunk(taps, Ci);
    const int np = (int)min((long long)bw_chunk(taps, Ci), (long long)Co * Ci - p0);
    const bool sn = LF(1) >= 0;
    float inv[NP], gw[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        float* norms = q == 0 ? norms0 : norms1;
        inv[q] = 1.f / norms[4 * layer + 2];
        gw[q] = 0.f;
        if (sn) {
            const float d = ws[(size_t)q * n_layers + layer];   // (sn_dotfold_kernel's sum of the layer's shares)
            gw[q] = d * inv[q];  // <G, Wbar>
            if (e[1] == 0 && threadIdx.x == 0) norms[4 * layer + 3] = d;
        }
    }
    if (taps == 1) {   // nine (co, ci) pairs per thread, no reordering: straight to the gradient buffer
        float* dst1 = grads + LF(0) + p0;
        float gv[NP][9], vv[NP][9], uu[NP][9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int j = min((int)threadIdx.x + 256 * k, np - 1);
            const long long pr = p0 + j;
            const int co = (int)(pr / Ci), ci = (int)(pr - (long long)co * Ci);
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                gv[q][k] = (q == 0 ? dwbar0 : dwbar1)[(size_t)LF(14) + (size_t)co * Kp + ci];
                vv[q][k] = sn ? (q == 0 ? uv0 : uv1)[LF(17) + ci] : 0.f;
                uu[q][k] = sn ? (q == 0 ? uv0 : uv1)[LF(16) + co] : 0.f;
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int j = (int)threadIdx.x + 256 * k;
            if (j >= np) break;
            float x = 0.f;
#pragma unroll
            for (int q = 0; q < NP; ++q) x += sn ? (gv[q][k] - uu[q][k] * gw[q] * vv[q][k]) * inv[q] : gv[q][k];
            if (LF(18)) atomicAdd(dst1 + j, x);
            else if (overwrite) dst1[j] = x;
            else dst1[j] += x;
        }
        return;
    }
    if (taps == 9 && (Ci & 3) == 0) {   // four consecutive ci per thread (see bw_chunk)
        if (4 * (int)threadIdx.x < np) {
            const long long pq = p0 + 4 * threadIdx.x;
            const int co = (int)(pq / Ci), ci = (int)(pq - (long long)co * Ci);
            const size_t goff = (size_t)LF(14) + (size_t)co * Kp + ci;
            float4 gv[NP][9], v4[NP][9];
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) gv[q][tap] = *reinterpret_cast<const float4*>((q == 0 ? dwbar0 : dwbar1) + goff + tap * Ci_p);
            float uc[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                uc[q] = sn ? (q == 0 ? uv0 : uv1)[LF(16) + co] * gw[q] : 0.f;
#pragma unroll
                for (int j = 0; j < 9; ++j)   // v[(ci + k) * 9 + tap], k < 4: 36 consecutive floats
                    v4[q][j] = sn ? *reinterpret_cast<const float4*>((q == 0 ? uv0 : uv1) + LF(17) + (size_t)ci * 9 + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float x[36];   // [k][tap] = W's order
#pragma unroll
            for (int m = 0; m < 36; ++m) x[m] = 0.f;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const float* vf = reinterpret_cast<const float*>(v4[q]);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float g4[4] = {gv[q][tap].x, gv[q][tap].y, gv[q][tap].z, gv[q][tap].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) x[k * 9 + tap] += sn ? (g4[k] - uc[q] * vf[k * 9 + tap]) * inv[q] : g4[k];
                }
            }
#pragma unroll
            for (int j = 0; j < 9; ++j) reinterpret_cast<float4*>(gl)[9 * threadIdx.x + j] = make_float4(x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]);
        }
    } else
    if ((int)threadIdx.x < np) {
        const long long pr = p0 + threadIdx.x;
        const int co = (int)(pr / Ci), ci = (int)(pr - (long long)co * Ci);
        const size_t goff = (size_t)LF(14) + (size_t)co * Kp + ci;
        float uc[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) uc[q] = sn ? (q == 0 ? uv0 : uv1)[LF(16) + co] * gw[q] : 0.f;
        const size_t voff = (size_t)LF(17) + (size_t)ci * taps;
        if (taps == 9) {   // (loads issued together, see sn_dot_kernel)
            float gv[NP][9];
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) gv[q][tap] = (q == 0 ? dwbar0 : dwbar1)[goff + tap * Ci_p];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                float x = 0.f;
#pragma unroll
                for (int q = 0; q < NP; ++q) x += sn ? (gv[q][tap] - uc[q] * (q == 0 ? uv0 : uv1)[voff + tap]) * inv[q] : gv[q][tap];
                gl[threadIdx.x * 9 + tap] = x;
            }
        } else {
            for (int tap = 0; tap < taps; ++tap) {
                float x = 0.f;
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const float g = (q == 0 ? dwbar0 : dwbar1)[goff + tap * Ci_p];
                    x += sn ? (g - uc[q] * (q == 0 ? uv0 : uv1)[voff + tap]) * inv[q] : g;
                }
                gl[threadIdx.x * taps + tap] = x;
            }
        }
    }
    __syncthreads();
    float* dst = grads + LF(0) + p0 * taps;
    if (LF(18)) {   // rows of a multiply-applied weight update the same gradient concurrently
        for (int j = threadIdx.x; j < np * taps; j += 256) atomicAdd(dst + j, gl[j]);
    } else if (overwrite) {   // the gradient buffer is known to be zero here (first flush after zero_grad): no read
        if ((np * taps & 3) == 0)   // (16-byte stores: p0 * taps and the parameter offsets are multiples of 4)
            for (int j = threadIdx.x; j < np * taps / 4; j += 256) reinterpret_cast<float4*>(dst)[j] = reinterpret_cast<const float4*>(gl)[j];
        else
            for (int j = threadIdx.x; j < np * taps; j += 256) dst[j] = gl[j];
    } else if ((np * taps) % (9 * 256) == 0) {   // full blocks: the read-modify-writes of a thread in batches of nine
        for (int b0 = 0; b0 < np * taps; b0 += 9 * 256) {
            float dv[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) dv[q] = dst[b0 + threadIdx.x + 256 * q];
#pragma unroll
            for (int q = 0; q < 9; ++q) dst[b0 + threadIdx.x + 256 * q] = dv[q] + gl[b0 + threadIdx.x + 256 * q];
        }
    } else {
        for (int j = threadIdx.x; j < np * taps; j += 256) dst[j] += gl[j];
    }
}

extern "C" int l2i_weights_prepare(const long long* layers, int n_layers, const int* tab_wtu, int n_wtu,
                                   const int* tab_wv, int n_wv, const int* tab_pack, int n_pack, const int* tab_fin,
                                   int n_fin, const float* params, float* sn_state, float* pass_uv, long long uv_len,
                                   float* norms, void* packed, int dtype, int training, int clear, const int* tab_tfold, int n_tfold,
                                   float* scratch, long long scratch_floats, long long npart_floats, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!layers || !params || !packed || !norms) return L2I_ERR_ARG;
    // scratch = [npart: one float per sn_wv block, at the layers' row-19 offsets | tpart: the W^T u partial rows, at the tables' offsets]
    if ((n_wv > 0 || n_wtu > 0) && (!scratch || npart_floats < 0 || npart_floats > scratch_floats || (training && n_wtu > 0 && (!tab_tfold || n_tfold <= 0)))) return L2I_ERR_ARG;
    float* npart = scratch;
    float* tpart = scratch ? scratch + npart_floats : nullptr;
    if (dtype != 0 && dtype != 1 && dtype != 3) return L2I_ERR_ARG;   // 3: bf16 with split (hi + lo) forward packs, see sn_pack_body
    // Round 6: nothing of norms / pass_uv is ACCUMULATED into any more (t = W^T u is written by sn_tfold_kernel, u and ||t||^2 by sn_wv_kernel,
    // ||W v||^2 / sigma by sn_finish_kernel, <G, W> by the backward) -- every value a later kernel reads has been stored first, so the clear that
    // rounds 1-5 needed in front of the atomics is skipped in train and eval mode alike (7 launches per iteration); `clear == 2` still forces it.
    if (clear == 2) {  // (a caller that wants defined padding)
        const long long gap = pass_uv - norms;   // adjacent buffers (layout2img_amd/arena.py PassCtx): one memset for both
        if (uv_len > 0 && gap >= 4LL * n_layers && gap <= 4LL * n_layers + 64) {
            if (l2i_zero_async(norms, sizeof(float) * (size_t)(gap + uv_len), stream) != hipSuccess) return L2I_ERR_LAUNCH;
        } else {
            if (l2i_zero_async(norms, sizeof(float) * 4 * n_layers, stream) != hipSuccess) return L2I_ERR_LAUNCH;
            if (uv_len > 0 && l2i_zero_async(pass_uv, sizeof(float) * uv_len, stream) != hipSuccess) return L2I_ERR_LAUNCH;
        }
    }
    if (training && n_wtu > 0) {
        hipLaunchKernelGGL(sn_wtu_kernel, dim3(n_wtu), dim3(256), 0, stream, layers, tab_wtu, params, sn_state, tpart);
        hipLaunchKernelGGL(sn_tfold_kernel, dim3(n_tfold), dim3(256), 0, stream, layers, tab_tfold, tpart, pass_uv);
    }
    static const int wv_r = getenv("L2I_SN_WV_R") ? atoi(getenv("L2I_SN_WV_R")) : 4;   // must match layout2img_amd/arena.py (rows per block = 4 R)
    const int wv_rows = 4 * (wv_r == 2 ? 2 : 4);
    if (n_wv > 0) {
        if (wv_r == 2)
            hipLaunchKernelGGL(sn_wv_kernel<2>, dim3(n_wv), dim3(256), 0, stream, layers, tab_wv, params, sn_state, pass_uv, norms, npart, training);
        else
            hipLaunchKernelGGL(sn_wv_kernel<4>, dim3(n_wv), dim3(256), 0, stream, layers, tab_wv, params, sn_state, pass_uv, norms, npart, training);
    }
    if (n_pack > 0) {
        const size_t lds = (dtype == 0 ? sizeof(float) : sizeof(bf16_t)) * PK_TCO * (32 * 9 + 2);   // >= 64 * (256 + 2) elements
        static bool ready = false;
        if (!ready) {
            (void)hipFuncSetAttribute((const void*)sn_pack_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)sn_pack_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            ready = true;
        }
        if (dtype == 0)
            hipLaunchKernelGGL(sn_pack_kernel<float>, dim3(n_pack), dim3(256), lds, stream, layers, tab_pack, params, npart, wv_rows,
                               (float*)packed, training);
        else if (dtype == 3) {
            static bool ready3 = false;
            if (!ready3) {
                (void)hipFuncSetAttribute((const void*)sn_pack_kernel<bf16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds));
                ready3 = true;
            }
            hipLaunchKernelGGL((sn_pack_kernel<bf16_t, true>), dim3(n_pack), dim3(256), 2 * lds, stream, layers, tab_pack, params, npart, wv_rows,
                               (bf16_t*)packed, training);
        } else
            hipLaunchKernelGGL(sn_pack_kernel<bf16_t>, dim3(n_pack), dim3(256), lds, stream, layers, tab_pack, params, npart, wv_rows,
                               (bf16_t*)packed, training);
    }
    if (n_fin > 0)
        hipLaunchKernelGGL(sn_finish_kernel, dim3(n_fin, 8), dim3(256), 0, stream, layers, tab_fin, sn_state, pass_uv, norms, npart, wv_rows,
                           training);
    return l2i_check_launch();
}

extern "C" int l2i_weights_backward2(const long long* layers, int n_layers, const int* tab_dot, int n_dot,
                                     const int* tab_apply, int n_apply, const float* params, const float* dwbar0,
                                     const float* pass_uv0, float* norms0, const float* dwbar1, const float* pass_uv1,
                                     float* norms1, float* grads, float* ws, int overwrite, const int* dot_range, int dot_base, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!layers || !params || !dwbar0 || !norms0 || !grads || !ws || (n_dot > 0 && !dot_range)) return L2I_ERR_ARG;
    const int np = dwbar1 ? 2 : 1;
    if (np == 2 && (!pass_uv1 || !norms1)) return L2I_ERR_ARG;
    // ws = [n_dot x np shares of <G, W> | np x n_layers sums], cleared again behind the launches (the workspace is all-zero between calls)
    const size_t used = (size_t)n_dot * np + (size_t)np * n_layers;
    if (used > 32 * 4 * 1024) return L2I_ERR_ARG;   // L2I_WS_FLOATS
    float* dsum = ws + (size_t)n_dot * np;
    if (n_dot > 0) {
        if (np == 2) {
            hipLaunchKernelGGL(sn_dot_kernel<2>, dim3(n_dot), dim3(256), 0, stream, layers, tab_dot, params, dwbar0, dwbar1, ws, n_layers);
            hipLaunchKernelGGL(sn_dotfold_kernel<2>, dim3(n_layers), dim3(256), 0, stream, dot_range, dot_base, n_dot, ws, dsum, n_layers);
        } else {
            hipLaunchKernelGGL(sn_dot_kernel<1>, dim3(n_dot), dim3(256), 0, stream, layers, tab_dot, params, dwbar0, dwbar0, ws, n_layers);
            hipLaunchKernelGGL(sn_dotfold_kernel<1>, dim3(n_layers), dim3(256), 0, stream, dot_range, dot_base, n_dot, ws, dsum, n_layers);
        }
    }
    if (n_apply > 0) {
        if (np == 2)
            hipLaunchKernelGGL(sn_apply_kernel<2>, dim3(n_apply), dim3(256), 0, stream, layers, tab_apply, dwbar0, dwbar1, pass_uv0, pass_uv1,
                               norms0, norms1, dsum, n_layers, grads, overwrite);
        else
            hipLaunchKernelGGL(sn_apply_kernel<1>, dim3(n_apply), dim3(256), 0, stream, layers, tab_apply, dwbar0, dwbar0, pass_uv0, pass_uv0,
                               norms0, norms0, dsum, n_layers, grads, overwrite);
    }
    if (n_dot > 0 && l2i_zero_async(ws, sizeof(float) * ((used + 3) & ~(size_t)3), stream) != hipSuccess) return L2I_ERR_LAUNCH;
    return l2i_check_launch();
}

extern "C" int l2i_weights_backward(const long long* layers, int n_layers, const int* tab_dot, int n_dot,
                                    const int* tab_apply, int n_apply, const float* params, const float* dwbar,
                                    const float* pass_uv, float* norms, float* grads, float* ws, const int* dot_range, int dot_base, void* stream_) {
    return l2i_weights_backward2(layers, n_layers, tab_dot, n_dot, tab_apply, n_apply, params, dwbar, pass_uv, norms, nullptr, nullptr,
                                 nullptr, grads, ws, 0, dot_range, dot_base, stream_);
}
