// Pyramid-pooling stages of the generator's PSP mask head (reference model/resnet_generator_app_v2.py:724-752):
//   priors = [upsample_bilinear_ac(stage_s(adaptive_avg_pool_s(feats)), HxW) for s in (1, 2, 3, 6)];  cat(priors + [feats])
// Both resamplings are fixed SPARSE linear maps over the pixels of one image: a pixel lies in at most 10 pooling bins
// (1 + 1 + 4 + 4: the bins of AdaptiveAvgPool2d overlap when the size does not divide) and a bilinear sample reads at
// most 4 bins of its stage; both also factor into a map along x and one along y. The tables come from the host side,
// extracted from what torch's own ops give for the identity basis (layout2img_amd/generator.py: psp_taps).
// As torch ops this was 8 thin bmm launches each way (M or K = 1..36 against K or M = 4096: 25-70 us each), a 277 MB
// f32 concat and its cast. Here every direction streams the big tensors once and the concat is written once, in the
// operand dtype of the 3x3 bottleneck convolution that reads it:
//   pixel <- bins (concat forward, pooling backward): per-pixel tap tables, the small side (50 x F, 50 x C) in LDS,
//       no data-dependent loops or branches (a version that looped over the dense 36-bin rows spent its time in
//       serialised LDS latency: 214 us for the concat against 30 us of HBM time);
//   bins <- pixels (pooling forward, concat backward): separable two-pass reductions with plain stores (see below).
#include "common.h"

#define PSP_CH 128      // pixels per workgroup
#define PSP_MAXNB 64
#define PSP_MAXTA 16

// taps of the chunk -> LDS (as int2: index, weight bits)
__device__ __forceinline__ void psp_load_taps(int2* dst, const int* __restrict__ idx, const float* __restrict__ w, size_t base, int n) {
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = make_int2(idx[base + i], __float_as_int(w[base + i]));
}

// ---------------------------------------------------------------- the two reductions over pixels, separably
// Both maps factor into a map along x and one along y (rectangular pooling bins; bilinear = linear x linear), so the
// reductions over the pixels of an image run as
//   pass 1, one workgroup per image row:  T[b,y,q,:] = sum_x wx[q,x] in[b,y,x,:]       q = x-bin, numbered stage after stage
//   pass 2 (tiny):                        out[b,k,:] = sum_y wy[k,y] T[b,y,xq[k],:]    k = bin (stage, ky, kx)
// with plain stores only. (LDS float atomics were tried for pass 1 on 2-D tap tables: ds_add_f32 ran at well under one
// lane per clock, 0.4 - 1 ms per launch; global atomics on the 50 x C sums at ~14 atomics/ns.) A thread issues all the
// loads of its row segment before it uses any of them.
#define PSP_MAXQ 16     // x-bins over all stages (1 + 2 + 3 + 6 = 12)
#define PSP_MAXW 128

// pooling pass 1: thread = (4 channels, one of the workgroup's 8 image rows); wx [NQ][W]. A bin weight read from LDS serves four
// multiply-adds (round 4's form -- one channel per thread, half a row each -- issued one LDS broadcast per multiply-add and spent
// its 49 us there, for a 67 MB read); a thread walks its whole row, eight 16-byte loads in flight, so nothing is combined across threads.
#define PSP_PR 2   // image rows per workgroup = one wave (8 rows in a 256-thread workgroup: 256 workgroups for the whole map, 50 us)
__global__ __launch_bounds__(32 * PSP_PR) void psp_pool_rows_kernel(const float* __restrict__ feats, const float* __restrict__ wx,
                                                            float* __restrict__ T, int H, int W, int C, int NQ) {
    __shared__ float wl[PSP_MAXQ * PSP_MAXW];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < NQ * W; i += 32 * PSP_PR) wl[i] = wx[i];
    __syncthreads();
    const int cq = threadIdx.x & 31, y = blockIdx.x * PSP_PR + (threadIdx.x >> 5);
    if (y >= H) return;
    const float* row = feats + ((size_t)(b * H + y) * W) * C;
    float* dst = T + ((size_t)(b * H + y) * NQ) * C;
    for (int c = 4 * cq; c < C; c += 128) {
        float4 acc[PSP_MAXQ];
#pragma unroll
        for (int q = 0; q < PSP_MAXQ; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int x0 = 0; x0 < W; x0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(row + (size_t)min(x0 + u, W - 1) * C + c);   // (unconditional loads: all 8 in flight)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool in = x0 + u < W;
#pragma unroll
                for (int q = 0; q < PSP_MAXQ; ++q)
                    if (q < NQ) {
                        const float w = in ? wl[q * W + min(x0 + u, W - 1)] : 0.f;
                        acc[q].x = fmaf(w, v[u].x, acc[q].x); acc[q].y = fmaf(w, v[u].y, acc[q].y);
                        acc[q].z = fmaf(w, v[u].z, acc[q].z); acc[q].w = fmaf(w, v[u].w, acc[q].w);
                    }
            }
        }
#pragma unroll
        for (int q = 0; q < PSP_MAXQ; ++q)
            if (q < NQ) *reinterpret_cast<float4*>(dst + (size_t)q * C + c) = acc[q];
    }
}

// pass 2: out[b,k,d] = sum_y wy[k,y] T[b,y,xq[k],d]   (T [B][H][NQ][D], wy [NB][H], xq [NB])
__global__ __launch_bounds__(256) void psp_rows_reduce_kernel(const float* __restrict__ T, const float* __restrict__ wy,
                                                              const int* __restrict__ xq, float* __restrict__ out, int H, int NQ,
                                                              int D, int NB) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NB * D) return;
    const int k = i / D, d = i - k * D;
    const float* src = T + ((size_t)b * H * NQ + xq[k]) * D + d;
    const float* w = wy + (size_t)k * H;
    float acc = 0.f;
#pragma unroll 8
    for (int y = 0; y < H; ++y) acc = fmaf(w[y], src[(size_t)y * NQ * D], acc);
    out[(size_t)b * NB * D + i] = acc;
}

// ---------------------------------------------------------------- dfeats[b,p,c] = add[b,p,c] + add2[b,p,c] + cat[b,p,off+c] + sum_t aw[p,t] dpooled[b, aidx[p,t], c]
// add / add2: f32 gradients other readers of feats left for this launch (either may be null); cat: the gradient of the concat
// tensor (T_, rows of `cat_w` elements), whose last C columns are the concat branch's part of dfeats -- read here instead of
// being copied out to an f32 tensor first. dop (optional): bf16 operand copy of the result.
template <int TA, typename T_>
__global__ __launch_bounds__(256) void psp_pool_bwd_kernel(const float* __restrict__ dpooled, const int* __restrict__ aidx,
                                                           const float* __restrict__ aw, const float* __restrict__ add,
                                                           const float* __restrict__ add2, const T_* __restrict__ cat, int cat_w, int cat_off,
                                                           float* __restrict__ dfeats, bf16_t* __restrict__ dop, int HW, int C, int NB) {
    extern __shared__ float sm[];                     // dP [NB][C] | taps [PSP_CH][TA]
    float* dP = sm;
    int2* taps = reinterpret_cast<int2*>(sm + NB * C);
    const int b = blockIdx.y, p0 = blockIdx.x * PSP_CH;
    for (int i = threadIdx.x; i < NB * C; i += 256) dP[i] = dpooled[(size_t)b * NB * C + i];
    psp_load_taps(taps, aidx, aw, (size_t)p0 * TA, PSP_CH * TA);
    __syncthreads();
    const int c4n = C >> 2;
    for (int t0 = threadIdx.x; t0 < PSP_CH * c4n; t0 += 256) {
        const int pl = t0 / c4n, cq = t0 - pl * c4n;
        const size_t o = ((size_t)b * HW + p0 + pl) * C + 4 * cq;
        float4 acc = add ? *reinterpret_cast<const float4*>(add + o) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (add2) {
            const float4 a2 = *reinterpret_cast<const float4*>(add2 + o);
            acc.x += a2.x; acc.y += a2.y; acc.z += a2.z; acc.w += a2.w;
        }
        if (cat) {
            float v[4];
            Op4<T_>::load(cat + ((size_t)b * HW + p0 + pl) * cat_w + cat_off + 4 * cq, v);
            acc.x += v[0]; acc.y += v[1]; acc.z += v[2]; acc.w += v[3];
        }
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            const int2 tp = taps[pl * TA + t];
            const float w = __int_as_float(tp.y);
            const float4 d = *reinterpret_cast<const float4*>(dP + tp.x * C + 4 * cq);
            acc.x = fmaf(w, d.x, acc.x); acc.y = fmaf(w, d.y, acc.y); acc.z = fmaf(w, d.z, acc.z); acc.w = fmaf(w, d.w, acc.w);
        }
        *reinterpret_cast<float4*>(dfeats + o) = acc;
        if (dop) {
            const float v[4] = {acc.x, acc.y, acc.z, acc.w};
            Op4<bf16_t>::store(dop + o, v);
        }
    }
}

// ---------------------------------------------------------------- cat[b,p,:] = [sum_t uw y[uidx] for each stage] ++ feats[b,p,:]
// cat rows are NS*F + C elements of T. Tasks of 4 consecutive columns (F % 4 == 0: a task lies inside one stage).
template <typename T>
__global__ __launch_bounds__(256) void psp_expand_fwd_kernel(const float* __restrict__ feats, const float* __restrict__ y,
                                                             const int* __restrict__ uidx, const float* __restrict__ uw,
                                                             T* __restrict__ cat, int HW, int C, int F, int NB, int NS) {
    extern __shared__ float sm[];                     // Y [NB][F] | taps [PSP_CH][NS][4]
    float* Y = sm;
    int2* taps = reinterpret_cast<int2*>(sm + NB * F);
    const int b = blockIdx.y, p0 = blockIdx.x * PSP_CH;
    for (int i = threadIdx.x; i < NB * F; i += 256) Y[i] = y[(size_t)b * NB * F + i];
    psp_load_taps(taps, uidx, uw, (size_t)p0 * NS * 4, PSP_CH * NS * 4);
    __syncthreads();
    const int Wd = NS * F + C;
    const int f4n = F >> 2, g_pri = NS * f4n;
    {   // the copy of feats into the last C columns: four 16-byte loads in flight per thread (one task at a time, mixed with the
        // interpolation tasks below, left every load waiting for its own round trip: 61 us for 205 MB; round 5)
        const int c4n = C >> 2, nt = PSP_CH * c4n;
        for (int t0 = threadIdx.x; t0 < nt; t0 += 1024) {
            float4 f[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = min(t0 + 256 * u, nt - 1);
                const int pl = t / c4n, c = 4 * (t - pl * c4n);
                f[u] = *reinterpret_cast<const float4*>(feats + ((size_t)b * HW + p0 + pl) * C + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + 256 * u;
                if (t >= nt) break;
                const int pl = t / c4n, c = 4 * (t - pl * c4n);
                const float v[4] = {f[u].x, f[u].y, f[u].z, f[u].w};
                Op4<T>::store(cat + ((size_t)b * HW + p0 + pl) * Wd + NS * F + c, v);
            }
        }
    }
    const int g_all = g_pri;
    for (int t0 = threadIdx.x; t0 < PSP_CH * g_all; t0 += 256) {
        const int pl = t0 / g_all, gq = t0 - pl * g_all;
        T* row = cat + ((size_t)b * HW + p0 + pl) * Wd;
        float v[4];
        {
            const int s = gq / f4n, j = 4 * (gq - s * f4n);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int2 tp = taps[(pl * NS + s) * 4 + t];
                const float w = __int_as_float(tp.y);
                const float4 yy = *reinterpret_cast<const float4*>(Y + tp.x * F + j);
                acc.x = fmaf(w, yy.x, acc.x); acc.y = fmaf(w, yy.y, acc.y); acc.z = fmaf(w, yy.z, acc.z); acc.w = fmaf(w, yy.w, acc.w);
            }
            v[0] = acc.x; v[1] = acc.y; v[2] = acc.z; v[3] = acc.w;
            Op4<T>::store(row + s * F + j, v);
        }
    }
}

// ---------------------------------------------------------------- backward of the above
//   dy[b,k,j] = sum_p U[p,k] g[b,p,s*F+j]  (k in stage s);   dfeats[b,p,c] = g[b,p,NS*F+c]
// pass 1 per image row: a thread owns a prior column (s, j), loads its W values of the row at once and forms the <= 6
// sums over x of its stage's x-bins: T[b,y,q,j] (wxt [NQ][W] = bilinear weight of x-bin q at column x; qoff [NS+1] = first
// x-bin of each stage); pass 2 is psp_rows_reduce_kernel with the bilinear weights along y.
template <typename T_, int WMAX>
__global__ __launch_bounds__(256) void psp_expand_rows_kernel(const T_* __restrict__ g, const float* __restrict__ wxt,
                                                              const int* __restrict__ qoff, float* __restrict__ T,
                                                              float* __restrict__ dfeats, int p_H, int W, int C, int F, int NQ, int NS) {
    __shared__ float wl[PSP_MAXQ * PSP_MAXW];
    __shared__ int qo[16];
    // thread = (4 consecutive prior columns of one stage, one of the workgroup's 2 image rows): 8-byte loads, and a bin weight read
    // from LDS serves four multiply-adds (one column per thread: one LDS read per multiply-add and 2-byte loads, 54 us; round 5)
    const int b = blockIdx.y, H = p_H;
    for (int i = threadIdx.x; i < NQ * W; i += 256) wl[i] = wxt[i];
    if (threadIdx.x <= NS) qo[threadIdx.x] = qoff[threadIdx.x];
    __syncthreads();
    const int Wd = NS * F + C;
    const int f4n = F >> 2, nquad = NS * f4n;
    const int quad = threadIdx.x & 127, y = blockIdx.x * 2 + (threadIdx.x >> 7);
    if (y < H && quad < nquad) {
        const T_* row = g + ((size_t)(b * H + y) * W) * Wd;
        float* dst = T + ((size_t)(b * H + y) * NQ) * F;
        const int s = quad / f4n, j = 4 * (quad - s * f4n), col = s * F + j;
        const int q0 = qo[s], nq = min(qo[s + 1] - q0, 8);
        float4 acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int x0 = 0; x0 < W; x0 += 16) {
            float v[16][4];
#pragma unroll
            for (int u = 0; u < 16; ++u) Op4<T_>::load(row + (size_t)min(x0 + u, W - 1) * Wd + col, v[u]);   // (unconditional: all 16 in flight)
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const bool in = x0 + u < W;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < nq) {
                        const float w = in ? wl[(q0 + q) * W + min(x0 + u, W - 1)] : 0.f;
                        acc[q].x = fmaf(w, v[u][0], acc[q].x); acc[q].y = fmaf(w, v[u][1], acc[q].y);
                        acc[q].z = fmaf(w, v[u][2], acc[q].z); acc[q].w = fmaf(w, v[u][3], acc[q].w);
                    }
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q < nq) *reinterpret_cast<float4*>(dst + (size_t)(q0 + q) * F + j) = acc[q];
    }
    const int c4n = C >> 2;
    if (dfeats)   // (null: l2i_psp_pool_bwd reads these columns of g itself)
    for (int rr = 0; rr < 2; ++rr) {
        const int yy = blockIdx.x * 2 + rr;
        if (yy >= H) break;
        const T_* row = g + ((size_t)(b * H + yy) * W) * Wd;
        for (int t = threadIdx.x; t < W * c4n; t += 256) {
            const int x = t / c4n, c = 4 * (t - x * c4n);
            float v[4];
            Op4<T_>::load(row + (size_t)x * Wd + NS * F + c, v);
            *reinterpret_cast<float4*>(dfeats + ((size_t)(b * H + yy) * W + x) * C + c) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

static bool psp_ok(int B, int HW, int C, int F, int NB) {
    if (B <= 0 || HW <= 0 || HW % PSP_CH || C <= 0 || C % 4 || C > 1024 || 256 % (C / 4) || NB <= 0 || NB > PSP_MAXNB) return false;
    return F >= 0 && F % 4 == 0;
}

extern "C" int l2i_psp_pool_fwd(const float* feats, const float* wx, const float* wy, const int* xq, float* pooled, float* rows,
                                int B, int H, int C, int NB, int NQ, void* stream) {
    if (!feats || !wx || !wy || !xq || !pooled || !rows || H <= 0 || H > PSP_MAXW || (H & 1) || NQ <= 0 || NQ > PSP_MAXQ ||
        !psp_ok(B, H * H, C, 0, NB))
        return L2I_ERR_ARG;
    hipLaunchKernelGGL(psp_pool_rows_kernel, dim3((H + PSP_PR - 1) / PSP_PR, B), dim3(32 * PSP_PR), 0, (hipStream_t)stream, feats, wx, rows, H, H, C, NQ);
    hipLaunchKernelGGL(psp_rows_reduce_kernel, dim3((NB * C + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, rows, wy, xq, pooled,
                       H, NQ, C, NB);
    return l2i_check_launch();
}

extern "C" int l2i_psp_pool_bwd(const float* dpooled, const int* aidx, const float* aw, int TA, const float* add, const float* add2,
                                const void* cat, int cat_w, int cat_off, int cat_dtype, float* dfeats, void* dfeats_op, int B, int HW, int C,
                                int NB, void* stream) {
    if (!dpooled || !aidx || !aw || !dfeats || !psp_ok(B, HW, C, 0, NB)) return L2I_ERR_ARG;
    if (cat && (cat_w < cat_off + C || cat_off % 4 || cat_w % 4 || (cat_dtype != 0 && cat_dtype != 1))) return L2I_ERR_ARG;
    const size_t lds = sizeof(float) * (size_t)NB * C + sizeof(int2) * (size_t)PSP_CH * TA;
    if (lds > 64 * 1024) return L2I_ERR_ARG;
    const dim3 grid(HW / PSP_CH, B);
#define PSP_PB(TA_, TT) hipLaunchKernelGGL((psp_pool_bwd_kernel<TA_, TT>), grid, dim3(256), lds, (hipStream_t)stream, dpooled, aidx, aw, add, add2, \
                                           (const TT*)cat, cat_w, cat_off, dfeats, (bf16_t*)dfeats_op, HW, C, NB)
    const bool b16 = cat && cat_dtype == 1;
    if (TA == 12) { if (b16) PSP_PB(12, bf16_t); else PSP_PB(12, float); }
    else if (TA == 16) { if (b16) PSP_PB(16, bf16_t); else PSP_PB(16, float); }
    else return L2I_ERR_ARG;
#undef PSP_PB
    return l2i_check_launch();
}

extern "C" int l2i_psp_expand_fwd(const float* feats, const float* y, const int* uidx, const float* uw, void* cat, int B, int HW,
                                  int C, int F, int NB, int n_stages, int dtype, void* stream) {
    if (!feats || !y || !uidx || !uw || !cat || n_stages <= 0 || n_stages > 8 || !psp_ok(B, HW, C, F, NB) || F == 0) return L2I_ERR_ARG;
    const size_t lds = sizeof(float) * (size_t)NB * F + sizeof(int2) * (size_t)PSP_CH * n_stages * 4;
    if (lds > 64 * 1024) return L2I_ERR_ARG;
    const dim3 grid(HW / PSP_CH, B);
    if (dtype == 1)
        hipLaunchKernelGGL(psp_expand_fwd_kernel<bf16_t>, grid, dim3(256), lds, (hipStream_t)stream, feats, y, uidx, uw,
                           (bf16_t*)cat, HW, C, F, NB, n_stages);
    else if (dtype == 0)
        hipLaunchKernelGGL(psp_expand_fwd_kernel<float>, grid, dim3(256), lds, (hipStream_t)stream, feats, y, uidx, uw,
                           (float*)cat, HW, C, F, NB, n_stages);
    else
        return L2I_ERR_ARG;
    return l2i_check_launch();
}

extern "C" int l2i_psp_expand_bwd(const void* g, const float* wxt, const float* wy, const int* xq, const int* qoff, float* dy,
                                  float* dfeats, float* rows, int B, int H, int C, int F, int NB, int NQ, int n_stages, int dtype,
                                  void* stream) {
    if (!g || !wxt || !wy || !xq || !qoff || !dy || !rows || H <= 0 || H > PSP_MAXW || NQ <= 0 || NQ > PSP_MAXQ ||
        n_stages <= 0 || n_stages > 8 || !psp_ok(B, H * H, C, F, NB) || F == 0)
        return L2I_ERR_ARG;
    if (n_stages * (F / 4) > 128) return L2I_ERR_ARG;   // (a thread per 4 prior columns, 128 per image row)
    const dim3 grid((H + 1) / 2, B);
    hipStream_t st = (hipStream_t)stream;
#define PSP_ROWS(TT, WM) hipLaunchKernelGGL((psp_expand_rows_kernel<TT, WM>), grid, dim3(256), 0, st, (const TT*)g, wxt, qoff, rows, dfeats, H, H, C, F, NQ, n_stages)
    if (dtype == 1) {
        if (H <= 32) PSP_ROWS(bf16_t, 32); else if (H <= 64) PSP_ROWS(bf16_t, 64); else PSP_ROWS(bf16_t, 128);
    } else if (dtype == 0) {
        if (H <= 32) PSP_ROWS(float, 32); else if (H <= 64) PSP_ROWS(float, 64); else PSP_ROWS(float, 128);
    } else {
        return L2I_ERR_ARG;
    }
#undef PSP_ROWS
    hipLaunchKernelGGL(psp_rows_reduce_kernel, dim3((NB * F + 255) / 256, B), dim3(256), 0, st, rows, wy, xq, dy, H, NQ, F, NB);
    return l2i_check_launch();
}
