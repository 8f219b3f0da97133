// Normalisation kernels (HBM-bound), NHWC f32 streams -> T operands.
//
// Replaces on the hot path:
//  * SpatialAdaptiveSynBatchNorm2d.forward (reference model/norm_module.py:163-186): affine-free
//    batch norm followed by the ISLA modulation
//        gamma(p,c) = 1 + sum_o m_o(p) W[b,o,c] / (sum_o m_o(p) + 1e-6),  beta likewise,
//    fused with the ReLU that always follows it (model/resnet_generator_app_v2.py:655-661).
//    The reference materialises two (b,o,C,h,w) products; here gamma/beta are formed per pixel
//    tile from an LDS copy of W,B and never touch HBM.
//  * SynchronizedBatchNorm2d / nn.BatchNorm2d / nn.InstanceNorm2d + ReLU
//    (model/sync_batchnorm/batchnorm.py:48-78, model/resnet_generator_app_v2.py:416-417,648-650,
//    735-746, model/mask_regression.py:66-80): same kernels, mode 1 (per-channel affine) / 2 (none),
//    statistics grouped per image for the instance norm.
//  * their backward passes (autograd of the above in the reference).
#include <stdlib.h>
#include "common.h"

#define NM_PT 64    // pixels per block
#define NM_CC 128   // channels per block
#define NM_MAXO 32

// ---------------------------------------------------------------- channel statistics
// x [rows][C] f32, rows grouped in consecutive runs of rows_per_group. sums/sqsums [G][C] += .
// Optionally also writes the operand-dtype copy of x (the cast a following MFMA kernel needs), so the
// bias-gradient reduction and the dY cast of a convolution's backward are one pass over dY.
// ~1024 blocks in total (4 per CU, 8 x 16-byte loads in flight per thread). Each block's per-channel partials end in
// one atomic per channel: into `sums`/`sqsums` directly when few blocks share an address (many groups), otherwise into
// the replicated workspace `ws` (common.h) that the last block folds into `sums`/`sqsums`.
template <typename T>
__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ x, long long rows, int C,
                                                            long long rows_per_group, int slabs_per_group,
                                                            float* __restrict__ sums, float* __restrict__ sqsums,
                                                            T* __restrict__ raw, float* __restrict__ part) {
    __shared__ float4 red[2][256];
    const int cols4 = C >> 2;
    const int group = blockIdx.x / slabs_per_group, slab = blockIdx.x % slabs_per_group;
    const long long slab_rows = (rows_per_group + slabs_per_group - 1) / slabs_per_group;
    const long long r0 = group * rows_per_group + slab * slab_rows;
    const long long r1 = min(group * rows_per_group + rows_per_group, r0 + slab_rows);
    // slabs_per_group > 1: this workgroup's sums are ROW (group, slab) of the partial matrix part[G * slabs][L] (stored; rows_fold adds the
    // rows in a fixed order: deterministic, round 6); one slab per group: the only contribution to its addresses, added in place
    const int L = sqsums ? 2 * C : C;
    float* const sdst = part ? part + (size_t)blockIdx.x * L : sums + (size_t)group * C;
    float* const qdst = !sqsums ? nullptr : part ? sdst + C : sqsums + (size_t)group * C;
    // column chunks of 256 float4 columns: blockIdx.y (the grouped projection's dY is 256 rows x 19 712 channels: with the chunks
    // walked in a loop the whole matrix was FOUR workgroups' work, 37 us for 30 MB)
    {
        const int cbase = blockIdx.y * 256;
        const int ncol = min(256, cols4 - cbase);
        const int TY = 256 / ncol;
        const int tx = threadIdx.x % ncol, ty = threadIdx.x / ncol;
        float4 s = make_float4(0, 0, 0, 0), q = make_float4(0, 0, 0, 0);
        if (ty < TY) {
            // eight rows per batch, the eight loads issued back to back (an `unroll 8` of the plain loop kept its per-row
            // bound checks between the loads and waited for each: 1.1 TB/s), then single rows
            const float* xc = x + 4 * (cbase + tx);
            auto acc1 = [&](const float4 v, long long r) {
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
                if (raw) {
                    const size_t off = (size_t)r * C + 4 * (cbase + tx);
                    if constexpr (sizeof(T) == 2) {
                        uint2 pk;
                        pk.x = f2bf2(v.x, v.y);
                        pk.y = f2bf2(v.z, v.w);
                        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(raw) + off) = pk;
                    } else {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(raw) + off) = v;
                    }
                }
            };
            long long r = r0 + ty;
            for (; r + 7LL * TY < r1; r += 8LL * TY) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(xc + (size_t)(r + (long long)u * TY) * C);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc1(v[u], r + (long long)u * TY);
            }
            for (; r < r1; r += TY) acc1(*reinterpret_cast<const float4*>(xc + (size_t)r * C), r);
        }
        __syncthreads();
        red[0][threadIdx.x] = s;
        red[1][threadIdx.x] = q;
        __syncthreads();
        if (threadIdx.x < ncol) {
            for (int j = 1; j < TY; ++j) {
                const float4 a = red[0][j * ncol + tx], b = red[1][j * ncol + tx];
                s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
                q.x += b.x; q.y += b.y; q.z += b.z; q.w += b.w;
            }
            float* so = sdst + 4 * (cbase + tx);
            if (part) {
                *reinterpret_cast<float4*>(so) = s;
                if (qdst) *reinterpret_cast<float4*>(qdst + 4 * (cbase + tx)) = q;
            } else {   // (atomic: a bias gradient that a second stream may add to at the same time; ONE add per address from this launch)
                atomicAdd(so + 0, s.x); atomicAdd(so + 1, s.y); atomicAdd(so + 2, s.z); atomicAdd(so + 3, s.w);
                if (qdst) {
                    float* qo = qdst + 4 * (cbase + tx);
                    atomicAdd(qo + 0, q.x); atomicAdd(qo + 1, q.y); atomicAdd(qo + 2, q.z); atomicAdd(qo + 3, q.w);
                }
            }
        }
    }
}

extern "C" int l2i_channel_stats(const float* x, long long rows, int C, long long rows_per_group, float* sums,
                                 float* sqsums, void* raw, int dtype, float* scratch, long long scratch_floats, void* stream) {
    if (!x || !sums || C % 4 || rows_per_group <= 0 || rows % rows_per_group) return L2I_ERR_ARG;
    const long long G = rows / rows_per_group;
    const int cchunks = (C / 4 + 255) / 256;
    const int L = sqsums ? 2 * C : C;
    long long slabs = (1024 + G * cchunks - 1) / (G * cchunks);
    const long long max_slabs = (rows_per_group + 15) / 16;   // (two 8-row batches per wave at least)
    if (slabs > max_slabs) slabs = max_slabs;
    if (slabs < 1) slabs = 1;
    // more than one slab per group: partial rows in the caller's scratch [G * slabs][L] + the fold's chunk rows behind them
    while (slabs > 1 && (!scratch || ((size_t)scratch & 15) || G * slabs * L + rows_fold_tmp_floats((int)slabs, L, (int)G) > scratch_floats)) slabs = scratch ? slabs / 2 : 1;
    if (G > 65535) return L2I_ERR_ARG;
    float* part = slabs > 1 ? scratch : nullptr;
    const dim3 grid((unsigned)(G * slabs), (unsigned)cchunks);
    if (dtype == 1)
        hipLaunchKernelGGL(channel_stats_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, x, rows, C, rows_per_group,
                           (int)slabs, sums, sqsums, (bf16_t*)raw, part);
    else if (dtype == 0)
        hipLaunchKernelGGL(channel_stats_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, rows, C, rows_per_group,
                           (int)slabs, sums, sqsums, (float*)raw, part);
    else
        return L2I_ERR_ARG;
    if (part) rows_fold(part, (int)slabs, L, (int)G, sums, sqsums, C, C, 2, part + G * slabs * L, (hipStream_t)stream);
    return l2i_check_launch();
}

// ---------------------------------------------------------------- modulated norm, forward
struct NormArgs {
    const float* x;        // [B][HW][C]
    const float* dy;       // bwd: gradient wrt the (post-ReLU) output
    float* dy_keep;        // bwd, O > 8 only: scratch copy of dy (dxhat may alias dy)
    const float* sums;     // [G][C]
    const float* sqsums;   // [G][C]
    const float* mask;     // [B][O][HW] or null
    const float* wproj;    // mode 0: [B][O][C] via strides; mode 1: [C]
    const float* bproj;
    void* out_op;          // T [B][HW][C] or null
    float* out_f32;        // fwd: optional f32 copy; bwd: dxhat
    float* s1; float* s2;  // bwd: [G][C]
    float* dwproj; float* dbproj; float* dmask;
    float* ws;             // bwd: replicated workspace for the per-channel totals (common.h) or null
    float* part;           // bwd, <= 8 objects: per-workgroup partial dW / dB rows, summed by norm_a8_finish_kernel (no atomics) -- or null
    float* run_mean; float* run_var; float momentum;   // fwd, train mode: running statistics updated in the same launch (or null)
    long long pstride_b, pstride_o;
    int B, HW, C, O, mode, relu, stat_stride;
    float count, eps;
    int dmask_store;       // bwd, <= 8 objects, C within one channel chunk: dmask = (plain stores) instead of += (atomics)
    // round 6, no float atomics when the caller lends scratch (l2i_norm_mod_bwd_a):
    float* spart;          // bwd: per-channel totals (s1, s2; affine mode also dW, dB) of a workgroup = row (image x segment) of [B * nseg][nval * C],
                           // channel chunks side by side, STORED; rows_fold adds the rows in order behind the launch. null: atomics / the workspace
    float* dmpart;         // bwd, several channel chunks: dmask contribution of chunk tc = row tc of [tiles_c][B * O * HW], STORED; rows_fold adds the chunks
};

__device__ __forceinline__ float4 f4mad(float a, float4 b, float4 c) {
    return make_float4(fmaf(a, b.x, c.x), fmaf(a, b.y, c.y), fmaf(a, b.z, c.z), fmaf(a, b.w, c.w));
}

// LDS carve: Wl[O][CC] Bl[O][CC] mn[O][PT] sinv[PT]
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void norm_mod_kernel(NormArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int O = p.mode == 0 ? p.O : 0;
    float* Wl = reinterpret_cast<float*>(smem);
    float* Bl = Wl + O * NM_CC;
    float* mn = Bl + O * NM_CC;
    float* sinv = mn + O * NM_PT;

    const int tiles_p = (p.HW + NM_PT - 1) / NM_PT;
    const int tiles_c = (p.C + NM_CC - 1) / NM_CC;
    int bid = blockIdx.x;
    const int tc = bid % tiles_c; bid /= tiles_c;
    const int tp = bid % tiles_p;
    const int b = bid / tiles_p;
    const int c0 = tc * NM_CC, p0 = tp * NM_PT;
    const int cc = min(NM_CC, p.C - c0);
    const int tid = threadIdx.x;

    if (O > 0) {
        for (int i = tid; i < O * NM_CC; i += 256) {
            const int o = i / NM_CC, c = i - o * NM_CC;
            float w = 0.f, bb = 0.f;
            if (c < cc) {
                const size_t off = (size_t)b * p.pstride_b + (size_t)o * p.pstride_o + c0 + c;
                w = p.wproj[off];
                bb = p.bproj[off];
            }
            Wl[i] = w;
            Bl[i] = bb;
        }
        if (tid < NM_PT) {
            const int px = p0 + tid;
            float S = 1e-6f;
            if (px < p.HW)
                for (int o = 0; o < O; ++o) S += p.mask[((size_t)b * O + o) * p.HW + px];
            const float inv = 1.f / S;
            sinv[tid] = inv;
            for (int o = 0; o < O; ++o)
                mn[o * NM_PT + tid] = px < p.HW ? p.mask[((size_t)b * O + o) * p.HW + px] * inv : 0.f;
        }
        __syncthreads();
    }

    const int cv = tid & 31, prow = tid >> 5;  // 32 float4 columns x 8 pixel rows
    const int c = c0 + 4 * cv;
    const bool con = 4 * cv < cc;
    float4 mean = make_float4(0, 0, 0, 0), istd = make_float4(1, 1, 1, 1);
    float4 aw = make_float4(1, 1, 1, 1), ab = make_float4(0, 0, 0, 0);
    if (con) {
        const size_t so = (size_t)b * p.stat_stride + c;
        const float4 s = *reinterpret_cast<const float4*>(p.sums + so);
        const float4 q = *reinterpret_cast<const float4*>(p.sqsums + so);
        const float ic = 1.f / p.count;
        mean = make_float4(s.x * ic, s.y * ic, s.z * ic, s.w * ic);
        istd.x = rsqrtf(fmaxf(q.x * ic - mean.x * mean.x, 0.f) + p.eps);
        istd.y = rsqrtf(fmaxf(q.y * ic - mean.y * mean.y, 0.f) + p.eps);
        istd.z = rsqrtf(fmaxf(q.z * ic - mean.z * mean.z, 0.f) + p.eps);
        istd.w = rsqrtf(fmaxf(q.w * ic - mean.w * mean.w, 0.f) + p.eps);
        if (p.mode == 1) {
            aw = *reinterpret_cast<const float4*>(p.wproj + c);
            ab = *reinterpret_cast<const float4*>(p.bproj + c);
        }
        if (!BWD && p.run_mean && b == 0 && tp == 0 && prow == 0) {
            // nn.BatchNorm2d train-mode side effect (momentum update with the UNBIASED batch variance), once per channel
            const float ub = p.count / fmaxf(p.count - 1.f, 1.f), mo = p.momentum;
            float4 rm = *reinterpret_cast<float4*>(p.run_mean + c), rv = *reinterpret_cast<float4*>(p.run_var + c);
            rm.x = (1.f - mo) * rm.x + mo * mean.x; rm.y = (1.f - mo) * rm.y + mo * mean.y;
            rm.z = (1.f - mo) * rm.z + mo * mean.z; rm.w = (1.f - mo) * rm.w + mo * mean.w;
            rv.x = (1.f - mo) * rv.x + mo * ub * fmaxf(q.x * ic - mean.x * mean.x, 0.f);
            rv.y = (1.f - mo) * rv.y + mo * ub * fmaxf(q.y * ic - mean.y * mean.y, 0.f);
            rv.z = (1.f - mo) * rv.z + mo * ub * fmaxf(q.z * ic - mean.z * mean.z, 0.f);
            rv.w = (1.f - mo) * rv.w + mo * ub * fmaxf(q.w * ic - mean.w * mean.w, 0.f);
            *reinterpret_cast<float4*>(p.run_mean + c) = rm;
            *reinterpret_cast<float4*>(p.run_var + c) = rv;
        }
    }
    float4 acc_s1 = make_float4(0, 0, 0, 0), acc_s2 = make_float4(0, 0, 0, 0);
    float4 acc_dw = make_float4(0, 0, 0, 0), acc_db = make_float4(0, 0, 0, 0);  // mode 1 (affine) grads
    T* OutOp = reinterpret_cast<T*>(p.out_op);

    float4 gk[NM_PT / 8], gxk[NM_PT / 8];  // bwd: g and g*xhat kept for the per-object pass
#pragma unroll
    for (int it = 0; it < NM_PT / 8; ++it) {
        const int pl = prow + 8 * it, px = p0 + pl;
        const bool on = con && px < p.HW;
        const size_t off = ((size_t)b * p.HW + px) * p.C + c;
        float4 xh = make_float4(0, 0, 0, 0), ga = aw, be = ab, y = make_float4(0, 0, 0, 0);
        if (on) {
            const float4 xv = *reinterpret_cast<const float4*>(p.x + off);
            xh = make_float4((xv.x - mean.x) * istd.x, (xv.y - mean.y) * istd.y, (xv.z - mean.z) * istd.z,
                             (xv.w - mean.w) * istd.w);
            if (p.mode == 0) {
                ga = make_float4(1, 1, 1, 1);
                be = make_float4(0, 0, 0, 0);
                for (int o = 0; o < O; ++o) {
                    const float m = mn[o * NM_PT + pl];
                    ga = f4mad(m, *reinterpret_cast<const float4*>(Wl + o * NM_CC + 4 * cv), ga);
                    be = f4mad(m, *reinterpret_cast<const float4*>(Bl + o * NM_CC + 4 * cv), be);
                }
            }
            y = make_float4(fmaf(ga.x, xh.x, be.x), fmaf(ga.y, xh.y, be.y), fmaf(ga.z, xh.z, be.z),
                            fmaf(ga.w, xh.w, be.w));
        }
        float4 g = make_float4(0, 0, 0, 0), gx = g;
        if (!BWD) {
            if (on) {
                if (p.relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
                if (p.out_f32) *reinterpret_cast<float4*>(p.out_f32 + off) = y;
                if (OutOp) {
                    if constexpr (sizeof(T) == 2) {
                        uint2 pk;
                        pk.x = f2bf2(y.x, y.y);
                        pk.y = f2bf2(y.z, y.w);
                        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(OutOp) + off) = pk;
                    } else {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(OutOp) + off) = y;
                    }
                }
            }
        } else {
            if (on) {
                const float4 d = *reinterpret_cast<const float4*>(p.dy + off);
                g.x = (!p.relu || y.x > 0.f) ? d.x : 0.f;
                g.y = (!p.relu || y.y > 0.f) ? d.y : 0.f;
                g.z = (!p.relu || y.z > 0.f) ? d.z : 0.f;
                g.w = (!p.relu || y.w > 0.f) ? d.w : 0.f;
                gx = make_float4(g.x * xh.x, g.y * xh.y, g.z * xh.z, g.w * xh.w);
                const float4 dxh = make_float4(g.x * ga.x, g.y * ga.y, g.z * ga.z, g.w * ga.w);
                *reinterpret_cast<float4*>(p.out_f32 + off) = dxh;
                acc_s1.x += dxh.x; acc_s1.y += dxh.y; acc_s1.z += dxh.z; acc_s1.w += dxh.w;
                acc_s2.x += dxh.x * xh.x; acc_s2.y += dxh.y * xh.y; acc_s2.z += dxh.z * xh.z; acc_s2.w += dxh.w * xh.w;
                if (p.mode == 1) {
                    acc_dw.x += gx.x; acc_dw.y += gx.y; acc_dw.z += gx.z; acc_dw.w += gx.w;
                    acc_db.x += g.x; acc_db.y += g.y; acc_db.z += g.z; acc_db.w += g.w;
                }
            }
            if (p.mode == 0 && p.dmask) {
                // o-independent part of dmask: -sum_c g (xh (gamma-1) + beta) / S ; zero for inactive lanes.
                // The shuffle is executed by every lane (block-uniform condition) so no lane reads a parked one.
                float t0 = gx.x * (ga.x - 1.f) + g.x * be.x + gx.y * (ga.y - 1.f) + g.y * be.y +
                           gx.z * (ga.z - 1.f) + g.z * be.z + gx.w * (ga.w - 1.f) + g.w * be.w;
#pragma unroll
                for (int s = 16; s > 0; s >>= 1) t0 += __shfl_xor(t0, s, 64);
                if (cv == 0 && px < p.HW) {
                    const float si = sinv[pl];
                    for (int o = 0; o < O; ++o) atomicAdd(p.dmask + ((size_t)b * O + o) * p.HW + px, -t0 * si);
                }
            }
        }
        gk[it] = g;
        gxk[it] = gx;
    }
    if (!BWD) return;

    // per-object gradients: dW[b,o,c] += sum_p gx * mn_o(p);  dB += sum_p g * mn_o(p);
    // dmask_o(p) += sum_c (gx W_oc + g B_oc) / S(p)
    if (p.mode == 0) {
        for (int o = 0; o < O; ++o) {
            const float4 wv = *reinterpret_cast<const float4*>(Wl + o * NM_CC + 4 * cv);
            const float4 bv = *reinterpret_cast<const float4*>(Bl + o * NM_CC + 4 * cv);
            float4 dw = make_float4(0, 0, 0, 0), db = dw;
#pragma unroll
            for (int it = 0; it < NM_PT / 8; ++it) {
                const int pl = prow + 8 * it, px = p0 + pl;
                const float m = mn[o * NM_PT + pl];
                dw = f4mad(m, gxk[it], dw);
                db = f4mad(m, gk[it], db);
                if (p.dmask) {
                    float a = gxk[it].x * wv.x + gxk[it].y * wv.y + gxk[it].z * wv.z + gxk[it].w * wv.w +
                              gk[it].x * bv.x + gk[it].y * bv.y + gk[it].z * bv.z + gk[it].w * bv.w;
#pragma unroll
                    for (int s = 16; s > 0; s >>= 1) a += __shfl_xor(a, s, 64);
                    if (cv == 0 && px < p.HW) atomicAdd(p.dmask + ((size_t)b * O + o) * p.HW + px, a * sinv[pl]);
                }
            }
            // reduce dw/db over the 8 pixel rows: lanes tid and tid^32 share cv within a wave; then across 4 waves
            dw.x += __shfl_xor(dw.x, 32, 64); dw.y += __shfl_xor(dw.y, 32, 64);
            dw.z += __shfl_xor(dw.z, 32, 64); dw.w += __shfl_xor(dw.w, 32, 64);
            db.x += __shfl_xor(db.x, 32, 64); db.y += __shfl_xor(db.y, 32, 64);
            db.z += __shfl_xor(db.z, 32, 64); db.w += __shfl_xor(db.w, 32, 64);
            if ((tid & 32) == 0 && con) {
                const size_t off = (size_t)b * p.pstride_b + (size_t)o * p.pstride_o + c;
                atomicAdd(p.dwproj + off + 0, dw.x); atomicAdd(p.dwproj + off + 1, dw.y);
                atomicAdd(p.dwproj + off + 2, dw.z); atomicAdd(p.dwproj + off + 3, dw.w);
                atomicAdd(p.dbproj + off + 0, db.x); atomicAdd(p.dbproj + off + 1, db.y);
                atomicAdd(p.dbproj + off + 2, db.z); atomicAdd(p.dbproj + off + 3, db.w);
            }
        }
    }
    // s1/s2 (+ affine grads)
    acc_s1.x += __shfl_xor(acc_s1.x, 32, 64); acc_s1.y += __shfl_xor(acc_s1.y, 32, 64);
    acc_s1.z += __shfl_xor(acc_s1.z, 32, 64); acc_s1.w += __shfl_xor(acc_s1.w, 32, 64);
    acc_s2.x += __shfl_xor(acc_s2.x, 32, 64); acc_s2.y += __shfl_xor(acc_s2.y, 32, 64);
    acc_s2.z += __shfl_xor(acc_s2.z, 32, 64); acc_s2.w += __shfl_xor(acc_s2.w, 32, 64);
    if (p.mode == 1) {
        acc_dw.x += __shfl_xor(acc_dw.x, 32, 64); acc_dw.y += __shfl_xor(acc_dw.y, 32, 64);
        acc_dw.z += __shfl_xor(acc_dw.z, 32, 64); acc_dw.w += __shfl_xor(acc_dw.w, 32, 64);
        acc_db.x += __shfl_xor(acc_db.x, 32, 64); acc_db.y += __shfl_xor(acc_db.y, 32, 64);
        acc_db.z += __shfl_xor(acc_db.z, 32, 64); acc_db.w += __shfl_xor(acc_db.w, 32, 64);
    }
    if ((tid & 32) == 0 && con) {
        const size_t so = (size_t)b * p.stat_stride + c;
        atomicAdd(p.s1 + so + 0, acc_s1.x); atomicAdd(p.s1 + so + 1, acc_s1.y);
        atomicAdd(p.s1 + so + 2, acc_s1.z); atomicAdd(p.s1 + so + 3, acc_s1.w);
        atomicAdd(p.s2 + so + 0, acc_s2.x); atomicAdd(p.s2 + so + 1, acc_s2.y);
        atomicAdd(p.s2 + so + 2, acc_s2.z); atomicAdd(p.s2 + so + 3, acc_s2.w);
        if (p.mode == 1) {
            atomicAdd(p.dwproj + c + 0, acc_dw.x); atomicAdd(p.dwproj + c + 1, acc_dw.y);
            atomicAdd(p.dwproj + c + 2, acc_dw.z); atomicAdd(p.dwproj + c + 3, acc_dw.w);
            atomicAdd(p.dbproj + c + 0, acc_db.x); atomicAdd(p.dbproj + c + 1, acc_db.y);
            atomicAdd(p.dbproj + c + 2, acc_db.z); atomicAdd(p.dbproj + c + 3, acc_db.w);
        }
    }
}

// ---------------------------------------------------------------- modulated norm, backward pass A
// One block = (image b, 128-channel chunk, a run of 32-pixel sub-tiles), two phases per sub-tile with different
// thread roles so that neither needs many registers (the block keeps 3 workgroups per CU resident; a single-role
// version needed 255 VGPRs = one wave per SIMD and ran at a tenth of the speed):
//   phase 1, thread = (pixel row, 4 channels): x-hat, gamma/beta, the ReLU gate, g = gated dy, gx = g * x-hat;
//            writes dxhat = g * gamma, accumulates the per-channel sums (s1, s2, affine grads) in registers and parks
//            g, gx in LDS;
//   phase 2, thread = (object, 4 channels), ISLA only: dW[b,o,c] += sum_p mn_o(p) gx(p,c), dB likewise with g --
//            each thread owns its accumulators for the whole run, no reduction -- and
//            part[o,p] = sum_c gx W_o + g B_o reduced over the 32 channel lanes (4 DPP steps + one permute);
//   then dmask[b,o,p] = (part[o,p] - sum_o' mn_o'(p) part[o',p]) / S(p)   (the second term is the "-1/S^2" path of
//   the mask normalisation, since gamma - 1 = sum_o mn_o W_o and beta = sum_o mn_o B_o).
// More than 8 objects (VG: 31) repeat phase 2 per chunk of 8 on the same parked tile.
#define NB_PX 32
#define NB_OC 8
#define NB_MAXCH 4   // object chunks (NM_MAXO / NB_OC)

__device__ __forceinline__ float sum32(float v) {   // sum over the 32 lanes of a half-wave, result in every lane
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
    v += __shfl_xor(v, 16, 64);
    return v;
}

__global__ __launch_bounds__(256) void norm_bwd_a_kernel(NormArgs p, int nseg, int seg_pixels) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int O = p.mode == 0 ? p.O : 0;
    float* Wl = reinterpret_cast<float*>(smem);
    float* Bl = Wl + O * NM_CC;
    float* mn = Bl + O * NM_CC;          // [O][NB_PX]
    float* sinv = mn + O * NB_PX;        // [NB_PX]
    float* partl = sinv + NB_PX;         // [O][NB_PX]
    float* gl = partl + O * NB_PX;       // [NB_PX][NM_CC]  (mode 0) -- also the end-of-run reduction buffer
    float* gxl = gl + NB_PX * NM_CC;     // [NB_PX][NM_CC]
    float4* red = reinterpret_cast<float4*>(gl);  // [4 values][8 pixel rows][32 cv]

    const int tiles_c = (p.C + NM_CC - 1) / NM_CC;
    int bid = blockIdx.x;
    const int tc = bid % tiles_c; bid /= tiles_c;
    const int seg = bid % nseg;
    const int b = bid / nseg;
    const int c0 = tc * NM_CC;
    const int cc = min(NM_CC, p.C - c0);
    const int tid = threadIdx.x;
    const int px_begin = seg * seg_pixels, px_end = min(p.HW, px_begin + seg_pixels);

    for (int i = tid; i < O * NM_CC; i += 256) {
        const int o = i / NM_CC, c = i - o * NM_CC;
        float w = 0.f, bb = 0.f;
        if (c < cc) {
            const size_t off = (size_t)b * p.pstride_b + (size_t)o * p.pstride_o + c0 + c;
            w = p.wproj[off];
            bb = p.bproj[off];
        }
        Wl[i] = w;
        Bl[i] = bb;
    }

    const int cv = tid & 31, prow = tid >> 5;
    const int c = c0 + 4 * cv;
    const bool con = 4 * cv < cc;
    float4 mean = make_float4(0, 0, 0, 0), istd = make_float4(1, 1, 1, 1);
    float4 aw = make_float4(1, 1, 1, 1), ab = make_float4(0, 0, 0, 0);
    if (con) {
        const size_t so = (size_t)b * p.stat_stride + c;
        const float4 s = *reinterpret_cast<const float4*>(p.sums + so);
        const float4 q = *reinterpret_cast<const float4*>(p.sqsums + so);
        const float ic = 1.f / p.count;
        mean = make_float4(s.x * ic, s.y * ic, s.z * ic, s.w * ic);
        istd.x = rsqrtf(fmaxf(q.x * ic - mean.x * mean.x, 0.f) + p.eps);
        istd.y = rsqrtf(fmaxf(q.y * ic - mean.y * mean.y, 0.f) + p.eps);
        istd.z = rsqrtf(fmaxf(q.z * ic - mean.z * mean.z, 0.f) + p.eps);
        istd.w = rsqrtf(fmaxf(q.w * ic - mean.w * mean.w, 0.f) + p.eps);
        if (p.mode == 1) {
            aw = *reinterpret_cast<const float4*>(p.wproj + c);
            ab = *reinterpret_cast<const float4*>(p.bproj + c);
        }
    }
    const float4 z4 = make_float4(0, 0, 0, 0);
    float4 acc_s1 = z4, acc_s2 = z4, acc_aw = z4, acc_ab = z4;
    // phase-2 role: object (tid >> 5) of each chunk, channels 4 * cv
    const int o2 = tid >> 5;
    float4 adw[NB_MAXCH], adb[NB_MAXCH];
#pragma unroll
    for (int k = 0; k < NB_MAXCH; ++k) { adw[k] = z4; adb[k] = z4; }

    for (int p0 = px_begin; p0 < px_end; p0 += NB_PX) {
        __syncthreads();   // previous sub-tile's LDS fully consumed (and Wl/Bl written, first time)
        if (O > 0 && tid < NB_PX) {
            const int px = p0 + tid;
            float S = 1e-6f;
            if (px < px_end)
                for (int o = 0; o < O; ++o) S += p.mask[((size_t)b * O + o) * p.HW + px];
            const float inv = 1.f / S;
            sinv[tid] = inv;
            for (int o = 0; o < O; ++o)
                mn[o * NB_PX + tid] = px < px_end ? p.mask[((size_t)b * O + o) * p.HW + px] * inv : 0.f;
        }
        // the x / dy loads of phase 1 do not depend on the mask: issue them before the barrier
        float4 xv[4], dv[4];
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
            const int px = p0 + prow + 8 * pi;
            xv[pi] = z4; dv[pi] = z4;
            if (con && px < px_end) {
                const size_t off = ((size_t)b * p.HW + px) * p.C + c;
                xv[pi] = *reinterpret_cast<const float4*>(p.x + off);
                dv[pi] = *reinterpret_cast<const float4*>(p.dy + off);
            }
        }
        __syncthreads();
        // ---- phase 1
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
            const int pl = prow + 8 * pi, px = p0 + pl;
            const bool on = con && px < px_end;
            float4 gg = z4, gxv = z4;
            if (on) {
                const float4 xh = make_float4((xv[pi].x - mean.x) * istd.x, (xv[pi].y - mean.y) * istd.y,
                                              (xv[pi].z - mean.z) * istd.z, (xv[pi].w - mean.w) * istd.w);
                float4 ga = aw, be = ab;
                if (p.mode == 0) {
                    ga = make_float4(1, 1, 1, 1);
                    be = z4;
                    for (int o = 0; o < O; ++o) {
                        const float m = mn[o * NB_PX + pl];
                        ga = f4mad(m, *reinterpret_cast<const float4*>(Wl + o * NM_CC + 4 * cv), ga);
                        be = f4mad(m, *reinterpret_cast<const float4*>(Bl + o * NM_CC + 4 * cv), be);
                    }
                }
                const float4 d = dv[pi];
                gg.x = (!p.relu || fmaf(ga.x, xh.x, be.x) > 0.f) ? d.x : 0.f;
                gg.y = (!p.relu || fmaf(ga.y, xh.y, be.y) > 0.f) ? d.y : 0.f;
                gg.z = (!p.relu || fmaf(ga.z, xh.z, be.z) > 0.f) ? d.z : 0.f;
                gg.w = (!p.relu || fmaf(ga.w, xh.w, be.w) > 0.f) ? d.w : 0.f;
                gxv = make_float4(gg.x * xh.x, gg.y * xh.y, gg.z * xh.z, gg.w * xh.w);
                const float4 dxh = make_float4(gg.x * ga.x, gg.y * ga.y, gg.z * ga.z, gg.w * ga.w);
                *reinterpret_cast<float4*>(p.out_f32 + ((size_t)b * p.HW + px) * p.C + c) = dxh;
                acc_s1.x += dxh.x; acc_s1.y += dxh.y; acc_s1.z += dxh.z; acc_s1.w += dxh.w;
                acc_s2.x += dxh.x * xh.x; acc_s2.y += dxh.y * xh.y; acc_s2.z += dxh.z * xh.z; acc_s2.w += dxh.w * xh.w;
                if (p.mode == 1) {
                    acc_aw.x += gxv.x; acc_aw.y += gxv.y; acc_aw.z += gxv.z; acc_aw.w += gxv.w;
                    acc_ab.x += gg.x; acc_ab.y += gg.y; acc_ab.z += gg.z; acc_ab.w += gg.w;
                }
            }
            if (p.mode == 0) {
                *reinterpret_cast<float4*>(gl + pl * NM_CC + 4 * cv) = gg;
                *reinterpret_cast<float4*>(gxl + pl * NM_CC + 4 * cv) = gxv;
            }
        }
        if (p.mode != 0) continue;
        __syncthreads();
        // ---- phase 2
#pragma unroll
        for (int k = 0; k < NB_MAXCH; ++k) {
            const int o = k * NB_OC + o2;
            if (k * NB_OC >= O) break;
            const bool oon = o < O;
            const int oc = oon ? o : 0;
            const float4 wv = *reinterpret_cast<const float4*>(Wl + oc * NM_CC + 4 * cv);
            const float4 bv = *reinterpret_cast<const float4*>(Bl + oc * NM_CC + 4 * cv);
#pragma unroll 4
            for (int pl = 0; pl < NB_PX; ++pl) {
                const float4 gxv = *reinterpret_cast<const float4*>(gxl + pl * NM_CC + 4 * cv);
                const float4 gv = *reinterpret_cast<const float4*>(gl + pl * NM_CC + 4 * cv);
                const float m = oon ? mn[oc * NB_PX + pl] : 0.f;
                adw[k] = f4mad(m, gxv, adw[k]);
                adb[k] = f4mad(m, gv, adb[k]);
                if (p.dmask) {
                    float part = gxv.x * wv.x + gxv.y * wv.y + gxv.z * wv.z + gxv.w * wv.w +
                                 gv.x * bv.x + gv.y * bv.y + gv.z * bv.z + gv.w * bv.w;
                    part = sum32(part);
                    if (cv == 0 && oon) partl[o * NB_PX + pl] = part;
                }
            }
        }
        if (p.dmask) {
            __syncthreads();
            for (int i = tid; i < O * NB_PX; i += 256) {
                const int o = i / NB_PX, pl = i - o * NB_PX;
                const int px = p0 + pl;
                if (px >= px_end) continue;
                float t0 = 0.f;
                for (int oo = 0; oo < O; ++oo) t0 = fmaf(mn[oo * NB_PX + pl], partl[oo * NB_PX + pl], t0);
                const float v = (partl[i] - t0) * sinv[pl];
                if (p.dmpart) p.dmpart[((size_t)tc * p.B * O + (size_t)b * O + o) * p.HW + px] = v;   // this chunk's row: rows_fold adds the chunks
                else atomicAdd(p.dmask + ((size_t)b * O + o) * p.HW + px, v);
            }
        }
    }

    // ---- per-object gradients: each thread owns (object, 4 channels). With p.part (the launcher lends scratch: round 6) the workgroup's values go
    // to ITS row (image, segment) of [B * nseg][2][O][C] with plain stores and norm_a_finish_kernel adds an image's segments in order; else one
    // atomic per value per block
    if (p.mode == 0 && con && p.part) {
        float* mine = p.part + (size_t)(b * nseg + seg) * (2 * (size_t)O * p.C);
#pragma unroll
        for (int k = 0; k < NB_MAXCH; ++k) {
            const int o = k * NB_OC + o2;
            if (o >= O) continue;
            *reinterpret_cast<float4*>(mine + (size_t)o * p.C + c) = adw[k];
            *reinterpret_cast<float4*>(mine + ((size_t)O + o) * p.C + c) = adb[k];
        }
    } else if (p.mode == 0 && con) {
#pragma unroll
        for (int k = 0; k < NB_MAXCH; ++k) {
            const int o = k * NB_OC + o2;
            if (o >= O) continue;
            const size_t off = (size_t)b * p.pstride_b + (size_t)o * p.pstride_o + c;
            atomicAdd(p.dwproj + off + 0, adw[k].x); atomicAdd(p.dwproj + off + 1, adw[k].y);
            atomicAdd(p.dwproj + off + 2, adw[k].z); atomicAdd(p.dwproj + off + 3, adw[k].w);
            atomicAdd(p.dbproj + off + 0, adb[k].x); atomicAdd(p.dbproj + off + 1, adb[k].y);
            atomicAdd(p.dbproj + off + 2, adb[k].z); atomicAdd(p.dbproj + off + 3, adb[k].w);
        }
    }
    // ---- per-channel sums: reduce the 8 pixel rows through LDS, then one atomic per value per block
    __syncthreads();
    const int nval = p.mode == 1 ? 4 : 2;
    red[(0 * 8 + prow) * 32 + cv] = acc_s1;
    red[(1 * 8 + prow) * 32 + cv] = acc_s2;
    if (p.mode == 1) {
        red[(2 * 8 + prow) * 32 + cv] = acc_aw;
        red[(3 * 8 + prow) * 32 + cv] = acc_ab;
    }
    __syncthreads();
    for (int i = tid; i < nval * 32; i += 256) {
        const int k = i >> 5, lc = i & 31;
        if (4 * lc >= cc) continue;
        float4 a = red[(k * 8) * 32 + lc];
        for (int w = 1; w < 8; ++w) {
            const float4 t = red[(k * 8 + w) * 32 + lc];
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        const int cch = c0 + 4 * lc;
        if (p.spart) {   // this workgroup's row of the partial matrix (one writer per value)
            *reinterpret_cast<float4*>(p.spart + (size_t)(blockIdx.x / tiles_c) * (nval * p.C) + k * p.C + cch) = a;
            continue;
        }
        float* dst;
        if (p.ws) {
            dst = ws_replica(p.ws, (blockIdx.x / tiles_c) % L2I_WS_R, nval * p.C) + k * p.C + cch;
        } else {
            const size_t so = (size_t)b * p.stat_stride + cch;
            dst = k == 0 ? p.s1 + so : k == 1 ? p.s2 + so : k == 2 ? p.dwproj + cch : p.dbproj + cch;
        }
        atomicAdd(dst + 0, a.x); atomicAdd(dst + 1, a.y); atomicAdd(dst + 2, a.z); atomicAdd(dst + 3, a.w);
    }
}

// ---------------------------------------------------------------- ISLA backward pass A, up to 8 objects: registers, not LDS
// Measured on the kernel above (tools/perf/norm_micro.py, 32 x 128^2 x 64 channels, 8 objects): 564 us against 153 us for
// the same tensor without the modulation, i.e. 4x the HBM time, spent on LDS traffic -- phase 1 re-reads W, B of every
// object for every pixel, phase 2 parks g, g*xhat in LDS and reads them back once per object -- and on half-idle lanes
// when C = 64 (a 128-channel chunk is hard-wired). Here a thread owns (pixel row, 4 channels) for both jobs:
//   * W, B of the 8 objects for its 4 channels live in registers (loaded once per workgroup run);
//   * gamma / beta, the ReLU gate, dxhat and s1, s2 as before;
//   * dW[o] += mn_o g xhat, dB[o] += mn_o g for all 8 objects from its OWN g, g*xhat registers (no parking), reduced over
//     the pixel rows of the workgroup through LDS once at the end of the run;
//   * dmask: part[o,p] = sum_c (g xhat W_o + g B_o) reduced over the channel lanes with DPP, combined as above.
// CV = float4 lanes across channels: 32 (128-channel chunks) or 16 (C = 64: every lane busy).
template <int CV>
__device__ __forceinline__ float sum_cv(float v) {   // sum over the CV channel lanes that share a pixel row
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
    if (CV == 32) v += __shfl_xor(v, 16, 64);
    return v;
}

template <int CV>
__global__ __launch_bounds__(256, 2) void norm_bwd_a8_kernel(NormArgs p, int nseg, int seg_pixels) {
    constexpr int CC = 4 * CV, PR = 256 / CV, NPI = NB_PX / PR;   // channels per chunk, pixel rows per pass, passes per sub-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* mn4 = reinterpret_cast<float4*>(smem);         // [NB_PX][2]: mn of objects 0-3 / 4-7 of a pixel
    float* sinv = reinterpret_cast<float*>(mn4 + 2 * NB_PX);   // [NB_PX]
    float* partl = sinv + NB_PX;                           // [8][NB_PX]
    float4* red = reinterpret_cast<float4*>(partl + 8 * NB_PX);   // [PR][8][CV] end-of-run reductions (32 KB)

    const int tiles_c = (p.C + CC - 1) / CC;
    int bid = blockIdx.x;
    const int tc = bid % tiles_c; bid /= tiles_c;
    const int seg = bid % nseg;
    const int b = bid / nseg;
    const int c0 = tc * CC;
    const int cc = min(CC, p.C - c0);
    const int tid = threadIdx.x;
    const int px_begin = seg * seg_pixels, px_end = min(p.HW, px_begin + seg_pixels);
    const int O = p.O;
    const int cv = tid % CV, prow = tid / CV;
    const int c = c0 + 4 * cv;
    const bool con = 4 * cv < cc;
    const float4 z4 = make_float4(0, 0, 0, 0);

    float4 Wr[8], Br[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        Wr[o] = z4; Br[o] = z4;
        if (con && o < O) {
            const size_t off = (size_t)b * p.pstride_b + (size_t)o * p.pstride_o + c;
            Wr[o] = *reinterpret_cast<const float4*>(p.wproj + off);
            Br[o] = *reinterpret_cast<const float4*>(p.bproj + off);
        }
    }
    float4 mean = z4, istd = make_float4(1, 1, 1, 1);
    if (con) {
        const size_t so = (size_t)b * p.stat_stride + c;
        const float4 s = *reinterpret_cast<const float4*>(p.sums + so);
        const float4 q = *reinterpret_cast<const float4*>(p.sqsums + so);
        const float ic = 1.f / p.count;
        mean = make_float4(s.x * ic, s.y * ic, s.z * ic, s.w * ic);
        istd.x = rsqrtf(fmaxf(q.x * ic - mean.x * mean.x, 0.f) + p.eps);
        istd.y = rsqrtf(fmaxf(q.y * ic - mean.y * mean.y, 0.f) + p.eps);
        istd.z = rsqrtf(fmaxf(q.z * ic - mean.z * mean.z, 0.f) + p.eps);
        istd.w = rsqrtf(fmaxf(q.w * ic - mean.w * mean.w, 0.f) + p.eps);
    }
    float4 acc_s1 = z4, acc_s2 = z4, adw[8], adb[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) { adw[o] = z4; adb[o] = z4; }

    // Rolling prefetch: a thread's x / dy rows and the mask values of the NEXT sub-tile are requested as soon as the registers
    // holding the current ones are free (x, dy: right after the row has been used; mask: right after it went to LDS), so
    // that the HBM latency of a sub-tile hides behind the arithmetic of the previous one. With two workgroups per CU
    // (228 VGPRs) and loads issued at the top of each sub-tile the kernel spent 9 us per 32-pixel sub-tile for 1.3 us of
    // arithmetic per wave. Addresses are clamped instead of branched around (values of rows / channels outside the tile
    // are never used: `on` below), so the loads are unconditional and issue back to back.
    const int cld = min(c, p.C - 4);
    auto row_ptr = [&](const float* base, int px) {
        return reinterpret_cast<const float4*>(base + ((size_t)b * p.HW + min(px, px_end - 1)) * p.C + cld);
    };
    float4 xv[NPI], dv[NPI];
    float mreg[8];
#pragma unroll
    for (int pi = 0; pi < NPI; ++pi) {
        xv[pi] = *row_ptr(p.x, px_begin + prow + PR * pi);
        dv[pi] = *row_ptr(p.dy, px_begin + prow + PR * pi);
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) mreg[o] = p.mask[((size_t)b * O + min(o, O - 1)) * p.HW + min(px_begin + (tid & (NB_PX - 1)), px_end - 1)];

    for (int p0 = px_begin; p0 < px_end; p0 += NB_PX) {
        __syncthreads();   // previous sub-tile's mn / partl consumed
        if (tid < NB_PX) {
            const int px = p0 + tid;
            float m[8];
            float S = 1e-6f;
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                m[o] = (o < O && px < px_end) ? mreg[o] : 0.f;
                S += m[o];
            }
            const float inv = 1.f / S;
            sinv[tid] = inv;
            mn4[2 * tid] = make_float4(m[0] * inv, m[1] * inv, m[2] * inv, m[3] * inv);
            mn4[2 * tid + 1] = make_float4(m[4] * inv, m[5] * inv, m[6] * inv, m[7] * inv);
#pragma unroll
            for (int o = 0; o < 8; ++o) mreg[o] = p.mask[((size_t)b * O + min(o, O - 1)) * p.HW + min(px + NB_PX, px_end - 1)];
        }
        __syncthreads();
#pragma unroll
        for (int pi = 0; pi < NPI; ++pi) {
            const int pl = prow + PR * pi, px = p0 + pl;
            const bool on = con && px < px_end;
            const float4 ma = mn4[2 * pl], mb = mn4[2 * pl + 1];
            const float m[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
            float4 gg = z4, gxv = z4;
            if (on) {
                const float4 xh = make_float4((xv[pi].x - mean.x) * istd.x, (xv[pi].y - mean.y) * istd.y,
                                              (xv[pi].z - mean.z) * istd.z, (xv[pi].w - mean.w) * istd.w);
                float4 ga = make_float4(1, 1, 1, 1), be = z4;
#pragma unroll
                for (int o = 0; o < 8; ++o) { ga = f4mad(m[o], Wr[o], ga); be = f4mad(m[o], Br[o], be); }
                const float4 d = dv[pi];
                gg.x = (!p.relu || fmaf(ga.x, xh.x, be.x) > 0.f) ? d.x : 0.f;
                gg.y = (!p.relu || fmaf(ga.y, xh.y, be.y) > 0.f) ? d.y : 0.f;
                gg.z = (!p.relu || fmaf(ga.z, xh.z, be.z) > 0.f) ? d.z : 0.f;
                gg.w = (!p.relu || fmaf(ga.w, xh.w, be.w) > 0.f) ? d.w : 0.f;
                gxv = make_float4(gg.x * xh.x, gg.y * xh.y, gg.z * xh.z, gg.w * xh.w);
                const float4 dxh = make_float4(gg.x * ga.x, gg.y * ga.y, gg.z * ga.z, gg.w * ga.w);
                *reinterpret_cast<float4*>(p.out_f32 + ((size_t)b * p.HW + px) * p.C + c) = dxh;
                acc_s1.x += dxh.x; acc_s1.y += dxh.y; acc_s1.z += dxh.z; acc_s1.w += dxh.w;
                acc_s2.x += dxh.x * xh.x; acc_s2.y += dxh.y * xh.y; acc_s2.z += dxh.z * xh.z; acc_s2.w += dxh.w * xh.w;
#pragma unroll
                for (int o = 0; o < 8; ++o) { adw[o] = f4mad(m[o], gxv, adw[o]); adb[o] = f4mad(m[o], gg, adb[o]); }
            }
            xv[pi] = *row_ptr(p.x, px + NB_PX);   // the next sub-tile's row into the registers just consumed
            dv[pi] = *row_ptr(p.dy, px + NB_PX);
            if (p.dmask) {   // (every lane of the row takes part in the DPP reduction; idle lanes carry zeros)
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    float part = gxv.x * Wr[o].x + gxv.y * Wr[o].y + gxv.z * Wr[o].z + gxv.w * Wr[o].w +
                                 gg.x * Br[o].x + gg.y * Br[o].y + gg.z * Br[o].z + gg.w * Br[o].w;
                    part = sum_cv<CV>(part);
                    if (cv == 0) partl[o * NB_PX + pl] = part;
                }
            }
        }
        if (p.dmask) {
            __syncthreads();
            for (int i = tid; i < O * NB_PX; i += 256) {
                const int o = i / NB_PX, pl = i - o * NB_PX;
                const int px = p0 + pl;
                if (px >= px_end) continue;
                const float4 ma = mn4[2 * pl], mb = mn4[2 * pl + 1];
                const float t0 = ma.x * partl[0 * NB_PX + pl] + ma.y * partl[1 * NB_PX + pl] + ma.z * partl[2 * NB_PX + pl] +
                                 ma.w * partl[3 * NB_PX + pl] + mb.x * partl[4 * NB_PX + pl] + mb.y * partl[5 * NB_PX + pl] +
                                 mb.z * partl[6 * NB_PX + pl] + mb.w * partl[7 * NB_PX + pl];
                float* dm = p.dmask + ((size_t)b * O + o) * p.HW + px;
                const float v = (partl[i] - t0) * sinv[pl];
                if (p.dmpart) p.dmpart[((size_t)tc * p.B * O + (size_t)b * O + o) * p.HW + px] = v;   // this chunk's row: rows_fold adds the chunks
                else if (p.dmask_store) *dm = v;   // one channel chunk: this thread is the only writer (4 M atomics of the 128^2 layer = 10 us)
                else atomicAdd(dm, v);
            }
        }
    }

    // ---- per-object gradients: sum the PR pixel-row partials through LDS (dW, then dB). With p.part the workgroup's 2 x 8 x CC
    // sums go to its own rows of `part` with plain stores and norm_a8_finish_kernel, the launch behind this one, adds the nseg
    // partial rows of an (image, channel chunk) into dW / dB: 2048 atomics per workgroup (1 M per launch) cost 5-11 us of a
    // 18-57 us launch (tools/perf/norm_shapes.py, round 5; a last-arriver sum inside this kernel needs an agent-scope release
    // per workgroup: 26 -> 71 us). One segment per image: plain read-modify-write here. Without p.part: atomics.
    if (p.part) {
        float4* mine = reinterpret_cast<float4*>(p.part) + ((size_t)(b * tiles_c + tc) * nseg + seg) * (2 * 8 * CV);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();
#pragma unroll
            for (int o = 0; o < 8; ++o) red[(prow * 8 + o) * CV + cv] = pass == 0 ? adw[o] : adb[o];
            __syncthreads();
            if (tid < 8 * CV) {   // 8 objects x CV float4 values per pass
                const int i = tid;
                const int o = i / CV, lc = i - o * CV;
                float4 a = red[o * CV + lc];
#pragma unroll
                for (int r = 1; r < PR; ++r) {
                    const float4 t = red[(r * 8 + o) * CV + lc];
                    a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
                }
                if (nseg == 1) {
                    if (o < O && 4 * lc < cc) {
                        float4* dst = reinterpret_cast<float4*>((pass == 0 ? p.dwproj : p.dbproj) + (size_t)b * p.pstride_b + (size_t)o * p.pstride_o + c0 + 4 * lc);
                        float4 d = *dst;
                        d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
                        *dst = d;
                    }
                } else {
                    mine[pass * 8 * CV + i] = a;
                }
            }
        }
    } else {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#pragma unroll
        for (int o = 0; o < 8; ++o) red[(prow * 8 + o) * CV + cv] = pass == 0 ? adw[o] : adb[o];
        __syncthreads();
        for (int i = tid; i < 8 * CV; i += 256) {
            const int o = i / CV, lc = i - o * CV;
            if (o >= O || 4 * lc >= cc) continue;
            float4 a = red[o * CV + lc];
            for (int r = 1; r < PR; ++r) {
                const float4 t = red[(r * 8 + o) * CV + lc];
                a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
            }
            float* dst = (pass == 0 ? p.dwproj : p.dbproj) + (size_t)b * p.pstride_b + (size_t)o * p.pstride_o + c0 + 4 * lc;
            atomicAdd(dst + 0, a.x); atomicAdd(dst + 1, a.y); atomicAdd(dst + 2, a.z); atomicAdd(dst + 3, a.w);
        }
    }
    }
    // ---- per-channel sums s1, s2
    __syncthreads();
    red[(0 * PR + prow) * CV + cv] = acc_s1;
    red[(1 * PR + prow) * CV + cv] = acc_s2;
    __syncthreads();
    for (int i = tid; i < 2 * CV; i += 256) {
        const int k = i / CV, lc = i - k * CV;
        if (4 * lc >= cc) continue;
        float4 a = red[(k * PR) * CV + lc];
        for (int r = 1; r < PR; ++r) {
            const float4 t = red[(k * PR + r) * CV + lc];
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        const int cch = c0 + 4 * lc;
        if (p.spart) {   // this workgroup's row of the partial matrix (one writer per value)
            *reinterpret_cast<float4*>(p.spart + (size_t)(blockIdx.x / tiles_c) * (2 * p.C) + k * p.C + cch) = a;
            continue;
        }
        float* dst;
        if (p.ws) dst = ws_replica(p.ws, (blockIdx.x / tiles_c) % L2I_WS_R, 2 * p.C) + k * p.C + cch;
        else dst = (k == 0 ? p.s1 : p.s2) + (size_t)b * p.stat_stride + cch;
        atomicAdd(dst + 0, a.x); atomicAdd(dst + 1, a.y); atomicAdd(dst + 2, a.z); atomicAdd(dst + 3, a.w);
    }
}

// Behind norm_bwd_a8_kernel on the same stream: blocks [0, nfold) fold the replicated s1 / s2 workspace (ws_fold_kernel's job),
// the others add the nseg partial dW / dB rows of every (image, channel chunk) into dwproj / dbproj (float4 per thread).
__global__ __launch_bounds__(256) void norm_a8_finish_kernel(WsFoldArgs f, int nfold, const float4* __restrict__ part, float* dw, float* db,
                                                             int B, int tiles_c, int nseg, int CV, int C, int O, long long psb, long long pso,
                                                             RowsFoldArgs sf, RowsFoldArgs df) {
    const int n_sf = sf.src ? sf.nbx : 0, n_df = df.src ? df.nbx : 0;
    if ((int)blockIdx.x < n_sf + n_df) {   // the ordered sums of the workgroups' s1 / s2 rows and of the channel chunks' dmask rows (round 6) ride on this launch
        __shared__ float4 fold_red[256];
        if ((int)blockIdx.x < n_sf) rows_fold2_body(sf, blockIdx.x, 0, fold_red);
        else rows_fold2_body(df, blockIdx.x - n_sf, 0, fold_red);
        return;
    }
    if ((int)blockIdx.x < n_sf + n_df + nfold) { ws_fold_body(f, blockIdx.x - n_sf - n_df); return; }
    const int idx = (blockIdx.x - n_sf - n_df - nfold) * 256 + threadIdx.x;
    const int row = 16 * CV;   // float4 values per partial row: [2 passes][8 objects][CV]
    const int r = idx % row, bt = idx / row;
    if (bt >= B * tiles_c) return;
    const int lc = r % CV, o = (r / CV) % 8, pass = r / (8 * CV);
    const int tc = bt % tiles_c, b = bt / tiles_c;
    const int c = tc * 4 * CV + 4 * lc;
    if (o >= O || c >= C) return;
    const float4* src = part + (size_t)bt * nseg * row + r;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = 0; s0 < nseg; s0 += 8) {   // eight partial rows in flight
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = s0 + u < nseg ? src[(size_t)(s0 + u) * row] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) { a.x += t[u].x; a.y += t[u].y; a.z += t[u].z; a.w += t[u].w; }
    }
    float4* dst = reinterpret_cast<float4*>((pass == 0 ? dw : db) + (size_t)b * psb + (size_t)o * pso + c);
    float4 d = *dst;
    d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
    *dst = d;
}

// Behind norm_bwd_a_kernel (more than 8 objects: VG layouts): dW / dB [b][o][c] += the sum of the image's nseg stored rows, in order.
__global__ __launch_bounds__(256) void norm_a_finish_kernel(const float* __restrict__ part, float* dw, float* db, int B, int nseg, int O, int C,
                                                            long long psb, long long pso) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c4n = C >> 2;
    if (idx >= (long long)B * 2 * O * c4n) return;
    const int c4 = (int)(idx % c4n), o = (int)((idx / c4n) % O), pass = (int)((idx / ((long long)c4n * O)) % 2), b = (int)(idx / ((long long)c4n * O * 2));
    const size_t row = 2 * (size_t)O * C;
    const float* src = part + (size_t)b * nseg * row + ((size_t)pass * O + o) * C + 4 * c4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = 0; s0 < nseg; ++s0) {
        const float4 t = *reinterpret_cast<const float4*>(src + (size_t)s0 * row);
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
    float* d = (pass == 0 ? dw : db) + (size_t)b * psb + (size_t)o * pso + 4 * c4;
    d[0] += a.x; d[1] += a.y; d[2] += a.z; d[3] += a.w;
}

static size_t norm_bwd_lds(const NormArgs& a) {
    const int O = a.mode == 0 ? a.O : 0;
    const size_t tile = a.mode == 0 ? (size_t)2 * NB_PX * NM_CC : (size_t)4 * 4 * 8 * 32;   // g, gx | reduction buffer (float4 x 4 x 8 x 32)
    return sizeof(float) * ((size_t)2 * O * NM_CC + (size_t)2 * O * NB_PX + NB_PX + tile) + 16;
}

// ISLA forward for layouts with at most 8 objects (COCO): the register-resident form. norm_mod_kernel re-reads the
// projections W[o][c], B[o][c] of every object from LDS for EVERY pixel (16 ds_read_b128 per 16 bytes of input: LDS traffic
// 16 x the HBM traffic -- 0.46 of the HBM rate, and half its lanes idle on the 64-channel layers, the largest ones). Here a
// thread owns 4 fixed channels, keeps their 8 + 8 projection rows in registers for a run of 256 pixels, and reads only the 8
// normalised mask values of a pixel (two broadcast ds_read_b128). CV float4 columns per block: 32 (>= 128 channels) or 16.
template <typename T, int CV>
__global__ __launch_bounds__(256) void norm_mod8_kernel(NormArgs p) {
    constexpr int PT = 256, ROWS = 256 / CV;
    __shared__ __attribute__((aligned(16))) float mnl[PT][8];
    const int tiles_p = (p.HW + PT - 1) / PT, tiles_c = (p.C + 4 * CV - 1) / (4 * CV);
    int bid = blockIdx.x;
    const int tc = bid % tiles_c; bid /= tiles_c;
    const int tp = bid % tiles_p, b = bid / tiles_p;
    const int p0 = tp * PT, tid = threadIdx.x;
    if (p.mode == 0) {   // normalised masks of this block's pixels: m_o / (sum_o m_o + 1e-6)
        const int px = p0 + tid;
        float m[8], S = 1e-6f;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            m[o] = (o < p.O && px < p.HW) ? p.mask[((size_t)b * p.O + o) * p.HW + px] : 0.f;
            S += m[o];
        }
        const float inv = 1.f / S;
        *reinterpret_cast<float4*>(&mnl[tid][0]) = make_float4(m[0] * inv, m[1] * inv, m[2] * inv, m[3] * inv);
        *reinterpret_cast<float4*>(&mnl[tid][4]) = make_float4(m[4] * inv, m[5] * inv, m[6] * inv, m[7] * inv);
    }
    const int cv = tid % CV, prow = tid / CV;
    const int c = tc * 4 * CV + 4 * cv;
    const bool con = c < p.C;
    float4 Wr[8], Br[8];
    float4 mean = make_float4(0, 0, 0, 0), istd = make_float4(1, 1, 1, 1);
    float4 ga0 = make_float4(1, 1, 1, 1), be0 = make_float4(0, 0, 0, 0);   // mode 1: the affine weight / bias; mode 2: identity
    if (p.mode == 1 && con) {
        ga0 = *reinterpret_cast<const float4*>(p.wproj + c);
        be0 = *reinterpret_cast<const float4*>(p.bproj + c);
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        Wr[o] = make_float4(0, 0, 0, 0); Br[o] = make_float4(0, 0, 0, 0);
        if (p.mode == 0 && con && o < p.O) {
            const size_t off = (size_t)b * p.pstride_b + (size_t)o * p.pstride_o + c;
            Wr[o] = *reinterpret_cast<const float4*>(p.wproj + off);
            Br[o] = *reinterpret_cast<const float4*>(p.bproj + off);
        }
    }
    if (con) {
        const size_t so = (size_t)b * p.stat_stride + c;
        const float4 s = *reinterpret_cast<const float4*>(p.sums + so);
        const float4 q = *reinterpret_cast<const float4*>(p.sqsums + so);
        const float ic = 1.f / p.count;
        mean = make_float4(s.x * ic, s.y * ic, s.z * ic, s.w * ic);
        const float4 var = make_float4(fmaxf(q.x * ic - mean.x * mean.x, 0.f), fmaxf(q.y * ic - mean.y * mean.y, 0.f),
                                       fmaxf(q.z * ic - mean.z * mean.z, 0.f), fmaxf(q.w * ic - mean.w * mean.w, 0.f));
        istd = make_float4(rsqrtf(var.x + p.eps), rsqrtf(var.y + p.eps), rsqrtf(var.z + p.eps), rsqrtf(var.w + p.eps));
        if (p.run_mean && b == 0 && tp == 0 && prow == 0) {
            // nn.BatchNorm2d train-mode side effect (momentum update with the UNBIASED batch variance), once per channel
            const float ub = p.count / fmaxf(p.count - 1.f, 1.f), mo = p.momentum;
            float4 rm = *reinterpret_cast<float4*>(p.run_mean + c), rv = *reinterpret_cast<float4*>(p.run_var + c);
            rm.x = (1.f - mo) * rm.x + mo * mean.x; rm.y = (1.f - mo) * rm.y + mo * mean.y;
            rm.z = (1.f - mo) * rm.z + mo * mean.z; rm.w = (1.f - mo) * rm.w + mo * mean.w;
            rv.x = (1.f - mo) * rv.x + mo * ub * var.x; rv.y = (1.f - mo) * rv.y + mo * ub * var.y;
            rv.z = (1.f - mo) * rv.z + mo * ub * var.z; rv.w = (1.f - mo) * rv.w + mo * ub * var.w;
            *reinterpret_cast<float4*>(p.run_mean + c) = rm;
            *reinterpret_cast<float4*>(p.run_var + c) = rv;
        }
    }
    __syncthreads();
    T* OutOp = reinterpret_cast<T*>(p.out_op);
    const int npx = min(PT, p.HW - p0);
    // four row loads in flight per thread (round 5: the loop was two loads deep and ran at 0.51 of the HBM rate)
    if (con)
    for (int pl0 = prow; pl0 < npx; pl0 += 4 * ROWS) {
      float4 xq[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
          xq[u] = pl0 + u * ROWS < npx ? *reinterpret_cast<const float4*>(p.x + ((size_t)b * p.HW + p0 + pl0 + u * ROWS) * p.C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pl = pl0 + u * ROWS;
        if (pl >= npx) break;
        const size_t off = ((size_t)b * p.HW + p0 + pl) * p.C + c;
        const float4 xv = xq[u];
        const float4 xh = make_float4((xv.x - mean.x) * istd.x, (xv.y - mean.y) * istd.y, (xv.z - mean.z) * istd.z, (xv.w - mean.w) * istd.w);
        float4 ga = ga0, be = be0;
        if (p.mode == 0) {
            const float4 m0 = *reinterpret_cast<const float4*>(&mnl[pl][0]), m1 = *reinterpret_cast<const float4*>(&mnl[pl][4]);
            ga = f4mad(m0.x, Wr[0], ga); be = f4mad(m0.x, Br[0], be);
            ga = f4mad(m0.y, Wr[1], ga); be = f4mad(m0.y, Br[1], be);
            ga = f4mad(m0.z, Wr[2], ga); be = f4mad(m0.z, Br[2], be);
            ga = f4mad(m0.w, Wr[3], ga); be = f4mad(m0.w, Br[3], be);
            ga = f4mad(m1.x, Wr[4], ga); be = f4mad(m1.x, Br[4], be);
            ga = f4mad(m1.y, Wr[5], ga); be = f4mad(m1.y, Br[5], be);
            ga = f4mad(m1.z, Wr[6], ga); be = f4mad(m1.z, Br[6], be);
            ga = f4mad(m1.w, Wr[7], ga); be = f4mad(m1.w, Br[7], be);
        }
        float4 y = make_float4(fmaf(ga.x, xh.x, be.x), fmaf(ga.y, xh.y, be.y), fmaf(ga.z, xh.z, be.z), fmaf(ga.w, xh.w, be.w));
        if (p.relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
        if (p.out_f32) *reinterpret_cast<float4*>(p.out_f32 + off) = y;
        if (OutOp) {
            if constexpr (sizeof(T) == 2) {
                uint2 pk;
                pk.x = f2bf2(y.x, y.y);
                pk.y = f2bf2(y.z, y.w);
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(OutOp) + off) = pk;
            } else {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(OutOp) + off) = y;
            }
        }
      }
    }
}

static size_t norm_lds(const NormArgs& a) {
    const int O = a.mode == 0 ? a.O : 0;
    return sizeof(float) * ((size_t)2 * O * NM_CC + (size_t)O * NM_PT + NM_PT) + 16;
}

static int norm_check(const NormArgs& a) {
    if (!a.x || !a.sums || !a.sqsums || a.C % 4 || a.B <= 0 || a.HW <= 0) return L2I_ERR_ARG;
    if (a.mode < 0 || a.mode > 2) return L2I_ERR_ARG;
    if (a.mode == 0 && (!a.mask || !a.wproj || !a.bproj || a.O < 1 || a.O > NM_MAXO)) return L2I_ERR_ARG;
    if (a.mode == 1 && (!a.wproj || !a.bproj)) return L2I_ERR_ARG;
    return L2I_OK;
}

extern "C" int l2i_norm_mod_fwd(const float* x, int B, int HW, int C, const float* sums, const float* sqsums, float count,
                                float eps, int stat_stride, const float* mask, int O, const float* wproj,
                                const float* bproj, long long pstride_b, long long pstride_o, int mode, int relu,
                                void* out_op, float* out_f32, int dtype, float* run_mean, float* run_var, float momentum,
                                void* stream) {
    NormArgs a = {};
    a.run_mean = stat_stride == 0 ? run_mean : nullptr; a.run_var = run_var; a.momentum = momentum;
    a.x = x; a.B = B; a.HW = HW; a.C = C; a.sums = sums; a.sqsums = sqsums; a.count = count; a.eps = eps;
    a.stat_stride = stat_stride; a.mask = mask; a.O = O; a.wproj = wproj; a.bproj = bproj;
    a.pstride_b = pstride_b; a.pstride_o = pstride_o; a.mode = mode; a.relu = relu; a.out_op = out_op; a.out_f32 = out_f32;
    if (norm_check(a) != L2I_OK || (!out_op && !out_f32)) return L2I_ERR_ARG;
    if (dtype != 0 && dtype != 1) return L2I_ERR_ARG;
    static const bool no_m8 = getenv("L2I_NORM_M8") && atoi(getenv("L2I_NORM_M8")) == 0;   // tuning: the LDS form for every layer
    if (((mode == 0 && O <= 8) || mode == 1 || mode == 2) && !no_m8) {   // COCO layouts (projections in registers),
                                                                                           // plain batch norms: norm_mod8_kernel
        const int cvw = C <= 64 ? 16 : 32;
        const int nb8 = B * ((HW + 255) / 256) * ((C + 4 * cvw - 1) / (4 * cvw));
        if (dtype == 1 && cvw == 32) hipLaunchKernelGGL((norm_mod8_kernel<bf16_t, 32>), dim3(nb8), dim3(256), 0, (hipStream_t)stream, a);
        else if (dtype == 1) hipLaunchKernelGGL((norm_mod8_kernel<bf16_t, 16>), dim3(nb8), dim3(256), 0, (hipStream_t)stream, a);
        else if (cvw == 32) hipLaunchKernelGGL((norm_mod8_kernel<float, 32>), dim3(nb8), dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((norm_mod8_kernel<float, 16>), dim3(nb8), dim3(256), 0, (hipStream_t)stream, a);
        return l2i_check_launch();
    }
    const int nblk = B * ((HW + NM_PT - 1) / NM_PT) * ((C + NM_CC - 1) / NM_CC);
    if (dtype == 0)
        hipLaunchKernelGGL((norm_mod_kernel<float, false>), dim3(nblk), dim3(256), norm_lds(a), (hipStream_t)stream, a);
    else if (dtype == 1)
        hipLaunchKernelGGL((norm_mod_kernel<bf16_t, false>), dim3(nblk), dim3(256), norm_lds(a), (hipStream_t)stream, a);
    else
        return L2I_ERR_ARG;
    return l2i_check_launch();
}

extern "C" int l2i_norm_mod_bwd_a(const float* x, const float* dy, int B, int HW, int C, const float* sums,
                                  const float* sqsums, float count, float eps, int stat_stride, const float* mask, int O,
                                  const float* wproj, const float* bproj, long long pstride_b, long long pstride_o,
                                  int mode, int relu, float* dxhat, float* s1, float* s2, float* dwproj, float* dbproj,
                                  float* dmask, float* dy_keep, float* ws, float* part, long long part_floats, int dmask_fresh, void* stream) {
    NormArgs a = {};
    a.x = x; a.dy = dy; a.B = B; a.HW = HW; a.C = C; a.sums = sums; a.sqsums = sqsums; a.count = count; a.eps = eps;
    a.stat_stride = stat_stride; a.mask = mask; a.O = O; a.wproj = wproj; a.bproj = bproj;
    a.pstride_b = pstride_b; a.pstride_o = pstride_o; a.mode = mode; a.relu = relu; a.out_f32 = dxhat;
    a.s1 = s1; a.s2 = s2; a.dwproj = dwproj; a.dbproj = dbproj; a.dmask = dmask; a.dy_keep = dy_keep;
    if (norm_check(a) != L2I_OK || !dy || !dxhat || !s1 || !s2) return L2I_ERR_ARG;
    if (mode != 2 && (!dwproj || !dbproj)) return L2I_ERR_ARG;
    (void)dy_keep;   // (formerly a scratch copy of dy for O > 8; object chunks now share the LDS tile)
    const int tiles_c = (C + NM_CC - 1) / NM_CC;
    const int subtiles = (HW + NB_PX - 1) / NB_PX;
    // Workgroups a launch aims for: 512 = one round of two per CU. Every workgroup ends its run with a reduction of its 16 + 2
    // per-channel accumulators through LDS and ~1000 atomics, so longer runs amortise it: same-box A/B of the training
    // iteration 24.35 ms (2048), 24.10 (1024), 23.91 (512). (L2I_NORM_WGS: tuning)
    static const int wg_target = getenv("L2I_NORM_WGS") ? atoi(getenv("L2I_NORM_WGS")) : 512;
    int nseg = wg_target / (B * tiles_c);
    if (nseg > subtiles) nseg = subtiles;
    if (nseg < 1) nseg = 1;
    const int seg_pixels = ((subtiles + nseg - 1) / nseg) * NB_PX;
    nseg = (HW + seg_pixels - 1) / seg_pixels;
    a.ws = (stat_stride == 0 && B * nseg > 32) ? ws : nullptr;   // batch statistics shared by many workgroups (unless their rows are stored: a.spart below)
    // <= 8 objects: dW / dB through per-workgroup partial rows (16 x 64 or 16 x 128 floats each) when the caller lends scratch
    const bool vec_ok = mode == 0 && dwproj && dbproj && ((uintptr_t)dwproj % 16 == 0) && ((uintptr_t)dbproj % 16 == 0) && pstride_b % 4 == 0 && pstride_o % 4 == 0;
    int fin_ns = 0, fin_tc = 0, fin_cv = 0;
    auto lend = [&](int ns, int tc, int cc) {
        a.part = nullptr;
        static const bool off = getenv("L2I_NORM_PART") && atoi(getenv("L2I_NORM_PART")) == 0;
        if (off || !vec_ok || !part) return;
        if (ns > 1 && (long long)B * tc * ns * 16 * cc > part_floats) return;
        a.part = part;
        if (ns > 1) { fin_ns = ns; fin_tc = tc; fin_cv = cc / 4; }
    };
    static bool ready = false;
    if (!ready) {   // up to 31 objects: ~72 KB of dynamic LDS
        (void)hipFuncSetAttribute((const void*)norm_bwd_a_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        ready = true;
    }
    static const bool no_a8 = getenv("L2I_NORM_A8") && atoi(getenv("L2I_NORM_A8")) == 0;   // tuning: the LDS version for every layer
    const bool a8 = mode == 0 && O <= 8 && !no_a8;
    // Round 6: the per-channel totals and the channel chunks' dmask contributions as STORED rows in the tail of the caller's scratch, added in a
    // fixed order by rows_fold behind the launch (the gradient of every layer below a normalisation was moving in its last bits from run to run
    // with the order of these float atomics). Geometry of the kernel that will run:
    const int nval = mode == 1 ? 4 : 2;
    int k_ns = nseg, k_tc = tiles_c;
    if (a8 && C <= 64) {
        k_tc = (C + 63) / 64;
        int ns = wg_target / (B * k_tc);
        if (ns > subtiles) ns = subtiles;
        if (ns < 1) ns = 1;
        const int sp = ((subtiles + ns - 1) / ns) * NB_PX;
        k_ns = (HW + sp - 1) / sp;
    }
    const long long n_dm = (long long)B * O * HW;
    long long tail = part_floats;   // floats of `part` still free (the dW / dB partial rows of the <= 8-object kernel keep the front)
    float* stat_tmp = nullptr; float* dm_tmp = nullptr;
    const int stat_rows = stat_stride == 0 ? B * k_ns : k_ns, stat_z = stat_stride == 0 ? 1 : B;
    if (part && !((uintptr_t)part & 15) && C % 4 == 0) {
        const long long need = (long long)B * k_ns * nval * C + rows_fold_tmp_floats(stat_rows, nval * C, stat_z);
        const long long front = a8 ? (long long)B * k_tc * k_ns * 16 * (C <= 64 ? 64 : 128) : (mode == 0 ? (long long)B * k_ns * 2 * O * C : 0);
        if (front + need <= tail) {
            tail = (tail - need) & ~3LL;
            a.spart = part + tail;
            stat_tmp = a.spart + (long long)B * k_ns * nval * C;
        }
        if (dmask && k_tc > 1 && n_dm % 4 == 0) {
            const long long needm = (long long)k_tc * n_dm;
            if (front + needm <= tail) { tail = (tail - needm) & ~3LL; a.dmpart = part + tail; }
        }
        part_floats = tail;   // (what `lend` below may still hand to the dW / dB rows)
    }
    (void)dm_tmp;
    if (dmask && dmask_fresh && !a.dmpart) {   // the caller's dmask is uninitialised: overwrite it where one workgroup owns a pixel, else clear it first
        if (a8 && C <= NM_CC) a.dmask_store = 1;
        else if (l2i_zero_async(dmask, sizeof(float) * (size_t)B * O * HW, (hipStream_t)stream) != hipSuccess) return L2I_ERR_LAUNCH;
    }
    if (a.spart) a.ws = nullptr;
    if (a8) {   // COCO layouts: the register-resident version
        const size_t lds8 = sizeof(float) * (8 * NB_PX + NB_PX + 8 * NB_PX) + 32 * 1024 + 16;
        if (C <= 64) {
            const int t64 = (C + 63) / 64;
            int ns = wg_target / (B * t64);
            if (ns > subtiles) ns = subtiles;
            if (ns < 1) ns = 1;
            const int sp = ((subtiles + ns - 1) / ns) * NB_PX;
            ns = (HW + sp - 1) / sp;
            a.ws = (stat_stride == 0 && B * ns > 32 && !a.spart) ? ws : nullptr;
            lend(ns, t64, 64);
            hipLaunchKernelGGL(norm_bwd_a8_kernel<16>, dim3(B * ns * t64), dim3(256), lds8, (hipStream_t)stream, a, ns, sp);
        } else {
            lend(nseg, tiles_c, 128);
            hipLaunchKernelGGL(norm_bwd_a8_kernel<32>, dim3(B * nseg * tiles_c), dim3(256), lds8, (hipStream_t)stream, a, nseg, seg_pixels);
        }
    } else {
        // more than 8 objects (VG layouts): the projections' dW / dB as stored rows per (image, segment) + an ordered finish launch (round 6)
        a.part = nullptr;
        const bool rows_ok = mode == 0 && part && !((uintptr_t)part & 15) && C % 4 == 0 && dwproj && dbproj && pstride_b % 4 == 0 && pstride_o % 4 == 0 &&
                             ((uintptr_t)dwproj % 16 == 0) && ((uintptr_t)dbproj % 16 == 0) && (long long)B * nseg * 2 * O * C <= part_floats;
        if (rows_ok) a.part = part;
        hipLaunchKernelGGL(norm_bwd_a_kernel, dim3(B * nseg * tiles_c), dim3(256), norm_bwd_lds(a), (hipStream_t)stream, a, nseg,
                           seg_pixels);
        if (rows_ok) {
            const long long n = (long long)B * 2 * O * (C / 4);
            hipLaunchKernelGGL(norm_a_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)part, dwproj, dbproj,
                               B, nseg, O, C, pstride_b, pstride_o);
        }
    }
    // The ordered folds behind the launch: s1 / s2 (+ the affine layer's dW / dB: four destinations of one job) and the channel chunks' dmask rows.
    // Direct jobs over one group ride on the finish launch when there is one, else they are ONE multi-job launch; others get their own launches.
    RowsFoldArgs sf = {}, df = {};
    const bool stats_direct = a.spart && stat_z == 1 && stat_rows <= L2I_FOLD_DIRECT;
    if (stats_direct) sf = rows_fold_args4(a.spart, stat_rows, nval * C, 1, s1, s2, mode == 1 ? dwproj : nullptr, mode == 1 ? dbproj : nullptr, C, C, 0, 2, nullptr, nval * C);
    if (a.dmpart) df = rows_fold_args4(a.dmpart, k_tc, (int)n_dm, 1, dmask, nullptr, nullptr, nullptr, (int)n_dm, (int)n_dm, 0, dmask_fresh ? 0 : 1, nullptr, (int)n_dm);
    if (fin_ns) {   // one launch: the workspace fold (if any), the sum of the partial dW / dB rows, and the folds above
        WsFoldArgs f = {};
        f.ws = a.ws; f.dst[0] = s1; f.dst[1] = s2; f.L = 2 * C; f.C = C;
        const int nfold = a.ws ? (2 * C + 255) / 256 : 0;
        const int nfin = (int)(((long long)B * fin_tc * 16 * fin_cv + 255) / 256);
        hipLaunchKernelGGL(norm_a8_finish_kernel, dim3(sf.nbx + df.nbx + nfold + nfin), dim3(256), 0, (hipStream_t)stream, f, nfold, (const float4*)part, dwproj,
                           dbproj, B, fin_tc, fin_ns, fin_cv, C, O, pstride_b, pstride_o, sf, df);
    } else {
        if (a.ws) ws_fold(a.ws, (mode == 1 ? 4 : 2) * C, C, s1, s2, dwproj, dbproj, (hipStream_t)stream);
        RowsFoldArgs none = {};
        rows_fold_multi(sf, df, none, (hipStream_t)stream);
    }
    if (a.spart && !stats_direct) {   // (per-image statistics, or more rows than one direct fold takes)
        rows_fold(a.spart, stat_rows, 2 * C, stat_z, s1, s2, C, stat_stride, 1, stat_tmp, (hipStream_t)stream, nullptr, nval * C);
        if (mode == 1)   // the affine layer's dW / dB: columns [2C, 4C) of the same rows, over ALL images
            rows_fold(a.spart + 2 * C, B * k_ns, 2 * C, 1, dwproj, dbproj, C, 0, 2, stat_tmp, (hipStream_t)stream, nullptr, nval * C);
    }
    return l2i_check_launch();
}

// ---------------------------------------------------------------- batch-norm backward, second pass
//   dx = invstd * (dxhat - s1/count - xhat * s2/count)      (+= into dx_out when accumulate)
__global__ __launch_bounds__(256) void norm_bwd_b_kernel(const float* __restrict__ x, const float* __restrict__ dxhat,
                                                         const float* __restrict__ sums, const float* __restrict__ sqsums,
                                                         const float* __restrict__ s1, const float* __restrict__ s2,
                                                         float* __restrict__ dx, long long rows, int C, long long rows_per_group,
                                                         float count, float eps, int accumulate, bf16_t* __restrict__ dx_op) {
    const int cols4 = C >> 2;
    const long long total = rows * cols4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / cols4;
        const int c = 4 * (int)(i - r * cols4);
        const size_t so = (size_t)(r / rows_per_group) * C + c;
        const float ic = 1.f / count;
        const float4 s = *reinterpret_cast<const float4*>(sums + so);
        const float4 q = *reinterpret_cast<const float4*>(sqsums + so);
        const float4 a1 = *reinterpret_cast<const float4*>(s1 + so);
        const float4 a2 = *reinterpret_cast<const float4*>(s2 + so);
        const float4 xv = *reinterpret_cast<const float4*>(x + r * C + c);
        const float4 d = *reinterpret_cast<const float4*>(dxhat + r * C + c);
        float4 o;
#define L2I_BWDB(f)                                                            \
        {                                                                      \
            const float m = s.f * ic;                                          \
            const float is = rsqrtf(fmaxf(q.f * ic - m * m, 0.f) + eps);       \
            const float xh = (xv.f - m) * is;                                  \
            o.f = is * (d.f - a1.f * ic - xh * a2.f * ic);                     \
        }
        L2I_BWDB(x) L2I_BWDB(y) L2I_BWDB(z) L2I_BWDB(w)
#undef L2I_BWDB
        float4* dst = reinterpret_cast<float4*>(dx + r * C + c);
        if (accumulate) {
            const float4 prev = *dst;
            o.x += prev.x; o.y += prev.y; o.z += prev.z; o.w += prev.w;
        }
        *dst = o;
        if (dx_op) {   // the bf16 operand copy of dx the producing convolution's backward reads (instead of a cast pass over dx)
            const float ov[4] = {o.x, o.y, o.z, o.w};
            Op4<bf16_t>::store(dx_op + r * C + c, ov);
        }
    }
}

extern "C" int l2i_norm_bwd_b(const float* x, const float* dxhat, const float* sums, const float* sqsums, const float* s1,
                              const float* s2, float* dx, long long rows, int C, long long rows_per_group, float count,
                              float eps, int accumulate, void* dx_op_bf16, void* stream) {
    if (!x || !dxhat || !sums || !sqsums || !s1 || !s2 || !dx || C % 4 || rows_per_group <= 0) return L2I_ERR_ARG;
    const long long total = rows * (C / 4);
    long long nblk = (total + 255) / 256;
    if (nblk > 4096) nblk = 4096;
    hipLaunchKernelGGL(norm_bwd_b_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, x, dxhat, sums, sqsums, s1,
                       s2, dx, rows, C, rows_per_group, count, eps, accumulate, reinterpret_cast<bf16_t*>(dx_op_bf16));
    return l2i_check_launch();
}
