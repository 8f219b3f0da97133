// Loss, optimiser and operand-cast kernels.
//
//  * hinge losses with fused backward (reference train_context_app_v2.py:159-172, 180-187):
//      mode 0: mean(relu(1 - x))   (D, real)      mode 1: mean(relu(1 + x))  (D, fake)
//      mode 2: -mean(x)            (G)
//    mean over rows with valid != 0 (the reference drops label-0 rows before the loss, model/
//    rcnn_discriminator_app.py:415-417). loss_out += weight * loss ; grad = weight * dloss/dx.
//    count_ptr (optional, device) overrides the divisor with the global row count under data parallelism.
//  * L1 pixel loss with fused backward (train_context_app_v2.py:143,184).
//  * Adam (train_context_app_v2.py:121,127: betas (0, 0.999), eps 1e-8) over one flat buffer.
//  * f32 stream -> T operand casts (raw and/or ReLU'd copy) feeding the MFMA kernels.
#include "common.h"

// Branch-free form: value = max(c + s*x, lo) with (s, c, lo) = (-1, 1, 0) | (+1, 1, 0) | (-1, 0, -inf) for modes
// 0 | 1 | 2. (A three-way `mode` select inside the loop was mis-compiled by hipcc 7.2 for mode 2.)
__global__ __launch_bounds__(256) void hinge_kernel(const float* __restrict__ x, const int* __restrict__ valid, int n, float s,
                                                    float c, float lo, float weight, const float* __restrict__ count_ptr,
                                                    float* __restrict__ loss_out, float* __restrict__ grad) {
    __shared__ float red[16];
    float cnt = 0.f, acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float ok = (!valid || valid[i] != 0) ? 1.f : 0.f;
        cnt += ok;
        acc += ok * fmaxf(c + s * x[i], lo);
    }
    cnt = block_sum(cnt, red);
    acc = block_sum(acc, red);
    if (count_ptr) cnt = *count_ptr;  // data-parallel: mean over the GLOBAL number of rows
    const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float ok = (!valid || valid[i] != 0) ? 1.f : 0.f;
        const float on = (c + s * x[i] > lo) ? 1.f : 0.f;
        grad[i] = ok * on * s * inv * weight;
    }
    if (threadIdx.x == 0) atomicAdd(loss_out, weight * acc * inv);
}

extern "C" int l2i_hinge_fwd_bwd(const float* x, const int* valid, int n, int mode, float weight, const float* count_ptr,
                                 float* loss_out, float* grad, void* stream) {
    if (!x || !loss_out || !grad || n < 0 || mode < 0 || mode > 2) return L2I_ERR_ARG;
    const float s = mode == 1 ? 1.f : -1.f, c = mode == 2 ? 0.f : 1.f, lo = mode == 2 ? -INFINITY : 0.f;
    hipLaunchKernelGGL(hinge_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, valid, n, s, c, lo, weight, count_ptr,
                       loss_out, grad);
    return l2i_check_launch();
}

__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                 float weight, float* __restrict__ loss_out, float* __restrict__ grad, float* __restrict__ part) {
    __shared__ float red[16];
    float acc = 0.f;
    const float gs = weight / (float)n;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float d = a[i] - b[i];
        acc += fabsf(d);
        grad[i] = d > 0.f ? gs : d < 0.f ? -gs : 0.f;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) {
        if (part) *reinterpret_cast<float4*>(part + 4 * blockIdx.x) = make_float4(acc * gs, 0.f, 0.f, 0.f);   // this workgroup's share: a 4-float row
        else atomicAdd(loss_out, acc * gs);
    }
}

// part (optional, 16-byte aligned, >= 4096 floats, contents undefined afterwards): the workgroups' shares of the loss are stored there and added in
// order by a fold launch -- the loss VALUE is then bit-identical from run to run like its gradient (round 6); null: one float atomic per workgroup.
extern "C" int l2i_l1_fwd_bwd(const float* a, const float* b, long long n, float weight, float* loss_out, float* grad, float* part,
                              void* stream) {
    if (!a || !b || !loss_out || !grad || n <= 0 || ((size_t)part & 15)) return L2I_ERR_ARG;
    long long nblk = (n + 255) / 256;
    if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL(l1_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, a, b, n, weight, loss_out, grad, part);
    if (part) {
        const RowsFoldArgs f = rows_fold_args4(part, (int)nblk, 4, 1, loss_out, nullptr, nullptr, nullptr, 4, 1, 0, 1, nullptr, 4);
        hipLaunchKernelGGL(rows_fold2_kernel, dim3(f.nbx, 1, 1), dim3(256), 0, (hipStream_t)stream, f);
    }
    return l2i_check_launch();
}

// torch.optim.Adam (no amsgrad, no weight decay): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n4, float b1, float b2, float eps,
                                                   float step_size, float inv_sqrt_bc2, float grad_scale, float lr,
                                                   const int* __restrict__ step_ptr) {
    if (step_ptr) {   // step count kept on the device (graph-captured iterations): same double-precision bias corrections
        const double t = (double)*step_ptr;
        step_size = (float)((double)lr / (1.0 - pow((double)b1, t)));
        inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow((double)b2, t)));
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
#define L2I_ADAM(f)                                                        \
        {                                                                  \
            const float gr = gg.f * grad_scale;                            \
            mm.f = b1 * mm.f + (1.f - b1) * gr;                            \
            vv.f = b2 * vv.f + (1.f - b2) * gr * gr;                       \
            pp.f -= step_size * mm.f / (sqrtf(vv.f) * inv_sqrt_bc2 + eps); \
        }
        L2I_ADAM(x) L2I_ADAM(y) L2I_ADAM(z) L2I_ADAM(w)
#undef L2I_ADAM
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
}

extern "C" int l2i_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                             float eps, int step, float grad_scale, const int* step_ptr, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || n % 4 || (!step_ptr && step < 1)) return L2I_ERR_ARG;
    float step_size = 0.f, inv_sqrt_bc2 = 0.f;
    if (!step_ptr) {
        const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
        step_size = (float)(lr / bc1);
        inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    }
    long long nblk = (n / 4 + 255) / 256;
    if (nblk > 2048) nblk = 2048;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n / 4, beta1, beta2,
                       eps, step_size, inv_sqrt_bc2, grad_scale, lr, step_ptr);
    return l2i_check_launch();
}

// f32 -> T copies (raw and/or relu). n multiple of 4.
template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ x, T* __restrict__ raw, T* __restrict__ act,
                                                   long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        if constexpr (sizeof(T) == 2) {
            if (raw) {
                uint2 pk;
                pk.x = f2bf2(v.x, v.y);
                pk.y = f2bf2(v.z, v.w);
                reinterpret_cast<uint2*>(raw)[i] = pk;
            }
            if (act) {
                uint2 pk;
                pk.x = f2bf2(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f));
                pk.y = f2bf2(fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                reinterpret_cast<uint2*>(act)[i] = pk;
            }
        } else {
            if (raw) reinterpret_cast<float4*>(raw)[i] = v;
            if (act)
                reinterpret_cast<float4*>(act)[i] = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
        }
    }
}

extern "C" int l2i_cast_op(const float* x, void* raw, void* act, long long n, int dtype, void* stream) {
    if (!x || (!raw && !act) || n <= 0 || n % 4) return L2I_ERR_ARG;
    long long nblk = (n / 4 + 255) / 256;
    if (nblk > 4096) nblk = 4096;
    if (dtype == 0)
        hipLaunchKernelGGL(cast_kernel<float>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, x, (float*)raw,
                           (float*)act, n / 4);
    else if (dtype == 1)
        hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)raw,
                           (bf16_t*)act, n / 4);
    else
        return L2I_ERR_ARG;
    return l2i_check_launch();
}

// Split operand of the "bf16x3" forward mode: x [rows][C] f32 (optionally ReLU'd) -> out [rows][3 C] bf16 = [hi | lo | hi] with
// hi = bf16(x), lo = bf16(x - hi). Against a weight pack laid out [w_hi | w_hi | w_lo] per tap (sn_pack_body SPLIT) the convolution
// kernels accumulate x_hi w_hi + x_lo w_hi + x_hi w_lo. C % 8 == 0; one thread per 8 channels.
__global__ __launch_bounds__(256) void split_cast_kernel(const float* __restrict__ x, bf16_t* __restrict__ out, long long n8, int C8, int relu) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const long long row = i / C8;
        const int c8 = (int)(i - row * C8);
        const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        float lo[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (relu) v[j] = fmaxf(v[j], 0.f);
            lo[j] = v[j] - bf2f(f2bf(v[j]));
        }
        uint4 hi4, lo4;
        hi4.x = f2bf2(v[0], v[1]); hi4.y = f2bf2(v[2], v[3]); hi4.z = f2bf2(v[4], v[5]); hi4.w = f2bf2(v[6], v[7]);
        lo4.x = f2bf2(lo[0], lo[1]); lo4.y = f2bf2(lo[2], lo[3]); lo4.z = f2bf2(lo[4], lo[5]); lo4.w = f2bf2(lo[6], lo[7]);
        bf16_t* d = out + (row * 3 * C8 + c8) * 8;
        *reinterpret_cast<uint4*>(d) = hi4;
        *reinterpret_cast<uint4*>(d + (long long)C8 * 8) = lo4;
        *reinterpret_cast<uint4*>(d + (long long)C8 * 16) = hi4;
    }
}
extern "C" int l2i_split_cast(const float* x, void* out3, long long rows, int C, int relu, void* stream) {
    if (!x || !out3 || rows <= 0 || C <= 0 || C % 8) return L2I_ERR_ARG;
    const long long n8 = rows * (C / 8);
    long long nblk = (n8 + 255) / 256;
    if (nblk > 8192) nblk = 8192;
    hipLaunchKernelGGL(split_cast_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, x, (bf16_t*)out3, n8, C / 8, relu);
    return l2i_check_launch();
}

// Measurement aid (tools/perf/phase_stamps.py): one lane stores the device's wall clock (100 MHz) -- launched between the phases of a
// captured iteration, the slots read back the REAL timeline of a graph replay, with no profiler attached.
__global__ void stamp_kernel(long long* slot) {
    if (threadIdx.x == 0) *slot = (long long)wall_clock64();
}
extern "C" int l2i_debug_stamp(long long* slot, void* stream) {
    if (!slot) return L2I_ERR_ARG;
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, slot);
    return l2i_check_launch();
}

// dx = g * [mask > 0] (+ add)   -- ReLU backward on f32 streams (mask: f32 pre- or post-activation values)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ mask,
                                                       const float* __restrict__ add, float* __restrict__ out, long long n4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 v = reinterpret_cast<const float4*>(g)[i];
        const float4 m = reinterpret_cast<const float4*>(mask)[i];
        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        if (add) {
            const float4 a = reinterpret_cast<const float4*>(add)[i];
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        reinterpret_cast<float4*>(out)[i] = v;
    }
}

extern "C" int l2i_relu_bwd(const float* g, const float* mask, const float* add, float* out, long long n, void* stream) {
    if (!g || !mask || !out || n <= 0 || n % 4) return L2I_ERR_ARG;
    long long nblk = (n / 4 + 255) / 256;
    if (nblk > 4096) nblk = 4096;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, g, mask, add, out, n / 4);
    return l2i_check_launch();
}

// ---------------------------------------------------------------- appearance head: Gram term without the Gram matrices
// Reference model/rcnn_discriminator_app.py:148-157: per ROI, F = relu(app_conv(obj)) viewed as (C, hw), Gram =
// F F^T / C (C x C), and the head applies Linear(2C -> 1) to [sum_rows(Gram)/C... ] -- only  w1 . (Gram 1) / C  of the
// Gram matrix ever reaches the output:  gram_term[r] = (1/C^2) sum_p (sum_c a[r,p,c]) (sum_c a[r,p,c] w1[c]),
// a = relu(x). One pass over x forward (keeping s[r,p], t[r,p]), one pass backward (dx, dw1).
//   x [R][HW][C] f32 (pre-ReLU), w [C];  fwd: out[r] += gram_term (out pre-zeroed), s,t [R][HW]
#define GH_POS 16   // positions per block (4 per wave)
// One workgroup of 16 waves per ROI (a wave per position, four positions each at HW = 64): the ROI's term is the sum of its positions in a
// fixed order and has ONE writer (rounds 1-5: four 4-wave workgroups per ROI and a float atomic each -- the appearance logit moved in its
// last bit from run to run). Same number of waves in flight as before.
__global__ __launch_bounds__(1024) void gram_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             float* __restrict__ out, float* __restrict__ s_keep,
                                                             float* __restrict__ t_keep, int HW, int C) {
    __shared__ float red[16];
    const int r = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float blk = 0.f;
    for (int pos = wave; pos < HW; pos += 16) {
        const float* row = x + ((size_t)r * HW + pos) * C;
        float s = 0.f, t = 0.f;
        for (int c = 4 * lane; c < C; c += 256) {
            const float4 v = *reinterpret_cast<const float4*>(row + c);
            const float4 ww = *reinterpret_cast<const float4*>(w + c);
            const float a0 = fmaxf(v.x, 0.f), a1 = fmaxf(v.y, 0.f), a2 = fmaxf(v.z, 0.f), a3 = fmaxf(v.w, 0.f);
            s += a0 + a1 + a2 + a3;
            t += a0 * ww.x + a1 * ww.y + a2 * ww.z + a3 * ww.w;
        }
        s = wave_sum(s);
        t = wave_sum(t);
        if (lane == 0) { s_keep[(size_t)r * HW + pos] = s; t_keep[(size_t)r * HW + pos] = t; }
        blk += s * t;
    }
    if (lane == 0) red[wave] = blk;
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) a += red[i];
        out[r] += a / ((float)C * (float)C);
    }
}

//   bwd: dx[r,p,c] = [x>0] g[r]/C^2 (t[r,p] + s[r,p] w[c]);  dw[c] += sum_{r,p} g[r]/C^2 s[r,p] a[r,p,c]  (via ws replicas)
__global__ __launch_bounds__(256) void gram_head_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ s_keep, const float* __restrict__ t_keep,
                                                            const float* __restrict__ g, float* __restrict__ dx, bf16_t* __restrict__ dx_op,
                                                            float* __restrict__ ws, int HW, int C, int parts) {
    extern __shared__ float dwl[];   // [4 waves][C]
    const int r = blockIdx.x / parts, part = blockIdx.x % parts;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float gs = g[r] / ((float)C * (float)C);
    for (int c = 4 * lane; c < C; c += 256) {
        const float4 ww = *reinterpret_cast<const float4*>(w + c);
        float4 acc = make_float4(0, 0, 0, 0);
        for (int pi = wave; pi < GH_POS; pi += 4) {
            const int pos = part * GH_POS + pi;
            if (pos >= HW) break;
            const size_t off = ((size_t)r * HW + pos) * C + c;
            const float4 v = *reinterpret_cast<const float4*>(x + off);
            const float s = s_keep[(size_t)r * HW + pos] * gs, t = t_keep[(size_t)r * HW + pos] * gs;
            float4 d;
            d.x = v.x > 0.f ? fmaf(s, ww.x, t) : 0.f; d.y = v.y > 0.f ? fmaf(s, ww.y, t) : 0.f;
            d.z = v.z > 0.f ? fmaf(s, ww.z, t) : 0.f; d.w = v.w > 0.f ? fmaf(s, ww.w, t) : 0.f;
            *reinterpret_cast<float4*>(dx + off) = d;
            if (dx_op) *reinterpret_cast<uint2*>(dx_op + off) = make_uint2(f2bf2(d.x, d.y), f2bf2(d.z, d.w));   // (the operand copy app_conv's conv2 backward reads)
            acc.x = fmaf(s, fmaxf(v.x, 0.f), acc.x); acc.y = fmaf(s, fmaxf(v.y, 0.f), acc.y);
            acc.z = fmaf(s, fmaxf(v.z, 0.f), acc.z); acc.w = fmaf(s, fmaxf(v.w, 0.f), acc.w);
        }
        *reinterpret_cast<float4*>(dwl + wave * C + c) = acc;
    }
    __syncthreads();
    // this workgroup's row of the partial matrix [R * parts][C] (stored; rows_fold adds the rows in order behind the launch: round 6 -- rounds 2-5
    // added into 32 replicas of the workspace with float atomics, and the head's weight gradient moved in its last bits from run to run)
    float* row = ws + (size_t)blockIdx.x * C;
    for (int c = threadIdx.x; c < C; c += 256) row[c] = (dwl[c] + dwl[C + c]) + (dwl[2 * C + c] + dwl[3 * C + c]);
}

extern "C" int l2i_gram_head_fwd(const float* x, const float* w, float* out, float* s_keep, float* t_keep, int R, int HW,
                                 int C, void* stream) {
    if (!x || !w || !out || !s_keep || !t_keep || C % 4 || R < 0) return L2I_ERR_ARG;
    if (R == 0) return L2I_OK;
    hipLaunchKernelGGL(gram_head_fwd_kernel, dim3(R), dim3(1024), 0, (hipStream_t)stream, x, w, out, s_keep, t_keep, HW, C);
    return l2i_check_launch();
}

extern "C" int l2i_gram_head_bwd(const float* x, const float* w, const float* s_keep, const float* t_keep, const float* g,
                                 float* dx, float* dw, float* scratch, long long scratch_floats, int R, int HW, int C, void* dx_op_bf16, void* stream) {
    if (!x || !w || !s_keep || !t_keep || !g || !dx || !dw || !scratch || ((size_t)scratch & 15) || C % 4 || C > 4096 || R < 0) return L2I_ERR_ARG;
    if (R == 0) return L2I_OK;
    const int parts = (HW + GH_POS - 1) / GH_POS;
    const long long rows = (long long)R * parts;
    if (rows * C + rows_fold_tmp_floats((int)rows, C, 1) > scratch_floats) return L2I_ERR_ARG;
    hipLaunchKernelGGL(gram_head_bwd_kernel, dim3(R * parts), dim3(256), sizeof(float) * 4 * C, (hipStream_t)stream, x, w,
                       s_keep, t_keep, g, dx, (bf16_t*)dx_op_bf16, scratch, HW, C, parts);
    rows_fold(scratch, (int)rows, C, 1, dw, nullptr, C, 0, 2, scratch + rows * C, (hipStream_t)stream);
    return l2i_check_launch();
}

// ---------------------------------------------------------------- bilinear resize of planar maps (align_corners = False)
// F.interpolate(mask, size=(H, W), mode="bilinear") as used on the (b, o, 64, 64) object masks at every ISLA norm and
// stage-mask blend (reference model/norm_module.py:172-173, model/resnet_generator_app_v2.py:465-470):
//   src = max((dst + 0.5) * in/out - 0.5, 0); i0 = floor(src); i1 = min(i0 + 1, in - 1); lerp.
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ in, float* __restrict__ out, long long N,
                                                              int h, int w, int H, int W) {
    const long long total = N * H * W;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const long long n = i / ((long long)W * H);
        const float fy = fmaxf(((float)y + 0.5f) * sy - 0.5f, 0.f), fx = fmaxf(((float)x + 0.5f) * sx - 0.5f, 0.f);
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const float* p = in + n * h * w;
        const float top = p[y0 * w + x0] * (1.f - lx) + p[y0 * w + x1] * lx;
        const float bot = p[y1 * w + x0] * (1.f - lx) + p[y1 * w + x1] * lx;
        out[i] = top * (1.f - ly) + bot * ly;
    }
}

extern "C" int l2i_resize_bilinear(const float* in, float* out, long long N, int h, int w, int H, int W, void* stream) {
    if (!in || !out || N < 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return L2I_ERR_ARG;
    const long long total = N * H * W;
    if (total == 0) return L2I_OK;
    long long nblk = (total + 255) / 256;
    if (nblk > 8192) nblk = 8192;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, in, out, N, h, w, H, W);
    return l2i_check_launch();
}

// ---------------------------------------------------------------- stage-mask blend of the generator
// reference model/resnet_generator_app_v2.py:465-470, per block with a mask head:
//   seman = sigmoid(gather(logits, y)) * nearest(bbox_mask, H);  a = sigmoid(alpha[y]);
//   stage = bilinear(bmask, H) * (1 - a) + seman * a
// as ONE forward and two backward launches instead of ~35 elementwise / gather / scatter_add / resize launches per
// stage. logits [B][H][H][Cp] f32 NHWC; bmask, boxm [B][O][S][S] planar f32 (S = 64); y [B][O] int64; alpha [Cp];
// out [B][O][H][H]. S = f * H with f = 1 or even: nearest reads (f*h, f*w); bilinear(align_corners=False) at an even
// integer factor is the lerp (weights 1/2) of the 2 x 2 pixels at f*h + f/2 - 1 + {0,1} -- written in the order
// resize_bilinear_kernel evaluates it so both give the same bits.
__device__ __forceinline__ float sm_sigmoid(float v) { return 1.f / (1.f + __expf(-v)); }

__global__ __launch_bounds__(256) void stage_mask_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ bmask,
                                                             const float* __restrict__ boxm, const float* __restrict__ alpha,
                                                             const long long* __restrict__ y, float* __restrict__ out,
                                                             float* __restrict__ keep, int O, int H, int Cp, int S) {
    const int bo = blockIdx.x, b = bo / O;
    const int f = S / H, off = f / 2 - 1;
    const int cls = (int)y[bo];
    const float a = sm_sigmoid(alpha[cls]);
    const float* bm = bmask + (size_t)bo * S * S;
    const float* xm = boxm + (size_t)bo * S * S;
    const size_t plane = (size_t)gridDim.x * H * H;
    for (int p = blockIdx.y * 256 + threadIdx.x; p < H * H; p += gridDim.y * 256) {
        const int h = p / H, w = p - h * H;
        float rb;
        if (f == 1) {
            rb = bm[p];
        } else {
            const float* q = bm + (size_t)(f * h + off) * S + f * w + off;
            const float top = q[0] * 0.5f + q[1] * 0.5f, bot = q[S] * 0.5f + q[S + 1] * 0.5f;
            rb = top * 0.5f + bot * 0.5f;
        }
        const float sg = sm_sigmoid(Cp ? logits[((size_t)b * H * H + p) * Cp + cls] : logits[(size_t)bo * H * H + p]);   // Cp = 0: logits already gathered per object, planar [B][O][H][H] (class_logits_fwd_kernel)
        const float m = xm[(size_t)(f * h) * S + f * w];
        out[(size_t)bo * H * H + p] = rb * (1.f - a) + (sg * m) * a;
        keep[(size_t)bo * H * H + p] = sg;
        keep[plane + (size_t)bo * H * H + p] = rb;
    }
}

// backward, planar part: gl[b,o,p] = g a m s(1-s) (the logit gradient before the per-class scatter),
// dbmask (every pixel of the S x S plane written), dalpha[y] += a(1-a) sum_p g (s m - rb).
__global__ __launch_bounds__(256) void stage_mask_bwd_planar_kernel(const float* __restrict__ g, const float* __restrict__ keep,
                                                                    const float* __restrict__ boxm, const float* __restrict__ alpha,
                                                                    const long long* __restrict__ y, float* __restrict__ gl,
                                                                    float* __restrict__ dbmask, float* __restrict__ dalpha,
                                                                    int O, int H, int S) {
    __shared__ float red[8];
    const int bo = blockIdx.x;
    const int f = S / H, off = f / 2 - 1;
    const int cls = (int)y[bo];
    const float a = sm_sigmoid(alpha[cls]);
    const float* xm = boxm + (size_t)bo * S * S;
    const float* gp = g + (size_t)bo * H * H;
    const size_t plane = (size_t)gridDim.x * H * H;
    float acc = 0.f;
    for (int p = threadIdx.x; p < H * H; p += 256) {
        const int h = p / H, w = p - h * H;
        const float gv = gp[p], sg = keep[(size_t)bo * H * H + p], rb = keep[plane + (size_t)bo * H * H + p];
        const float m = xm[(size_t)(f * h) * S + f * w];
        gl[(size_t)bo * H * H + p] = gv * a * m * sg * (1.f - sg);
        acc += gv * (sg * m - rb);
    }
    float* db = dbmask + (size_t)bo * S * S;
    for (int q = threadIdx.x; q < S * S; q += 256) {
        const int Y = q / S, X = q - Y * S;
        float v;
        if (f == 1) {
            v = gp[q] * (1.f - a);
        } else {
            const int ry = Y % f - off, rx = X % f - off;
            v = (ry == 0 || ry == 1) && (rx == 0 || rx == 1) ? 0.25f * (1.f - a) * gp[(Y / f) * H + X / f] : 0.f;
        }
        db[q] = v;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) dalpha[bo] = acc * a * (1.f - a);   // (dalpha: here the per-slot shares [B * O]; stage_mask_dalpha_kernel adds a class's slots in order)
}

// dalpha[class] += the shares of the slots that carry the class, in slot order: one thread per class (round 6; one float atomic per slot before).
__global__ __launch_bounds__(256) void stage_mask_dalpha_kernel(const float* __restrict__ share, const long long* __restrict__ y, float* __restrict__ dalpha, int BO, int n) {
    __shared__ __attribute__((aligned(16))) int ys[1024];
    __shared__ __attribute__((aligned(16))) float sh[1024];
    const int c = blockIdx.x * 256 + threadIdx.x;
    float s = 0.f;
    for (int r0 = 0; r0 < BO; r0 += 1024) {   // (slots staged in LDS: the scan is 256 LDS reads per thread, not 256 dependent global loads)
        const int nr = min(1024, BO - r0);
        __syncthreads();
        for (int i = threadIdx.x; i < nr; i += 256) { ys[i] = (int)y[r0 + i]; sh[i] = share[r0 + i]; }
        __syncthreads();
        // (four slots per LDS read, eight reads in flight: one slot per iteration was a chain of 256 LDS round trips, 15 us for 1 us of work)
        const int4* y4 = reinterpret_cast<const int4*>(ys);
        const float4* s4 = reinterpret_cast<const float4*>(sh);
        int r = 0;
#pragma unroll 4
        for (; r + 4 <= nr; r += 4) {
            const int4 yy = y4[r >> 2];
            const float4 ss = s4[r >> 2];
            s += yy.x == c ? ss.x : 0.f; s += yy.y == c ? ss.y : 0.f; s += yy.z == c ? ss.z : 0.f; s += yy.w == c ? ss.w : 0.f;
        }
        for (; r < nr; ++r) s += ys[r] == c ? sh[r] : 0.f;
    }
    if (c < n && s != 0.f) dalpha[c] += s;
}

// backward, logits: dlogits[b,p,c] = sum_o [y[b,o] == c] gl[b,o,p] -- every element written (no zero fill + scatter_add).
__global__ __launch_bounds__(256) void stage_mask_bwd_logits_kernel(const float* __restrict__ gl, const long long* __restrict__ y,
                                                                    float* __restrict__ dlogits, int O, int HH, int Cp, long long total4) {
    const int c4n = Cp / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const int c = 4 * (int)(i % c4n);
        const long long bp = i / c4n;
        const int b = (int)(bp / HH), p = (int)(bp - (long long)b * HH);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int o = 0; o < O; ++o) {
            const int d = (int)y[b * O + o] - c;
            if (d >= 0 && d < 4) {
                const float t = gl[((size_t)b * O + o) * HH + p];
                v.x += d == 0 ? t : 0.f; v.y += d == 1 ? t : 0.f; v.z += d == 2 ? t : 0.f; v.w += d == 3 ? t : 0.f;
            }
        }
        *reinterpret_cast<float4*>(dlogits + 4 * i) = v;
    }
}

static bool stage_mask_geom_ok(int B, int O, int H, int Cp, int S) {
    if (B <= 0 || O <= 0 || H <= 0 || S <= 0 || Cp < 0 || Cp % 4 || S % H) return false;   // (Cp = 0: planar gathered logits)
    const int f = S / H;
    return f == 1 || f % 2 == 0;
}

extern "C" int l2i_stage_mask_fwd(const float* logits, const float* bmask, const float* boxm, const float* alpha,
                                  const long long* y, float* out, float* keep, int B, int O, int H, int Cp, int S, void* stream) {
    if (!logits || !bmask || !boxm || !alpha || !y || !out || !keep || !stage_mask_geom_ok(B, O, H, Cp, S)) return L2I_ERR_ARG;
    int parts = (H * H + 1023) / 1024;
    hipLaunchKernelGGL(stage_mask_fwd_kernel, dim3(B * O, parts), dim3(256), 0, (hipStream_t)stream, logits, bmask, boxm, alpha, y,
                       out, keep, O, H, Cp, S);
    return l2i_check_launch();
}

extern "C" int l2i_stage_mask_bwd(const float* g, const float* keep, const float* boxm, const float* alpha, const long long* y,
                                  float* gl, float* dlogits, float* dbmask, float* dalpha, int B, int O, int H, int Cp, int S,
                                  float* share, int n_alpha, void* stream) {
    if (!g || !keep || !boxm || !alpha || !y || !gl || (!dlogits && Cp) || !dbmask || !dalpha || !share || n_alpha <= 0 || !stage_mask_geom_ok(B, O, H, Cp, S))
        return L2I_ERR_ARG;
    hipLaunchKernelGGL(stage_mask_bwd_planar_kernel, dim3(B * O), dim3(256), 0, (hipStream_t)stream, g, keep, boxm, alpha, y, gl,
                       dbmask, share, O, H, S);
    hipLaunchKernelGGL(stage_mask_dalpha_kernel, dim3((n_alpha + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)share, y, dalpha, B * O, n_alpha);
    if (Cp == 0) return l2i_check_launch();   // planar logits: gl IS their gradient (no dense [B][H][H][Cp] tensor exists)
    const long long total4 = (long long)B * H * H * (Cp / 4);
    long long nblk = (total4 + 255) / 256;
    if (nblk > 16384) nblk = 16384;
    hipLaunchKernelGGL(stage_mask_bwd_logits_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, gl, y, dlogits, O,
                       H * H, Cp, total4);
    return l2i_check_launch();
}

// ---------------------------------------------------------------- class-gathered logits of the generator's mask heads
// reference model/resnet_generator_app_v2.py:643-651 + :465-466: every mask head ends in Conv2d(100, 184, 1) and the ONLY reader of its
// 184-channel result is  seman = gather(m, 1, y)  -- the <= 8 channels of the image's own object classes. Computing all 184 (a 96 MB
// f32 tensor at 64 x 64, its bias-gradient pass, a dense mostly-zero gradient, three GEMM launches each way) is work nobody reads:
//   lg[b,o,p] = bias[y[b,o]] + sum_c a[b,p,c] W[y[b,o],c]                                   (forward: 8 x 100 MACs per pixel)
//   da[b,p,c] = sum_o gl[b,o,p] W[y_o,c];  dW[y_o,c] += sum_p gl[b,o,p] a[b,p,c];  dbias[y_o] += sum_p gl[b,o,p]     (backward)
// a [B][HH][Cp] f32 NHWC (Cp >= C, multiple of 4), W [classes][ldw] f32, lg / gl planar [B][O][HH], O <= 8.
#define CL_O 8
__global__ __launch_bounds__(256) void class_logits_fwd_kernel(const float* __restrict__ a, const float* __restrict__ w, const float* __restrict__ bias,
                                                               const long long* __restrict__ y, float* __restrict__ lg, int O, int HH, int Cp, int C, int ldw) {
    __shared__ float wt[128][CL_O];   // [channel][object]: a pixel's 8 dot products read two broadcast float4 per channel
    __shared__ float bs[CL_O];
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < 128 * CL_O; i += 256) {
        const int c = i / CL_O, o = i % CL_O;
        wt[c][o] = (c < C && o < O) ? w[(size_t)y[b * O + o] * ldw + c] : 0.f;
    }
    if (threadIdx.x < CL_O) bs[threadIdx.x] = (threadIdx.x < O && bias) ? bias[y[b * O + threadIdx.x]] : 0.f;
    __syncthreads();
    for (int p = blockIdx.y * 256 + threadIdx.x; p < HH; p += gridDim.y * 256) {
        const float4* row = reinterpret_cast<const float4*>(a + ((size_t)b * HH + p) * Cp);
        float acc[CL_O];
#pragma unroll
        for (int o = 0; o < CL_O; ++o) acc[o] = bs[o];
        for (int c4 = 0; c4 < (C + 3) / 4; ++c4) {
            const float4 v = row[c4];
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 w0 = *reinterpret_cast<const float4*>(&wt[4 * c4 + k][0]), w1 = *reinterpret_cast<const float4*>(&wt[4 * c4 + k][4]);
                acc[0] = fmaf(e[k], w0.x, acc[0]); acc[1] = fmaf(e[k], w0.y, acc[1]); acc[2] = fmaf(e[k], w0.z, acc[2]); acc[3] = fmaf(e[k], w0.w, acc[3]);
                acc[4] = fmaf(e[k], w1.x, acc[4]); acc[5] = fmaf(e[k], w1.y, acc[5]); acc[6] = fmaf(e[k], w1.z, acc[6]); acc[7] = fmaf(e[k], w1.w, acc[7]);
            }
        }
#pragma unroll
        for (int o = 0; o < CL_O; ++o)
            if (o < O) lg[((size_t)b * O + o) * HH + p] = acc[o];
    }
}

// backward: a thread owns FOUR consecutive channels (one float4 of a pixel row: Cp / 4 <= 32 lanes per pixel, PG = 256 / (Cp / 4)
// pixels per block iteration), loops over the block's pixels with four row loads in flight: a and da move as 16-byte accesses of whole
// rows, the 8 gradients of a pixel are LDS broadcasts, a thread keeps its 8 x 4 block of dW in registers; the PG partial blocks are
// combined through LDS and added with one atomic per (class, channel); the bias gradient is summed by the threads that stage gl.
// (first version: one channel per thread, one dependent 4-byte load per iteration, 8 threads summing the bias serially: 64 us per launch.)
__global__ __launch_bounds__(256) void class_logits_bwd_kernel(const float* __restrict__ a, const float* __restrict__ w, const long long* __restrict__ y,
                                                               const float* __restrict__ gl, float* __restrict__ da, float* __restrict__ dw,
                                                               float* __restrict__ dbias, int O, int HH, int Cp, int C, int ldw, int per) {
    __shared__ float gs[CL_O][256];
    __shared__ float red[CL_O][128];
    const int b = blockIdx.x, L4 = Cp >> 2, PG = 256 / L4;
    const int c4 = threadIdx.x % L4, pg = threadIdx.x / L4, c = 4 * c4;
    const bool act = pg < PG;
    const int p0 = blockIdx.y * per, p1 = min(HH, p0 + per);
    float wr[CL_O][4], acc[CL_O][4], gpart[CL_O];
    int cls[CL_O];
#pragma unroll
    for (int o = 0; o < CL_O; ++o) {
        cls[o] = o < O ? (int)y[b * O + o] : 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { wr[o][k] = (o < O && c + k < C) ? w[(size_t)cls[o] * ldw + c + k] : 0.f; acc[o][k] = 0.f; }
        gpart[o] = 0.f;
    }
    for (int q0 = p0; q0 < p1; q0 += 256) {
        const int nq = min(256, p1 - q0);
        __syncthreads();
#pragma unroll
        for (int o = 0; o < CL_O; ++o) {
            const float v = (o < O && (int)threadIdx.x < nq) ? gl[((size_t)b * O + o) * HH + q0 + threadIdx.x] : 0.f;
            gs[o][threadIdx.x] = v;
            gpart[o] += v;
        }
        __syncthreads();
        if (act) {
            const float4* arow = reinterpret_cast<const float4*>(a + ((size_t)b * HH + q0) * Cp) + c4;
            float4* drow = reinterpret_cast<float4*>(da + ((size_t)b * HH + q0) * Cp) + c4;
            for (int q = pg; q < nq; q += 4 * PG) {
                float4 av[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) av[u] = q + u * PG < nq ? arow[(size_t)(q + u * PG) * L4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int qq = q + u * PG;
                    if (qq < nq) {
                        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int o = 0; o < CL_O; ++o) {
                            const float g_ = gs[o][qq];
                            d.x = fmaf(g_, wr[o][0], d.x); d.y = fmaf(g_, wr[o][1], d.y); d.z = fmaf(g_, wr[o][2], d.z); d.w = fmaf(g_, wr[o][3], d.w);
                            acc[o][0] = fmaf(g_, av[u].x, acc[o][0]); acc[o][1] = fmaf(g_, av[u].y, acc[o][1]);
                            acc[o][2] = fmaf(g_, av[u].z, acc[o][2]); acc[o][3] = fmaf(g_, av[u].w, acc[o][3]);
                        }
                        drow[(size_t)qq * L4] = d;
                    }
                }
            }
        }
    }
    // combine the PG pixel groups: group after group adds its block into red[o][channel] (Cp <= 128)
    __syncthreads();
    for (int i = threadIdx.x; i < CL_O * 128; i += 256) (&red[0][0])[i] = 0.f;
    __syncthreads();
    for (int g_ = 0; g_ < PG; ++g_) {
        if (act && pg == g_) {
#pragma unroll
            for (int o = 0; o < CL_O; ++o)
#pragma unroll
                for (int k = 0; k < 4; ++k) red[o][c + k] += acc[o][k];
        }
        __syncthreads();
    }
    // this workgroup's rows of `tmp` ([parts][B * O][128]): STORED (round 6; rounds 4-5 added the `parts` workgroups of an image into one row with
    // float atomics). The finish kernel adds the parts and the slots of a class in a fixed order. (Adding into dw[class] directly would put every
    // image's padding slots (class 0) on the same 100 addresses: ~3000 same-address atomics = 90 us.)
    float* mine = dw + ((size_t)blockIdx.y * gridDim.x * O + (size_t)b * O) * 128;
    for (int i = threadIdx.x; i < CL_O * 128; i += 256) {
        const int o = i >> 7, ch = i & 127;
        if (o < O && ch < 127) mine[(size_t)o * 128 + ch] = ch < C ? red[o][ch] : 0.f;
    }
    {   // column 127 of the row: the bias gradient (zero when there is no bias) -- wave-level sums of the staging threads' partials
        __syncthreads();
#pragma unroll
        for (int o = 0; o < CL_O; ++o) {
            float v = gpart[o];
#pragma unroll
            for (int sh = 32; sh > 0; sh >>= 1) v += __shfl_down(v, sh, 64);
            if ((threadIdx.x & 63) == 0) red[o][threadIdx.x >> 6] = v;
        }
        __syncthreads();
        if (threadIdx.x < CL_O && (int)threadIdx.x < O)
            mine[(size_t)threadIdx.x * 128 + 127] = dbias ? (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]) : 0.f;
    }
}

// second kernel: one workgroup per CLASS gathers the rows of the (image, object) slots that carry it -- no atomics, fixed summation order.
// 128 channels x 8 slot phases: the padding class is carried by ~100 slots, and a single thread walking them is 100 dependent loads (36 us measured).
__global__ __launch_bounds__(1024) void class_logits_bwd_finish_kernel(const float* __restrict__ tmp, const long long* __restrict__ y, float* __restrict__ dw,
                                                                       float* __restrict__ dbias, int BO, int C, int ldw, int parts) {
    __shared__ int ys[1024];
    __shared__ float part[8][128];
    const int k = blockIdx.x, c = threadIdx.x & 127, ph = threadIdx.x >> 7;
    float v = 0.f;
    for (int r0 = 0; r0 < BO; r0 += 1024) {
        const int n = min(1024, BO - r0);
        __syncthreads();
        if ((int)threadIdx.x < n) ys[threadIdx.x] = (int)y[r0 + threadIdx.x];
        __syncthreads();
        for (int r = ph; r < n; r += 8)
            if (ys[r] == k)
                for (int pt = 0; pt < parts; ++pt) v += tmp[((size_t)pt * BO + r0 + r) * 128 + c];
    }
    part[ph][c] = v;
    __syncthreads();
    if (ph == 0) {
        v = ((part[0][c] + part[1][c]) + (part[2][c] + part[3][c])) + ((part[4][c] + part[5][c]) + (part[6][c] + part[7][c]));
        if (c < C) dw[(size_t)k * ldw + c] += v;
        if (c == 127 && dbias) dbias[k] += v;
    }
}

extern "C" int l2i_class_logits_fwd(const float* a, const float* w, const float* bias, const long long* y, float* lg, int B, int O, int HH,
                                    int Cp, int C, int ldw, void* stream) {
    if (!a || !w || !y || !lg || B <= 0 || O <= 0 || O > CL_O || HH <= 0 || C <= 0 || C > 128 || Cp < C || Cp % 4 || ldw < C) return L2I_ERR_ARG;
    int parts = (HH + 255) / 256;
    if (parts > 16) parts = 16;
    hipLaunchKernelGGL(class_logits_fwd_kernel, dim3(B, parts), dim3(256), 0, (hipStream_t)stream, a, w, bias, y, lg, O, HH, Cp, C, ldw);
    return l2i_check_launch();
}

// dw [classes][ldw] and dbias [classes] are ADDED to (one workgroup per class, no atomics: deterministic). tmp: f32 scratch of
// (l2i_class_logits_bwd_parts(HH) + 1) * B * O * 128 floats (contents undefined before and after): the per-(pixel part, image, object) rows the first kernel
// stores (column 127: the bias gradient).
extern "C" int l2i_class_logits_bwd(const float* a, const float* w, const long long* y, const float* gl, float* da, float* dw, float* dbias,
                                    float* tmp, int classes, int B, int O, int HH, int Cp, int C, int ldw, void* stream) {
    if (!a || !w || !y || !gl || !da || !dw || !tmp || classes <= 0 || B <= 0 || O <= 0 || O > CL_O || HH <= 0 || C <= 0 || C > 126 || Cp < C || Cp > 128 ||
        Cp % 4 || ldw < C)
        return L2I_ERR_ARG;
    const int per = HH >= 2048 ? 128 : 256;   // pixels per workgroup (more, shorter workgroups on the large maps: the loop is latency-bound)
    const int parts = (HH + per - 1) / per;
    hipLaunchKernelGGL(class_logits_bwd_kernel, dim3(B, parts), dim3(256), 0, (hipStream_t)stream, a, w, y, gl, da, tmp, dbias ? tmp : nullptr, O, HH, Cp, C, ldw, per);
    const float* rows = tmp;
    if (parts > 1) {   // the pixel parts of every (image, object) slot first, in order (slab `parts` of tmp), then the slots of a class
        float* sum = tmp + (size_t)parts * B * O * 128;
        rows_fold(tmp, parts, B * O * 128, 1, sum, nullptr, B * O * 128, 0, 0, nullptr, (hipStream_t)stream);
        rows = sum;
    }
    hipLaunchKernelGGL(class_logits_bwd_finish_kernel, dim3(classes), dim3(1024), 0, (hipStream_t)stream, rows, y, dw, dbias, B * O, C, ldw, 1);
    return l2i_check_launch();
}
extern "C" int l2i_class_logits_bwd_parts(int HH) { return (HH + (HH >= 2048 ? 128 : 256) - 1) / (HH >= 2048 ? 128 : 256); }

// ---------------------------------------------------------------- projection heads of the discriminator
// reference model/rcnn_discriminator_app.py:127-129 (image head) and :160-166 (object head):
//   f[r,c] = scale * sum_p relu(x[r,p,c]);  out[r] = sum_c f[r,c] (wl[c] + E[y[r],c]) + bias      (E, y optional)
// one launch each way instead of relu / sum / linear / index_select / mul / sum / add and their ~20 backward launches.
// wl and E are read straight from the pass's packed operand copies (T = bf16 or f32; E rows emb_stride apart) and the
// weight gradients are added straight into the pass's f32 dW accumulators (demb rows demb_stride apart).
template <typename T>
__global__ __launch_bounds__(256) void proj_head_fwd_kernel(const float* __restrict__ x, const T* __restrict__ wl,
                                                            const T* __restrict__ emb, int emb_stride, const long long* __restrict__ y,
                                                            const float* __restrict__ bias, float scale, float* __restrict__ out,
                                                            float* __restrict__ feat, int HW, int C) {
    __shared__ float red[8];
    const int r = blockIdx.x;
    const T* e = emb ? emb + (size_t)y[r] * emb_stride : nullptr;
    float acc = 0.f;
    for (int c = 4 * threadIdx.x; c < C; c += 1024) {
        float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* col = x + (size_t)r * HW * C + c;
        for (int p = 0; p < HW; ++p) {
            const float4 v = *reinterpret_cast<const float4*>(col + (size_t)p * C);
            f.x += fmaxf(v.x, 0.f); f.y += fmaxf(v.y, 0.f); f.z += fmaxf(v.z, 0.f); f.w += fmaxf(v.w, 0.f);
        }
        f.x *= scale; f.y *= scale; f.z *= scale; f.w *= scale;
        *reinterpret_cast<float4*>(feat + (size_t)r * C + c) = f;
        float w0 = OpT<T>::to(wl[c]), w1 = OpT<T>::to(wl[c + 1]), w2 = OpT<T>::to(wl[c + 2]), w3 = OpT<T>::to(wl[c + 3]);
        if (e) { w0 += OpT<T>::to(e[c]); w1 += OpT<T>::to(e[c + 1]); w2 += OpT<T>::to(e[c + 2]); w3 += OpT<T>::to(e[c + 3]); }
        acc += f.x * w0 + f.y * w1 + f.z * w2 + f.w * w3;
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) out[r] = acc + (bias ? bias[0] : 0.f);
}

// dst[c] += sum_r g[r] M[row(r)][c] for the 16 columns c0..c0+15, by one 256-thread block: 16 row slices x 16 columns
// (64-byte row segments), the loads of a slice independent of each other; the slices are combined in LDS.
// rows: optional row indirection (class ids), else row(r) = r.
template <typename T>
__device__ __forceinline__ void colsum16(const T* __restrict__ M, int ld, const long long* __restrict__ rows,
                                         const float* __restrict__ g, int R, int C, int c0, float* __restrict__ dst, float* red) {
    const int cl = threadIdx.x & 15, rs = threadIdx.x >> 4, c = c0 + cl;
    float t = 0.f;
    if (c < C) {
#pragma unroll 4
        for (int r = rs; r < R; r += 16) {
            const size_t row = rows ? (size_t)rows[r] : (size_t)r;
            t = fmaf(g[r], OpT<T>::to(M[row * ld + c]), t);
        }
    }
    red[rs * 16 + cl] = t;
    __syncthreads();
    if (rs == 0 && c < C) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += red[k * 16 + cl];
        atomicAdd(dst + c, a);
    }
}

// blocks [0, R): dx of row r and dE[y[r]] += g[r] f[r];  blocks [R, R + ceil(C/16)): dwl[c] += sum_r g[r] f[r,c], and the
// first of them dbias += sum_r g[r]. (A transposed pass over the saved f instead of R-way atomics per channel.)
template <typename T>
__global__ __launch_bounds__(256) void proj_head_bwd_kernel(const float* __restrict__ x, const T* __restrict__ wl,
                                                            const T* __restrict__ emb, int emb_stride, const long long* __restrict__ y,
                                                            const float* __restrict__ g, const float* __restrict__ feat, float scale,
                                                            float* __restrict__ dx, float* __restrict__ dwl, float* __restrict__ demb,
                                                            int demb_stride, float* __restrict__ dbias, int R, int HW, int C, bf16_t* __restrict__ dx_op) {
    __shared__ float red[256];
    if ((int)blockIdx.x >= R) {
        if (dwl) colsum16<float>(feat, C, nullptr, g, R, C, ((int)blockIdx.x - R) * 16, dwl, red);
        if ((int)blockIdx.x == R && dbias) {
            __syncthreads();
            float t = 0.f;
            for (int r = threadIdx.x; r < R; r += 256) t += g[r];
            t = block_sum(t, red);
            if (threadIdx.x == 0) atomicAdd(dbias, t);
        }
        return;
    }
    const int r = blockIdx.x;
    const float gr = g[r], gs = gr * scale;
    const long long cls = emb ? y[r] : 0;
    const T* e = emb ? emb + (size_t)cls * emb_stride : nullptr;
    for (int c = 4 * threadIdx.x; c < C; c += 1024) {
        float w0 = OpT<T>::to(wl[c]), w1 = OpT<T>::to(wl[c + 1]), w2 = OpT<T>::to(wl[c + 2]), w3 = OpT<T>::to(wl[c + 3]);
        if (e) { w0 += OpT<T>::to(e[c]); w1 += OpT<T>::to(e[c + 1]); w2 += OpT<T>::to(e[c + 2]); w3 += OpT<T>::to(e[c + 3]); }
        w0 *= gs; w1 *= gs; w2 *= gs; w3 *= gs;
        const size_t base = (size_t)r * HW * C + c;
        for (int p = 0; p < HW; ++p) {
            const float4 v = *reinterpret_cast<const float4*>(x + base + (size_t)p * C);
            float4 d;
            d.x = v.x > 0.f ? w0 : 0.f; d.y = v.y > 0.f ? w1 : 0.f; d.z = v.z > 0.f ? w2 : 0.f; d.w = v.w > 0.f ? w3 : 0.f;
            *reinterpret_cast<float4*>(dx + base + (size_t)p * C) = d;
            if (dx_op) *reinterpret_cast<uint2*>(dx_op + base + (size_t)p * C) = make_uint2(f2bf2(d.x, d.y), f2bf2(d.z, d.w));
        }
    }
    if (demb) {
        // dE[class] += sum over the rows that carry the class of g[r] f[r]: the FIRST such row's workgroup adds them all, in row order, and is the
        // only writer of dE[class] in this launch (round 6; rounds 2-5: one float atomic per row and channel, whose order changed from run to run)
        __shared__ int rlist[L2I_CLASS_LIST];
        __shared__ int wsum[4];
        if (!class_first(y, r, cls)) return;
        const int nl = class_rows(y, r, R, cls, rlist, wsum);
        for (int c = 4 * threadIdx.x; c < C; c += 1024) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            int i = 0;
            for (; i + 4 <= nl; i += 4) {   // (four rows in flight; the padding class is carried by ~100 rows)
                const int r0 = rlist[i], r1 = rlist[i + 1], r2 = rlist[i + 2], r3 = rlist[i + 3];
                const float g0 = g[r0], g1 = g[r1], g2 = g[r2], g3 = g[r3];
                const float4 f0 = *reinterpret_cast<const float4*>(feat + (size_t)r0 * C + c), f1 = *reinterpret_cast<const float4*>(feat + (size_t)r1 * C + c);
                const float4 f2 = *reinterpret_cast<const float4*>(feat + (size_t)r2 * C + c), f3 = *reinterpret_cast<const float4*>(feat + (size_t)r3 * C + c);
                a.x = fmaf(g0, f0.x, a.x); a.y = fmaf(g0, f0.y, a.y); a.z = fmaf(g0, f0.z, a.z); a.w = fmaf(g0, f0.w, a.w);
                a.x = fmaf(g1, f1.x, a.x); a.y = fmaf(g1, f1.y, a.y); a.z = fmaf(g1, f1.z, a.z); a.w = fmaf(g1, f1.w, a.w);
                a.x = fmaf(g2, f2.x, a.x); a.y = fmaf(g2, f2.y, a.y); a.z = fmaf(g2, f2.z, a.z); a.w = fmaf(g2, f2.w, a.w);
                a.x = fmaf(g3, f3.x, a.x); a.y = fmaf(g3, f3.y, a.y); a.z = fmaf(g3, f3.z, a.z); a.w = fmaf(g3, f3.w, a.w);
            }
            for (; i < nl; ++i) {
                const int rr = rlist[i];
                const float g2 = g[rr];
                const float4 f = *reinterpret_cast<const float4*>(feat + (size_t)rr * C + c);
                a.x = fmaf(g2, f.x, a.x); a.y = fmaf(g2, f.y, a.y); a.z = fmaf(g2, f.z, a.z); a.w = fmaf(g2, f.w, a.w);
            }
            float* d = demb + (size_t)cls * demb_stride + c;
            d[0] += a.x; d[1] += a.y; d[2] += a.z; d[3] += a.w;
        }
    }
}

extern "C" int l2i_proj_head_fwd(const float* x, const void* wl, const void* emb, int emb_stride, const long long* y,
                                 const float* bias, float scale, float* out, float* feat, int R, int HW, int C, int dtype,
                                 void* stream) {
    if (!x || !wl || !out || !feat || (emb && !y) || C % 4 || R < 0 || HW <= 0) return L2I_ERR_ARG;
    if (R == 0) return L2I_OK;
    if (dtype == 1)
        hipLaunchKernelGGL(proj_head_fwd_kernel<bf16_t>, dim3(R), dim3(256), 0, (hipStream_t)stream, x, (const bf16_t*)wl,
                           (const bf16_t*)emb, emb_stride, y, bias, scale, out, feat, HW, C);
    else
        hipLaunchKernelGGL(proj_head_fwd_kernel<float>, dim3(R), dim3(256), 0, (hipStream_t)stream, x, (const float*)wl,
                           (const float*)emb, emb_stride, y, bias, scale, out, feat, HW, C);
    return l2i_check_launch();
}

extern "C" int l2i_proj_head_bwd(const float* x, const void* wl, const void* emb, int emb_stride, const long long* y,
                                 const float* g, const float* feat, float scale, float* dx, float* dwl, float* demb,
                                 int demb_stride, float* dbias, int R, int HW, int C, int dtype, void* dx_op_bf16, void* stream) {
    if (!x || !wl || !g || !feat || !dx || (emb && !y) || (demb && !emb) || C % 4 || R < 0 || HW <= 0 || (demb && R > L2I_CLASS_LIST)) return L2I_ERR_ARG;
    if (R == 0) return L2I_OK;
    const int extra = (dwl || dbias) ? (C + 15) / 16 : 0;
    if (dtype == 1)
        hipLaunchKernelGGL(proj_head_bwd_kernel<bf16_t>, dim3(R + extra), dim3(256), 0, (hipStream_t)stream, x, (const bf16_t*)wl,
                           (const bf16_t*)emb, emb_stride, y, g, feat, scale, dx, dwl, demb, demb_stride, dbias, R, HW, C, (bf16_t*)dx_op_bf16);
    else
        hipLaunchKernelGGL(proj_head_bwd_kernel<float>, dim3(R + extra), dim3(256), 0, (hipStream_t)stream, x, (const float*)wl,
                           (const float*)emb, emb_stride, y, g, feat, scale, dx, dwl, demb, demb_stride, dbias, R, HW, C, (bf16_t*)dx_op_bf16);
    return l2i_check_launch();
}

// The class-embedding term of the appearance head (reference model/rcnn_discriminator_app.py:154-157):
//   out[r] = sum_c E[y[r],c] w2[c] + bias;  bwd: dE[y[r],c] += g[r] w2[c], dw2[c] += sum_r g[r] E[y[r],c], dbias += sum_r g[r].
template <typename T>
__global__ __launch_bounds__(256) void emb_dot_fwd_kernel(const T* __restrict__ emb, int emb_stride, const long long* __restrict__ y,
                                                          const T* __restrict__ w2, const float* __restrict__ bias,
                                                          float* __restrict__ out, int R, int C) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r >= R) return;
    const T* e = emb + (size_t)y[r] * emb_stride;
    float t = 0.f;
    for (int c = lane; c < C; c += 64) t = fmaf(OpT<T>::to(e[c]), OpT<T>::to(w2[c]), t);
    t = wave_sum(t);
    if (lane == 0) out[r] = t + (bias ? bias[0] : 0.f);
}

// blocks [0, R): dE[y[r]][c] += g[r] w2[c];  blocks [R, R + ceil(C/16)): dw2[c] += sum_r g[r] E[y[r]][c]; the first of them dbias.
template <typename T>
__global__ __launch_bounds__(256) void emb_dot_bwd_kernel(const T* __restrict__ emb, int emb_stride, const long long* __restrict__ y,
                                                          const T* __restrict__ w2, const float* __restrict__ g,
                                                          float* __restrict__ demb, int demb_stride, float* __restrict__ dw2,
                                                          float* __restrict__ dbias, int R, int C) {
    __shared__ float red[256];
    if ((int)blockIdx.x < R) {   // dE[class][c] += (sum of g over the class's rows) w2[c]: the first row of a class adds for all of them, in row order
        const int r = blockIdx.x;   // -- one writer per class row, no float atomics (round 6)
        const long long cls = y[r];
        __shared__ int rlist[L2I_CLASS_LIST];
        __shared__ int wsum[4];
        if (!class_first(y, r, cls)) return;
        const int nl = class_rows(y, r, R, cls, rlist, wsum);
        float s = 0.f;
        for (int i = 0; i < nl; ++i) s += g[rlist[i]];
        if (s == 0.f) return;
        float* d = demb + (size_t)cls * demb_stride;
        for (int c = threadIdx.x; c < C; c += 256) d[c] += s * OpT<T>::to(w2[c]);
        return;
    }
    colsum16<T>(emb, emb_stride, y, g, R, C, ((int)blockIdx.x - R) * 16, dw2, red);
    if ((int)blockIdx.x == R && dbias) {
        __syncthreads();
        float s = 0.f;
        for (int r = threadIdx.x; r < R; r += 256) s += g[r];
        s = block_sum(s, red);
        if (threadIdx.x == 0) atomicAdd(dbias, s);
    }
}

extern "C" int l2i_emb_dot_fwd(const void* emb, int emb_stride, const long long* y, const void* w2, const float* bias, float* out,
                               int R, int C, int dtype, void* stream) {
    if (!emb || !y || !w2 || !out || R < 0 || C <= 0) return L2I_ERR_ARG;
    if (R == 0) return L2I_OK;
    if (dtype == 1)
        hipLaunchKernelGGL(emb_dot_fwd_kernel<bf16_t>, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)emb,
                           emb_stride, y, (const bf16_t*)w2, bias, out, R, C);
    else
        hipLaunchKernelGGL(emb_dot_fwd_kernel<float>, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, (const float*)emb,
                           emb_stride, y, (const float*)w2, bias, out, R, C);
    return l2i_check_launch();
}

extern "C" int l2i_emb_dot_bwd(const void* emb, int emb_stride, const long long* y, const void* w2, const float* g, float* demb,
                               int demb_stride, float* dw2, float* dbias, int R, int C, int dtype, void* stream) {
    if (!emb || !y || !w2 || !g || !demb || !dw2 || R < 0 || C <= 0 || R > L2I_CLASS_LIST) return L2I_ERR_ARG;
    if (R == 0) return L2I_OK;
    if (dtype == 1)
        hipLaunchKernelGGL(emb_dot_bwd_kernel<bf16_t>, dim3(R + (C + 15) / 16), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)emb,
                           emb_stride, y, (const bf16_t*)w2, g, demb, demb_stride, dw2, dbias, R, C);
    else
        hipLaunchKernelGGL(emb_dot_bwd_kernel<float>, dim3(R + (C + 15) / 16), dim3(256), 0, (hipStream_t)stream, (const float*)emb,
                           emb_stride, y, (const float*)w2, g, demb, demb_stride, dw2, dbias, R, C);
    return l2i_check_launch();
}

extern "C" int l2i_version(void) { return 1; }
