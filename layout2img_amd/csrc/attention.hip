// Object-context attention core (one workgroup per image), forward and backward.
//
// Replaces box_attention (reference model/resnet_generator_app_v2.py:79-120; VG variant
// model/resnet_generator_vg.py:77-122) for h = 1 head, d = 308:
//   S_ij = q_i . k_j / sqrt(d);  key j with label 0 -> S_ij = -1e9 (masked_fill, :102-103)
//   COCO: L_ij = log(max(geo_ij, 1e-6)) + S_ij (:113)      VG: L_ij = S_ij (vg :115)
//   P = softmax_j(L);  out_i = sum_j P_ij v_j
// q, k, v tiles live in LDS; the row softmax is a wavefront shuffle reduction (o <= 64 keys, one
// lane per key); QK^T and PV are f32 VALU dot products (o x o x 308 per image: latency-bound work).
#include "common.h"

#define AT_MAXO 64

struct AttnArgs {
    const float* q; const float* k; const float* v;  // [B][O][D]
    const float* geo;                                 // [B][O][O] or null
    const int* keyvalid;                              // [B][O] or null
    float* out;                                       // [B][O][D]
    float* prob;                                      // [B][O][O] (saved for backward)
    const float* dout;                                // bwd
    float* dq; float* dk; float* dv; float* dgeo;     // bwd outputs
    int B, O, D;
    int ld, ldd;                                      // row strides (floats) of q / k / v and of dq / dk / dv: D, or the width of
                                                      // the grouped projection's (rows, 3 Dp) result the three are slices of
    float scale;
};

template <bool BWD>
__global__ __launch_bounds__(256) void attn_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int O = p.O, D = p.D, b = blockIdx.x;
    float* qs = reinterpret_cast<float*>(smem);  // [O][D]
    float* ks = qs + O * D;
    float* vs = ks + O * D;
    float* Ps = vs + O * D;       // [O][O]  probabilities
    float* dSs = Ps + O * O;      // [O][O]  (bwd) dS
    float* dos = dSs + O * O;     // [O][D]  (bwd) dout
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t base = (size_t)b * O * D;
    for (int i = tid; i < O * D; i += 256) {
        const int r = i / D, c = i - r * D;
        const size_t src = ((size_t)b * O + r) * p.ld + c;
        qs[i] = p.q[src];
        ks[i] = p.k[src];
        vs[i] = p.v[src];
        if (BWD) dos[i] = p.dout[base + i];
    }
    __syncthreads();
    if (!BWD) {
        // scores: one wave per (i, j) pair
        for (int pr = wave; pr < O * O; pr += 4) {
            const int i = pr / O, j = pr - i * O;
            float d = 0.f;
            for (int t = lane; t < D; t += 64) d += qs[i * D + t] * ks[j * D + t];
            d = wave_sum(d);
            if (lane == 0) {
                float s = d * p.scale;
                if (p.keyvalid && p.keyvalid[b * O + j] == 0) s = -1e9f;
                if (p.geo) s += logf(fmaxf(p.geo[((size_t)b * O + i) * O + j], 1e-6f));
                Ps[pr] = s;
            }
        }
        __syncthreads();
        // softmax: one wave per row, one lane per key
        for (int i = wave; i < O; i += 4) {
            const float s = lane < O ? Ps[i * O + lane] : -INFINITY;
            const float m = wave_max(s);
            const float e = lane < O ? expf(s - m) : 0.f;
            const float z = wave_sum(e);
            if (lane < O) {
                const float pv = e / z;
                Ps[i * O + lane] = pv;
                p.prob[((size_t)b * O + i) * O + lane] = pv;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < O * D; idx += 256) {
            const int i = idx / D, t = idx - i * D;
            float a = 0.f;
            for (int j = 0; j < O; ++j) a += Ps[i * O + j] * vs[j * D + t];
            p.out[base + idx] = a;
        }
    } else {
        for (int i = tid; i < O * O; i += 256) Ps[i] = p.prob[(size_t)b * O * O + i];
        __syncthreads();
        // dP_ij = dout_i . v_j ; dS_ij = P_ij (dP_ij - sum_j' P_ij' dP_ij')
        for (int pr = wave; pr < O * O; pr += 4) {
            const int i = pr / O, j = pr - i * O;
            float d = 0.f;
            for (int t = lane; t < D; t += 64) d += dos[i * D + t] * vs[j * D + t];
            d = wave_sum(d);
            if (lane == 0) dSs[pr] = d;
        }
        __syncthreads();
        for (int i = wave; i < O; i += 4) {
            const float pv = lane < O ? Ps[i * O + lane] : 0.f;
            const float dp = lane < O ? dSs[i * O + lane] : 0.f;
            const float dotv = wave_sum(pv * dp);
            if (lane < O) {
                const float ds = pv * (dp - dotv);
                dSs[i * O + lane] = ds;
                if (p.dgeo) {
                    const float g = p.geo[((size_t)b * O + i) * O + lane];
                    p.dgeo[((size_t)b * O + i) * O + lane] = g >= 1e-6f ? ds / g : 0.f;
                }
            }
        }
        __syncthreads();
        for (int idx = tid; idx < O * D; idx += 256) {
            const int i = idx / D, t = idx - i * D;
            float aq = 0.f, ak = 0.f, av = 0.f;
            for (int j = 0; j < O; ++j) {
                aq += dSs[i * O + j] * ks[j * D + t];   // dq_i = scale * sum_j dS_ij k_j
                ak += dSs[j * O + i] * qs[j * D + t];   // dk_i = scale * sum_j dS_ji q_j
                av += Ps[j * O + i] * dos[j * D + t];   // dv_i = sum_j P_ji dout_j
            }
            const size_t dst = ((size_t)b * O + i) * p.ldd + t;
            p.dq[dst] = aq * p.scale;
            p.dk[dst] = ak * p.scale;
            p.dv[dst] = av;
        }
    }
}

static size_t attn_lds(int O, int D, bool bwd) {
    return sizeof(float) * ((size_t)(bwd ? 4 : 3) * O * D + 2 * O * O);
}

extern "C" int l2i_box_attention_fwd(const float* q, const float* k, const float* v, const float* geo, const int* keyvalid,
                                     float* out, float* prob, int B, int O, int D, int ld, float scale, void* stream) {
    if (!q || !k || !v || !out || !prob || O < 1 || O > AT_MAXO || B < 1 || ld < D) return L2I_ERR_ARG;
    if (attn_lds(O, D, false) > 160 * 1024) return L2I_ERR_ARG;
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.geo = geo; a.keyvalid = keyvalid; a.out = out; a.prob = prob;
    a.B = B; a.O = O; a.D = D; a.ld = ld; a.ldd = D; a.scale = scale;
    const size_t lds = attn_lds(O, D, false);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)attn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return L2I_ERR_LAUNCH;
    hipLaunchKernelGGL(attn_kernel<false>, dim3(B), dim3(256), lds, (hipStream_t)stream, a);
    return l2i_check_launch();
}

extern "C" int l2i_box_attention_bwd(const float* q, const float* k, const float* v, const float* geo, const float* prob,
                                     const float* dout, float* dq, float* dk, float* dv, float* dgeo, int B, int O, int D,
                                     int ld, int ldd, float scale, void* stream) {
    if (!q || !k || !v || !prob || !dout || !dq || !dk || !dv || O < 1 || O > AT_MAXO || B < 1 || ld < D || ldd < D) return L2I_ERR_ARG;
    if (dgeo && !geo) return L2I_ERR_ARG;
    if (attn_lds(O, D, true) > 160 * 1024) return L2I_ERR_ARG;
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v; a.geo = geo; a.prob = const_cast<float*>(prob); a.dout = dout;
    a.dq = dq; a.dk = dk; a.dv = dv; a.dgeo = dgeo; a.B = B; a.O = O; a.D = D; a.ld = ld; a.ldd = ldd; a.scale = scale;
    const size_t lds = attn_lds(O, D, true);
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void*)attn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return L2I_ERR_LAUNCH;
    hipLaunchKernelGGL(attn_kernel<true>, dim3(B), dim3(256), lds, (hipStream_t)stream, a);
    return l2i_check_launch();
}
