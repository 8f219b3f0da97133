// Layout-side glue of the generator as a handful of kernels (gfx950): everything here is small, latency-bound work
// over (batch x objects) that the reference expresses as chains of 10-35 elementwise torch ops each -- ~140 stock
// launches per generator forward and as many in its backward (profiles/r03_base_gfwd_census.txt), a quarter of the
// forward's time. One kernel per reference function instead:
//
//   l2i_box_geometry_*   BoxRelationalEmbedding + WGs Linear(64,1) + ReLU   (model/resnet_generator_app_v2.py:17-76,175-180)
//   l2i_layout_masks_*   sigmoid + masks_to_layout (grid_sample) + bbox_mask (utils/bilinear.py:137-192, app_v2.py:697-721)
//   l2i_add_layernorm_*  residual add (+ the h = 1 "concat heads" shuffle, :197-198) + nn.LayerNorm (:199-214)
//   l2i_latent_*         label embedding lookup + concat with z             (app_v2.py:437-441)
//   l2i_fc_to_nhwc_*     .view(N, C, 4, 4) of a Linear output as the NHWC stream / operand the convolutions read (:453, mask_regression.py:87)
//   l2i_tanh_nchw_*      tanh + NHWC -> NCHW of the to-RGB result            (:497-499)
//   l2i_psp_stages_*     PSP pyramid stages: 1x1 conv + BatchNorm2d + ReLU on the pooled bins (:741-746)
#include "common.h"

// ------------------------------------------------------------------------------------------------ box geometry
// geo[b,i,j] = relu(wg . emb(b,i,j) + bias); emb = [sin(100 pos_c dim_k), cos(...)], pos = (log|dcx / w_i|, log|dcy / h_i|,
// log(w_i / w_j), log(h_i / h_j)) with the xywh boxes read as corner boxes (as the reference does), c-major / k-minor.
__device__ __forceinline__ void geo_pos(const float* __restrict__ bbox, int b, int O, int i, int j, float (&pos)[4]) {
    const float* bi = bbox + ((size_t)b * O + i) * 4;
    const float* bj = bbox + ((size_t)b * O + j) * 4;
    const float cxi = (bi[0] + bi[2]) * 0.5f, cyi = (bi[1] + bi[3]) * 0.5f, wi = (bi[2] - bi[0]) + 1.0f, hi = (bi[3] - bi[1]) + 1.0f;
    const float cxj = (bj[0] + bj[2]) * 0.5f, cyj = (bj[1] + bj[3]) * 0.5f, wj = (bj[2] - bj[0]) + 1.0f, hj = (bj[3] - bj[1]) + 1.0f;
    pos[0] = logf(fmaxf(fabsf((cxi - cxj) / wi), 1e-3f));
    pos[1] = logf(fmaxf(fabsf((cyi - cyj) / hi), 1e-3f));
    pos[2] = logf(wi / wj);
    pos[3] = logf(hi / hj);
}

__global__ __launch_bounds__(256) void box_geometry_fwd_kernel(const float* __restrict__ bbox, const float* __restrict__ dim_mat,
                                                               const float* __restrict__ wg, const float* __restrict__ wg_bias,
                                                               float* __restrict__ geo, int B, int O) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * O * O) return;
    const int j = idx % O, i = (idx / O) % O, b = idx / (O * O);
    float pos[4];
    geo_pos(bbox, b, O, i, j, pos);
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float mul = (100.0f * pos[c]) * dim_mat[k];
            float sn, cs;
            sincosf(mul, &sn, &cs);
            acc = fmaf(wg[c * 8 + k], sn, acc);
            acc = fmaf(wg[32 + c * 8 + k], cs, acc);
        }
    geo[idx] = fmaxf(acc + wg_bias[0], 0.f);
}

// dwg[f] += sum over pairs with geo > 0 of dgeo * emb_f (block f < 64), dbias += sum dgeo (block 64)
__global__ __launch_bounds__(256) void box_geometry_bwd_kernel(const float* __restrict__ bbox, const float* __restrict__ dim_mat,
                                                               const float* __restrict__ geo, const float* __restrict__ dgeo,
                                                               float* __restrict__ dwg, float* __restrict__ dbias, int B, int O) {
    __shared__ float red[16];
    const int f = blockIdx.x;
    float acc = 0.f;
    for (int idx = threadIdx.x; idx < B * O * O; idx += 256) {
        if (!(geo[idx] > 0.f)) continue;
        const float g = dgeo[idx];
        if (f == 64) { acc += g; continue; }
        const int j = idx % O, i = (idx / O) % O, b = idx / (O * O);
        float pos[4];
        geo_pos(bbox, b, O, i, j, pos);
        const int fc = f & 31, c = fc >> 3, k = fc & 7;
        const float mul = (100.0f * pos[c]) * dim_mat[k];
        acc = fmaf(g, f < 32 ? sinf(mul) : cosf(mul), acc);
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) {
        if (f == 64) dbias[0] += acc;
        else dwg[f] += acc;
    }
}

extern "C" int l2i_box_geometry_fwd(const float* bbox, const float* dim_mat, const float* wg, const float* wg_bias, float* geo, int B,
                                    int O, void* stream) {
    if (!bbox || !dim_mat || !wg || !wg_bias || !geo || B < 1 || O < 1) return L2I_ERR_ARG;
    const int n = B * O * O;
    hipLaunchKernelGGL(box_geometry_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, bbox, dim_mat, wg, wg_bias, geo, B, O);
    return l2i_check_launch();
}
extern "C" int l2i_box_geometry_bwd(const float* bbox, const float* dim_mat, const float* geo, const float* dgeo, float* dwg,
                                    float* dbias, int B, int O, void* stream) {
    if (!bbox || !dim_mat || !geo || !dgeo || !dwg || !dbias) return L2I_ERR_ARG;
    hipLaunchKernelGGL(box_geometry_bwd_kernel, dim3(65), dim3(256), 0, (hipStream_t)stream, bbox, dim_mat, geo, dgeo, dwg, dbias, B, O);
    return l2i_check_launch();
}

// ------------------------------------------------------------------------------------------------ layout masks
// One workgroup per object (b, o): bmask = grid_sample(sigmoid(m), box-relative grid) (bilinear, zeros outside,
// align_corners = False) and boxm = the hard rectangle indicator, both H x H. `lin` = torch.linspace(0, 1, H) as torch
// computes it (passed in so that the sample positions are bit-identical to the reference's).
// m: [N][M][M] logits with element stride m_stride (channel 0 of a padded NHWC conv result).
#define LM_MAXM 32
__global__ __launch_bounds__(256) void layout_masks_fwd_kernel(const float* __restrict__ m, int m_stride, const float* __restrict__ bbox,
                                                               const float* __restrict__ lin, float* __restrict__ bmask,
                                                               float* __restrict__ boxm, int M, int H) {
    __shared__ float sm[LM_MAXM * LM_MAXM];
    const int n = blockIdx.x;
    for (int i = threadIdx.x; i < M * M; i += 256) sm[i] = 1.0f / (1.0f + expf(-m[((size_t)n * M * M + i) * m_stride]));
    __syncthreads();
    const float x0 = bbox[n * 4], y0 = bbox[n * 4 + 1], ww = bbox[n * 4 + 2], hh = bbox[n * 4 + 3];
    for (int p = threadIdx.x; p < H * H; p += 256) {
        const int y = p / H, x = p - y * H;
        const float X = (lin[x] - x0) / ww, Y = (lin[y] - y0) / hh;
        if (boxm) boxm[(size_t)n * H * H + p] = (X < 0.f || X > 1.f || Y < 0.f || Y > 1.f) ? 0.f : 1.f;
        const float gx = X * 2.f - 1.f, gy = Y * 2.f - 1.f;
        const float ix = ((gx + 1.f) * M - 1.f) / 2.f, iy = ((gy + 1.f) * M - 1.f) / 2.f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x_w = (int)fx, y_n = (int)fy, x_e = x_w + 1, y_s = y_n + 1;
        const float nw = ((fx + 1.f) - ix) * ((fy + 1.f) - iy), ne = (ix - fx) * ((fy + 1.f) - iy);
        const float sw = ((fx + 1.f) - ix) * (iy - fy), se = (ix - fx) * (iy - fy);
        // (NaN / inf coordinates of degenerate boxes: the int casts are garbage then, the range tests reject them)
        const bool vx_w = ix == ix && fx >= 0.f && fx <= (float)(M - 1), vx_e = ix == ix && fx >= -1.f && fx <= (float)(M - 2);
        const bool vy_n = iy == iy && fy >= 0.f && fy <= (float)(M - 1), vy_s = iy == iy && fy >= -1.f && fy <= (float)(M - 2);
        float o = 0.f;
        if (vx_w && vy_n) o += sm[y_n * M + x_w] * nw;
        if (vx_e && vy_n) o += sm[y_n * M + x_e] * ne;
        if (vx_w && vy_s) o += sm[y_s * M + x_w] * sw;
        if (vx_e && vy_s) o += sm[y_s * M + x_e] * se;
        bmask[(size_t)n * H * H + p] = o;
    }
}

// dm[n][i] (element stride d_stride; the other d_stride - 1 floats of each group are written as zeros) =
// sigmoid'(m) * sum over pixels of g * bilinear weight
__global__ __launch_bounds__(256) void layout_masks_bwd_kernel(const float* __restrict__ m, int m_stride, const float* __restrict__ bbox,
                                                               const float* __restrict__ lin, const float* __restrict__ g,
                                                               float* __restrict__ dm, int d_stride, int M, int H) {
    // The adjoint of the bilinear sampling is SEPARABLE: pixel (y, x) feeds cells (y_n | y_s, x_w | x_e) with weights wy * wx. Round 6: a
    // gather in two fixed-order passes -- T[y][cx] = sum_x wx(x, cx) g[y][x], acc[cy][cx] = sum_y wy(y, cy) T[y][cx] -- every value has one
    // writer (rounds 2-5 scattered four LDS float atomics per pixel, whose order over the four waves changed the sums' last bits from run to run).
    extern __shared__ float lm_smem[];
    float* acc = lm_smem;                 // [M][M]
    float* T = acc + M * M;               // [H][M]
    float* wl = T + H * M;                // [2 axes][H]: weight of the low cell (x_w / y_n), 0 when it is outside
    float* wh = wl + 2 * H;               // [2 axes][H]: weight of the high cell
    int* cl = reinterpret_cast<int*>(wh + 2 * H);   // [2 axes][H]: index of the low cell (may be -1)
    const int n = blockIdx.x;
    const float x0 = bbox[n * 4], y0 = bbox[n * 4 + 1], ww = bbox[n * 4 + 2], hh = bbox[n * 4 + 3];
    for (int i = threadIdx.x; i < 2 * H; i += 256) {
        const int ax = i / H, q = i - ax * H;   // axis 0: x, 1: y
        const float U = (lin[q] - (ax ? y0 : x0)) / (ax ? hh : ww);
        const float gq = U * 2.f - 1.f;
        const float iq = ((gq + 1.f) * M - 1.f) / 2.f;
        const float fq = floorf(iq);
        const bool v_lo = iq == iq && fq >= 0.f && fq <= (float)(M - 1), v_hi = iq == iq && fq >= -1.f && fq <= (float)(M - 2);
        wl[i] = v_lo ? (fq + 1.f) - iq : 0.f;
        wh[i] = v_hi ? iq - fq : 0.f;
        cl[i] = (v_lo || v_hi) ? (int)fq : -4;
    }
    __syncthreads();
    // pixel range [lo, hi) per axis and cell: the low-cell index is monotone along an axis, so the pixels that reach cell c are contiguous
    int* lo = cl + 2 * H;   // [2 axes][M]
    int* hi = lo + 2 * M;
    for (int i = threadIdx.x; i < 2 * M; i += 256) {
        const int ax = i / M, c = i - ax * M;
        int l = H, h = 0;
        for (int q = 0; q < H; ++q) {
            const int k = cl[ax * H + q];
            if (k == c || k + 1 == c) { l = min(l, q); h = max(h, q + 1); }
        }
        lo[i] = l; hi[i] = h;
    }
    __syncthreads();
    const float* gp = g + (size_t)n * H * H;
    for (int i = threadIdx.x; i < H * M; i += 256) {
        const int y = i / M, cx = i - y * M;
        float t = 0.f;
        for (int x = lo[cx]; x < hi[cx]; ++x) {
            const int c = cl[x];
            const float w = (c == cx ? wl[x] : 0.f) + (c + 1 == cx ? wh[x] : 0.f);
            t = fmaf(w, gp[y * H + x], t);
        }
        T[i] = t;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < M * M; i += 256) {
        const int cy = i / M, cx = i - cy * M;
        float a = 0.f;
        for (int y = lo[M + cy]; y < hi[M + cy]; ++y) {
            const int c = cl[H + y];
            const float w = (c == cy ? wl[H + y] : 0.f) + (c + 1 == cy ? wh[H + y] : 0.f);
            a = fmaf(w, T[y * M + cx], a);
        }
        acc[i] = a;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < M * M; i += 256) {
        const float s = 1.0f / (1.0f + expf(-m[((size_t)n * M * M + i) * m_stride]));
        float* d = dm + ((size_t)n * M * M + i) * d_stride;
        d[0] = acc[i] * s * (1.0f - s);
        for (int k = 1; k < d_stride; ++k) d[k] = 0.f;
    }
}

extern "C" int l2i_layout_masks_fwd(const float* m, int m_stride, const float* bbox, const float* lin, float* bmask, float* boxm, int N,
                                    int M, int H, void* stream) {
    if (!m || !bbox || !lin || !bmask || N < 1 || M < 1 || M > LM_MAXM || H < 1 || m_stride < 1) return L2I_ERR_ARG;
    hipLaunchKernelGGL(layout_masks_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, m, m_stride, bbox, lin, bmask, boxm, M, H);
    return l2i_check_launch();
}
extern "C" int l2i_layout_masks_bwd(const float* m, int m_stride, const float* bbox, const float* lin, const float* g, float* dm,
                                    int d_stride, int N, int M, int H, void* stream) {
    if (!m || !bbox || !lin || !g || !dm || N < 1 || M < 1 || M > LM_MAXM || H < 1 || m_stride < 1 || d_stride < 1) return L2I_ERR_ARG;
    const size_t lds = sizeof(float) * ((size_t)M * M + (size_t)H * M + 6 * (size_t)H + 4 * (size_t)M);
    if (lds > 64 * 1024) return L2I_ERR_ARG;
    hipLaunchKernelGGL(layout_masks_bwd_kernel, dim3(N), dim3(256), lds, (hipStream_t)stream, m, m_stride, bbox, lin, g, dm, d_stride, M, H);
    return l2i_check_launch();
}

// ------------------------------------------------------------------------------------------------ add + LayerNorm
// y[r, :D] = LayerNorm(a'[r] + b[r]) * gamma + beta (eps inside the sqrt, biased variance: nn.LayerNorm), y[r, D:ldy] = 0.
// a' = a, or (perm_O > 0) the reference's h = 1 "concat heads" shuffle of a per-image (O, D) matrix:
// a'[img, r, c] = a[img, (r D + c) % O, (r D + c) / O]  (x.transpose(1, 2).contiguous().view(B, -1, D), app_v2.py:197-198).
// One wave per row. Optional operand-dtype copy of y (what the next Linear reads).
struct AlnArgs {
    const float* a; const float* b; const float* gamma; const float* beta;
    float* y; void* y_op; float* mean; float* rstd;
    const float* dy; float* da; float* db; float* dgamma; float* dbeta;
    int rows, D, lda, ldb, ldy, perm_O, op_dtype;
    float eps;
    float* gpart; int Dp;   // bwd: per-workgroup rows of dgamma / dbeta in the caller's scratch (null: one float atomic per channel and workgroup)
};
__device__ __forceinline__ float aln_a(const AlnArgs& p, int r, int c) {
    if (p.perm_O <= 0) return p.a[(size_t)r * p.lda + c];
    const int img = r / p.perm_O, rr = r - img * p.perm_O;
    const int idx = rr * p.D + c;
    return p.a[((size_t)img * p.perm_O + idx % p.perm_O) * p.lda + idx / p.perm_O];
}
#define ALN_MAXE 8   // elements per lane: D <= 512
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(AlnArgs p) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.rows) return;
    float s[ALN_MAXE];
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < ALN_MAXE; ++e) {
        const int c = lane + 64 * e;
        s[e] = c < p.D ? aln_a(p, r, c) + p.b[(size_t)r * p.ldb + c] : 0.f;
        sum += s[e];
    }
    const float mean = wave_sum(sum) / p.D;
    float sq = 0.f;
#pragma unroll
    for (int e = 0; e < ALN_MAXE; ++e) {
        const int c = lane + 64 * e;
        const float d = c < p.D ? s[e] - mean : 0.f;
        sq = fmaf(d, d, sq);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / p.D + p.eps);
    if (lane == 0) { p.mean[r] = mean; p.rstd[r] = rstd; }
#pragma unroll
    for (int e = 0; e < ALN_MAXE; ++e) {
        const int c = lane + 64 * e;
        if (c >= p.ldy) continue;
        const float v = c < p.D ? (s[e] - mean) * rstd * p.gamma[c] + p.beta[c] : 0.f;
        p.y[(size_t)r * p.ldy + c] = v;
        if (p.y_op) {
            if (p.op_dtype == 1) reinterpret_cast<bf16_t*>(p.y_op)[(size_t)r * p.ldy + c] = f2bf(v);
            else reinterpret_cast<float*>(p.y_op)[(size_t)r * p.ldy + c] = v;
        }
    }
}
// ds = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma; db[r, :D] = ds (pad columns zero); da = ds through the inverse
// shuffle (in a's (rows, lda) layout, pad columns zero); dgamma += sum_r dy xhat, dbeta += sum_r dy (atomics, 4 rows per block)
#define ALN_BWD_RPW 1
__global__ __launch_bounds__(256) void add_layernorm_bwd_kernel(AlnArgs p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dg[ALN_MAXE], dbt[ALN_MAXE];
#pragma unroll
    for (int e = 0; e < ALN_MAXE; ++e) { dg[e] = 0.f; dbt[e] = 0.f; }
    for (int q = 0; q < ALN_BWD_RPW; ++q) {   // (one row per wave: four rows in turn were four dependent round trips, 23 us for 256 rows on 16 workgroups)
        const int r = blockIdx.x * (4 * ALN_BWD_RPW) + wave * ALN_BWD_RPW + q;
        if (r >= p.rows) break;
        const float mean = p.mean[r], rstd = p.rstd[r];
        float xh[ALN_MAXE], g[ALN_MAXE];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < ALN_MAXE; ++e) {
            const int c = lane + 64 * e;
            if (c < p.D) {
                xh[e] = (aln_a(p, r, c) + p.b[(size_t)r * p.ldb + c] - mean) * rstd;
                const float dyv = p.dy[(size_t)r * p.ldy + c];
                g[e] = dyv * p.gamma[c];
                dg[e] = fmaf(dyv, xh[e], dg[e]);
                dbt[e] += dyv;
            } else { xh[e] = 0.f; g[e] = 0.f; }
            s1 += g[e];
            s2 = fmaf(g[e], xh[e], s2);
        }
        s1 = wave_sum(s1) / p.D;
        s2 = wave_sum(s2) / p.D;
#pragma unroll
        for (int e = 0; e < ALN_MAXE; ++e) {
            const int c = lane + 64 * e;
            if (c < p.ldb && p.db) p.db[(size_t)r * p.ldb + c] = c < p.D ? rstd * (g[e] - s1 - xh[e] * s2) : 0.f;
            if (p.da && p.perm_O <= 0 && c >= p.D && c < p.lda) p.da[(size_t)r * p.lda + c] = 0.f;   // pad columns of a padded Linear result
            if (c < p.D && p.da) {
                const float ds = rstd * (g[e] - s1 - xh[e] * s2);
                if (p.perm_O <= 0) p.da[(size_t)r * p.lda + c] = ds;
                else {
                    const int img = r / p.perm_O, rr = r - img * p.perm_O;
                    const int idx = rr * p.D + c;
                    p.da[((size_t)img * p.perm_O + idx % p.perm_O) * p.lda + idx / p.perm_O] = ds;
                }
            }
        }
    }
    // combine the four waves' partial dgamma / dbeta in LDS, one atomic per channel and block
    __shared__ float red[2][4][64 * ALN_MAXE];
#pragma unroll
    for (int e = 0; e < ALN_MAXE; ++e) { red[0][wave][lane + 64 * e] = dg[e]; red[1][wave][lane + 64 * e] = dbt[e]; }
    __syncthreads();
    if (p.gpart) {   // this workgroup's row [dgamma (Dp) | dbeta (Dp)] of the partial matrix: stored, added in order by rows_fold (round 6)
        float* row = p.gpart + (size_t)blockIdx.x * 2 * p.Dp;
        for (int c = threadIdx.x; c < p.Dp; c += 256) {
            row[c] = c < p.D ? (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]) : 0.f;
            row[p.Dp + c] = c < p.D ? (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]) : 0.f;
        }
        return;
    }
    for (int c = threadIdx.x; c < p.D; c += 256) {
        atomicAdd(p.dgamma + c, (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]));
        atomicAdd(p.dbeta + c, (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]));
    }
}
extern "C" int l2i_add_layernorm_fwd(const float* a, int lda, const float* b, int ldb, const float* gamma, const float* beta, float eps,
                                     float* y, int ldy, void* y_op, int op_dtype, float* mean, float* rstd, int rows, int D, int perm_O,
                                     void* stream) {
    if (!a || !b || !gamma || !beta || !y || !mean || !rstd || rows < 1 || D < 1 || D > 64 * ALN_MAXE || ldy < D || ldy > 64 * ALN_MAXE ||
        lda < D || ldb < D || (perm_O > 0 && rows % perm_O))
        return L2I_ERR_ARG;
    AlnArgs p = {};
    p.a = a; p.b = b; p.gamma = gamma; p.beta = beta; p.y = y; p.y_op = y_op; p.mean = mean; p.rstd = rstd;
    p.rows = rows; p.D = D; p.lda = lda; p.ldb = ldb; p.ldy = ldy; p.perm_O = perm_O; p.op_dtype = op_dtype; p.eps = eps;
    hipLaunchKernelGGL(add_layernorm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    return l2i_check_launch();
}
extern "C" int l2i_add_layernorm_bwd(const float* a, int lda, const float* b, int ldb, const float* gamma, const float* mean,
                                     const float* rstd, const float* dy, int ldy, float* da, float* db, float* dgamma, float* dbeta,
                                     int rows, int D, int perm_O, float* scratch, long long scratch_floats, void* stream) {
    if (!a || !b || !gamma || !mean || !rstd || !dy || !dgamma || !dbeta || rows < 1 || D < 1 || D > 64 * ALN_MAXE || ldb > 64 * ALN_MAXE)
        return L2I_ERR_ARG;
    AlnArgs p = {};
    p.a = a; p.b = b; p.gamma = gamma; p.mean = const_cast<float*>(mean); p.rstd = const_cast<float*>(rstd); p.dy = dy;
    p.da = da; p.db = db; p.dgamma = dgamma; p.dbeta = dbeta;
    p.rows = rows; p.D = D; p.lda = lda; p.ldb = ldb; p.ldy = ldy; p.perm_O = perm_O;
    const int nblk = (rows + 4 * ALN_BWD_RPW - 1) / (4 * ALN_BWD_RPW);
    p.Dp = (D + 3) & ~3;
    p.gpart = (scratch && !((size_t)scratch & 15) && (long long)nblk * 2 * p.Dp + rows_fold_tmp_floats(nblk, 2 * p.Dp, 1) <= scratch_floats) ? scratch : nullptr;
    hipLaunchKernelGGL(add_layernorm_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, p);
    if (p.gpart) {   // columns [0, D) -> dgamma, [Dp, Dp + D) -> dbeta: two segments of one fold
        if (nblk <= L2I_FOLD_DIRECT) {
            const RowsFoldArgs f = rows_fold_args4(p.gpart, nblk, 2 * p.Dp, 1, dgamma, dbeta, nullptr, nullptr, p.Dp, D, 0, 1, nullptr, 2 * p.Dp);
            hipLaunchKernelGGL(rows_fold2_kernel, dim3(f.nbx, 1, 1), dim3(256), 0, (hipStream_t)stream, f);
        } else {
            float* tmp = p.gpart + (size_t)nblk * 2 * p.Dp;
            rows_fold(p.gpart, nblk, p.Dp, 1, dgamma, nullptr, D, 0, 1, tmp, (hipStream_t)stream, nullptr, 2 * p.Dp);
            rows_fold(p.gpart + p.Dp, nblk, p.Dp, 1, dbeta, nullptr, D, 0, 1, tmp, (hipStream_t)stream, nullptr, 2 * p.Dp);
        }
    }
    return l2i_check_launch();
}

// ------------------------------------------------------------------------------------------------ latent
// out[r] = [z[r] (Z floats) | emb[y[r]] (E floats) | zeros up to ld]; optional operand-dtype copy.
__global__ __launch_bounds__(256) void latent_fwd_kernel(const float* __restrict__ z, const float* __restrict__ emb,
                                                         const long long* __restrict__ y, float* __restrict__ out, void* out_op,
                                                         int op_dtype, int* __restrict__ keyvalid, int rows, int Z, int E, int ld) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)rows * ld) return;
    const int r = (int)(idx / ld), c = (int)(idx - (long long)r * ld);
    if (keyvalid && c == 0) keyvalid[r] = y[r] != 0 ? 1 : 0;   // the attention's key mask (label 0 = padding slot)
    const float v = c < Z ? z[(size_t)r * Z + c] : (c < Z + E ? emb[(size_t)y[r] * E + (c - Z)] : 0.f);
    out[idx] = v;
    if (out_op) {
        if (op_dtype == 1) reinterpret_cast<bf16_t*>(out_op)[idx] = f2bf(v);
        else reinterpret_cast<float*>(out_op)[idx] = v;
    }
}
__global__ __launch_bounds__(256) void latent_bwd_kernel(const float* __restrict__ g, const long long* __restrict__ y,
                                                         float* __restrict__ demb, int rows, int Z, int E, int ld) {
    // dE[class] += the embedding part of the gradient rows that carry the class: the workgroup of the FIRST such row adds them all in row order
    // and is the only writer of dE[class] (round 6; one float atomic per row and column before: the order changed from run to run)
    __shared__ int rlist[L2I_CLASS_LIST];
    __shared__ int wsum[4];
    const int r = blockIdx.x;
    const long long cls = y[r];
    if (!class_first(y, r, cls)) return;
    const int nl = class_rows(y, r, rows, cls, rlist, wsum);
    for (int c = threadIdx.x; c < E; c += 256) {
        float s = 0.f;
        for (int i = 0; i < nl; ++i) s += g[(size_t)rlist[i] * ld + Z + c];
        demb[(size_t)cls * E + c] += s;
    }
}
extern "C" int l2i_latent_fwd(const float* z, const float* emb, const long long* y, float* out, void* out_op, int op_dtype, int* keyvalid,
                              int rows, int Z, int E, int ld, void* stream) {
    if (!z || !emb || !y || !out || rows < 1 || ld < Z + E) return L2I_ERR_ARG;
    const long long n = (long long)rows * ld;
    hipLaunchKernelGGL(latent_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, z, emb, y, out, out_op,
                       op_dtype, keyvalid, rows, Z, E, ld);
    return l2i_check_launch();
}
extern "C" int l2i_latent_bwd(const float* g, const long long* y, float* demb, int rows, int Z, int E, int ld, void* stream) {
    if (!g || !y || !demb || rows < 1 || rows > L2I_CLASS_LIST) return L2I_ERR_ARG;
    hipLaunchKernelGGL(latent_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, g, y, demb, rows, Z, E, ld);
    return l2i_check_launch();
}

// ------------------------------------------------------------------------------------------------ Linear output -> NHWC
// in [N][C * P] (a Linear's rows, (c, p) order = .view(N, C, 4, 4)) -> out [N][P][C] f32 and / or operand dtype.
// bwd: g [N][P][C] f32 -> din [N][C * P] f32 (+ operand-dtype copy: the Linear's weight-gradient operand).
__global__ __launch_bounds__(256) void fc_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, void* out_op, int op_dtype,
                                                         long long total, int C, int P, int inverse) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    // idx enumerates the DESTINATION: forward (n, p, c), inverse (n, c, p)
    const long long n = idx / ((long long)C * P);
    const int rem = (int)(idx - n * C * P);
    int src;
    if (!inverse) { const int pp = rem / C, c = rem - pp * C; src = c * P + pp; }
    else { const int c = rem / P, pp = rem - c * P; src = pp * C + c; }
    const float v = in[n * C * P + src];
    if (out) out[idx] = v;
    if (out_op) {
        if (op_dtype == 1) reinterpret_cast<bf16_t*>(out_op)[idx] = f2bf(v);
        else reinterpret_cast<float*>(out_op)[idx] = v;
    }
}
extern "C" int l2i_fc_to_nhwc(const float* in, float* out, void* out_op, int op_dtype, long long N, int C, int P, int inverse, void* stream) {
    if (!in || (!out && !out_op) || N < 1 || C < 1 || P < 1) return L2I_ERR_ARG;
    const long long total = N * C * P;
    hipLaunchKernelGGL(fc_to_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, out_op, op_dtype,
                       total, C, P, inverse);
    return l2i_check_launch();
}

// ------------------------------------------------------------------------------------------------ tanh + NHWC -> NCHW
// img[b][c][p] = tanh(pre[b][p][c]) for c < C (pre has Cp >= C channels per pixel).
// bwd: dpre[b][p][c] = (1 - img^2) dimg for c < C, 0 for the pad channels (+ operand-dtype copy for the to-RGB conv's backward).
__global__ __launch_bounds__(256) void tanh_nchw_fwd_kernel(const float* __restrict__ pre, float* __restrict__ img, long long total, int C,
                                                            int Cp, int HW) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // destination (b, c, p)
    if (idx >= total) return;
    const int pp = (int)(idx % HW);
    const long long bc = idx / HW;
    const int c = (int)(bc % C);
    const long long b = bc / C;
    img[idx] = tanhf(pre[(b * HW + pp) * Cp + c]);
}
__global__ __launch_bounds__(256) void tanh_nchw_bwd_kernel(const float* __restrict__ img, const float* __restrict__ g, float* __restrict__ dpre,
                                                            void* dpre_op, int op_dtype, long long total, int C, int Cp, int HW) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // destination (b, p, c) over Cp channels
    if (idx >= total) return;
    const int c = (int)(idx % Cp);
    const long long bp = idx / Cp;
    const int pp = (int)(bp % HW);
    const long long b = bp / HW;
    float v = 0.f;
    if (c < C) {
        const long long s = (b * C + c) * HW + pp;
        const float t = img[s];
        v = (1.0f - t * t) * g[s];
    }
    dpre[idx] = v;
    if (dpre_op) {
        if (op_dtype == 1) reinterpret_cast<bf16_t*>(dpre_op)[idx] = f2bf(v);
        else reinterpret_cast<float*>(dpre_op)[idx] = v;
    }
}
extern "C" int l2i_tanh_nchw_fwd(const float* pre, float* img, long long B, int C, int Cp, int HW, void* stream) {
    if (!pre || !img || B < 1 || C < 1 || Cp < C || HW < 1) return L2I_ERR_ARG;
    const long long total = B * C * HW;
    hipLaunchKernelGGL(tanh_nchw_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pre, img, total, C, Cp, HW);
    return l2i_check_launch();
}
extern "C" int l2i_tanh_nchw_bwd(const float* img, const float* g, float* dpre, void* dpre_op, int op_dtype, long long B, int C, int Cp,
                                 int HW, void* stream) {
    if (!img || !g || !dpre || B < 1 || C < 1 || Cp < C || HW < 1) return L2I_ERR_ARG;
    const long long total = B * Cp * HW;
    hipLaunchKernelGGL(tanh_nchw_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, g, dpre, dpre_op,
                       op_dtype, total, C, Cp, HW);
    return l2i_check_launch();
}

// ------------------------------------------------------------------------------------------------ PSP pyramid stages
// For each stage s (bins k in [off_s, off_s + n_s) of every image; rows = B n_s):
//   raw[b,k,f] = sum_c pooled[b,k,c] W_s[f,c];  train: (mean, var) over the stage's rows per f, running stats updated
//   (momentum, unbiased variance) -- eval: the running statistics;  y = relu((raw - mean) rstd gamma + beta).
// One workgroup per (stage, 4 output channels).
// W[s]: [F][C], gamma / beta / running_*[s]: [F] (one pointer per stage: the modules' own parameters); keeps raw [B][NB][F]
// and (mean, rstd) [S][2][F] for the backward.
#define PSP_FT 4
struct PspStArgs {
    const float* pooled; const float* W[8]; const float* gamma[8]; const float* beta[8]; float* rmean[8]; float* rvar[8];   // per stage: [F][C] / [F]
    float* raw; float* y; float* stat;
    const float* dy; float* draw; float* dgamma; float* dbeta;
    int B, NB, C, F, S, training;
    int off[8], ns[8];
    float eps, momentum;
};
__global__ __launch_bounds__(1024) void psp_stages_fwd_kernel(PspStArgs p) {
    __shared__ float ws[PSP_FT][512];
    __shared__ float red[16];
    // (the raw values go to p.raw and are read back by the SAME thread in the later passes: no size limit, no fence needed)
    const int s = blockIdx.y, f0 = blockIdx.x * PSP_FT, tid = threadIdx.x;
    const int ns = p.ns[s], off = p.off[s], rows = p.B * ns;
    for (int i = tid; i < PSP_FT * p.C; i += blockDim.x) {
        const int ff = i / p.C, c = i - ff * p.C;
        ws[ff][c] = f0 + ff < p.F ? p.W[s][(size_t)(f0 + ff) * p.C + c] : 0.f;
    }
    __syncthreads();
    float sum[PSP_FT] = {0.f, 0.f, 0.f, 0.f};
    for (int r = tid; r < rows; r += blockDim.x) {
        const int b = r / ns, k = r - b * ns;
        const float* x = p.pooled + ((size_t)b * p.NB + off + k) * p.C;
        float a[PSP_FT] = {0.f, 0.f, 0.f, 0.f};
        int c = 0;
        for (; c + 32 <= p.C; c += 32) {   // eight 16-byte loads of the row in flight (one at a time: 32 dependent round trips per row, 40 us)
            float4 xq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) xq[u] = *reinterpret_cast<const float4*>(x + c + 4 * u);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int ff = 0; ff < PSP_FT; ++ff)
                    a[ff] = fmaf(xq[u].w, ws[ff][c + 4 * u + 3], fmaf(xq[u].z, ws[ff][c + 4 * u + 2], fmaf(xq[u].y, ws[ff][c + 4 * u + 1], fmaf(xq[u].x, ws[ff][c + 4 * u], a[ff]))));
        }
        for (; c < p.C; c += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(x + c);
#pragma unroll
            for (int ff = 0; ff < PSP_FT; ++ff)
                a[ff] = fmaf(xv.w, ws[ff][c + 3], fmaf(xv.z, ws[ff][c + 2], fmaf(xv.y, ws[ff][c + 1], fmaf(xv.x, ws[ff][c], a[ff]))));
        }
        float* rw = p.raw + ((size_t)b * p.NB + off + k) * p.F + f0;
#pragma unroll
        for (int ff = 0; ff < PSP_FT; ++ff) {
            if (f0 + ff < p.F) rw[ff] = a[ff];
            sum[ff] += a[ff];
        }
    }
    float mean[PSP_FT], rstd[PSP_FT];
    if (p.training) {
#pragma unroll
        for (int ff = 0; ff < PSP_FT; ++ff) mean[ff] = block_sum(sum[ff], red) / rows;
        float sq[PSP_FT] = {0.f, 0.f, 0.f, 0.f};
        for (int r = tid; r < rows; r += blockDim.x) {
            const int b = r / ns, k = r - b * ns;
            const float* rw = p.raw + ((size_t)b * p.NB + off + k) * p.F + f0;
#pragma unroll
            for (int ff = 0; ff < PSP_FT; ++ff)
                if (f0 + ff < p.F) { const float d = rw[ff] - mean[ff]; sq[ff] = fmaf(d, d, sq[ff]); }
        }
#pragma unroll
        for (int ff = 0; ff < PSP_FT; ++ff) {
            const float var = block_sum(sq[ff], red) / rows;
            rstd[ff] = 1.0f / sqrtf(var + p.eps);
            if (tid == 0 && f0 + ff < p.F) {
                const int o = f0 + ff;
                p.rmean[s][o] = (1.f - p.momentum) * p.rmean[s][o] + p.momentum * mean[ff];
                p.rvar[s][o] = (1.f - p.momentum) * p.rvar[s][o] + p.momentum * (rows > 1 ? var * rows / (rows - 1) : var);
            }
        }
    } else {
        __syncthreads();
#pragma unroll
        for (int ff = 0; ff < PSP_FT; ++ff) {
            const int o = min(f0 + ff, p.F - 1);
            mean[ff] = p.rmean[s][o];
            rstd[ff] = 1.0f / sqrtf(p.rvar[s][o] + p.eps);
        }
    }
    if (tid == 0)
#pragma unroll
        for (int ff = 0; ff < PSP_FT; ++ff)
            if (f0 + ff < p.F) { p.stat[((size_t)s * 2) * p.F + f0 + ff] = mean[ff]; p.stat[((size_t)s * 2 + 1) * p.F + f0 + ff] = rstd[ff]; }
    for (int r = tid; r < rows; r += blockDim.x) {
        const int b = r / ns, k = r - b * ns;
        const size_t o = ((size_t)b * p.NB + off + k) * p.F + f0;
#pragma unroll
        for (int ff = 0; ff < PSP_FT; ++ff)
            if (f0 + ff < p.F) {
                const float v = p.raw[o + ff];
                p.y[o + ff] = fmaxf((v - mean[ff]) * rstd[ff] * p.gamma[s][f0 + ff] + p.beta[s][f0 + ff], 0.f);
            }
    }
}
// BatchNorm + ReLU backward per (stage, 4 channels): draw = gamma rstd (g - mean(g) - xhat mean(g xhat)), g = dy 1[y > 0];
// dgamma / dbeta written (each (stage, channel) belongs to one workgroup).
__global__ __launch_bounds__(1024) void psp_stages_bwd_bn_kernel(PspStArgs p) {
    __shared__ float red[16];
    const int s = blockIdx.y, f0 = blockIdx.x * PSP_FT, tid = threadIdx.x;
    const int ns = p.ns[s], off = p.off[s], rows = p.B * ns;
    float mean[PSP_FT], rstd[PSP_FT], gam[PSP_FT], bet[PSP_FT];
#pragma unroll
    for (int ff = 0; ff < PSP_FT; ++ff) {
        const int f = min(f0 + ff, p.F - 1);
        mean[ff] = p.stat[((size_t)s * 2) * p.F + f];
        rstd[ff] = p.stat[((size_t)s * 2 + 1) * p.F + f];
        gam[ff] = p.gamma[s][f];
        bet[ff] = p.beta[s][f];
    }
    float s1[PSP_FT] = {0.f, 0.f, 0.f, 0.f}, s2[PSP_FT] = {0.f, 0.f, 0.f, 0.f};
    for (int r = tid; r < rows; r += blockDim.x) {
        const int b = r / ns, k = r - b * ns;
        const size_t o = ((size_t)b * p.NB + off + k) * p.F + f0;
#pragma unroll
        for (int ff = 0; ff < PSP_FT; ++ff)
            if (f0 + ff < p.F) {
                const float xh = (p.raw[o + ff] - mean[ff]) * rstd[ff];
                const float g = xh * gam[ff] + bet[ff] > 0.f ? p.dy[o + ff] : 0.f;
                s1[ff] += g;
                s2[ff] = fmaf(g, xh, s2[ff]);
            }
    }
#pragma unroll
    for (int ff = 0; ff < PSP_FT; ++ff) {
        s1[ff] = block_sum(s1[ff], red);
        s2[ff] = block_sum(s2[ff], red);
        if (tid == 0 && f0 + ff < p.F) { p.dgamma[(size_t)s * p.F + f0 + ff] = s2[ff]; p.dbeta[(size_t)s * p.F + f0 + ff] = s1[ff]; }
    }
    for (int r = tid; r < rows; r += blockDim.x) {
        const int b = r / ns, k = r - b * ns;
        const size_t o = ((size_t)b * p.NB + off + k) * p.F + f0;
#pragma unroll
        for (int ff = 0; ff < PSP_FT; ++ff)
            if (f0 + ff < p.F) {
                const float xh = (p.raw[o + ff] - mean[ff]) * rstd[ff];
                const float g = xh * gam[ff] + bet[ff] > 0.f ? p.dy[o + ff] : 0.f;
                p.draw[o + ff] = p.training ? gam[ff] * rstd[ff] * (g - s1[ff] / rows - xh * s2[ff] / rows) : gam[ff] * rstd[ff] * g;
            }
    }
}
// dpooled[b,k,c] = sum_f draw[b,k,f] W_s[f,c] (blockIdx.y == 0: one thread per (row, 4 channels)) and
// dW_s[f,c] = sum_rows draw[.,f] pooled[.,c] (blockIdx.y == 1: one thread per (s, f, c))
__global__ __launch_bounds__(256) void psp_stages_bwd_mm_kernel(PspStArgs p, float* __restrict__ dpooled, float* __restrict__ dW) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.y == 0) {
        const int C4 = p.C / 4;
        if (idx >= (long long)p.B * p.NB * C4) return;
        const int c = (int)(idx % C4) * 4;
        const long long row = idx / C4;   // b * NB + bin
        const int bin = (int)(row % p.NB);
        const float* w = p.W[0] + c;
#pragma unroll
        for (int k = 1; k < 8; ++k)
            if (k < p.S && bin >= p.off[k]) w = p.W[k] + c;   // (static indices: no scratch copy of the argument arrays)
        const float* __restrict__ d = p.draw + row * p.F;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int f = 0; f < p.F; ++f) {
            const float dv = d[f];
            const float4 wv = *reinterpret_cast<const float4*>(w + (size_t)f * p.C);
            a.x = fmaf(dv, wv.x, a.x); a.y = fmaf(dv, wv.y, a.y); a.z = fmaf(dv, wv.z, a.z); a.w = fmaf(dv, wv.w, a.w);
        }
        *reinterpret_cast<float4*>(dpooled + row * p.C + c) = a;
    } else {
        // 16 (stage, f, c) outputs per workgroup, the stage's rows split over 16 slices of 16 lanes (a serial walk over the up to
        // 1152 rows is one dependent load round trip per row: measured 390 us; four slices of a wave each: 54 us) and combined in LDS
        __shared__ float red[16][16];
        const int l16 = threadIdx.x & 15, slice = threadIdx.x >> 4;
        const long long o = (long long)blockIdx.x * 16 + l16;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const bool on = o < (long long)p.S * p.F * p.C;
        if (on) {
            const int c = (int)(o % p.C);
            const int f = (int)((o / p.C) % p.F), s = (int)(o / ((long long)p.C * p.F));
            int ns = 0, off = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k == s) { ns = p.ns[k]; off = p.off[k]; }   // (static indices: no scratch copy of the argument arrays)
            const int rows = p.B * ns;
            const float* dr = p.draw + f;
            const float* pl = p.pooled + c;
            int r = slice;
            for (; r + 48 < rows; r += 64) {
                const int r1 = r + 16, r2 = r + 32, r3 = r + 48;
                const size_t q0 = (size_t)(r / ns) * p.NB + off + r % ns, q1 = (size_t)(r1 / ns) * p.NB + off + r1 % ns;
                const size_t q2 = (size_t)(r2 / ns) * p.NB + off + r2 % ns, q3 = (size_t)(r3 / ns) * p.NB + off + r3 % ns;
                const float d0 = dr[q0 * p.F], d1 = dr[q1 * p.F], d2 = dr[q2 * p.F], d3 = dr[q3 * p.F];
                const float x0 = pl[q0 * p.C], x1 = pl[q1 * p.C], x2 = pl[q2 * p.C], x3 = pl[q3 * p.C];
                a0 = fmaf(d0, x0, a0); a1 = fmaf(d1, x1, a1); a2 = fmaf(d2, x2, a2); a3 = fmaf(d3, x3, a3);
            }
            for (; r < rows; r += 16) {
                const size_t q0 = (size_t)(r / ns) * p.NB + off + r % ns;
                a0 = fmaf(dr[q0 * p.F], pl[q0 * p.C], a0);
            }
        }
        red[slice][l16] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (slice == 0 && on) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k][l16];
            dW[o] = t;
        }
    }
}
static int psp_fill(PspStArgs& p, int B, int NB, int C, int F, int S, const int* sizes) {
    if (S < 1 || S > 8 || C % 4 || C > 512) return L2I_ERR_ARG;
    int off = 0;
    for (int s = 0; s < S; ++s) {
        p.off[s] = off; p.ns[s] = sizes[s] * sizes[s];
        off += p.ns[s];
    }
    if (off != NB) return L2I_ERR_ARG;
    p.B = B; p.NB = NB; p.C = C; p.F = F; p.S = S;
    return L2I_OK;
}
extern "C" int l2i_psp_stages_fwd(const float* pooled, const float* const* W, const float* const* gamma, const float* const* beta,
                                  float* const* rmean, float* const* rvar, float* raw, float* y, float* stat, int B, int NB, int C, int F,
                                  int S, const int* sizes, int training, float eps, float momentum, void* stream) {
    if (!pooled || !W || !gamma || !beta || !rmean || !rvar || !raw || !y || !stat || !sizes) return L2I_ERR_ARG;
    PspStArgs p = {};
    if (psp_fill(p, B, NB, C, F, S, sizes) != L2I_OK) return L2I_ERR_ARG;
    for (int s = 0; s < S; ++s) {
        if (!W[s] || !gamma[s] || !beta[s] || !rmean[s] || !rvar[s]) return L2I_ERR_ARG;
        p.W[s] = W[s]; p.gamma[s] = gamma[s]; p.beta[s] = beta[s]; p.rmean[s] = rmean[s]; p.rvar[s] = rvar[s];
    }
    p.pooled = pooled; p.raw = raw; p.y = y; p.stat = stat;
    p.training = training; p.eps = eps; p.momentum = momentum;
    // (1024 threads: the largest stage has B x 36 rows and its batch statistics need all of them in one workgroup)
    hipLaunchKernelGGL(psp_stages_fwd_kernel, dim3((F + PSP_FT - 1) / PSP_FT, S), dim3(1024), 0, (hipStream_t)stream, p);
    return l2i_check_launch();
}
extern "C" int l2i_psp_stages_bwd(const float* pooled, const float* const* W, const float* const* gamma, const float* const* beta,
                                  const float* raw, const float* stat, const float* dy, float* draw, float* dpooled, float* dW, float* dgamma,
                                  float* dbeta, int B, int NB, int C, int F, int S, const int* sizes, int training, void* stream) {
    if (!pooled || !W || !gamma || !beta || !raw || !stat || !dy || !draw || !dpooled || !dW || !dgamma || !dbeta || !sizes) return L2I_ERR_ARG;
    PspStArgs p = {};
    if (psp_fill(p, B, NB, C, F, S, sizes) != L2I_OK) return L2I_ERR_ARG;
    for (int s = 0; s < S; ++s) {
        if (!W[s] || !gamma[s] || !beta[s]) return L2I_ERR_ARG;
        p.W[s] = W[s]; p.gamma[s] = gamma[s]; p.beta[s] = beta[s];
    }
    p.pooled = pooled; p.raw = const_cast<float*>(raw); p.stat = const_cast<float*>(stat);
    p.dy = dy; p.draw = draw; p.dgamma = dgamma; p.dbeta = dbeta; p.training = training;
    hipLaunchKernelGGL(psp_stages_bwd_bn_kernel, dim3((F + PSP_FT - 1) / PSP_FT, S), dim3(1024), 0, (hipStream_t)stream, p);
    const long long n0 = ((long long)B * NB * (C / 4) + 255) / 256, n1 = ((long long)S * F * C + 15) / 16;   // workgroups of the two roles
    const long long nmax = n0 > n1 ? n0 : n1;
    hipLaunchKernelGGL(psp_stages_bwd_mm_kernel, dim3((unsigned)nmax, 2), dim3(256), 0, (hipStream_t)stream, p, dpooled, dW);
    return l2i_check_launch();
}

// ------------------------------------------------------------------------------------------------ ROI layout of the discriminator
// CombineDiscriminator128_app.forward's box bookkeeping (model/rcnn_discriminator_app.py:402-417) + the ordering of
// ResnetDiscriminator128_app.forward (:131-146) without the host-synchronising nonzero(): the R = b*o rows are COMPACTED
// on the device -- real ROIs first in the reference's output order (large ROIs, then small ones, original order within
// each), padding rows (label 0) behind them -- i.e. a stable sort by key = 2 [label == 0] + [small] (two_scale) that the
// host side used to do with ~26 torch launches incl. a radix sort. One workgroup of 1024 threads walks the rows in chunks of
// 1024 (any R: Visual Genome layouts at a per-GPU batch above 33 have more than 1024 rows): a first sweep counts the rows
// of each key, a second one places them (chunk order = original order within a key).
__global__ __launch_bounds__(1024) void roi_layout_kernel(const float* __restrict__ bbox, const long long* __restrict__ label, float size,
                                                          int two_scale, int o, int R, float* __restrict__ rois, long long* __restrict__ y,
                                                          int* __restrict__ valid, int* __restrict__ count) {
    __shared__ int wtot[4][16];
    __shared__ int total[4], base[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    auto row_key = [&](int i, float (&r)[5], long long& lab) {
        const float* bb = bbox + (size_t)i * 4;
        r[0] = (float)(i / o);
        r[1] = bb[0] * size; r[2] = bb[1] * size; r[3] = (bb[0] + bb[2]) * size; r[4] = (bb[1] + bb[3]) * size;
        lab = label[i];
        return (lab != 0 ? 0 : 2) + ((two_scale && (r[3] - r[1]) < 64.f && (r[4] - r[2]) < 64.f) ? 1 : 0);
    };
    if (t < 4) { total[t] = 0; base[t] = 0; }
    __syncthreads();
    if (R > 1024) {   // (one chunk: the totals come out of the placement sweep's own ballots)
        int mine[4] = {0, 0, 0, 0};
        for (int i = t; i < R; i += 1024) {
            float r[5];
            long long lab;
            ++mine[row_key(i, r, lab)];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int v = mine[k];
            for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
            if (lane == 0 && v) atomicAdd(&total[k], v);
        }
        __syncthreads();
    }
    for (int c0 = 0; c0 < R; c0 += 1024) {
        const int i = c0 + t;
        float r[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        long long lab = 0;
        const int key = i < R ? row_key(i, r, lab) : -1;
        int before[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long m = __ballot(key == k);
            before[k] = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wtot[k][wave] = __popcll(m);
        }
        __syncthreads();
        if (R <= 1024 && t < 4) {
            int tot = 0;
            for (int w_ = 0; w_ < 16; ++w_) tot += wtot[t][w_];
            total[t] = tot;
        }
        if (R <= 1024) __syncthreads();
        if (i < R) {
            int pos = base[key] + before[key];
            for (int w_ = 0; w_ < wave; ++w_) pos += wtot[key][w_];
            for (int k = 0; k < key; ++k) pos += total[k];
            float* d = rois + (size_t)pos * 5;
            d[0] = r[0]; d[1] = r[1]; d[2] = r[2]; d[3] = r[3]; d[4] = r[4];
            y[pos] = lab;
            valid[pos] = lab != 0 ? 1 : 0;
        }
        __syncthreads();
        if (t < 4) {
            int tot = 0;
            for (int w_ = 0; w_ < 16; ++w_) tot += wtot[t][w_];
            base[t] += tot;
        }
        __syncthreads();
    }
    if (t == 0) count[0] = total[0] + total[1];
}
extern "C" int l2i_roi_layout(const float* bbox, const long long* label, float size, int two_scale, int o, int R, float* rois, long long* y,
                              int* valid, int* count, void* stream) {
    if (!bbox || !label || !rois || !y || !valid || !count || R < 1 || o < 1) return L2I_ERR_ARG;
    hipLaunchKernelGGL(roi_layout_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, bbox, label, size, two_scale, o, R, rois, y, valid, count);
    return l2i_check_launch();
}

// ------------------------------------------------------------------------------------------------ discriminator input image
// img [B][C][H][W] f32 -> x [B][H][W][Cp] (channels C..Cp zero: the padded NHWC stream the first block reads) and, optionally,
// its 2x2 average xs [B][H/2][W/2][Cp] (OptimizedBlock pools the shortcut BEFORE its 1x1 conv, rcnn_discriminator_app.py:311-314),
// each as f32 + operand copy. One thread per (b, y/2, x/2): the 2x2 quad of every channel.
// bwd: dimg[b][c][y][x] = dx[b][y][x][c] + 0.25 dxs[b][y/2][x/2][c] (either may be NULL).
__global__ __launch_bounds__(256) void image_nhwc_fwd_kernel(const float* __restrict__ img, float* __restrict__ x, void* x_op, float* __restrict__ xs,
                                                             void* xs_op, int op_dtype, long long nquad, int C, int Cp, int H, int W) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nquad) return;
    const int W2 = W >> 1, H2 = H >> 1;
    const int x2 = (int)(q % W2), y2 = (int)((q / W2) % H2);
    const long long b = q / ((long long)W2 * H2);
    if (Cp == 8) {   // the padded RGB image: whole pixels (32 bytes f32 / 16 bytes bf16) per store instead of 4-byte scatters
        float v[4][8];   // [pixel of the quad][channel]
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float2 t0 = make_float2(0.f, 0.f), t1 = t0;
            if (c < C) {
                const float* s = img + ((b * C + c) * H + 2 * y2) * W + 2 * x2;
                t0 = *reinterpret_cast<const float2*>(s);
                t1 = *reinterpret_cast<const float2*>(s + W);
            }
            v[0][c] = t0.x; v[1][c] = t0.y; v[2][c] = t1.x; v[3][c] = t1.y;
        }
        float a[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) a[c] = ((v[0][c] + v[1][c]) + (v[2][c] + v[3][c])) * 0.25f;
        auto put = [&](float* f32, void* op, long long o, const float (&w)[8]) {
            *reinterpret_cast<float4*>(f32 + o) = make_float4(w[0], w[1], w[2], w[3]);
            *reinterpret_cast<float4*>(f32 + o + 4) = make_float4(w[4], w[5], w[6], w[7]);
            if (op) {
                if (op_dtype == 1) {
                    uint4 pk;
                    pk.x = f2bf2(w[0], w[1]); pk.y = f2bf2(w[2], w[3]); pk.z = f2bf2(w[4], w[5]); pk.w = f2bf2(w[6], w[7]);
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(op) + o) = pk;
                } else {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(op) + o) = make_float4(w[0], w[1], w[2], w[3]);
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(op) + o + 4) = make_float4(w[4], w[5], w[6], w[7]);
                }
            }
        };
#pragma unroll
        for (int k = 0; k < 4; ++k) put(x, x_op, ((b * H + 2 * y2 + (k >> 1)) * W + 2 * x2 + (k & 1)) * 8, v[k]);
        if (xs) put(xs, xs_op, ((b * H2 + y2) * W2 + x2) * 8, a);
        return;
    }
    for (int c = 0; c < Cp; ++c) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < C) {
            const float* s = img + ((b * C + c) * H + 2 * y2) * W + 2 * x2;
            v[0] = s[0]; v[1] = s[1]; v[2] = s[W]; v[3] = s[W + 1];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long long o = ((b * H + 2 * y2 + (k >> 1)) * W + 2 * x2 + (k & 1)) * Cp + c;
            x[o] = v[k];
            if (x_op) {
                if (op_dtype == 1) reinterpret_cast<bf16_t*>(x_op)[o] = f2bf(v[k]);
                else reinterpret_cast<float*>(x_op)[o] = v[k];
            }
        }
        if (xs) {
            const float a = ((v[0] + v[1]) + (v[2] + v[3])) * 0.25f;
            const long long o = ((b * H2 + y2) * W2 + x2) * Cp + c;
            xs[o] = a;
            if (xs_op) {
                if (op_dtype == 1) reinterpret_cast<bf16_t*>(xs_op)[o] = f2bf(a);
                else reinterpret_cast<float*>(xs_op)[o] = a;
            }
        }
    }
}
__global__ __launch_bounds__(256) void image_nhwc_bwd_kernel(const float* __restrict__ dx, const float* __restrict__ dxs, float* __restrict__ dimg,
                                                             long long total, int C, int Cp, int H, int W) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // destination (b, c, y, x)
    if (idx >= total) return;
    const int xx = (int)(idx % W), yy = (int)((idx / W) % H);
    const int c = (int)((idx / ((long long)W * H)) % C);
    const long long b = idx / ((long long)W * H * C);
    float v = 0.f;
    if (dx) v = dx[((b * H + yy) * W + xx) * Cp + c];
    if (dxs) v += 0.25f * dxs[((b * (H >> 1) + (yy >> 1)) * (W >> 1) + (xx >> 1)) * Cp + c];
    dimg[idx] = v;
}
extern "C" int l2i_image_nhwc_fwd(const float* img, float* x, void* x_op, float* xs, void* xs_op, int op_dtype, long long B, int C, int Cp,
                                  int H, int W, void* stream) {
    if (!img || !x || B < 1 || C < 1 || Cp < C || (H & 1) || (W & 1)) return L2I_ERR_ARG;
    const long long nquad = B * (H >> 1) * (W >> 1);
    hipLaunchKernelGGL(image_nhwc_fwd_kernel, dim3((unsigned)((nquad + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img, x, x_op, xs, xs_op,
                       op_dtype, nquad, C, Cp, H, W);
    return l2i_check_launch();
}
extern "C" int l2i_image_nhwc_bwd(const float* dx, const float* dxs, float* dimg, long long B, int C, int Cp, int H, int W, void* stream) {
    if ((!dx && !dxs) || !dimg || B < 1) return L2I_ERR_ARG;
    const long long total = B * C * H * W;
    hipLaunchKernelGGL(image_nhwc_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dx, dxs, dimg, total, C, Cp,
                       H, W);
    return l2i_check_launch();
}

// ------------------------------------------------------------------------------------------------ adjoint of the bilinear resize
// dx [N][h][w] = sum over the output pixels of g [N][H][W] times their bilinear (align_corners = False) weights: the
// transpose of l2i_resize_bilinear (F.interpolate of the object masks), in GATHER form -- one thread per input pixel walks
// the few output rows / columns whose two taps can reach it and evaluates the forward's own tap formula (no atomics:
// the scatter form serialised 16 output pixels on every address when downsampling).
__device__ __forceinline__ void rb_taps(int Y, float scale, int h, int& y0, int& y1, float& ly) {
    const float fy = fmaxf(((float)Y + 0.5f) * scale - 0.5f, 0.f);
    y0 = min((int)fy, h - 1);
    y1 = min(y0 + 1, h - 1);
    ly = fy - (float)y0;
}
__global__ __launch_bounds__(256) void resize_bilinear_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, long long total, int h, int w,
                                                                  int H, int W) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // input pixel (n, y, x)
    if (idx >= total) return;
    const int x = (int)(idx % w), y = (int)((idx / w) % h);
    const long long n = idx / ((long long)w * h);
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    // output rows Y whose taps {y0, y0 + 1} can contain y: (Y + 0.5) sy - 0.5 in (y - 1, y + 1)
    const int Ylo = max(0, (int)floorf(((float)y - 0.5f) / sy - 0.5f) - 1), Yhi = min(H - 1, (int)ceilf(((float)y + 1.5f) / sy - 0.5f) + 1);
    const int Xlo = max(0, (int)floorf(((float)x - 0.5f) / sx - 0.5f) - 1), Xhi = min(W - 1, (int)ceilf(((float)x + 1.5f) / sx - 0.5f) + 1);
    const float* gp = g + n * H * W;
    float acc = 0.f;
    for (int Y = Ylo; Y <= Yhi; ++Y) {
        int y0, y1; float ly;
        rb_taps(Y, sy, h, y0, y1, ly);
        const float wy = (y0 == y ? 1.f - ly : 0.f) + (y1 == y ? ly : 0.f);
        if (wy == 0.f) continue;
        for (int X = Xlo; X <= Xhi; ++X) {
            int x0, x1; float lx;
            rb_taps(X, sx, w, x0, x1, lx);
            const float wx = (x0 == x ? 1.f - lx : 0.f) + (x1 == x ? lx : 0.f);
            acc = fmaf(gp[Y * W + X], wy * wx, acc);
        }
    }
    dx[idx] = acc;
}
extern "C" int l2i_resize_bilinear_bwd(const float* g, float* dx, long long N, int h, int w, int H, int W, void* stream) {
    if (!g || !dx || N < 1) return L2I_ERR_ARG;
    const long long total = N * h * w;
    hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, dx, total, h, w, H, W);
    return l2i_check_launch();
}

// ------------------------------------------------------------------------------------------------ Dropout2d scale
// out[b][p][c] = in[b][p][c] * (u[b][c] >= prob ? 1 / (1 - prob) : 0): nn.Dropout2d (whole channels per sample) given the
// uniform draws u (model/resnet_generator_app_v2.py:739); the same launch is its backward (in = dy).
__global__ __launch_bounds__(256) void channel_dropout_kernel(const float* __restrict__ in, const float* __restrict__ u, float* __restrict__ out,
                                                              long long total4, int C, int HW, float prob) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int C4 = C >> 2;
    const int c = (int)(idx % C4) * 4;
    const long long b = idx / ((long long)C4 * HW);
    const float4 v = reinterpret_cast<const float4*>(in)[idx];
    const float4 uu = *reinterpret_cast<const float4*>(u + b * C + c);
    const float k = 1.0f / (1.0f - prob);
    float4 o;
    o.x = uu.x >= prob ? v.x * k : 0.f; o.y = uu.y >= prob ? v.y * k : 0.f;
    o.z = uu.z >= prob ? v.z * k : 0.f; o.w = uu.w >= prob ? v.w * k : 0.f;
    reinterpret_cast<float4*>(out)[idx] = o;
}
extern "C" int l2i_channel_dropout(const float* in, const float* u, float* out, long long B, int HW, int C, float prob, void* stream) {
    if (!in || !u || !out || B < 1 || (C & 3) || prob < 0.f || prob >= 1.f) return L2I_ERR_ARG;
    const long long total4 = B * HW * (C >> 2);
    hipLaunchKernelGGL(channel_dropout_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, u, out, total4, C, HW, prob);
    return l2i_check_launch();
}

// ------------------------------------------------------------------------------------------------ mask regression: IN + ReLU + bilinear x2
// The step between two convolutions of the mask regressor (reference model/mask_regression.py:64-95): InstanceNorm2d (no affine,
// biased variance, eps) -> ReLU -> F.interpolate(scale_factor=2, mode="bilinear", align_corners=False) on the per-object maps
// x [N][S][S][C] (N = b*o objects, S = 4 or 8). It ran as channel_stats + normalise + a batched GEMM with the resampling matrix +
// the operand cast (and their four backward counterparts); here one thread owns one (object, channel) plane in registers:
// statistics, normalisation, ReLU and the separable 2-tap interpolation, written once as the f32 stream and once as the
// operand copy the next convolution reads. Backward: the adjoint of the interpolation row by row, the ReLU gate and the
// instance-norm backward dx = istd (g - mean(g) - xhat mean(g xhat)), from x alone (nothing else is saved).
__device__ __forceinline__ void up2_tap(int o, int S, int& i0, int& i1, float& l) {   // align_corners = False, scale 2
    const float src = fmaxf(((float)o + 0.5f) * 0.5f - 0.5f, 0.f);
    i0 = (int)src;
    i1 = min(i0 + 1, S - 1);
    l = src - (float)i0;
}
template <int S>
__global__ __launch_bounds__(256) void in_relu_up2_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, void* out_op, int op_dtype,
                                                             int C, float eps) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const long long n = blockIdx.x;
    const float* xp = x + n * S * S * C + c;
    float v[S * S];
#pragma unroll
    for (int i = 0; i < S * S; ++i) v[i] = xp[(long long)i * C];
    float mean = 0.f;
#pragma unroll
    for (int i = 0; i < S * S; ++i) mean += v[i];
    mean *= 1.f / (S * S);
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < S * S; ++i) var = fmaf(v[i] - mean, v[i] - mean, var);
    const float istd = rsqrtf(var * (1.f / (S * S)) + eps);
#pragma unroll
    for (int i = 0; i < S * S; ++i) v[i] = fmaxf((v[i] - mean) * istd, 0.f);
    float* op = out + n * 4 * S * S * C + c;
    // gridDim.z workgroups share an (object, channel chunk): each forms the statistics again (64 values from L2) and writes its share of
    // the 2S output rows -- one workgroup per object was 256 workgroups of four waves for 117 MB of stores (37 us)
    const int rows_per = (2 * S + (int)gridDim.z - 1) / (int)gridDim.z, oy_lo = (int)blockIdx.z * rows_per, oy_hi = min(2 * S, oy_lo + rows_per);
#pragma unroll
    for (int oy = 0; oy < 2 * S; ++oy) {
        if (oy < oy_lo || oy >= oy_hi) continue;
        int y0, y1; float ly;
        up2_tap(oy, S, y0, y1, ly);
        float row[S];
#pragma unroll
        for (int i = 0; i < S; ++i) row[i] = (1.f - ly) * v[y0 * S + i] + ly * v[y1 * S + i];
#pragma unroll
        for (int ox = 0; ox < 2 * S; ++ox) {
            int x0, x1; float lx;
            up2_tap(ox, S, x0, x1, lx);
            const float a = (1.f - lx) * row[x0] + lx * row[x1];
            const long long o = (long long)(oy * 2 * S + ox) * C;
            op[o] = a;
            if (out_op) {
                const long long oo = n * 4 * S * S * C + c + o;
                if (op_dtype == 1) reinterpret_cast<bf16_t*>(out_op)[oo] = f2bf(a);
                else reinterpret_cast<float*>(out_op)[oo] = a;
            }
        }
    }
}
template <int S>
__global__ __launch_bounds__(256) void in_relu_up2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ dx,
                                                             void* dx_op, int op_dtype, int C, float eps) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= C) return;
    const long long n = blockIdx.x;
    const float* xp = x + n * S * S * C + c;
    float v[S * S], d[S * S];
#pragma unroll
    for (int i = 0; i < S * S; ++i) { v[i] = xp[(long long)i * C]; d[i] = 0.f; }
    float mean = 0.f;
#pragma unroll
    for (int i = 0; i < S * S; ++i) mean += v[i];
    mean *= 1.f / (S * S);
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < S * S; ++i) var = fmaf(v[i] - mean, v[i] - mean, var);
    const float istd = rsqrtf(var * (1.f / (S * S)) + eps);
#pragma unroll
    for (int i = 0; i < S * S; ++i) v[i] = (v[i] - mean) * istd;   // xhat
    const float* gp = g + n * 4 * S * S * C + c;
#pragma unroll
    for (int oy = 0; oy < 2 * S; ++oy) {   // adjoint of the interpolation: along x inside the row, then the row into its two source rows
        int y0, y1; float ly;
        up2_tap(oy, S, y0, y1, ly);
        float t[S];
#pragma unroll
        for (int i = 0; i < S; ++i) t[i] = 0.f;
#pragma unroll
        for (int ox = 0; ox < 2 * S; ++ox) {
            int x0, x1; float lx;
            up2_tap(ox, S, x0, x1, lx);
            const float gv = gp[(long long)(oy * 2 * S + ox) * C];
            t[x0] = fmaf(1.f - lx, gv, t[x0]);
            t[x1] = fmaf(lx, gv, t[x1]);
        }
#pragma unroll
        for (int i = 0; i < S; ++i) {
            d[y0 * S + i] = fmaf(1.f - ly, t[i], d[y0 * S + i]);
            d[y1 * S + i] = fmaf(ly, t[i], d[y1 * S + i]);
        }
    }
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int i = 0; i < S * S; ++i) {
        d[i] = v[i] > 0.f ? d[i] : 0.f;   // ReLU gate
        m1 += d[i];
        m2 = fmaf(d[i], v[i], m2);
    }
    m1 *= 1.f / (S * S);
    m2 *= 1.f / (S * S);
    float* dp = dx + n * S * S * C + c;
#pragma unroll
    for (int i = 0; i < S * S; ++i) {
        const float r = istd * (d[i] - m1 - v[i] * m2);
        dp[(long long)i * C] = r;
        if (dx_op) {   // the operand copy the producing convolution's backward reads as its dY
            const long long oo = n * S * S * C + c + (long long)i * C;
            if (op_dtype == 1) reinterpret_cast<bf16_t*>(dx_op)[oo] = f2bf(r);
            else reinterpret_cast<float*>(dx_op)[oo] = r;
        }
    }
}
// Plain bilinear x2 (align_corners = False) of NHWC maps x [N][S][S][C] -> [N][2S][2S][C] + the operand copy the next convolution reads:
// F.interpolate(x, size, mode="bilinear") between the convolutions of the VG generator's MaskRegressNet (reference model/mask_regression.py:
// 20-33,42-58), whose normalisation is a BatchNorm (ops.norm_act) rather than the InstanceNorm the kernel above folds in. It ran as a
// batched torch.matmul with the dense resampling matrix (rocBLAS) through round 5. One thread = four channels of one output pixel.
// Backward (the adjoint): an input pixel gathers from the <= 4 x 4 output pixels whose taps can name it -- every dx value has one writer.
__global__ __launch_bounds__(256) void up2_nhwc_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, void* out_op, int op_dtype,
                                                          long long total4, int S, int C4) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int c4 = (int)(idx % C4);
    long long r = idx / C4;
    const int ox = (int)(r % (2 * S)); r /= 2 * S;
    const int oy = (int)(r % (2 * S));
    const long long n = r / (2 * S);
    int y0, y1, x0, x1; float ly, lx;
    up2_tap(oy, S, y0, y1, ly);
    up2_tap(ox, S, x0, x1, lx);
    const float4* xp = reinterpret_cast<const float4*>(x) + n * S * S * C4 + c4;
    const float4 a = xp[(long long)(y0 * S + x0) * C4], b = xp[(long long)(y0 * S + x1) * C4];
    const float4 c = xp[(long long)(y1 * S + x0) * C4], d = xp[(long long)(y1 * S + x1) * C4];
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
    const float v[4] = {w00 * a.x + w01 * b.x + w10 * c.x + w11 * d.x, w00 * a.y + w01 * b.y + w10 * c.y + w11 * d.y,
                        w00 * a.z + w01 * b.z + w10 * c.z + w11 * d.z, w00 * a.w + w01 * b.w + w10 * c.w + w11 * d.w};
    reinterpret_cast<float4*>(out)[idx] = make_float4(v[0], v[1], v[2], v[3]);
    if (out_op) {
        if (op_dtype == 1) Op4<bf16_t>::store(reinterpret_cast<bf16_t*>(out_op) + 4 * idx, v);
        else Op4<float>::store(reinterpret_cast<float*>(out_op) + 4 * idx, v);
    }
}
__global__ __launch_bounds__(256) void up2_nhwc_bwd_kernel(const float* __restrict__ g, float* __restrict__ dx, void* dx_op, int op_dtype,
                                                          long long total4, int S, int C4) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total4) return;
    const int c4 = (int)(idx % C4);
    long long r = idx / C4;
    const int ix = (int)(r % S); r /= S;
    const int iy = (int)(r % S);
    const long long n = r / S;
    const float4* gp = reinterpret_cast<const float4*>(g) + n * 4 * S * S * C4 + c4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int oy = max(0, 2 * iy - 1); oy <= min(2 * S - 1, 2 * iy + 2); ++oy) {
        int y0, y1; float ly;
        up2_tap(oy, S, y0, y1, ly);
        const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
        if (wy == 0.f) continue;
        for (int ox = max(0, 2 * ix - 1); ox <= min(2 * S - 1, 2 * ix + 2); ++ox) {
            int x0, x1; float lx;
            up2_tap(ox, S, x0, x1, lx);
            const float w = wy * ((x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f));
            if (w == 0.f) continue;
            const float4 v = gp[(long long)(oy * 2 * S + ox) * C4];
            acc[0] = fmaf(w, v.x, acc[0]); acc[1] = fmaf(w, v.y, acc[1]); acc[2] = fmaf(w, v.z, acc[2]); acc[3] = fmaf(w, v.w, acc[3]);
        }
    }
    reinterpret_cast<float4*>(dx)[idx] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (dx_op) {
        if (op_dtype == 1) Op4<bf16_t>::store(reinterpret_cast<bf16_t*>(dx_op) + 4 * idx, acc);
        else Op4<float>::store(reinterpret_cast<float*>(dx_op) + 4 * idx, acc);
    }
}
extern "C" int l2i_up2_nhwc_fwd(const float* x, float* out, void* out_op, int op_dtype, long long N, int S, int C, void* stream) {
    if (!x || !out || N < 1 || S < 1 || C < 4 || C % 4 || (op_dtype != 0 && op_dtype != 1)) return L2I_ERR_ARG;
    const long long total4 = N * 4 * S * S * (C / 4);
    hipLaunchKernelGGL(up2_nhwc_fwd_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, out_op, op_dtype, total4, S, C / 4);
    return l2i_check_launch();
}
extern "C" int l2i_up2_nhwc_bwd(const float* g, float* dx, void* dx_op, int op_dtype, long long N, int S, int C, void* stream) {
    if (!g || !dx || N < 1 || S < 1 || C < 4 || C % 4 || (op_dtype != 0 && op_dtype != 1)) return L2I_ERR_ARG;
    const long long total4 = N * S * S * (C / 4);
    hipLaunchKernelGGL(up2_nhwc_bwd_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, dx, dx_op, op_dtype, total4, S, C / 4);
    return l2i_check_launch();
}

extern "C" int l2i_in_relu_up2_fwd(const float* x, float* out, void* out_op, int op_dtype, long long N, int S, int C, float eps, void* stream) {
    if (!x || !out || N < 1 || C < 1 || (S != 4 && S != 8) || (op_dtype != 0 && op_dtype != 1)) return L2I_ERR_ARG;
    const dim3 grid((unsigned)N, (unsigned)((C + 255) / 256), S == 8 ? 4u : 2u);
    if (S == 4) hipLaunchKernelGGL(in_relu_up2_fwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, x, out, out_op, op_dtype, C, eps);
    else hipLaunchKernelGGL(in_relu_up2_fwd_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, x, out, out_op, op_dtype, C, eps);
    return l2i_check_launch();
}
extern "C" int l2i_in_relu_up2_bwd(const float* x, const float* g, float* dx, void* dx_op, int op_dtype, long long N, int S, int C, float eps,
                                   void* stream) {
    if (!x || !g || !dx || N < 1 || C < 1 || (S != 4 && S != 8) || (op_dtype != 0 && op_dtype != 1)) return L2I_ERR_ARG;
    const dim3 grid((unsigned)N, (unsigned)((C + 255) / 256));
    if (S == 4) hipLaunchKernelGGL(in_relu_up2_bwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, x, g, dx, dx_op, op_dtype, C, eps);
    else hipLaunchKernelGGL(in_relu_up2_bwd_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, x, g, dx, dx_op, op_dtype, C, eps);
    return l2i_check_launch();
}
