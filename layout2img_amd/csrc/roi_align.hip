// ROIAlign forward/backward, NHWC, two-scale selection in one launch.
//
// Replaces torchvision.ops.RoIAlign((8,8), 1/4, 0) / ((8,8), 1/8, 0) as constructed at reference
// model/rcnn_discriminator_app.py:98-99 and called at :139,143, including the small/large routing of
// :131-134 (an ROI whose width AND height are < thr pixels pools from the fine map, otherwise from
// the coarse map). Algorithm restated from torchvision's published roi_align (aligned=False):
//   roi = box * scale; roi_w = max(x2-x1, 1), roi_h likewise; bin = roi/pooled;
//   grid = sampling_ratio > 0 ? sampling_ratio : ceil(roi / pooled);
//   sample (iy, ix) of bin (ph, pw): y = y1 + ph*bin_h + (iy+.5)*bin_h/grid_h, x likewise;
//   y < -1 or y > H (same for x): 0; else clamp to >= 0, low = (int)y, if low >= H-1: low = high = H-1,
//   y = low; bilinear of the 4 neighbours; bin value = mean over samples.
// Rows with valid[r] == 0 (padding slots, label 0) produce zeros and receive no gradient, which lets
// the caller keep a fixed R = b*o instead of the reference's host-synchronising nonzero() (:415).
#include "common.h"

struct RoiArgs {
    const float* feat_s; const float* feat_l;  // [B][Hs][Ws][C], [B][Hl][Wl][C] (feat_l may be null)
    float* dfeat_s; float* dfeat_l;            // backward targets (atomic accumulate)
    const float* rois;                         // [R][5] = batch, x1, y1, x2, y2 (image pixels)
    const int* valid;                          // [R] or null
    float* out;                                // fwd: [R][P][P][C]; bwd: incoming gradient
    bf16_t* out_raw; bf16_t* out_relu;         // fwd, optional: bf16 operand copies of out / relu(out) (what the two ROI heads read)
    int R, C, P, Hs, Ws, Hl, Wl, sampling;
    float scale_s, scale_l, thr;
};

template <bool BWD>
__global__ __launch_bounds__(128) void roi_align_kernel(RoiArgs p) {
    const int bin = blockIdx.x % (p.P * p.P), r = blockIdx.x / (p.P * p.P);
    const int ph = bin / p.P, pw = bin % p.P;
    const int cols4 = p.C >> 2;
    float* o = p.out + ((size_t)r * p.P * p.P + bin) * p.C;
    const bool ok = !p.valid || p.valid[r] != 0;
    if (!ok) {
        if (!BWD)
            for (int c4 = threadIdx.x; c4 < cols4; c4 += 128) {
                reinterpret_cast<float4*>(o)[c4] = make_float4(0, 0, 0, 0);
                const size_t oo = ((size_t)r * p.P * p.P + bin) * p.C + 4 * c4;
                if (p.out_raw) *reinterpret_cast<uint2*>(p.out_raw + oo) = make_uint2(0u, 0u);
                if (p.out_relu) *reinterpret_cast<uint2*>(p.out_relu + oo) = make_uint2(0u, 0u);
            }
        return;
    }
    const float* roi = p.rois + 5 * r;
    const int b = (int)roi[0];
    const bool small = !p.feat_l || ((roi[3] - roi[1]) < p.thr && (roi[4] - roi[2]) < p.thr);
    const float scale = small ? p.scale_s : p.scale_l;
    const int H = small ? p.Hs : p.Hl, W = small ? p.Ws : p.Wl;
    const float* feat = (small ? p.feat_s : p.feat_l) + (size_t)b * H * W * p.C;
    float* dfeat = BWD ? (small ? p.dfeat_s : p.dfeat_l) + (size_t)b * H * W * p.C : nullptr;
    const float x1 = roi[1] * scale, y1 = roi[2] * scale, x2 = roi[3] * scale, y2 = roi[4] * scale;
    const float roi_w = fmaxf(x2 - x1, 1.f), roi_h = fmaxf(y2 - y1, 1.f);
    const float bin_h = roi_h / p.P, bin_w = roi_w / p.P;
    const int gh = p.sampling > 0 ? p.sampling : (int)ceilf(roi_h / p.P);
    const int gw = p.sampling > 0 ? p.sampling : (int)ceilf(roi_w / p.P);
    const float inv_count = 1.f / fmaxf((float)(gh * gw), 1.f);

    for (int c4 = threadIdx.x; c4 < cols4; c4 += 128) {
        float4 acc = make_float4(0, 0, 0, 0);
        float4 g = make_float4(0, 0, 0, 0);
        if (BWD) {
            g = reinterpret_cast<const float4*>(o)[c4];
            g.x *= inv_count; g.y *= inv_count; g.z *= inv_count; g.w *= inv_count;
        }
        for (int iy = 0; iy < gh; ++iy) {
            float y = y1 + ph * bin_h + (iy + 0.5f) * bin_h / gh;
            for (int ix = 0; ix < gw; ++ix) {
                float x = x1 + pw * bin_w + (ix + 0.5f) * bin_w / gw;
                if (y < -1.f || y > (float)H || x < -1.f || x > (float)W) continue;
                float yy = fmaxf(y, 0.f), xx = fmaxf(x, 0.f);
                int yl = (int)yy, xl = (int)xx, yh, xh;
                if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else yh = yl + 1;
                if (xl >= W - 1) { xh = xl = W - 1; xx = (float)xl; } else xh = xl + 1;
                const float ly = yy - yl, lx = xx - xl, hy = 1.f - ly, hx = 1.f - lx;
                const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                const size_t o1 = ((size_t)yl * W + xl) * p.C + 4 * c4, o2 = ((size_t)yl * W + xh) * p.C + 4 * c4;
                const size_t o3 = ((size_t)yh * W + xl) * p.C + 4 * c4, o4 = ((size_t)yh * W + xh) * p.C + 4 * c4;
                if (!BWD) {
                    const float4 v1 = *reinterpret_cast<const float4*>(feat + o1);
                    const float4 v2 = *reinterpret_cast<const float4*>(feat + o2);
                    const float4 v3 = *reinterpret_cast<const float4*>(feat + o3);
                    const float4 v4 = *reinterpret_cast<const float4*>(feat + o4);
                    acc.x += w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x;
                    acc.y += w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
                    acc.z += w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z;
                    acc.w += w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
                } else {
                    const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        atomicAdd(dfeat + o1 + e, w1 * gv[e]);
                        atomicAdd(dfeat + o2 + e, w2 * gv[e]);
                        atomicAdd(dfeat + o3 + e, w3 * gv[e]);
                        atomicAdd(dfeat + o4 + e, w4 * gv[e]);
                    }
                }
            }
        }
        if (!BWD) {
            acc.x *= inv_count; acc.y *= inv_count; acc.z *= inv_count; acc.w *= inv_count;
            reinterpret_cast<float4*>(o)[c4] = acc;
            const size_t oo = ((size_t)r * p.P * p.P + bin) * p.C + 4 * c4;
            if (p.out_raw) *reinterpret_cast<uint2*>(p.out_raw + oo) = make_uint2(f2bf2(acc.x, acc.y), f2bf2(acc.z, acc.w));
            if (p.out_relu)
                *reinterpret_cast<uint2*>(p.out_relu + oo) = make_uint2(f2bf2(fmaxf(acc.x, 0.f), fmaxf(acc.y, 0.f)), f2bf2(fmaxf(acc.z, 0.f), fmaxf(acc.w, 0.f)));
        }
    }
}

static int roi_fill(RoiArgs& a, const float* feat_s, const float* feat_l, const float* rois, const int* valid, int R, int C,
                    int P, int Hs, int Ws, float scale_s, int Hl, int Wl, float scale_l, float thr, int sampling) {
    if (!feat_s || !rois || R < 0 || C % 4 || P <= 0) return L2I_ERR_ARG;
    a.feat_s = feat_s; a.feat_l = feat_l; a.rois = rois; a.valid = valid; a.R = R; a.C = C; a.P = P;
    a.Hs = Hs; a.Ws = Ws; a.scale_s = scale_s; a.Hl = Hl; a.Wl = Wl; a.scale_l = scale_l; a.thr = thr;
    a.sampling = sampling;
    return L2I_OK;
}

extern "C" int l2i_roi_align_fwd(const float* feat_s, const float* feat_l, const float* rois, const int* valid, float* out,
                                 int R, int C, int P, int Hs, int Ws, float scale_s, int Hl, int Wl, float scale_l,
                                 float thr, int sampling, void* out_raw_bf16, void* out_relu_bf16, void* stream) {
    RoiArgs a = {};
    if (roi_fill(a, feat_s, feat_l, rois, valid, R, C, P, Hs, Ws, scale_s, Hl, Wl, scale_l, thr, sampling) || !out)
        return L2I_ERR_ARG;
    a.out = out; a.out_raw = (bf16_t*)out_raw_bf16; a.out_relu = (bf16_t*)out_relu_bf16;
    if (R == 0) return L2I_OK;
    hipLaunchKernelGGL(roi_align_kernel<false>, dim3(R * P * P), dim3(128), 0, (hipStream_t)stream, a);
    return l2i_check_launch();
}

// Backward, separable form: one workgroup per (ROI, 32-channel chunk). The bilinear scatter of ROIAlign is separable
// -- every sample's weight on feature pixel (y, x) is hat_y * hat_x, bins form a regular grid, and a sample is
// dropped iff its y OR its x is outside [-1, size] -- so with
//   Wy[y][ph] = sum over the gh sample rows of bin row ph of their weight on feature row y   (Wx likewise)
// the gradient of the ROI's footprint is  D[y][x][c] = sum_pw Wx[x][pw] * ( sum_ph Wy[y][ph] * G[ph][pw][c] ) / count:
// two small dense products out of LDS, no LDS atomics, and ONE global atomic per touched (pixel, channel). The
// footprint is at most RB_F x RB_F pixels (an ROI spans <= 16 feature pixels on either map by the routing rule);
// larger ones (not produced by this path) scatter sample by sample.
#define RB_F 20
#define RB_C 32
#define RB_P 8

__global__ __launch_bounds__(256) void roi_align_bwd_sep_kernel(RoiArgs p, int chunks) {
    __shared__ float G[RB_P * RB_P * RB_C];   // [bin][c], already / count
    __shared__ float T[RB_F * RB_P * RB_C];   // [y][pw][c]
    __shared__ float Wy[RB_F * RB_P], Wx[RB_F * RB_P];
    const int r = blockIdx.x / chunks, c0 = (blockIdx.x % chunks) * RB_C;
    if (p.valid && p.valid[r] == 0) return;
    const float* roi = p.rois + 5 * r;
    const int b = (int)roi[0];
    const bool small = !p.dfeat_l || ((roi[3] - roi[1]) < p.thr && (roi[4] - roi[2]) < p.thr);
    const float scale = small ? p.scale_s : p.scale_l;
    const int H = small ? p.Hs : p.Hl, W = small ? p.Ws : p.Wl;
    float* dfeat = (small ? p.dfeat_s : p.dfeat_l) + (size_t)b * H * W * p.C;
    const float x1 = roi[1] * scale, y1 = roi[2] * scale, x2 = roi[3] * scale, y2 = roi[4] * scale;
    const float roi_w = fmaxf(x2 - x1, 1.f), roi_h = fmaxf(y2 - y1, 1.f);
    const float bin_h = roi_h / p.P, bin_w = roi_w / p.P;
    const int gh = p.sampling > 0 ? p.sampling : (int)ceilf(roi_h / p.P);
    const int gw = p.sampling > 0 ? p.sampling : (int)ceilf(roi_w / p.P);
    const float inv_count = 1.f / fmaxf((float)(gh * gw), 1.f);
    const int fy0 = max((int)floorf(y1), 0), fx0 = max((int)floorf(x1), 0);
    const int fy1 = min((int)floorf(y1 + roi_h) + 1, H - 1), fx1 = min((int)floorf(x1 + roi_w) + 1, W - 1);
    const int fh = fy1 - fy0 + 1, fw = fx1 - fx0 + 1;
    const int P = p.P;
    if (fh <= 0 || fw <= 0) return;
    if (fh > RB_F || fw > RB_F) {  // oversize footprint: scatter directly (never taken on this path)
        const int c = threadIdx.x & (RB_C - 1), part = threadIdx.x / RB_C;
        const bool con = c0 + c < p.C;
        for (int bin = part; bin < P * P; bin += 256 / RB_C) {
            const int ph = bin / P, pw = bin % P;
            const float g = con ? p.out[((size_t)r * P * P + bin) * p.C + c0 + c] * inv_count : 0.f;
            for (int iy = 0; iy < gh; ++iy)
                for (int ix = 0; ix < gw; ++ix) {
                    const float y = y1 + ph * bin_h + (iy + 0.5f) * bin_h / gh, x = x1 + pw * bin_w + (ix + 0.5f) * bin_w / gw;
                    if (y < -1.f || y > (float)H || x < -1.f || x > (float)W || !con) continue;
                    float yy = fmaxf(y, 0.f), xx = fmaxf(x, 0.f);
                    int yl = (int)yy, xl = (int)xx, yh, xh;
                    if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else yh = yl + 1;
                    if (xl >= W - 1) { xh = xl = W - 1; xx = (float)xl; } else xh = xl + 1;
                    const float ly = yy - yl, lx = xx - xl, hy = 1.f - ly, hx = 1.f - lx;
                    atomicAdd(dfeat + ((size_t)yl * W + xl) * p.C + c0 + c, hy * hx * g);
                    atomicAdd(dfeat + ((size_t)yl * W + xh) * p.C + c0 + c, hy * lx * g);
                    atomicAdd(dfeat + ((size_t)yh * W + xl) * p.C + c0 + c, ly * hx * g);
                    atomicAdd(dfeat + ((size_t)yh * W + xh) * p.C + c0 + c, ly * lx * g);
                }
        }
        return;
    }
    for (int i = threadIdx.x; i < P * P * RB_C; i += 256) {
        const int bin = i / RB_C, c = i - bin * RB_C;
        G[i] = c0 + c < p.C ? p.out[((size_t)r * P * P + bin) * p.C + c0 + c] * inv_count : 0.f;
    }
    for (int i = threadIdx.x; i < 2 * RB_F * RB_P; i += 256) {
        const bool isx = i >= RB_F * RB_P;
        const int j = isx ? i - RB_F * RB_P : i;
        const int pix = j / RB_P, pb = j - pix * RB_P;
        const int n = isx ? fw : fh, g = isx ? gw : gh, lim = isx ? W : H, f0 = isx ? fx0 : fy0;
        const float start = isx ? x1 : y1, bsz = isx ? bin_w : bin_h;
        float w = 0.f;
        if (pix < n && pb < P)
            for (int s = 0; s < g; ++s) {
                const float v = start + pb * bsz + (s + 0.5f) * bsz / g;
                if (v < -1.f || v > (float)lim) continue;
                float vv = fmaxf(v, 0.f);
                int lo = (int)vv, hi;
                if (lo >= lim - 1) { hi = lo = lim - 1; vv = (float)lo; } else hi = lo + 1;
                const float l = vv - lo, h = 1.f - l;
                if (lo - f0 == pix) w += h;
                if (hi - f0 == pix) w += l;
            }
        (isx ? Wx : Wy)[j] = w;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < fh * P * RB_C; i += 256) {
        const int c = i & (RB_C - 1), rest = i / RB_C;
        const int pw = rest % P, y = rest / P;
        float acc = 0.f;
        for (int ph = 0; ph < P; ++ph) acc = fmaf(Wy[y * RB_P + ph], G[(ph * P + pw) * RB_C + c], acc);
        T[(y * RB_P + pw) * RB_C + c] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < fh * fw * RB_C; i += 256) {
        const int c = i & (RB_C - 1), px = i / RB_C;
        const int y = px / fw, x = px - y * fw;
        float acc = 0.f;
        for (int pw = 0; pw < P; ++pw) acc = fmaf(Wx[x * RB_P + pw], T[(y * RB_P + pw) * RB_C + c], acc);
        if (acc != 0.f && c0 + c < p.C) atomicAdd(dfeat + ((size_t)(fy0 + y) * W + fx0 + x) * p.C + c0 + c, acc);
    }
}

// Backward, gather form (round 5): one workgroup per (image, map, 8 feature rows, 32 channels) OWNS its output tile -- 8 rows x
// <= 32 columns x 32 channels, 32 accumulators per thread -- and walks the image's ROIs on that map (<= o of them, found with a
// ballot over the R rows: order = row order, so the sum is deterministic). Per ROI the separable products of the kernel above:
//   G (64 bins x 32 channels) -> LDS;  T[y][pw][c] = sum_ph Wy[y][ph] G[ph][pw][c]  for the tile's rows;
//   acc[y][x][c] += sum_pw Wx[x][pw] T[y][pw][c].
// Every pixel of the gradient maps is written exactly once with plain stores: no atomics (20 M per pass before), no zero fill
// of the 67 + 17 MB maps before the launch, and the bf16 operand copies the following weight / data gradient launches read
// come out of the same stores. Needs map widths <= 32, P == 8, C % 32 == 0 (128^2 and 64^2 images); else the scatter form.
#define RG_ROWS 8
#define RG_C 32
#define RG_LIST 1024   // ROIs of one image on one map a workgroup can list: the launcher takes the gather form only for R <= RG_LIST
__global__ __launch_bounds__(256, 4) void roi_align_bwd_gather_kernel(RoiArgs p, bf16_t* __restrict__ dop_s, bf16_t* __restrict__ dop_l, int chunks,
                                                                   int tiles_s, int tiles_l) {
    __shared__ float Gs[64 * RG_C];            // [bin][c], already / count
    __shared__ float T[RG_ROWS * 8 * RG_C];    // [y][pw][c]
    __shared__ float Wy[RG_ROWS * 8], Wx[32 * 8];
    __shared__ int list[RG_LIST];
    __shared__ int wcnt[4], nl;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    const int c0 = (bid % chunks) * RG_C; bid /= chunks;
    const int tiles = tiles_s + tiles_l;
    const int tt = bid % tiles, b = bid / tiles;
    const bool small = tt < tiles_s;
    const int row0 = (small ? tt : tt - tiles_s) * RG_ROWS;
    const int H = small ? p.Hs : p.Hl, W = small ? p.Ws : p.Wl;
    const float scale = small ? p.scale_s : p.scale_l;
    // ---- the ROIs of image b routed to this map, in row order
    if (tid == 0) nl = 0;
    __syncthreads();
    for (int base = 0; base < p.R; base += 256) {
        const int r = base + tid;
        bool f = false;
        if (r < p.R && (!p.valid || p.valid[r] != 0)) {
            const float* roi = p.rois + 5 * r;
            const bool sm = !p.dfeat_l || ((roi[3] - roi[1]) < p.thr && (roi[4] - roi[2]) < p.thr);
            f = (int)roi[0] == b && sm == small;
        }
        const unsigned long long m = __ballot(f);
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int off = nl;
        for (int w_ = 0; w_ < wave; ++w_) off += wcnt[w_];
        off += __popcll(m & ((1ull << lane) - 1ull));
        if (f && off < RG_LIST) list[off] = r;
        __syncthreads();
        if (tid == 0) nl = min(RG_LIST, nl + wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3]);
        __syncthreads();
    }
    const int n_roi = nl;
    const int c = tid & (RG_C - 1), xg = tid >> 5;   // thread: channel c, columns xg + 8 k, all 8 rows
    float acc[RG_ROWS][4];
#pragma unroll
    for (int y = 0; y < RG_ROWS; ++y)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[y][k] = 0.f;
    const int P = p.P;   // == 8
    struct Geo { float x1, y1, bin_w, bin_h, inv_count; int gh, gw; bool hit; };
    auto geo = [&](int r) {
        Geo q;
        const float* roi = p.rois + 5 * r;
        q.x1 = roi[1] * scale; q.y1 = roi[2] * scale;
        const float x2 = roi[3] * scale, y2 = roi[4] * scale;
        const float roi_w = fmaxf(x2 - q.x1, 1.f), roi_h = fmaxf(y2 - q.y1, 1.f);
        q.bin_h = roi_h / P; q.bin_w = roi_w / P;
        q.gh = p.sampling > 0 ? p.sampling : (int)ceilf(roi_h / P);
        q.gw = p.sampling > 0 ? p.sampling : (int)ceilf(roi_w / P);
        q.inv_count = 1.f / fmaxf((float)(q.gh * q.gw), 1.f);
        const int fy0 = max((int)floorf(q.y1), 0), fy1 = min((int)floorf(q.y1 + roi_h) + 1, H - 1);
        q.hit = !(fy1 < row0 || fy0 >= row0 + RG_ROWS);   // the footprint touches this tile's rows
        return q;
    };
    // the ROIs whose footprint touches the tile, one after the other; the NEXT one's 64 x 32 gradient values are requested before the
    // current one's products are formed (the launch is four workgroups per CU deep: its latency has to hide inside the workgroup)
    float gq[8];
    int li = 0;
    Geo cur = {};
    for (; li < n_roi; ++li) {
        cur = geo(list[li]);
        if (cur.hit) break;
    }
    if (li < n_roi) {
#pragma unroll
        for (int j = 0; j < 8; ++j) gq[j] = p.out[((size_t)list[li] * 64 + (tid >> 5) + 8 * j) * p.C + c0 + c];
    }
    while (li < n_roi) {
        __syncthreads();   // the previous ROI's Gs / T / Wy / Wx are no longer read
#pragma unroll
        for (int j = 0; j < 8; ++j) Gs[((tid >> 5) + 8 * j) * RG_C + c] = gq[j] * cur.inv_count;
        int ln = li + 1;
        Geo nxt = {};
        for (; ln < n_roi; ++ln) {
            nxt = geo(list[ln]);
            if (nxt.hit) break;
        }
        if (ln < n_roi) {
#pragma unroll
            for (int j = 0; j < 8; ++j) gq[j] = p.out[((size_t)list[ln] * 64 + (tid >> 5) + 8 * j) * p.C + c0 + c];
        }
        for (int i = tid; i < RG_ROWS * 8 + W * 8; i += 256) {
            const bool isx = i >= RG_ROWS * 8;
            const int j = isx ? i - RG_ROWS * 8 : i;
            const int pix = (j >> 3) + (isx ? 0 : row0), pb = j & 7;   // absolute feature row / column
            const int g = isx ? cur.gw : cur.gh, lim = isx ? W : H;
            const float start = isx ? cur.x1 : cur.y1, bsz = isx ? cur.bin_w : cur.bin_h;
            float w = 0.f;
            for (int s_ = 0; s_ < g; ++s_) {
                const float v = start + pb * bsz + (s_ + 0.5f) * bsz / g;
                if (v < -1.f || v > (float)lim) continue;
                float vv = fmaxf(v, 0.f);
                int lo = (int)vv, hi;
                if (lo >= lim - 1) { hi = lo = lim - 1; vv = (float)lo; } else hi = lo + 1;
                const float l = vv - lo, h = 1.f - l;
                if (lo == pix) w += h;
                if (hi == pix) w += l;
            }
            (isx ? Wx : Wy)[j] = w;
        }
        __syncthreads();
        {   // T[y = xg][pw][c]
            float t[8];
#pragma unroll
            for (int pw = 0; pw < 8; ++pw) t[pw] = 0.f;
#pragma unroll
            for (int ph = 0; ph < 8; ++ph) {
                const float wy = Wy[xg * 8 + ph];
#pragma unroll
                for (int pw = 0; pw < 8; ++pw) t[pw] = fmaf(wy, Gs[(ph * 8 + pw) * RG_C + c], t[pw]);
            }
#pragma unroll
            for (int pw = 0; pw < 8; ++pw) T[(xg * 8 + pw) * RG_C + c] = t[pw];
        }
        __syncthreads();
        float wx[4][8];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int pw = 0; pw < 8; ++pw) wx[k][pw] = xg + 8 * k < W ? Wx[(xg + 8 * k) * 8 + pw] : 0.f;
#pragma unroll
        for (int y = 0; y < RG_ROWS; ++y) {
            float t[8];
#pragma unroll
            for (int pw = 0; pw < 8; ++pw) t[pw] = T[(y * 8 + pw) * RG_C + c];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int pw = 0; pw < 8; ++pw) acc[y][k] = fmaf(wx[k][pw], t[pw], acc[y][k]);
        }
        li = ln; cur = nxt;
    }
    float* dfeat = (small ? p.dfeat_s : p.dfeat_l) + (size_t)b * H * W * p.C;
    bf16_t* dop = small ? dop_s : dop_l;
    if (dop) dop += (size_t)b * H * W * p.C;
#pragma unroll
    for (int y = 0; y < RG_ROWS; ++y) {
        if (row0 + y >= H) break;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = xg + 8 * k;
            if (x >= W) continue;
            const size_t o = ((size_t)(row0 + y) * W + x) * p.C + c0 + c;
            dfeat[o] = acc[y][k];
            if (dop) dop[o] = f2bf(acc[y][k]);
        }
    }
}

extern "C" int l2i_roi_align_bwd(const float* rois, const int* valid, const float* dout, float* dfeat_s, float* dfeat_l,
                                 int R, int C, int P, int Hs, int Ws, float scale_s, int Hl, int Wl, float scale_l,
                                 float thr, int sampling, int B, int fresh, void* dfeat_s_bf16, void* dfeat_l_bf16, void* stream) {
    RoiArgs a = {};
    // the gradient maps stand in for the feature maps (only "is there a coarse map" is read from them)
    if (roi_fill(a, dfeat_s, dfeat_l, rois, valid, R, C, P, Hs, Ws, scale_s, Hl, Wl, scale_l, thr, sampling) || !dout)
        return L2I_ERR_ARG;
    a.dfeat_s = dfeat_s; a.dfeat_l = dfeat_l; a.out = const_cast<float*>(dout);
    if ((dfeat_s_bf16 || dfeat_l_bf16) && !fresh) return L2I_ERR_ARG;   // (the copies are of the complete maps)
    if (P > RB_P) return L2I_ERR_ARG;
    // fresh: the maps are uninitialised memory and receive the gradient (=, not +=). The gather form writes every pixel itself.
    static const bool no_gather = getenv("L2I_ROI_GATHER") && atoi(getenv("L2I_ROI_GATHER")) == 0;
    if (fresh && B <= 0) return L2I_ERR_ARG;
    if (fresh && !no_gather && P == 8 && C % RG_C == 0 && Ws <= 32 && (!dfeat_l || Wl <= 32) && Hs > 0 && R <= RG_LIST) {
        const int chunks = C / RG_C;
        const int tiles_s = (Hs + RG_ROWS - 1) / RG_ROWS, tiles_l = dfeat_l ? (Hl + RG_ROWS - 1) / RG_ROWS : 0;
        hipLaunchKernelGGL(roi_align_bwd_gather_kernel, dim3((unsigned)(B * (tiles_s + tiles_l) * chunks)), dim3(256), 0, (hipStream_t)stream, a,
                           (bf16_t*)dfeat_s_bf16, (bf16_t*)dfeat_l_bf16, chunks, tiles_s, tiles_l);
        return l2i_check_launch();
    }
    if (dfeat_s_bf16 || dfeat_l_bf16) return L2I_ERR_ARG;   // (only the gather form writes the copies: refuse rather than leave them unwritten)
    if (fresh) {
        if (l2i_zero_async(dfeat_s, sizeof(float) * (size_t)B * Hs * Ws * C, (hipStream_t)stream) != hipSuccess) return L2I_ERR_LAUNCH;
        if (dfeat_l && l2i_zero_async(dfeat_l, sizeof(float) * (size_t)B * Hl * Wl * C, (hipStream_t)stream) != hipSuccess) return L2I_ERR_LAUNCH;
    }
    if (R == 0) return L2I_OK;
    const int chunks = (C + RB_C - 1) / RB_C;
    hipLaunchKernelGGL(roi_align_bwd_sep_kernel, dim3(R * chunks), dim3(256), 0, (hipStream_t)stream, a, chunks);
    return l2i_check_launch();
}
