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
    int chunk_major;     // K order (see issue_tiles)
    int nks;             // K-steps in total
    unsigned x_bytes, w_bytes;
    int splits, ks_per;  // split-K: `out` pre-zeroed, partials combined with f32 atomics
    // halo kernel geometry (conv_halo_kernel): sub-patches of PHs x PW pixels, halo rows h = sp*SUBH + hy*P + hx
    int PHs, sub_shift, P, SUBH, HR, HWd, halo_pieces;
    unsigned long long* dbg;   // profiling build only: per-wave phase cycle totals
    float alpha;
};

__device__ __forceinline__ void idx2pix(int idx, int hw_shift, int lin, int& py, int& px) {
    if (lin) { py = idx; px = 0; return; }
    const int q = idx >> 2, s = idx & 3;
    const int qy = q >> hw_shift, qx = q & ((1 << hw_shift) - 1);
    py = 2 * qy + (s >> 1);
    px = 2 * qx + (s & 1);
}

// Epilogue shared by the convolution kernels: alpha, bias, ReLU-backward mask, residual, optional operand copies,
// optional 2x2 average pool (in-register: the 4 GEMM rows of a quad are one lane's registers 4g..4g+3), split-K atomics.
template <typename T, int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16_t (&acc)[TM][TN], int wrow, int wcol, int lane,
                                              int tile_r, int tile_c, int n0, int split, int rows_total) {
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

// Tile geometry: BM x BN outputs per workgroup of WM x WN waves (each wave a (BM/WM) x (BN/WN) block of 32x32
// MFMA tiles), NS-stage LDS ring.
// HK = 1 halves the K-step (64-byte LDS rows): twice as many, half as large ring stages, so that two workgroups per
// CU can each keep three tiles in flight within the 160 KB of LDS (the DMA round trip under load is ~1.5 us).
template <typename T, int BM, int BN, int WM, int WN, int NS, int HK>
__global__ __launch_bounds__(WM* WN * 64) void conv_igemm_kernel(ConvArgs p) {
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

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nblk = p.tiles_m * p.tiles_n;
    const int split = blockIdx.x / nblk;
    const int bid = xcd_remap(blockIdx.x - split * nblk, nblk);
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int tile_c = tile_m % p.tiles_c, tile_r = tile_m / p.tiles_c;
    const int n0 = tile_n * BN;
    const int rows_total = p.B * p.Ho;
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
        const int b = r / p.Ho;
        const bool rv = r < rows_total;
        const int y = r - b * p.Ho, x = tile_c * p.PW + px;
        a_y[q] = y;
        a_x[q] = x;
        a_off[q] = (unsigned)(((b * p.Hi + (y >> p.up2)) * p.Wi + (x >> p.up2)) * p.Ci) * SZ;
        unsigned m = 0;
        for (int t = 0; t < p.KH * p.KH; ++t) {
            const int yy = y + t / p.KH - pad, xx = x + t % p.KH - pad;
            if (rv && yy >= 0 && yy < p.Ho && xx >= 0 && xx < p.Wo) m |= 1u << t;
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
    const u32x4_t rsrc_x = make_rsrc(p.x, p.x_bytes), rsrc_w = make_rsrc(p.w, p.w_bytes);
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
            const int tap = k0 / p.Ci, ci = k0 - tap * p.Ci;
            const int ky = tap / p.KH, kx = tap - ky * p.KH;
            const bool kvalid = k0 < p.K;
#pragma unroll
            for (int q = 0; q < AP; ++q) {
                const int ys = (a_y[q] + ky - pad) >> p.up2, xs = (a_x[q] + kx - pad) >> p.up2;
                const int y0 = a_y[q] >> p.up2, x0 = a_x[q] >> p.up2;
                const unsigned off = a_off[q] + (unsigned)((((ys - y0) * p.Wi + (xs - x0)) * p.Ci + ci) * SZ);
                const unsigned voff = (kvalid && ((a_mask[q] >> tap) & 1u)) ? off : OOB;
                buf_load_lds16(rsrc_x, voff, stage + q * RPP * ROWB + wbase);
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
    const int nks = min(p.nks, ks0 + p.ks_per);
    // NS-stage LDS ring. Tiles ks .. ks+NS-2 are in flight while tile ks is consumed: each wave waits with a COUNTED
    // s_waitcnt vmcnt for its own DMA of tile ks (its LPT newest-but-... loads may stay outstanding), then a raw
    // s_barrier makes every wave's part of the tile visible and proves the stage about to be refilled is no longer
    // read. (__syncthreads() would drain vmcnt to 0 and serialise the ring.) One K-step of a single workgroup is
    // otherwise bound by the ~1.5 us DMA round trip, not by its 0.2 us of MFMA work.
    constexpr int LPT = AP + BP;  // LDS-DMA instructions per thread per tile
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

    conv_epilogue<T, TM, TN>(p, acc, wrow, wcol, lane, tile_r, tile_c, n0, split, rows_total);
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
// for all taps and tile shapes used (exhaustive check: scratch/halo_check.py). B tiles: as in the kernel above.
template <int BM, int BN, int WM, int WN, int NSB>
__global__ __launch_bounds__(WM* WN * 64) void conv_halo_kernel(ConvArgs p) {
    typedef bf16_t T;
    constexpr int THREADS = WM * WN * 64, NW = WM * WN;
    constexpr int BK = 64, SZ = 2;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int RPP = THREADS / 8, BP = BN / RPP;
    constexpr int HPMAX = 7;   // halo pieces per wave (1 KB each): <= 50 pieces / 8 waves, 23 / 4 waves
    constexpr unsigned OOB = 0x80000000u;
    static_assert(BN % RPP == 0, "tile geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int nblk = p.tiles_m * p.tiles_n;
    const int split = blockIdx.x / nblk;
    const int bid = xcd_remap(blockIdx.x - split * nblk, nblk);
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int tile_c = tile_m % p.tiles_c, tile_r = tile_m / p.tiles_c;
    const int n0 = tile_n * BN;
    const int rows_total = p.B * p.Ho;
    const u32x4_t rsrc_x = make_rsrc(p.x, p.x_bytes), rsrc_w = make_rsrc(p.w, p.w_bytes);
    const unsigned smem_addr = lds_addr_of(smem);
    const unsigned halo_bytes = (unsigned)p.halo_pieces * 1024u;
    const unsigned bring_addr = smem_addr + 2u * halo_bytes;
    const char* bring = smem + 2u * halo_bytes;

    // ---- this thread's lane of each halo piece (fixed for the whole K loop; the chunk base is added per chunk)
    unsigned hoff[HPMAX];
    int hlim[HPMAX];
#pragma unroll
    for (int q = 0; q < HPMAX; ++q) {
        const int piece = wv + NW * q;
        const int h = piece * 8 + (lane >> 3), pch = lane & 7;
        const int sp = h / p.SUBH, rem = h - sp * p.SUBH;
        const int hy = rem / p.P, hx = rem - hy * p.P;
        const int gr0 = tile_r * p.PH + sp * p.PHs;          // global output row (b*Ho + y) of the sub-patch origin
        const int b = gr0 / p.Ho, y0 = gr0 - b * p.Ho, x0 = tile_c * p.PW;
        const int iy = (y0 >> p.up2) + hy - 1, ix = (x0 >> p.up2) + hx - 1;
        const bool ok = h < p.HR && hx < p.HWd && gr0 < rows_total && iy >= 0 && iy < p.Hi && ix >= 0 && ix < p.Wi;
        const int lch = pch ^ (((hx >> 1) + 4 * hy + 2 * sp) & 7);
        hoff[q] = ok ? (unsigned)(((b * p.Hi + iy) * p.Wi + ix) * p.Ci + lch * 8) * SZ : OOB;
        hlim[q] = p.Ci - lch * 8;   // this lane's channels exist in chunk cb iff cb < hlim
    }
    // ---- B rows (weights): as in conv_igemm_kernel
    const int lrow = tid >> 3;
    const int lchunk = (tid & 7) ^ ig2_swz(lrow);
    unsigned b_off[BP];
#pragma unroll
    for (int q = 0; q < BP; ++q) b_off[q] = (unsigned)((n0 + lrow + RPP * q) * p.Kpad + lchunk * 8) * SZ;
    const unsigned wbase = (unsigned)wv * 1024u;

    // ---- A fragment rows of this lane: pixel (sub-patch, y, x) of GEMM row wrow + 32 i + (lane & 31)
    const int wrow = (wave / WN) * (BM / WM), wcol = (wave % WN) * (BN / WN);
    int a_sp[TM], a_py[TM], a_px[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int py, px;
        idx2pix(wrow + i * 32 + (lane & 31), p.hw_shift, 0, py, px);
        a_sp[i] = py >> p.sub_shift;
        a_py[i] = py & (p.PHs - 1);
        a_px[i] = px;
    }
    const int hh = lane >> 5;
    const int brow = wcol + (lane & 31);

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto issue_halo_piece = [&](int q, unsigned buf_addr, int cb) {   // piece q of this wave, chunk base cb
        if (wv + NW * q < p.halo_pieces)
            buf_load_lds16(rsrc_x, cb < hlim[q] ? hoff[q] + (unsigned)cb * SZ : OOB, buf_addr + (unsigned)(wv + NW * q) * 1024u);
    };
    auto issue_b = [&](unsigned stage_addr, int tap, int cb) {
        const unsigned kadd = (unsigned)(tap * p.Ci + cb) * SZ;
#pragma unroll
        for (int q = 0; q < BP; ++q) buf_load_lds16(rsrc_w, b_off[q] + kadd, stage_addr + (unsigned)(q * RPP) * 128u + wbase);
    };

    // K range of this split: whole chunks (ks_per is a multiple of 9)
    const int ks0 = split * p.ks_per, ks1 = min(p.nks, ks0 + p.ks_per);
    if (ks0 < ks1) {
        int cb = (ks0 / 9) * BK, tap = 0, cpar = 0;
#pragma unroll
        for (int q = 0; q < HPMAX; ++q) issue_halo_piece(q, smem_addr, cb);
        // B ring of NSB stages: tiles ks+1 .. ks+NSB-1 are in flight while tile ks is consumed. c1 / c2 = LDS-DMA
        // instructions this wave issued one / two steps ago: they may still be outstanding at the next wait.
        int pf_tap = 0, pf_cb = cb;   // (tap, chunk base) of the next B tile to issue
        int c1 = 0, c2 = 0;
#pragma unroll
        for (int s = 0; s < NSB - 1; ++s) {
            int n = 0;
            if (ks0 + s < ks1) {
                issue_b(bring_addr + (unsigned)s * (BN * 128u), pf_tap, pf_cb);
                n = BP;
                if (++pf_tap == 9) { pf_tap = 0; pf_cb += BK; }
            }
            if (s > 0) { c2 = c1; c1 = n; }   // tile 0 itself must have landed at the first wait
        }
#ifdef L2I_PROF
        unsigned long long pa = 0, pb = 0, pc = 0, pd = 0;
#endif
        for (int ks = ks0; ks < ks1; ++ks) {
            const int it = ks - ks0;
#ifdef L2I_PROF
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#endif
            {
                const int allow = NSB >= 4 ? c1 + c2 : NSB == 3 ? c1 : 0;
                switch (allow) {   // s_waitcnt takes an immediate
                    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
                    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
                    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
                    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
                    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                }
            }
#ifdef L2I_PROF
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
            __builtin_amdgcn_s_barrier();   // everyone's part of tile ks (and, at tap 0, of the halo) landed; the stage and halo buffer refilled below are free
            asm volatile("" ::: "memory");
#ifdef L2I_PROF
            const unsigned long long t2 = __builtin_amdgcn_s_memtime();
#endif
            // weights NSB-1 taps ahead, and one piece per wave of the NEXT chunk's halo
            {
                int n = 0;
                if (ks + NSB - 1 < ks1) {
                    issue_b(bring_addr + (unsigned)((it + NSB - 1) % NSB) * (BN * 128u), pf_tap, pf_cb);
                    n = BP;
                    if (++pf_tap == 9) { pf_tap = 0; pf_cb += BK; }
                }
                if (ks + 9 - tap < ks1 && tap < HPMAX && wv + NW * tap < p.halo_pieces) {   // a next chunk exists and this wave has a piece `tap`
                    const unsigned nbuf = smem_addr + (unsigned)(cpar ^ 1) * halo_bytes;
                    ++n;
                    switch (tap) {   // (constant indices keep hoff / hlim in registers)
                        case 0: issue_halo_piece(0, nbuf, cb + BK); break;
                        case 1: issue_halo_piece(1, nbuf, cb + BK); break;
                        case 2: issue_halo_piece(2, nbuf, cb + BK); break;
                        case 3: issue_halo_piece(3, nbuf, cb + BK); break;
                        case 4: issue_halo_piece(4, nbuf, cb + BK); break;
                        case 5: issue_halo_piece(5, nbuf, cb + BK); break;
                        default: issue_halo_piece(6, nbuf, cb + BK); break;
                    }
                }
                c2 = c1; c1 = n;
            }
#ifdef L2I_PROF
            const unsigned long long t3 = __builtin_amdgcn_s_memtime();
#endif
            // ---- MFMA step for (tap, chunk): A from the halo at the tap's shift, B from the ring
            const char* hb = smem + (unsigned)cpar * halo_bytes;
            const char* bs = bring + (unsigned)(it % NSB) * (BN * 128u);
            const int ky = tap / 3, kx = tap - ky * 3;
            int a_row[TM], a_swz[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                int hy, hx;
                if (p.up2) { hy = ((a_py[i] + ky - 1) >> 1) + 1; hx = ((a_px[i] + kx - 1) >> 1) + 1; }
                else { hy = a_py[i] + ky; hx = a_px[i] + kx; }
                a_row[i] = (a_sp[i] * p.SUBH + hy * p.P + hx) * 128;
                a_swz[i] = ((hx >> 1) + 4 * hy + 2 * a_sp[i]) & 7;
            }
            {   // all 16 fragment reads first, then the MFMAs (see Mma2::step)
                bf16x8_t a[4][TM], b[4][TN];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[kk][i] = *reinterpret_cast<const bf16x8_t*>(hb + a_row[i] + (((kk * 2 + hh) ^ a_swz[i]) << 4));
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[kk][j] = *reinterpret_cast<const bf16x8_t*>(bs + ig2_off<0>(brow + j * 32, kk * 2 + hh));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, a[kk][i]),
                                __builtin_bit_cast(__attribute__((ext_vector_type(8))) __bf16, b[kk][j]), acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef L2I_PROF
            {
                const unsigned long long t4 = __builtin_amdgcn_s_memtime();
                pa += t1 - t0; pb += t2 - t1; pc += t3 - t2; pd += t4 - t3;
            }
#endif
            if (++tap == 9) { tap = 0; cb += BK; cpar ^= 1; }
        }
#ifdef L2I_PROF
        if (p.dbg && lane == 0) {
            unsigned long long* d = p.dbg + ((size_t)blockIdx.x * NW + wave) * 4;
            d[0] = pa; d[1] = pb; d[2] = pc; d[3] = pd;
        }
#endif
    }
    conv_epilogue<T, TM, TN>(p, acc, wrow, wcol, lane, tile_r, tile_c, n0, split, rows_total);
}

static int ilog2(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

static unsigned long long* g_conv_dbg = nullptr;   // profiling builds (-DL2I_PROF): per-wave phase cycle totals of the halo kernel
extern "C" int l2i_debug_set_buffer(void* p) { g_conv_dbg = (unsigned long long*)p; return L2I_OK; }
static int g_split_target = 512;   // tuning hook: workgroups a split-K launch aims for
// Launch one instantiation; LDS rings above 64 KB need the opt-in attribute (set once per instantiation).
template <typename T, int BM, int BN, int WM, int WN, int NS, int HK = 0>
static int launch_cfg(ConvArgs a, hipStream_t stream) {
    constexpr int BK = Mma<T>::BK >> HK;
    constexpr size_t lds = (size_t)NS * (BM + BN) * (HK ? 64 : 128);
    a.chunk_major = (a.KH == 3 && a.Ci >= BK) ? 1 : 0;
    a.nks = a.chunk_major ? 9 * ((a.Ci + BK - 1) / BK) : a.Kpad / BK;
    a.PH = BM / a.PW;
    if (!a.lin && (a.PH & 1)) return L2I_ERR_ARG;
    const int rows = a.B * a.Ho;
    a.tiles_m = ((rows + a.PH - 1) / a.PH) * a.tiles_c;
    a.tiles_n = (a.Co + BN - 1) / BN;
    const int nblk = a.tiles_m * a.tiles_n;
    const int nks = a.nks;
    // split-K for small grids with a long reduction (D block5/6, G res1/res2, ROI heads): fill the 256 CUs
    int splits = 1;
    if (a.out && !a.out_op && !a.out_op_raw && nblk < 192 && nks >= 16) {
        splits = (g_split_target + nblk - 1) / nblk;
        if (splits > nks / 16) splits = nks / 16;   // >= 16 K-steps per split: shorter ones are all prologue + atomic epilogue
        if (splits < 1) splits = 1;
    }
    a.ks_per = (nks + splits - 1) / splits;
    a.splits = (nks + a.ks_per - 1) / a.ks_per;
    if (a.splits > 1) {
        const size_t bytes = sizeof(float) * (size_t)a.B * (a.Ho >> a.pool2) * (a.Wo >> a.pool2) * a.Co;
        if (hipMemsetAsync(a.out, 0, bytes, stream) != hipSuccess) return L2I_ERR_LAUNCH;
    }
    static bool ready = false;
    if (!ready) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<T, BM, BN, WM, WN, NS, HK>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        ready = true;
    }
    (void)BK;
    hipLaunchKernelGGL((conv_igemm_kernel<T, BM, BN, WM, WN, NS, HK>), dim3(nblk * a.splits), dim3(WM * WN * 64), lds, stream, a);
    return l2i_check_launch();
}

// Halo kernel launch (bf16, 3x3, Ci >= 64, Wo >= 8, no upsample into 8-wide maps). Returns -100 when the shape is
// not covered so that the caller falls through to the generic kernel.
template <int BM, int BN, int WM, int WN, int NSB>
static int launch_halo(ConvArgs a, hipStream_t stream) {
    a.PH = BM / a.PW;
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
    const size_t lds = (size_t)2 * a.halo_pieces * 1024 + (size_t)NSB * BN * 128;
    if (lds > 160 * 1024) return -100;
    const int nchunks = (a.Ci + 63) / 64;
    a.nks = 9 * nchunks;
    const int rows = a.B * a.Ho;
    a.tiles_m = ((rows + a.PH - 1) / a.PH) * a.tiles_c;
    a.tiles_n = (a.Co + BN - 1) / BN;
    const int nblk = a.tiles_m * a.tiles_n;
    int splits = 1;
    if (a.out && !a.out_op && !a.out_op_raw && nblk < 192 && nchunks >= 4) {
        splits = (g_split_target + nblk - 1) / nblk;
        if (splits > nchunks / 2) splits = nchunks / 2;   // >= 18 K-steps per split
        if (splits < 1) splits = 1;
    }
    const int cper = (nchunks + splits - 1) / splits;
    a.ks_per = 9 * cper;
    a.splits = (nchunks + cper - 1) / cper;
    if (a.splits > 1) {
        const size_t bytes = sizeof(float) * (size_t)a.B * (a.Ho >> a.pool2) * (a.Wo >> a.pool2) * a.Co;
        if (hipMemsetAsync(a.out, 0, bytes, stream) != hipSuccess) return L2I_ERR_LAUNCH;
    }
    static bool ready = false;
    if (!ready) {
        (void)hipFuncSetAttribute((const void*)conv_halo_kernel<BM, BN, WM, WN, NSB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        ready = true;
    }
    hipLaunchKernelGGL((conv_halo_kernel<BM, BN, WM, WN, NSB>), dim3(nblk * a.splits), dim3(WM * WN * 64), lds, stream, a);
    return l2i_check_launch();
}

static int g_conv_cfg_override = -1;  // tuning hook (l2i_set_conv_config): -1 = heuristic
extern "C" int l2i_set_conv_config(int cfg) {
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
    a.x_bytes = (unsigned)((size_t)a.B * a.Hi * a.Wi * a.Ci * sizeof(T));
    a.w_bytes = (unsigned)((size_t)((a.Co + 127) / 128 * 128) * a.Kpad * sizeof(T));
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
    // Heuristic from scratch/conv_tune.py on MI355X (TFLOP/s, bf16): big-tile configs pay only when their grid still
    // fills the 256 CUs; a single wave of 128x128 tiles (one workgroup per CU) prefers the 8-wave deep ring.
    const long long M = (long long)a.B * a.Ho * a.Wo;
    if (sizeof(T) == 2 && a.KH == 3 && a.Ci >= 64 && !a.lin && a.Wo >= 8 && !(a.up2 && a.Wo < 16) && a.Ho >= 2 &&
        (g_conv_cfg_override < 0 || g_conv_cfg_override >= 10)) {
        // halo kernel: 256x128 / 8 waves where its grid still fills the chip (and always for 8-wide maps, whose
        // 128-row tile would not fit two workgroups per CU), else 128x128 or 128x64 / 4 waves, two workgroups per CU
        const long long t256 = ((M + 255) / 256) * ((a.Co + 127) / 128);
        int hc = a.Co <= 64 ? 1 : ((a.Wo < 16 || t256 >= 256) ? 2 : 0);
        if (g_conv_cfg_override >= 10) hc = g_conv_cfg_override - 10;
        if (a.Wo < 16 && hc < 2) hc = 2;
        int rc;
        switch (hc) {
            case 1: rc = launch_halo<128, 64, 2, 2, 2>(a, stream); break;
            case 2: rc = launch_halo<256, 128, 4, 2, 2>(a, stream); break;
            case 3: rc = launch_halo<256, 128, 4, 2, 4>(a, stream); break;    // 3 weight tiles in flight
            case 4: rc = launch_halo<128, 128, 2, 2, 4>(a, stream); break;    // one workgroup per CU, deep ring
            case 5: rc = launch_halo<256, 128, 4, 2, 3>(a, stream); break;
            default: rc = launch_halo<128, 128, 2, 2, 2>(a, stream); break;
        }
        if (rc != -100) return rc;
    }
    const long long t128 = ((M + 127) / 128) * ((a.Co + 127) / 128);
    const long long t256x128 = ((M + 255) / 256) * ((a.Co + 127) / 128);
    const long long t256x256 = ((M + 255) / 256) * ((a.Co + 255) / 256);
    int cfg;
    if (a.Co <= 64) cfg = 1;
    else if (!a.lin && a.Co % 256 == 0 && t256x256 >= 240 && a.nks >= 144) cfg = 4;   // 1024-channel 3x3 ROI heads
    else if (!a.lin && a.Co % 128 == 0 && t256x128 >= 384 && a.nks >= 72) cfg = 3;
    else if (t128 >= 192 && t128 <= 288) cfg = 2;
    else cfg = 0;
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
extern "C" int l2i_conv2d_fwd(const void* x, const void* w, const float* bias, const float* res,
                              const void* relu_mask, float* out, void* out_op, void* out_op_raw, int dtype, int B, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH,
                              int up2, int pool2, int relu_op, int Kpad, float alpha, void* stream) {
    if (!x || !w || (!out && !out_op && !out_op_raw)) return L2I_ERR_ARG;
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.res = res; a.out = out; a.out_op = out_op; a.out_op_raw = out_op_raw; a.relu_mask = relu_mask;
    a.B = B; a.Hi = Hi; a.Wi = Wi; a.Ci = Ci; a.Ho = Ho; a.Wo = Wo; a.Co = Co; a.KH = KH;
    a.up2 = up2 ? 1 : 0; a.pool2 = pool2 ? 1 : 0; a.relu_op = relu_op ? 1 : 0;
    a.Kpad = Kpad; a.alpha = alpha; a.dbg = g_conv_dbg;
    if (dtype == 0) return launch_conv<float>(a, (hipStream_t)stream);
    if (dtype == 1) return launch_conv<bf16_t>(a, (hipStream_t)stream);
    return L2I_ERR_ARG;
}

// Debug aid: co-resident workgroups per CU the runtime computes for a few instantiations (scratch/occupancy.py).
extern "C" int l2i_debug_occupancy(int which, int lds_bytes) {
    int n = -1;
    hipError_t e = hipSuccess;
    switch (which) {
        case 0: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)conv_igemm_kernel<bf16_t, 128, 128, 2, 2, 2, 0>, 256, lds_bytes); break;
        case 10: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)conv_halo_kernel<128, 128, 2, 2, 2>, 256, lds_bytes); break;
        case 12: e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void*)conv_halo_kernel<256, 128, 4, 2, 2>, 512, lds_bytes); break;
        default: break;
    }
    return e == hipSuccess ? n : -(int)e;
}
