// Shared device helpers for the layout2img gfx950 kernels.
// gfx950 only: 64-wide wavefronts, MFMA, 160 KiB LDS. No portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define L2I_OK 0
#define L2I_ERR_ARG (-1)
#define L2I_ERR_LAUNCH (-2)

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment, 4 VGPRs
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// round-to-nearest-even f32 -> bf16: gfx950's v_cvt_pk_bf16_f32 (one instruction per PAIR; the integer form -- add 0x7fff +
// lsb, shift, NaN select -- was ~6 VALU per element and showed up in every epilogue / normalisation kernel that writes operands)
typedef float l2i_f2_t __attribute__((ext_vector_type(2)));
typedef __bf16 l2i_b2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f2bf2(float lo, float hi) {   // two bf16 packed in one dword (lo in bits 0-15)
    l2i_f2_t v;
    v[0] = lo; v[1] = hi;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, l2i_b2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(f2bf2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

template <typename T> struct OpT;
template <> struct OpT<bf16_t> {
    static constexpr int EPG = 8;  // elements per 16-byte group
    __device__ static __forceinline__ bf16_t from(float f) { return f2bf(f); }
    __device__ static __forceinline__ float to(bf16_t v) { return bf2f(v); }
};
template <> struct OpT<float> {
    static constexpr int EPG = 4;
    __device__ static __forceinline__ float from(float f) { return f; }
    __device__ static __forceinline__ float to(float v) { return v; }
};

// four consecutive operand elements <-> four floats (8- / 16-byte accesses)
template <typename T> struct Op4;
template <> struct Op4<bf16_t> {
    __device__ static __forceinline__ void load(const bf16_t* p, float (&v)[4]) {
        const uint2 u = *reinterpret_cast<const uint2*>(p);
        v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
        v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    }
    __device__ static __forceinline__ void store(bf16_t* p, const float (&v)[4]) {
        uint2 u;
        u.x = f2bf2(v[0], v[1]);
        u.y = f2bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = u;
    }
};
template <> struct Op4<float> {
    __device__ static __forceinline__ void load(const float* p, float (&v)[4]) {
        const float4 u = *reinterpret_cast<const float4*>(p);
        v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
    }
    __device__ static __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); red must hold 16 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += red[i];
    __syncthreads();  // red[] may be reused by the caller's next reduction
    return r;
}

// ---------------------------------------------------------------- replicated accumulation workspace
// Device-scope atomics on ONE address serialise at ~0.09 us each on this chip (measured: they resolve at the memory
// side of the eight per-XCD L2s), so a reduction in which every workgroup adds its partial to the same per-channel
// totals costs (#workgroups x 0.09 us) of pure tail -- 45 us for 512 workgroups, against 10-20 us of streaming.
// Instead a workgroup adds into replica (block % L2I_WS_R) of a small workspace and a one-wave-per-64-values fold
// kernel, launched behind it on the same stream, adds the replica sums to the destination and zeroes the replicas
// again. (An in-kernel "last workgroup folds" needs an agent-scope release per workgroup, which writes the dirty L2
// back each time: measured 2x slower than the plain atomics.) Layout: ws[r*L + i] = replica r of value i.
#define L2I_WS_R 32
#ifndef L2I_WS_FLOATS
#define L2I_WS_FLOATS (32 * 4 * 1024)   // include/l2i.h
#endif
__device__ __forceinline__ float* ws_replica(float* ws, int r, int L) { return ws + (size_t)r * L; }

// value i = row k (= i / C) x channel: added to dst[k][i % C]
struct WsFoldArgs { float* ws; float* dst[4]; int L, C; };
__device__ __forceinline__ void ws_fold_body(const WsFoldArgs& p, int block) {
    const int i = block * 256 + threadIdx.x;
    if (i >= p.L) return;
    float* q = p.ws + i;
    float v[L2I_WS_R];
#pragma unroll
    for (int r = 0; r < L2I_WS_R; ++r) v[r] = q[(size_t)r * p.L];   // 32 independent loads in flight
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < L2I_WS_R; ++r) { t += v[r]; q[(size_t)r * p.L] = 0.f; }
    const int k = i / p.C;
    float* d = p.dst[k];
    if (d) atomicAdd(d + (i - k * p.C), t);   // (atomic: two passes of one network may fold into the same gradient from two streams)
}
static __global__ __launch_bounds__(256) void ws_fold_kernel(WsFoldArgs p) { ws_fold_body(p, blockIdx.x); }
static inline void ws_fold(float* ws, int L, int C, float* d0, float* d1, float* d2, float* d3, hipStream_t stream) {
    WsFoldArgs a;
    a.ws = ws; a.dst[0] = d0; a.dst[1] = d1; a.dst[2] = d2; a.dst[3] = d3; a.L = L; a.C = C;
    hipLaunchKernelGGL(ws_fold_kernel, dim3((L + 255) / 256), dim3(256), 0, stream, a);
}

// ---------------------------------------------------------------- one writer per class row (round 6)
// Gradients of embedding-like tables (dE[y[r]] += g[r] f[r]) were scattered with one float atomic per row and column: rows of the same
// class met in an order that changed from run to run. Now the workgroup of the FIRST row that carries a class adds all of the class's
// rows itself, in ascending row order, and is the only writer of dE[class] in the launch. Helpers for 256-thread workgroups:
// class_first(): true when no row in front of r carries r's class; class_rows(): the rows >= r with the class, ascending, in LDS.
#define L2I_CLASS_LIST 2048
__device__ __forceinline__ bool class_first(const long long* __restrict__ y, int r, long long cls) {
    int hit = 0;
    for (int rr = threadIdx.x; rr < r; rr += 256) hit |= (y[rr] == cls);
    return !__syncthreads_or(hit);
}
__device__ __forceinline__ int class_rows(const long long* __restrict__ y, int r0, int R, long long cls, int* list /* [L2I_CLASS_LIST] */, int* wsum /* [4] */) {
    int n = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = r0; base < R; base += 256) {
        const int rr = base + (int)threadIdx.x;
        const bool m = rr < R && y[rr] == cls;
        const unsigned long long b = __ballot(m);
        if (lane == 0) wsum[wave] = __popcll(b);
        __syncthreads();
        int off = n;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        const int pos = off + __popcll(b & ((1ull << lane) - 1ull));
        if (m && pos < L2I_CLASS_LIST) list[pos] = rr;
        n += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    return n < L2I_CLASS_LIST ? n : L2I_CLASS_LIST;   // (callers refuse R > L2I_CLASS_LIST on the host)
}

// ---------------------------------------------------------------- deterministic column sums of stored partial rows (round 6)
// The replicated workspace above still ends in float atomics: which workgroup's partial lands first in a replica changes from run to
// run, so a per-channel total -- a batch statistic -- moves in its last bits, a ReLU gate further on flips, and two runs of the same f32
// training step differ by ~1e-3 of the gradient. Where the partial sums of a launch have a natural owner (a wave of a convolution tile,
// a slab of a statistics pass) they are STORED as rows part[z][r][0..L) in the stream's scratch instead and summed here in a fixed
// order: dst[z * zs + i] (+)= sum_r part[z][r][i]. One small launch up to 16384 rows, two above (row chunks -> tmp[z][chunk][L], chunks -> dst). Plain stores also run at several times the rate of float atomics on this chip (DESIGN A.3 4.1b).
// The pairwise order inside a thread, across the row lanes of a workgroup and across the chunks is fixed by (R, L) alone.
#define L2I_FOLD_CHUNKS 64
static __global__ __launch_bounds__(256) void rows_fold1_kernel(const float* __restrict__ part, int R, int L, int rc, float* __restrict__ tmp, int ld) {
    __shared__ float4 red[256];
    const int L4 = L >> 2, cbase = blockIdx.x * 64, ncol = min(64, L4 - cbase), TY = 256 / ncol;
    const int tx = threadIdx.x % ncol, ty = threadIdx.x / ncol;
    const int r0 = blockIdx.y * rc, r1 = min(R, r0 + rc);
    const float* src = part + (size_t)blockIdx.z * R * ld + 4 * (cbase + tx);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ty < TY) {
        int r = r0 + ty;
        for (; r + 3 * TY < r1; r += 4 * TY) {
            const float4 a = *reinterpret_cast<const float4*>(src + (size_t)r * ld), b = *reinterpret_cast<const float4*>(src + (size_t)(r + TY) * ld);
            const float4 c = *reinterpret_cast<const float4*>(src + (size_t)(r + 2 * TY) * ld), d = *reinterpret_cast<const float4*>(src + (size_t)(r + 3 * TY) * ld);
            s.x += (a.x + b.x) + (c.x + d.x); s.y += (a.y + b.y) + (c.y + d.y); s.z += (a.z + b.z) + (c.z + d.z); s.w += (a.w + b.w) + (c.w + d.w);
        }
        for (; r < r1; r += TY) {
            const float4 a = *reinterpret_cast<const float4*>(src + (size_t)r * ld);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < ncol) {
        for (int j = 1; j < TY; ++j) { const float4 a = red[j * ncol + tx]; s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w; }
        *reinterpret_cast<float4*>(tmp + ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * L + 4 * (cbase + tx)) = s;
    }
}
// mode 0: dst = total; 1: dst += total (plain: the caller owns dst on this stream); 2: atomicAdd (a gradient that two passes of one
// network, on two streams, may fold into at the same time: one add per address and launch).
// A workgroup = cl float4 columns x (256 / cl) row lanes (cl a power of two <= 16, chosen by the host so that narrow matrices still make
// many workgroups and tall ones many row lanes): a thread adds every (256 / cl)-th row, four loads in flight, the row lanes meet in a
// fixed-order tree in LDS. Up to L2I_FOLD_DIRECT rows in this one launch. The body is a device function so that a kernel which runs behind
// the producer anyway (wgrad_reduce_kernel, norm_a8_finish_kernel) can carry a fold in extra workgroups instead of one more launch.
#define L2I_FOLD_DIRECT 16384
struct RowsFoldArgs {
    const float* src; int R, L;
    float* dst[4]; int seg, segv;   // column i = segment k (= i / seg) x position j: added to dst[k][j] when j < segv and dst[k] is not null
    long long zs; int mode; float* dup; int ld; int cl; int nbx;   // dup: a second destination of segment 0; nbx: workgroups along x
};
__device__ __forceinline__ void rows_fold2_body(const RowsFoldArgs& p, int bx, int bz, float4* red) {
    const int cl = p.cl, rl = 256 / cl;
    const int tx = threadIdx.x & (cl - 1), ty = threadIdx.x / cl;
    const int c4 = bx * cl + tx;   // float4 column
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool on = 4 * c4 < p.L;
    if (on) {
        const float* q = p.src + (size_t)bz * p.R * p.ld + 4 * c4;
        int r = ty;
        for (; r + 3 * rl < p.R; r += 4 * rl) {
            const float4 a = *reinterpret_cast<const float4*>(q + (size_t)r * p.ld), b = *reinterpret_cast<const float4*>(q + (size_t)(r + rl) * p.ld);
            const float4 c = *reinterpret_cast<const float4*>(q + (size_t)(r + 2 * rl) * p.ld), d = *reinterpret_cast<const float4*>(q + (size_t)(r + 3 * rl) * p.ld);
            s.x += (a.x + b.x) + (c.x + d.x); s.y += (a.y + b.y) + (c.y + d.y); s.z += (a.z + b.z) + (c.z + d.z); s.w += (a.w + b.w) + (c.w + d.w);
        }
        for (; r < p.R; r += rl) {
            const float4 a = *reinterpret_cast<const float4*>(q + (size_t)r * p.ld);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = rl >> 1; st >= 1; st >>= 1) {   // row lane ty += row lane ty + st: the same tree whatever the timing
        if (ty < st) {
            const float4 a = red[threadIdx.x], b = red[threadIdx.x + st * cl];
            red[threadIdx.x] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        }
        __syncthreads();
    }
    if (ty == 0 && on) {
        const float4 t = red[threadIdx.x];
        const float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * c4 + e;
            const int k = i / p.seg, jcol = i - k * p.seg;
            if (k > 3 || !p.dst[k] || jcol >= p.segv) continue;   // (padding columns of the partial rows: no destination)
            float* d = p.dst[k] + jcol + (size_t)bz * p.zs;
            if (p.mode == 2) atomicAdd(d, v[e]);
            else *d = p.mode ? *d + v[e] : v[e];
            if (p.dup && k == 0) {   // a second destination of the same sums (a block's conv2 and shortcut share dY: one bias gradient, two biases)
                if (p.mode == 2) atomicAdd(p.dup + jcol, v[e]);
                else p.dup[jcol] = p.mode ? p.dup[jcol] + v[e] : v[e];
            }
        }
    }
}
static __global__ __launch_bounds__(256) void rows_fold2_kernel(RowsFoldArgs p) {
    __shared__ float4 red[256];
    rows_fold2_body(p, blockIdx.x, blockIdx.z, red);
}
// up to three independent folds (one group each) in one launch: blockIdx.y picks the job
static __global__ __launch_bounds__(256) void rows_fold_multi_kernel(RowsFoldArgs a, RowsFoldArgs b, RowsFoldArgs c) {
    __shared__ float4 red[256];
    const RowsFoldArgs& p = blockIdx.y == 0 ? a : (blockIdx.y == 1 ? b : c);
    if (!p.src || (int)blockIdx.x >= p.nbx) return;
    rows_fold2_body(p, blockIdx.x, 0, red);
}
// floats of tmp a fold of (R rows, L columns, Z groups) needs
static inline long long rows_fold_tmp_floats(int R, int L, int Z) { return R <= L2I_FOLD_DIRECT ? 0 : (long long)Z * L2I_FOLD_CHUNKS * L; }
// The direct fold's arguments (R <= L2I_FOLD_DIRECT): columns [0, C0) go to dst0[z * zs + i], columns [C0, L) to dst1[z * zs + i - C0]
// (dst1 null: nowhere); L % 4 == 0, part 16-byte aligned; ld: floats between two rows (0: L), a group's rows are R * ld apart.
static inline RowsFoldArgs rows_fold_args4(const float* part, int R, int L, int Z, float* d0, float* d1, float* d2, float* d3, int seg, int segv, long long zs,
                                           int mode, float* dup, int ld) {
    RowsFoldArgs a;
    a.src = part; a.R = R; a.L = L; a.dst[0] = d0; a.dst[1] = d1; a.dst[2] = d2; a.dst[3] = d3; a.seg = seg > 0 ? seg : L; a.segv = segv;
    a.zs = zs; a.mode = mode; a.dup = dup; a.ld = ld > 0 ? ld : L;
    int cl = 16;
    const int L4 = L / 4;
    while (cl > 1 && (long long)((L4 + cl - 1) / cl) * Z < 128 && 256 / cl < R) cl >>= 1;
    a.cl = cl; a.nbx = (L4 + cl - 1) / cl;
    return a;
}
static inline RowsFoldArgs rows_fold_args(const float* part, int R, int L, int Z, float* dst0, float* dst1, int C0, long long zs, int mode, float* dup, int ld) {
    return rows_fold_args4(part, R, L, Z, dst0, dst1, nullptr, nullptr, dst1 ? C0 : L, C0, zs, mode, dup, ld);
}
// several direct folds (R <= L2I_FOLD_DIRECT, one group each) as ONE launch; jobs with src == null are skipped
static inline void rows_fold_multi(const RowsFoldArgs& a, const RowsFoldArgs& b, const RowsFoldArgs& c, hipStream_t stream) {
    int nbx = 0, ny = 0;
    const RowsFoldArgs* js[3] = {&a, &b, &c};
    for (int k = 0; k < 3; ++k)
        if (js[k]->src) { ny = k + 1; if (js[k]->nbx > nbx) nbx = js[k]->nbx; }
    if (ny) hipLaunchKernelGGL(rows_fold_multi_kernel, dim3(nbx, ny, 1), dim3(256), 0, stream, a, b, c);
}
static inline void rows_fold(const float* part, int R, int L, int Z, float* dst0, float* dst1, int C0, long long zs, int mode, float* tmp, hipStream_t stream,
                             float* dup = nullptr, int ld = 0) {
    if (ld <= 0) ld = L;
    if (R > L2I_FOLD_DIRECT) {
        const int rc = (R + L2I_FOLD_CHUNKS - 1) / L2I_FOLD_CHUNKS, nch = (R + rc - 1) / rc;
        hipLaunchKernelGGL(rows_fold1_kernel, dim3((L / 4 + 63) / 64, nch, Z), dim3(256), 0, stream, part, R, L, rc, tmp, ld);
        part = tmp; R = nch; ld = L;
    }
    const RowsFoldArgs a = rows_fold_args(part, R, L, Z, dst0, dst1, C0, zs, mode, dup, ld);
    hipLaunchKernelGGL(rows_fold2_kernel, dim3(a.nbx, 1, Z), dim3(256), 0, stream, a);
}

// ---------------------------------------------------------------- clearing a buffer from inside the library
// NOT hipMemsetAsync: a memset node captured into a HIP graph in front of a kernel that accumulates into the same buffer with
// atomics (split-K results, weight-gradient slices) is not ordered / executed reliably on replay with this runtime -- measured
// (tools/parity/graph_idempotence.py): the FIRST replay of such a launch is correct (fresh graph-pool memory happens to be zero), every
// later one adds onto stale or half-cleared contents (relative errors 0.15 ... 1e27). A kernel node is ordered like any other.
static __global__ __launch_bounds__(256) void l2i_zero_kernel(float4* __restrict__ p4, long long n4, float* __restrict__ tail, int ntail) {
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x, step = (long long)gridDim.x * 256;
    for (long long i = i0; i < n4; i += step) p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i0 < ntail) tail[i0] = 0.f;
}
static inline hipError_t l2i_zero_async(void* p, size_t bytes, hipStream_t stream) {   // p 4-byte aligned, bytes a multiple of 4
    if (bytes == 0) return hipSuccess;
    if (((size_t)p & 3) || (bytes & 3)) return hipErrorInvalidValue;
    float* f = (float*)p;
    size_t nf = bytes / 4, head = 0;
    while (((size_t)(f + head) & 15) && head < nf) ++head;            // floats in front of the first 16-byte boundary
    if (head) { hipLaunchKernelGGL(l2i_zero_kernel, dim3(1), dim3(256), 0, stream, (float4*)nullptr, 0LL, f, (int)head); }
    const size_t n4 = (nf - head) / 4, ntail = (nf - head) % 4;
    long long nblk = (long long)((n4 + 255) / 256);
    if (nblk > 2048) nblk = 2048;
    if (nblk < 1) nblk = 1;
    hipLaunchKernelGGL(l2i_zero_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, (float4*)(f + head), (long long)n4, f + head + 4 * n4, (int)ntail);
    return hipGetLastError();
}

// ---------------------------------------------------------------- optional per-launch timing (l2i_timing / l2i_timing_read)
// bench.py's roofline leg needs the duration of every conv / weight-gradient launch inside a timed iteration. A pair of
// hipEventRecord calls around a launch also brackets the launch latency (~4 us per launch, 1.2 ms per iteration over the
// ~250 conv launches: measured 550 TFLOP/s where rocprofv3's kernel durations give 596). hipExtLaunchKernelGGL attaches the
// start / stop events to the DISPATCH itself, so their timestamps are the kernel's own begin and end -- what rocprofv3
// reports -- on the stream the kernel is launched on.
#include <hip/hip_ext.h>
#include <vector>
struct L2iTimer {
    bool on = false;
    std::vector<hipEvent_t> start[2], stop[2];   // class 0: l2i_conv2d_fwd launches, 1: l2i_conv2d_wgrad launches
};
inline L2iTimer g_l2i_timer;
#define L2I_LAUNCH(CLS, KERNEL, GRID, BLOCK, LDS, STREAM, ...)                                                         \
    do {                                                                                                               \
        if (g_l2i_timer.on) {                                                                                          \
            hipEvent_t e0_, e1_;                                                                                       \
            (void)hipEventCreate(&e0_);                                                                                \
            (void)hipEventCreate(&e1_);                                                                                \
            hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, e0_, e1_, 0, __VA_ARGS__);                         \
            g_l2i_timer.start[CLS].push_back(e0_);                                                                     \
            g_l2i_timer.stop[CLS].push_back(e1_);                                                                      \
        } else {                                                                                                       \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__);                                         \
        }                                                                                                              \
    } while (0)

static inline int l2i_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? L2I_OK : L2I_ERR_LAUNCH;
}

// ---------------------------------------------------------------- wave-level timestamps (builds with -DL2I_TRACE only)
// tools/perf/conv_trace.py: slot 0 kernel entry, 1 before the reduction loop, 2 after it, 3 end of the epilogue --
// s_memrealtime ticks (100 MHz) of every wave of the LAST launch of a translation unit, plus (XCC_ID << 16 | HW_ID) of the
// wave. L2I_TRACE_DEFINE(tag) in a .hip file defines its buffers and the readers l2i_trace_read_<tag> / l2i_trace_ids_<tag>.
#ifdef L2I_TRACE
#define L2I_TRACE_WAVES (8192 * 8)
#define L2I_TRACE_DEFINE(TAG)                                                                                          \
    __device__ long long g_l2i_trace[L2I_TRACE_WAVES * 4];                                                             \
    __device__ unsigned g_l2i_trace_id[L2I_TRACE_WAVES];                                                               \
    extern "C" int l2i_trace_read_##TAG(long long* host, int nwaves) {                                                 \
        return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_l2i_trace), sizeof(long long) * 4 * (size_t)nwaves) == hipSuccess ? L2I_OK : L2I_ERR_LAUNCH; \
    }                                                                                                                  \
    extern "C" int l2i_trace_ids_##TAG(unsigned* host, int nwaves) {                                                   \
        return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_l2i_trace_id), sizeof(unsigned) * (size_t)nwaves) == hipSuccess ? L2I_OK : L2I_ERR_LAUNCH; \
    }
#define L2I_TR(SLOT)                                                                                                   \
    do {                                                                                                               \
        const int w_ = (int)blockIdx.x * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);                             \
        if ((threadIdx.x & 63) == 0 && w_ < L2I_TRACE_WAVES) {                                                         \
            const long long t_ = (long long)wall_clock64();                                                            \
            if ((SLOT) == 0) {                                                                                         \
                const unsigned xcc_ = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));                            \
                const unsigned hw_ = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));                             \
                g_l2i_trace_id[w_] = (xcc_ << 16) | (hw_ & 0xffffu);                                                   \
            }                                                                                                          \
            g_l2i_trace[w_ * 4 + (SLOT)] = t_;                                                                         \
        }                                                                                                              \
    } while (0)
#else
#define L2I_TRACE_DEFINE(TAG)
#define L2I_TR(SLOT) do { } while (0)
#endif

