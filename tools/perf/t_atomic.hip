#include <hip/hip_runtime.h>
#include <stdio.h>
// throughput of f32 atomicAdd (no return) to distinct, lane-consecutive addresses vs plain store vs load-add-store
__global__ void k_atomic(float* p, long n, int reps) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int r = 0; r < reps; ++r) { long j = (i + (long)r * gridDim.x * blockDim.x) % n; atomicAdd(p + j, 1.0f); }
}
__global__ void k_store(float* p, long n, int reps) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int r = 0; r < reps; ++r) { long j = (i + (long)r * gridDim.x * blockDim.x) % n; p[j] = 1.0f; }
}
__global__ void k_rmw(float* p, long n, int reps) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int r = 0; r < reps; ++r) { long j = (i + (long)r * gridDim.x * blockDim.x) % n; p[j] += 1.0f; }
}
// strided like an MFMA C-tile row store: lanes 0..31 consecutive floats, lanes 32..63 another row
__global__ void k_atomic_tile(float* p, long n, int ld, int reps) {
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long base = ((long)blockIdx.x * 4 + wave) * 32 * ld;
    for (int r = 0; r < reps; ++r) {
        long j = (base + (long)(r % 16 + (lane >> 5) * 16) * ld + (r / 16) * 32 + (lane & 31)) % n;
        atomicAdd(p + j, 1.0f);
    }
}
int main() {
    long n = 64L << 20; float* p; hipMalloc(&p, n * 4); hipMemset(p, 0, n * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int reps = 64, blocks = 4096, thr = 256;
    for (int v = 0; v < 4; ++v) {
        for (int it = 0; it < 2; ++it) {
            hipEventRecord(a);
            if (v == 0) k_atomic<<<blocks, thr>>>(p, n, reps);
            if (v == 1) k_store<<<blocks, thr>>>(p, n, reps);
            if (v == 2) k_rmw<<<blocks, thr>>>(p, n, reps);
            if (v == 3) k_atomic_tile<<<blocks, thr>>>(p, n, 9216, reps);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        double ops = (double)blocks * thr * reps;
        printf("variant %d: %.3f ms  %.1f Gops/s  %.2f TB/s(4B)\n", v, ms, ops / ms / 1e6, ops * 4 / ms / 1e9);
    }
    return 0;
}
