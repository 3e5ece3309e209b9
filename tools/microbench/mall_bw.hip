// Micro-benchmark (not product code): is a re-read of a buffer that fits the 256 MiB Infinity Cache
// faster than an HBM stream, and do cache-served reads overlap with HBM writes?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define GLOBAL __attribute__((address_space(1)))
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// reads `passes` times over x[0..n), optionally writes wfrac/16 of the bytes it reads to y
template <int WSIXTEENTHS>
__global__ __launch_bounds__(256, 8) void k_rw(const float* x, float* y, long n, int passes, float* out) {
    float acc = 0;
    const long step = (long)gridDim.x * 4096;
    for (int p = 0; p < passes; ++p) {
        for (long c = (long)blockIdx.x * 4096; c + 4096 <= n; c += step) {
            f32x4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *(const GLOBAL f32x4*)(x + c + 4 * (k * 256 + threadIdx.x));
#pragma unroll
            for (int k = 0; k < 4; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
            if (WSIXTEENTHS > 0) {
                // write WSIXTEENTHS/16 of the bytes: the first WSIXTEENTHS/4 vectors
#pragma unroll
                for (int k = 0; k < WSIXTEENTHS / 4; ++k) *(GLOBAL f32x4*)(y + (long)p * 0 + c / 4096 * (1024 * WSIXTEENTHS / 4) + 4 * (k * 256 + threadIdx.x)) = v[k];
            }
        }
    }
    if (acc == 123.456f) out[0] = acc;
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    f(); CHK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < reps; ++r) { CHK(hipEventRecord(a)); f(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b)); float ms; CHK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best;
}
int main() {
    const long big = 1L << 30;   // floats: 4 GiB
    float *x, *y, *out;
    CHK(hipMalloc(&x, big * 4)); CHK(hipMalloc(&y, big * 4)); CHK(hipMalloc(&out, 64));
    CHK(hipMemset(x, 0x3c, big * 4)); CHK(hipMemset(y, 0, big * 4));
    for (long mb : {16L, 32L, 64L, 128L, 192L, 512L, 2048L}) {
        const long n = mb * 1024 * 1024 / 4;
        const int passes = (int)(4096 / mb);      // 4 GiB of reads in total
        for (int g : {2048}) {
            float t0 = timeit([&] { hipLaunchKernelGGL(k_rw<0>, dim3(g), dim3(256), 0, 0, x, y, n, passes, out); }, 5);
            float t1 = timeit([&] { hipLaunchKernelGGL(k_rw<8>, dim3(g), dim3(256), 0, 0, x, y, n, passes, out); }, 5);
            const double rd = (double)n * 4 * passes;
            printf("footprint %4ld MiB x %3d passes: read-only %.0f us = %.2f TB/s | read + write 1/2 as much (fresh lines each pass over the footprint): %.0f us = %.2f TB/s read, %.2f TB/s read+write\n",
                   mb, passes, t0 * 1e3, rd / t0 / 1e9, t1 * 1e3, rd / t1 / 1e9, rd * 1.5 / t1 / 1e9);
        }
    }
    return 0;
}
