// Micro-benchmark (not product): what a plain float4 copy reaches on this GPU (MI355X_MICROARCH.md quotes 6.29 TB/s),
// for several launch shapes -- the yardstick for the codec's mixed read+write streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define GLOBAL __attribute__((address_space(1)))
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// each workgroup copies chunks of U*4 KB; STRIDED: chunks dealt round-robin; else one contiguous range per workgroup
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(const float* __restrict__ x, float* __restrict__ y, long n, int strided) {
    const long chunk = (long)U * 1024;                 // floats per chunk
    const long nchunks = n / chunk;
    long c0, c1, step;
    if (strided) { c0 = blockIdx.x; c1 = nchunks; step = gridDim.x; }
    else { c0 = nchunks * blockIdx.x / gridDim.x; c1 = nchunks * (blockIdx.x + 1) / gridDim.x; step = 1; }
    for (long c = c0; c < c1; c += step) {
        f32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const GLOBAL f32x4* p = (const GLOBAL f32x4*)(x + c * chunk + 4 * (k * 256 + threadIdx.x));
            v[k] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            GLOBAL f32x4* q = (GLOBAL f32x4*)(y + c * chunk + 4 * (k * 256 + threadIdx.x));
            if (NT) __builtin_nontemporal_store(v[k], q); else *q = v[k];
        }
    }
}
// the classic one element (16 B) per thread, huge grid
__global__ __launch_bounds__(256) void k_copy1(const float* __restrict__ x, float* __restrict__ y, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) ((GLOBAL f32x4*)y)[i] = ((const GLOBAL f32x4*)x)[i];
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    f(); f(); CHK(hipDeviceSynchronize());
    float tot = 0;
    for (int r = 0; r < reps; ++r) { CHK(hipEventRecord(a)); f(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b)); float ms; CHK(hipEventElapsedTime(&ms, a, b)); tot += ms; }
    return tot / reps;
}
int main() {
    const long n = 1L << 30;   // floats: 4 GiB in, 4 GiB out
    float *x, *y;
    CHK(hipMalloc(&x, n * 4)); CHK(hipMalloc(&y, n * 4));
    CHK(hipMemset(x, 0x3c, n * 4)); CHK(hipMemset(y, 0, n * 4));
    const double bytes = 8.0 * n;
    { float t = timeit([&] { hipLaunchKernelGGL(k_copy1, dim3((unsigned)(n / 4 / 256)), dim3(256), 0, 0, x, y, n / 4); }, 5);
      printf("one float4 per thread, grid %ld: %.1f us = %.2f TB/s\n", n / 4 / 256, t * 1e3, bytes / t / 1e9); }
    { float t = timeit([&] { CHK(hipMemcpyAsync(y, x, n * 4, hipMemcpyDeviceToDevice, 0)); }, 5);
      printf("hipMemcpy D2D: %.1f us = %.2f TB/s\n", t * 1e3, bytes / t / 1e9); }
    for (int strided : {1, 0})
        for (int g : {256, 512, 1024, 2048, 4096, 8192}) {
#define RUN(U, NT) { float t = timeit([&] { hipLaunchKernelGGL((k_copy<U, NT>), dim3(g), dim3(256), 0, 0, x, y, n, strided); }, 5); \
    printf("  %s grid %5d  %d x 16 B in flight per lane%s: %8.1f us = %.2f TB/s\n", strided ? "round-robin" : "contiguous ", g, U, NT ? ", nt" : "    ", t * 1e3, bytes / t / 1e9); }
            RUN(2, false) RUN(4, false) RUN(8, false) RUN(16, false) RUN(8, true)
        }
    return 0;
}
