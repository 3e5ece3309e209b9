// Micro-benchmark (not product code): HBM read bandwidth of a min/max reduction over float32 under
// different workgroup->address mappings, to choose the mapping of the reduce / pack kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define GLOBAL __attribute__((address_space(1)))
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int V>   // V float4 loads per thread per step
__global__ __launch_bounds__(256) void k_contig(const float* x, long n, float* out) {
    const long per = (n / (V * 1024)) / gridDim.x * (V * 1024);
    long c = per * blockIdx.x; const long e = c + per;
    float mn = 1e30f, mx = -1e30f;
    for (; c < e; c += V * 1024) {
        f32x4 v[V];
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] = *(const GLOBAL f32x4*)(x + c + 4 * (k * 256 + threadIdx.x));
#pragma unroll
        for (int k = 0; k < V; ++k) { mn = fminf(fminf(mn, v[k].x), fminf(v[k].y, fminf(v[k].z, v[k].w))); mx = fmaxf(fmaxf(mx, v[k].x), fmaxf(v[k].y, fmaxf(v[k].z, v[k].w))); }
    }
    if (mn == 123.456f) out[0] = mx;
}
template <int V>
__global__ __launch_bounds__(256) void k_stride(const float* x, long n, float* out) {
    float mn = 1e30f, mx = -1e30f;
    const long step = (long)gridDim.x * V * 1024;
    for (long c = (long)blockIdx.x * V * 1024; c + V * 1024 <= n; c += step) {
        f32x4 v[V];
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] = *(const GLOBAL f32x4*)(x + c + 4 * (k * 256 + threadIdx.x));
#pragma unroll
        for (int k = 0; k < V; ++k) { mn = fminf(fminf(mn, v[k].x), fminf(v[k].y, fminf(v[k].z, v[k].w))); mx = fmaxf(fmaxf(mx, v[k].x), fmaxf(v[k].y, fmaxf(v[k].z, v[k].w))); }
    }
    if (mn == 123.456f) out[0] = mx;
}

struct FS { float mn, mx, mxlo, mnhi; bool bad; };
__device__ __forceinline__ void fadd(FS& s, float x, float xh) {
    s.bad |= !(fabsf(x) < INFINITY);
    s.mn = fminf(s.mn, x); s.mx = fmaxf(s.mx, x);
    const bool lo = x < xh;
    s.mxlo = fmaxf(s.mxlo, lo ? x : -INFINITY);
    s.mnhi = fminf(s.mnhi, lo ? INFINITY : x);
}
template <int V>
__global__ __launch_bounds__(256, 8) void k_fstat(const float* x, long n, float* out, float xh) {
    const long per = (n / (V * 1024)) / gridDim.x * (V * 1024);
    long c = per * blockIdx.x; const long e = c + per;
    FS s = {INFINITY, -INFINITY, -INFINITY, INFINITY, false};
    for (; c < e; c += V * 1024) {
        f32x4 v[V];
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] = *(const GLOBAL f32x4*)(x + c + 4 * (k * 256 + threadIdx.x));
#pragma unroll
        for (int k = 0; k < V; ++k) { fadd(s, v[k].x, xh); fadd(s, v[k].y, xh); fadd(s, v[k].z, xh); fadd(s, v[k].w, xh); }
    }
    if (s.mn == 123.456f || s.bad) out[0] = s.mx + s.mxlo + s.mnhi;
}
__global__ void k_fill(float* x, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        x[i] = (h >> 8) * (250.0f / 16777216.0f);
    }
}

template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    f(); CHK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < reps; ++r) { CHK(hipEventRecord(a)); f(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b)); float ms; CHK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best;
}

int main(int argc, char** argv) {
    long n = argc > 1 ? atol(argv[1]) : 100000000L;
    n = n / 16384 * 16384;
    float *x, *out;
    CHK(hipMalloc(&x, n * 4)); CHK(hipMalloc(&out, 64));
    CHK(hipMemset(x, 0x3c, n * 4));
    if (argc > 2) { hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, x, n); CHK(hipDeviceSynchronize()); printf("random fill\n"); }
    for (int g : {1024, 2048}) {
        float t5 = timeit([&] { hipLaunchKernelGGL(k_fstat<4>, dim3(g), dim3(256), 0, 0, x, n, out, 125.0f); }, 10);
        printf("n=%ld grid=%5d  fstat V4 %.1f us (%.2f TB/s)\n", n, g, t5 * 1e3, n * 4 / t5 / 1e9);
    }
    int grids[] = {1024, 2048};
    for (int g : grids) {
        float t1 = timeit([&] { hipLaunchKernelGGL(k_contig<2>, dim3(g), dim3(256), 0, 0, x, n, out); }, 10);
        float t2 = timeit([&] { hipLaunchKernelGGL(k_contig<4>, dim3(g), dim3(256), 0, 0, x, n, out); }, 10);
        float t3 = timeit([&] { hipLaunchKernelGGL(k_stride<2>, dim3(g), dim3(256), 0, 0, x, n, out); }, 10);
        float t4 = timeit([&] { hipLaunchKernelGGL(k_stride<4>, dim3(g), dim3(256), 0, 0, x, n, out); }, 10);
        printf("n=%ld grid=%5d  contig V2 %.1f us (%.2f TB/s)  contig V4 %.1f us (%.2f TB/s)  stride V2 %.1f us (%.2f TB/s)  stride V4 %.1f us (%.2f TB/s)\n",
               n, g, t1 * 1e3, n * 4 / t1 / 1e9, t2 * 1e3, n * 4 / t2 / 1e9, t3 * 1e3, n * 4 / t3 / 1e9, t4 * 1e3, n * 4 / t4 / 1e9);
    }
    return 0;
}
