// Micro-benchmark (not product code): times the library's own kernels directly (no C-ABI, no torch).
#include "../../minnow_amd/csrc/mnw_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using namespace mnw;
__global__ void k_fill(float* x, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        x[i] = 100.0f + (h >> 8) * (50.0f / 16777216.0f);
    }
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    f(); CHK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < reps; ++r) { CHK(hipEventRecord(a)); f(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b)); float ms; CHK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best;
}
#include <unistd.h>
template <typename F> float timeit_idle(F f, int reps, int us) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    f(); CHK(hipDeviceSynchronize());
    float best = 1e9, tot = 0;
    for (int r = 0; r < reps; ++r) { usleep(us); CHK(hipEventRecord(a)); f(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b)); float ms; CHK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; tot += ms; }
    printf("   [idle %d us between launches: best %.1f us avg %.1f us]\n", us, best * 1e3, tot / reps * 1e3);
    return best;
}
int main(int argc, char** argv) {
    long n = argc > 1 ? atol(argv[1]) : 100000000L;
    float* x; uint8_t* out; float* y; DevJob* dj; uint64_t* stats; Meta* meta; JobSync* sy;
    CHK(hipMalloc(&x, n * 4)); CHK(hipMalloc(&out, n * 4 + 64)); CHK(hipMalloc(&y, n * 4));
    CHK(hipMalloc(&dj, sizeof(DevJob))); CHK(hipMalloc(&stats, 64 * 8 * 8)); CHK(hipMalloc(&meta, sizeof(Meta))); CHK(hipMalloc(&sy, sizeof(JobSync))); CHK(hipMemset(sy, 0, sizeof(JobSync)));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, x, n); CHK(hipDeviceSynchronize());
    DevJob j; memset(&j, 0, sizeof j);
    j.src = x; j.dst = out; j.n = n; j.cap = n * 4; j.pixels = 250000; j.low = 0; j.high = 250; j.dxq = 0.001f; j.rdx = 1.0 / (double)0.001f; j.xh = 125.0f;
    j.kind = 11; j.periodic = 1; j.mode = kModeStats; j.tile0 = 0;
    CHK(hipMemcpy(dj, &j, sizeof j, hipMemcpyHostToDevice));
    CHK(hipMemset(stats, 0, 64 * 64));
    long tiles = (n + kTile - 1) / kTile; long tilesd = (n + kTileD - 1) / kTileD;
    for (int us : {0, 50, 500, 5000}) timeit_idle([&] { CHK(hipMemsetAsync(sy, 0, sizeof(JobSync), 0)); launch_reduce_f32(0, 2048, dj, 1, tiles, stats, sy, meta); }, 10, us);
    {
        hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        float t = timeit([&] { CHK(hipMemsetAsync(sy, 0, sizeof(JobSync), st)); launch_reduce_f32(st, 2048, dj, 1, tiles, stats, sy, meta); CHK(hipStreamSynchronize(st)); }, 10);
        printf("   [non-blocking stream, events on null stream incl. sync: %.1f us]\n", t * 1e3);
        hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
        float best = 1e9;
        for (int r = 0; r < 10; ++r) { CHK(hipEventRecord(a, st)); CHK(hipMemsetAsync(sy, 0, sizeof(JobSync), st)); launch_reduce_f32(st, 2048, dj, 1, tiles, stats, sy, meta); CHK(hipEventRecord(b, st)); CHK(hipEventSynchronize(b)); float ms; CHK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
        printf("   [non-blocking stream, events on that stream: %.1f us]\n", best * 1e3);
        best = 1e9;
        for (int r = 0; r < 10; ++r) { CHK(hipMemsetAsync(stats, 0, 64 * 64, st)); CHK(hipEventRecord(a, st)); CHK(hipMemsetAsync(sy, 0, sizeof(JobSync), st)); launch_reduce_f32(st, 2048, dj, 1, tiles, stats, sy, meta); CHK(hipEventRecord(b, st)); CHK(hipEventSynchronize(b)); float ms; CHK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
        printf("   [memsetAsync + event + kernel + event: %.1f us]\n", best * 1e3);
    }
    for (int g : {1024, 2048}) {
        float t1 = timeit([&] { CHK(hipMemsetAsync(sy, 0, sizeof(JobSync), 0)); launch_reduce_f32(0, g, dj, 1, tiles, stats, sy, meta); }, 10);
        float t2 = timeit([&] { launch_pack_f32(0, g > 1024 ? 1024 : g, dj, 1, tiles, sy); }, 10);
        Meta m; CHK(hipMemcpy(&m, meta, sizeof m, hipMemcpyDeviceToHost));
        DevJob d2 = j; d2.src = out; d2.dst = y; d2.cap = m.nbytes; d2.min = m.min; d2.bits = (int)m.bits; d2.mode = kModeExplicit; d2.seed = 1;
        DevJob* dj2; CHK(hipMalloc(&dj2, sizeof(DevJob))); CHK(hipMemcpy(dj2, &d2, sizeof d2, hipMemcpyHostToDevice));
        float t3 = timeit([&] { launch_unpack_f32(0, g, true, false, dj2, 1, tilesd); }, 10);
        printf("n=%ld grid=%4d reduce %.1f us (%.2f TB/s)  pack %.1f us (bits %ld, %.2f TB/s rd+wr)  unpack %.1f us (%.2f TB/s)\n", n, g,
               t1 * 1e3, n * 4 / t1 / 1e9, t2 * 1e3, (long)m.bits, (n * 4 + m.nbytes) / t2 / 1e9, t3 * 1e3, (n * 4 + m.nbytes) / t3 / 1e9);
        CHK(hipFree(dj2));
    }
    return 0;
}
