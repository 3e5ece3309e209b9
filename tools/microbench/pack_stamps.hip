// Diagnostic (not product): per-phase cycle sums of the pack pass, wave 0 of every workgroup.
#define MNW_STAMPS 1
#include "../../minnow_amd/csrc/mnw_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using namespace mnw;
__global__ void k_fill(float* x, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        x[i] = (h >> 8) * (249.9f / 16777216.0f);
    }
}
int main(int argc, char** argv) {
    long n = argc > 1 ? atol(argv[1]) : 16777216L;
    int grid = argc > 2 ? atoi(argv[2]) : 1024;
    float* x; uint8_t* out; DevJob* dj; uint64_t* stats; Meta* meta; unsigned long long* sb; JobSync* sy;
    CHK(hipMalloc(&x, n * 4)); CHK(hipMalloc(&out, n * 4 + 64));
    CHK(hipMalloc(&dj, sizeof(DevJob))); CHK(hipMalloc(&stats, kStatSlots * 64)); CHK(hipMalloc(&meta, sizeof(Meta))); CHK(hipMalloc(&sy, sizeof(JobSync))); CHK(hipMemset(sy, 0, sizeof(JobSync)));
    CHK(hipMalloc(&sb, 8 * 8 * 4096)); CHK(hipMemset(sb, 0, 8 * 8 * 4096));
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &sb, sizeof(sb)));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, x, n); CHK(hipDeviceSynchronize());
    DevJob j; memset(&j, 0, sizeof j);
    j.src = x; j.dst = out; j.n = n; j.cap = n * 4; j.pixels = 250000; j.low = 0; j.high = 250; j.dxq = 0.001f; j.rdx = 1.0 / (double)0.001f; j.xh = 125.0f;
    j.kind = 11; j.periodic = 1; j.mode = kModeStats; j.tile0 = 0;
    CHK(hipMemcpy(dj, &j, sizeof j, hipMemcpyHostToDevice));
    CHK(hipMemset(stats, 0, kStatSlots * 64));
    long tiles = (n + kTile - 1) / kTile;
    launch_reduce_f32(0, 2048, dj, 1, tiles, stats, sy, meta);
    for (int r = 0; r < 3; ++r) launch_pack_f32(0, grid, dj, 1, tiles, sy);   // warm
    CHK(hipDeviceSynchronize());
    CHK(hipMemset(sb, 0, 8 * 8 * 4096));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    CHK(hipEventRecord(a)); launch_pack_f32(0, grid, dj, 1, tiles, sy); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h(8 * grid); CHK(hipMemcpy(h.data(), sb, 8 * 8 * grid, hipMemcpyDeviceToHost));
    const char* names[8] = {"preamble", "wait loads+quantize+stage", "prefetch issue+barrier", "transposed read+barrier", "pack+barrier", "read-out+stores+barrier", "", ""};
    printf("n=%ld grid=%d tiles/WG=%.1f kernel %.1f us (stamped build)\n", n, grid, (double)tiles / grid, ms * 1e3);
    double tot = 0;
    for (int i = 0; i < 6; ++i) { double s = 0; for (int w = 0; w < grid; ++w) s += h[w * 8 + i]; tot += s / grid; }
    double cyc = 0, rt = 0; for (int w = 0; w < grid; ++w) { cyc += h[w * 8 + 6]; rt += h[w * 8 + 7]; }
    const double ghz = cyc / rt * 0.1;
    printf("  workgroup lifetime mean %.0f cycles = %.2f us  -> shader clock %.2f GHz\n", cyc / grid, rt / grid / 100.0, ghz);
    for (int i = 0; i < 6; ++i) {
        double s = 0; for (int w = 0; w < grid; ++w) s += h[w * 8 + i];
        printf("  %-28s mean %9.0f cycles/WG  (%.1f%%)  = %.2f us\n", names[i], s / grid, 100.0 * s / grid / tot, s / grid / ghz / 1e3);
    }
    return 0;
}
