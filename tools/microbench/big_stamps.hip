// Diagnostic (not product): the single-read encode kernel on large blocks (two-level record sweep): kernel time, per-phase
// cycle sums of wave 0, and what sweep_big_f32 spends -- calls, poll iterations, cycles in its two phases.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off big_stamps.hip -o big_stamps && ./big_stamps n items_per_block [grid]
#define MNW_STAMPS 1
#include "../../minnow_amd/csrc/mnw_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using namespace mnw;
__global__ void k_fill(float* x, long n) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        x[i] = (h >> 8) * (249.9f / 16777216.0f);
    }
}
int main(int argc, char** argv) {
    long n = argc > 1 ? atol(argv[1]) : 100000000L;
    long B = (argc > 2 ? atol(argv[2]) : 2048) * (long)kItem;
    int grid = argc > 3 ? atoi(argv[3]) : 1024;
    float* x; uint8_t* out; unsigned long long* sb;
    CHK(hipMalloc(&x, n * 4)); CHK(hipMalloc(&out, n * 4 + 64));
    CHK(hipMalloc(&sb, 8 * 8 * 4096)); CHK(hipMemset(sb, 0, 8 * 8 * 4096));
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &sb, sizeof(sb)));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, x, n); CHK(hipDeviceSynchronize());
    int nj = (int)((n + B - 1) / B);
    std::vector<DevJob> hj(nj); long items = 0, groups = 0, max_ni = 0;
    for (int k = 0; k < nj; ++k) {
        DevJob& j = hj[k]; memset(&j, 0, sizeof j);
        j.src = x + k * B; j.dst = out + k * B * 4; j.n = (k == nj - 1) ? n - k * B : B; j.cap = j.n * 4;
        j.pixels = 250000; j.low = 0; j.high = 250; j.dxq = 0.001f; j.rdx = 1.0 / (double)0.001f; j.xh = 125.0f; j.kind = 11; j.periodic = 1;
        const long ni = (j.n + kItem - 1) / kItem;
        j.tile0 = items; j.grp0 = groups; items += ni; { const long R = (ni + kGroupItems - 1) / kGroupItems; groups += R * R; } if (ni > max_ni) max_ni = ni;
    }
    DevJob* dj; ItemRec* recs; uint32_t* ab; Meta* dm; int* ij;
    std::vector<int> hij(items); for (int k = 0; k < nj; ++k) for (long i = hj[k].tile0; i < hj[k].tile0 + (hj[k].n + kItem - 1) / kItem; ++i) hij[i] = k;
    CHK(hipMalloc(&ij, 4 * items)); CHK(hipMemcpy(ij, hij.data(), 4 * items, hipMemcpyHostToDevice));
    CHK(hipMalloc(&dj, sizeof(DevJob) * nj)); CHK(hipMalloc(&recs, sizeof(ItemRec) * (items + groups))); CHK(hipMalloc(&ab, 128)); CHK(hipMalloc(&dm, sizeof(Meta) * nj));
    CHK(hipMemcpy(dj, hj.data(), sizeof(DevJob) * nj, hipMemcpyHostToDevice)); CHK(hipMemset(recs, 0, sizeof(ItemRec) * (items + groups))); CHK(hipMemset(ab, 0, 128));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    uint32_t gen = 1;
    const int weighted = max_ni > grid ? 0 : 1, big = max_ni > kMaxSweep ? 1 : 0;
    auto launch = [&]() { CHK(hipMemset(ab, 0, 128)); launch_encode_reg_f32(0, grid, dj, nj, items, ij, recs, recs + items, gen++, ab, ab + 8, dm, weighted, -1, 0, big); };
    for (int r = 0; r < 3; ++r) launch();
    CHK(hipDeviceSynchronize()); CHK(hipMemset(sb, 0, 8 * 8 * 4096));
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}; CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_big_diag), z, sizeof z));
    CHK(hipMemset(ab, 0, 128)); CHK(hipDeviceSynchronize()); CHK(hipEventRecord(a)); launch(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h(8 * grid); CHK(hipMemcpy(h.data(), sb, 8 * 8 * grid, hipMemcpyDeviceToHost));
    Meta m0; CHK(hipMemcpy(&m0, dm, sizeof m0, hipMemcpyDeviceToHost)); uint32_t abv; CHK(hipMemcpy(&abv, ab, 4, hipMemcpyDeviceToHost));
    const char* names[6] = {"loads wait+lane stats", "reduce+fold+barrier+publish", "descriptor|quantize+barrier", "rebase+pack+barrier", "store+barrier", "park+issue next"};
    double cyc = 0, rt = 0; for (int w = 0; w < grid; ++w) { cyc += h[w * 8 + 6]; rt += h[w * 8 + 7]; }
    const double ghz = cyc / rt * 0.1;
    printf("n=%ld items/block=%ld blocks=%d items=%ld grid=%d items/WG=%.1f weighted=%d big=%d: kernel %.1f us (stamped build) bits=%ld abort=%u\n", n, B / kItem, nj, items, grid,
           (double)items / grid, weighted, big, ms * 1e3, (long)m0.bits, abv);
    printf("  workgroup lifetime mean %.2f us, shader clock %.2f GHz\n", rt / grid / 100.0, ghz);
    for (int i = 0; i < 6; ++i) {
        double s2 = 0; for (int w = 0; w < grid; ++w) s2 += h[w * 8 + i];
        printf("  %-26s %.2f us per WG = %.2f us per item\n", names[i], s2 / grid / ghz / 1e3, s2 / grid / ghz / 1e3 / ((double)items / grid));
    }
    unsigned long long d[8]; CHK(hipMemcpyFromSymbol(d, HIP_SYMBOL(g_big_diag), sizeof d));
    if (d[0]) printf("  sweep_big_f32: %llu calls, %.2f group polls and %.2f block polls per call, %.2f us per call (groups %.2f us, block %.2f us)\n", d[0],
                     (double)d[1] / d[0], (double)d[2] / d[0], d[3] / (double)d[0] / ghz / 1e3, d[4] / (double)d[0] / ghz / 1e3, d[5] / (double)d[0] / ghz / 1e3);
    return 0;
}
