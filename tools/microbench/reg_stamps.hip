// Diagnostic (not product): per-phase cycle sums of the single-read encode kernel, wave 0 of every workgroup.
#define MNW_STAMPS 1
#include "../../minnow_amd/csrc/mnw_kernels.hip"
#include <algorithm>
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
    long B = argc > 2 ? atol(argv[2]) : (1L << 20);
    int grid = argc > 3 ? atoi(argv[3]) : 1024;
    int skip = argc > 4 ? atoi(argv[4]) : 0;
    float* x; uint8_t* out; unsigned long long* sb;
    CHK(hipMalloc(&x, n * 4)); CHK(hipMalloc(&out, n * 4 + 64));
    CHK(hipMalloc(&sb, 8 * 8 * 4096)); CHK(hipMemset(sb, 0, 8 * 8 * 4096));
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &sb, sizeof(sb)));
    unsigned long long* sabs; CHK(hipMalloc(&sabs, 16 * 4096)); CHK(hipMemset(sabs, 0, 16 * 4096));
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_abs), &sabs, sizeof(sabs)));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, x, n); CHK(hipDeviceSynchronize());
    int nj = (int)((n + B - 1) / B);
    std::vector<DevJob> hj(nj); long items = 0;
    for (int k = 0; k < nj; ++k) {
        DevJob& j = hj[k]; memset(&j, 0, sizeof j);
        j.src = x + k * B; j.dst = out + k * B * 4; j.n = (k == nj - 1) ? n - k * B : B; j.cap = j.n * 4;
        j.pixels = 250000; j.low = 0; j.high = 250; j.dxq = 0.001f; j.rdx = 1.0 / (double)0.001f; j.xh = 125.0f; j.kind = 11; j.periodic = 1;
        j.tile0 = items; items += (j.n + kItem - 1) / kItem;
    }
    DevJob* dj; ItemRec* recs; uint32_t* ab; Meta* dm; int* ij;
    std::vector<int> hij(items); for (int k = 0; k < nj; ++k) for (long i = hj[k].tile0; i < hj[k].tile0 + (hj[k].n + kItem - 1) / kItem; ++i) hij[i] = k;
    CHK(hipMalloc(&ij, 4 * items)); CHK(hipMemcpy(ij, hij.data(), 4 * items, hipMemcpyHostToDevice));
    CHK(hipMalloc(&dj, sizeof(DevJob) * nj)); CHK(hipMalloc(&recs, sizeof(ItemRec) * items)); CHK(hipMalloc(&ab, 128)); CHK(hipMalloc(&dm, sizeof(Meta) * nj));
    CHK(hipMemcpy(dj, hj.data(), sizeof(DevJob) * nj, hipMemcpyHostToDevice)); CHK(hipMemset(recs, 0, sizeof(ItemRec) * items)); CHK(hipMemset(ab, 0, 128));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    uint32_t gen = 1;
    for (int r = 0; r < 3; ++r) { CHK(hipMemset(ab, 0, 128)); launch_encode_reg_f32(0, grid, dj, nj, items, ij, recs, recs + items, gen++, ab, ab + 8, dm, 1, -1, 0, 0); }
    CHK(hipDeviceSynchronize()); CHK(hipMemset(sb, 0, 8 * 8 * 4096));
    CHK(hipMemset(ab, 0, 128)); CHK(hipDeviceSynchronize()); CHK(hipEventRecord(a)); launch_encode_reg_f32(0, grid, dj, nj, items, ij, recs, recs + items, gen++, ab, ab + 8, dm, 1, -1, 0, 0); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h(8 * grid); CHK(hipMemcpy(h.data(), sb, 8 * 8 * grid, hipMemcpyDeviceToHost));
    Meta m0; CHK(hipMemcpy(&m0, dm, sizeof m0, hipMemcpyDeviceToHost)); uint32_t abv; CHK(hipMemcpy(&abv, ab, 4, hipMemcpyDeviceToHost));
    const char* names[6] = {"loads wait+lane stats", "reduce+fold+barrier+publish", "descriptor|quantize+barrier", "rebase+pack+barrier", "store+barrier", "park+issue next"};
    double cyc = 0, rt = 0; for (int w = 0; w < grid; ++w) { cyc += h[w * 8 + 6]; rt += h[w * 8 + 7]; }
    const double ghz = cyc / rt * 0.1;
    printf("n=%ld block=%ld items=%ld grid=%d items/WG=%.1f kernel %.1f us (stamped build) bits=%ld abort=%u\n", n, B, items, grid, (double)items / grid, ms * 1e3, (long)m0.bits, abv);
    printf("  workgroup lifetime mean %.2f us, shader clock %.2f GHz\n", rt / grid / 100.0, ghz);
    for (int i = 0; i < 6; ++i) {
        double s2 = 0; for (int w = 0; w < grid; ++w) s2 += h[w * 8 + i];
        printf("  %-26s %.2f us per WG = %.2f us per item\n", names[i], s2 / grid / ghz / 1e3, s2 / grid / ghz / 1e3 / ((double)items / grid));
    }
    { std::vector<unsigned long long> ha(2 * grid); CHK(hipMemcpy(ha.data(), sabs, 16 * grid, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull, smax = 0, emin = ~0ull, emax = 0;
      for (int w = 0; w < grid; ++w) { t0 = std::min(t0, ha[2 * w]); }
      for (int w = 0; w < grid; ++w) { smax = std::max(smax, ha[2 * w] - t0); emin = std::min(emin, ha[2 * w + 1] - t0); emax = std::max(emax, ha[2 * w + 1] - t0); }
      for (int x = 0; x < 8; ++x) { double se = 0; int c = 0; unsigned long long mx = 0, mn = ~0ull; for (int w = x; w < grid; w += 8) { se += ha[2 * w + 1] - t0; ++c; mx = std::max(mx, ha[2 * w + 1] - t0); mn = std::min(mn, ha[2 * w + 1] - t0); }
        printf("    w%%8=%d: end mean %.1f min %.1f max %.1f us\n", x, se / c / 100.0, mn / 100.0, mx / 100.0); }
      for (int qd = 0; qd < 8; ++qd) { double se = 0, sl = 0; int c = 0; for (int w = qd * grid / 8; w < (qd + 1) * grid / 8; ++w) { se += ha[2 * w + 1] - t0; sl += ha[2 * w + 1] - ha[2 * w]; ++c; }
        printf("    ids [%d,%d): end mean %.1f us, lifetime mean %.1f us\n", qd * grid / 8, (qd + 1) * grid / 8, se / c / 100.0, sl / c / 100.0); }
      int hist[16] = {0}; for (int w = 0; w < grid; ++w) { int b = (int)((ha[2 * w + 1] - t0) / 1000); if (b > 15) b = 15; hist[b]++; }
      printf("    end-time histogram (10 us bins):"); for (int b = 0; b < 16; ++b) printf(" %d", hist[b]); printf("\n");
      printf("    first 64 WG end times:"); for (int w = 0; w < 64; ++w) printf(" %.0f", (ha[2 * w + 1] - t0) / 100.0); printf("\n");
      printf("  workgroup starts within %.2f us; ends between %.2f and %.2f us after the first start (skip=%d)\n", smax / 100.0, emin / 100.0, emax / 100.0, skip); }
    unsigned long long slow = 0; CHK(hipMemcpyFromSymbol(&slow, HIP_SYMBOL(g_slow_sweeps), sizeof slow));
    printf("  slow-path sweeps (all launches) %llu of %ld items per launch\n", slow, items);
    return 0;
}
