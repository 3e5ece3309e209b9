// Diagnostic (not product): fused encode kernel with phases switched off, to price each phase.
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
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    f(); CHK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < reps; ++r) { CHK(hipEventRecord(a)); f(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b)); float ms; CHK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best;
}
int main(int argc, char** argv) {
    long n = argc > 1 ? atol(argv[1]) : 100000000L;
    float* x; uint8_t* out;
    CHK(hipMalloc(&x, n * 4)); CHK(hipMalloc(&out, n * 4 + 64));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, x, n); CHK(hipDeviceSynchronize());
    DevJob j; memset(&j, 0, sizeof j);
    j.pixels = 250000; j.low = 0; j.high = 250; j.dxq = 0.001f; j.rdx = 1.0 / (double)0.001f; j.xh = 125.0f; j.kind = 11; j.periodic = 1; j.mode = kModeStats;
    for (int ng : {1, 6, 24}) {
        std::vector<DevJob> hj(ng); std::vector<int64_t> gt(ng + 1); std::vector<int> gj(ng + 1);
        long per = (n / ng) / kTile * kTile; int64_t tl = 0;
        for (int k = 0; k < ng; ++k) {
            hj[k] = j; hj[k].src = x + k * per; hj[k].dst = out + k * per * 4; hj[k].n = (k == ng - 1) ? n - k * per : per; hj[k].cap = hj[k].n * 4;
            hj[k].tile0 = tl; gt[k] = tl; gj[k] = k; tl += (hj[k].n + kTile - 1) / kTile;
        }
        gt[ng] = tl; gj[ng] = ng;
        DevJob* djs; int64_t* dgt; int* dgj; uint64_t* dst_; JobSync* dsy; uint32_t* dab; Meta* dm;
        CHK(hipMalloc(&djs, sizeof(DevJob) * ng)); CHK(hipMalloc(&dgt, 8 * (ng + 1))); CHK(hipMalloc(&dgj, 4 * (ng + 1)));
        CHK(hipMalloc(&dst_, ng * kStatSlots * kStatWords * 8)); CHK(hipMalloc(&dsy, sizeof(JobSync) * ng)); CHK(hipMalloc(&dab, 128)); CHK(hipMalloc(&dm, sizeof(Meta) * ng));
        CHK(hipMemcpy(djs, hj.data(), sizeof(DevJob) * ng, hipMemcpyHostToDevice)); CHK(hipMemcpy(dgt, gt.data(), 8 * (ng + 1), hipMemcpyHostToDevice));
        CHK(hipMemcpy(dgj, gj.data(), 4 * (ng + 1), hipMemcpyHostToDevice));
        for (int grid : {512, 1024}) for (int skip : {0, 2, 1}) {
            CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_skip), &skip, sizeof(int)));
            float t = timeit([&] {
                if (skip != 1) { CHK(hipMemsetAsync(dst_, 0, ng * kStatSlots * kStatWords * 8, 0)); CHK(hipMemsetAsync(dsy, 0, sizeof(JobSync) * ng, 0)); }   // pack-only reuses the published words
                CHK(hipMemsetAsync(dab, 0, 128, 0));
                launch_encode_f32(0, grid, djs, ng, dgt, dgj, ng, dst_, dsy, dab, dm); }, 8);
            std::vector<Meta> hm(ng); CHK(hipMemcpy(hm.data(), dm, sizeof(Meta) * ng, hipMemcpyDeviceToHost)); uint32_t ab; CHK(hipMemcpy(&ab, dab, 4, hipMemcpyDeviceToHost));
            printf("n=%ld groups=%2d grid=%4d %-12s %.1f us  (bits %ld status %ld abort %u)\n", n, ng, grid,
                   skip == 0 ? "reduce+pack" : skip == 2 ? "reduce only" : "pack only", t * 1e3, (long)hm[0].bits, (long)hm[0].status, ab);
        }
    }
    return 0;
}
