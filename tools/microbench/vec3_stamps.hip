// Diagnostic (not product): k_encode_reg_vec3 on a C4-shaped file cell -- kernel time and per-phase cycle sums of wave 0.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off vec3_stamps.hip -o vec3_stamps && ./vec3_stamps [n_file sub_cells grid]
#define MNW_STAMPS 1
#include "../../minnow_amd/csrc/mnw_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using namespace mnw;
__global__ void k_fill(float* x, long n, int nf) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < 3 * n; i += gridDim.x * 256L) {
        const long p = i / 3; const int k = (int)(i % 3);
        const int c = k == 0 ? (int)(p % nf) : k == 1 ? (int)((p / nf) % nf) : (int)(p / ((long)nf * nf));
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        float v = c * (250.0f / nf) + ((h >> 8) * (1.0f / 16777216.0f) - 0.5f) * 8.0f;
        if (v < 0) v += 250.0f; if (v >= 250.0f) v -= 250.0f;
        x[i] = v;
    }
}
int main(int argc, char** argv) {
    const long nf = argc > 1 ? atol(argv[1]) : 500, S = argc > 2 ? atol(argv[2]) : 5;
    int grid = argc > 3 ? atoi(argv[3]) : 1024;
    const int weighted = argc > 4 ? atoi(argv[4]) : 0;
    const int deep = argc > 5 ? atoi(argv[5]) : 5;
    const long n = nf * nf * nf, s3 = S * S * S, nsub = nf / S, nsub3 = nsub * nsub * nsub;
    float* x; uint8_t* out; unsigned long long* sb;
    const long slot = (4 * nsub3 + 15) / 16 * 16;
    CHK(hipMalloc(&x, n * 12)); CHK(hipMalloc(&out, 3 * s3 * slot));
    CHK(hipMalloc(&sb, 8 * 8 * 4096)); CHK(hipMemset(sb, 0, 8 * 8 * 4096));
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &sb, sizeof(sb)));
    hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, x, n, (int)nf); CHK(hipDeviceSynchronize());
    std::vector<DevJob> hj(3 * s3);
    for (int k = 0; k < 3; ++k) for (long sc = 0; sc < s3; ++sc) {
        DevJob& j = hj[k * s3 + sc]; memset(&j, 0, sizeof j);
        j.src = x; j.dst = out + (k * s3 + sc) * slot; j.n = nsub3; j.cap = slot;
        j.pixels = 250000; j.low = 0; j.high = 250; j.dxq = 0.001f; j.rdx = 1.0 / (double)0.001f; j.xh = 125.0f; j.kind = 11; j.periodic = 1;
    }
    const long vt = vec3_tiles(nf, S), nrec = vec3_reg_recs(nf, S);
    if (vt < grid) grid = (int)vt;
    DevJob* dj; ItemRec* recs; uint32_t* ab; Meta* dm;
    CHK(hipMalloc(&dj, sizeof(DevJob) * hj.size())); CHK(hipMalloc(&recs, sizeof(ItemRec) * nrec)); CHK(hipMalloc(&ab, 128)); CHK(hipMalloc(&dm, sizeof(Meta) * hj.size()));
    CHK(hipMemcpy(dj, hj.data(), sizeof(DevJob) * hj.size(), hipMemcpyHostToDevice)); CHK(hipMemset(recs, 0, sizeof(ItemRec) * nrec)); CHK(hipMemset(ab, 0, 128));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    uint32_t gen = 1;
    auto launch = [&]() { launch_encode_reg_vec3(0, grid, x, nf, S, dj, recs, gen++, ab, ab + 8, dm, -1, weighted, deep); };
    for (int r = 0; r < 3; ++r) launch();
    unsigned long long* vd; CHK(hipMalloc(&vd, 8 * 8 * 4096)); CHK(hipMemset(vd, 0, 8 * 8 * 4096));
    CHK(hipDeviceSynchronize()); CHK(hipMemset(sb, 0, 8 * 8 * 4096));
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_v3_diag), &vd, sizeof(vd)));
    unsigned long long* vtr; CHK(hipMalloc(&vtr, 8 * (3 * vt + 2 * s3))); CHK(hipMemset(vtr, 0, 8 * (3 * vt + 2 * s3)));
    CHK(hipMemcpyToSymbol(HIP_SYMBOL(g_v3_trace), &vtr, sizeof(vtr)));
    CHK(hipDeviceSynchronize()); CHK(hipEventRecord(a)); launch(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h(8 * grid); CHK(hipMemcpy(h.data(), sb, 8 * 8 * grid, hipMemcpyDeviceToHost));
    Meta m0; CHK(hipMemcpy(&m0, dm, sizeof m0, hipMemcpyDeviceToHost)); uint32_t abv; CHK(hipMemcpy(&abv, ab, 4, hipMemcpyDeviceToHost));
    const char* names[6] = {"descriptor wait", "plane read+re-base+pack+stores", "barrier 1", "park", "floats wait+quantize+statistics", "requests+barrier 2+publish(+duty)"};
    double cyc = 0, rt = 0; for (int w = 0; w < grid; ++w) { cyc += h[w * 8 + 6]; rt += h[w * 8 + 7]; }
    const double ghz = cyc / rt * 0.1, items = (double)vt;
    printf("n_file=%ld sub_cells=%ld vtiles=%ld grid=%d weighted=%d deep=%d items/WG=%.1f: kernel %.1f us (stamped build) bits=%ld abort=%u\n", nf, S, vt, grid, weighted, deep, items / grid, ms * 1e3,
           (long)m0.bits, abv);
    printf("  workgroup lifetime mean %.2f us, shader clock %.2f GHz\n", rt / grid / 100.0, ghz);
    for (int i = 0; i < 6; ++i) {
        double s2 = 0; for (int w = 0; w < grid; ++w) s2 += h[w * 8 + i];
        printf("  %-28s %.2f us per WG = %.2f us per item\n", names[i], s2 / grid / ghz / 1e3, s2 / grid / ghz / 1e3 / (items / grid));
    }
    { std::vector<unsigned long long> hc(8 * grid); CHK(hipMemcpy(hc.data(), vd, 8 * 8 * grid, hipMemcpyDeviceToHost));
      unsigned long long dg[8] = {0}; for (int w = 0; w < grid; ++w) for (int i = 0; i < 8; ++i) dg[i] += hc[w * 8 + i];
      for (int q8 = 0; q8 < 8; ++q8) { unsigned long long m = 0, pl = 0; for (int w = q8 * grid / 8; w < (q8 + 1) * grid / 8; ++w) { m += hc[w * 8 + 1]; pl += hc[w * 8 + 2]; }
        printf("    workgroups [%d, %d): descriptor not there yet %llu times, %llu polls\n", q8 * grid / 8, (q8 + 1) * grid / 8, m, pl); }
      printf("  this launch, wave 0 of every workgroup: descriptor uses %llu, of which not there yet %llu, poll iterations %llu; sweeps %llu, sweep iterations %llu\n", dg[0], dg[1], dg[2], dg[3], dg[4]); }
    {   // time line: per sub-cell, the last record's publication -> sentinel seen -> descriptors published; per item, when the descriptor was looked for / there
        const long T = vt / s3;
        std::vector<unsigned long long> tr(3 * vt + 2 * s3); CHK(hipMemcpy(tr.data(), vtr, 8 * tr.size(), hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull; for (long i = 0; i < vt; ++i) if (tr[i] && tr[i] < t0) t0 = tr[i];
        double s_pub_span = 0, s_sent = 0, s_desc = 0, s_first = 0, s_wait = 0, s_lookslack = 0; long nwait = 0, nlate = 0;
        for (long sc = 0; sc < s3; ++sc) {
            unsigned long long mx = 0, mn = ~0ull; for (long t = 0; t < T; ++t) { mx = std::max(mx, tr[sc * T + t]); mn = std::min(mn, tr[sc * T + t]); }
            s_pub_span += (mx - mn) / 100.0; s_sent += ((double)tr[3 * vt + 2 * sc] - (double)mx) / 100.0; s_desc += ((double)tr[3 * vt + 2 * sc + 1] - (double)mx) / 100.0;
            for (long t = 0; t < T; ++t) {
                const double look = tr[vt + sc * T + t], got = tr[2 * vt + sc * T + t], dsc = tr[3 * vt + 2 * sc + 1];
                if (look == 0) continue;
                s_first += (look - tr[sc * T + t]) / 100.0; ++nwait; s_wait += (got - look) / 100.0; s_lookslack += (look - dsc) / 100.0; if (look < dsc) ++nlate;
            }
        }
        printf("  time line (us, means over %ld sub-cells): first..last record of a sub-cell %.2f; last record -> sentinel seen %.2f; last record -> descriptors published %.2f\n",
               s3, s_pub_span / s3, s_sent / s3, s_desc / s3);
        printf("  per item: publication -> descriptor looked for %.2f; look -> there %.2f; descriptor published -> looked for %.2f (looked for before it was published: %ld of %ld)\n",
               s_first / nwait, s_wait / nwait, s_lookslack / nwait, nlate, nwait);
    }
    // spread of the workgroup lifetimes by quarter of the grid
    for (int q = 0; q < 4; ++q) { double s2 = 0; for (int w = q * grid / 4; w < (q + 1) * grid / 4; ++w) s2 += h[w * 8 + 7]; printf("  quarter %d: lifetime %.1f us\n", q, s2 / (grid / 4) / 100.0); }
    return 0;
}
