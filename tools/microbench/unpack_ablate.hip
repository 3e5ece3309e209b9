// Diagnostic (not product): k_unpack_f32 with and without the dither arithmetic.
#include "../../minnow_amd/csrc/mnw_kernels.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using namespace mnw;
int main(int argc, char** argv) {
    long n = argc > 1 ? atol(argv[1]) : 1000000000L;
    int bits = 18;
    uint8_t* in; float* y; DevJob* dj;
    long nbytes = (n * bits + 7) / 8;
    CHK(hipMalloc(&in, nbytes + 64)); CHK(hipMalloc(&y, n * 4)); CHK(hipMalloc(&dj, sizeof(DevJob)));
    CHK(hipMemset(in, 0x5a, nbytes + 64));
    DevJob j; memset(&j, 0, sizeof j);
    j.src = in; j.dst = y; j.n = n; j.cap = nbytes; j.pixels = 250000; j.low = 0; j.high = 250; j.dxq = 0.001f; j.kind = 11; j.periodic = 1;
    j.mode = kModeExplicit; j.min = 0; j.bits = bits; j.seed = 1; j.tile0 = 0;
    CHK(hipMemcpy(dj, &j, sizeof j, hipMemcpyHostToDevice));
    long tiles = (n + kTileD - 1) / kTileD;
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    for (int g : {1024, 2048, 4096}) {
        float best = 1e9;
        for (int r = 0; r < 6; ++r) { CHK(hipEventRecord(a)); launch_unpack_f32(0, g, true, false, dj, 1, tiles); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b)); float ms; CHK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
        printf("%s n=%ld grid=%d unpack %.1f us (%.2f TB/s)\n",
#ifdef MNW_DIAG_NO_DITHER
               "NO-DITHER",
#else
               "full     ",
#endif
               n, g, best * 1e3, (n * 4.0 + nbytes) / best / 1e9);
    }
    return 0;
}
