// Micro-benchmark (not product code): can the intermediate of a two-pass encode live in the Infinity Cache?
// Pass A reads a chunk of X (HBM) and writes the same number of bytes to a scratch buffer that every chunk reuses; pass B reads the
// scratch and writes 9/16 of the bytes to Z (HBM).  If the scratch stays in the 256 MB memory-side cache its traffic never reaches
// HBM and the pair costs what ONE read of X and one write of Z cost.  Against: A' reads X (no scratch), B' reads X again and writes Z
// (the two-pass encode as it is).   hipcc -O3 --offload-arch=gfx950 mall_stage.hip -o mall_stage && ./mall_stage
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define GLOBAL __attribute__((address_space(1)))
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// one 16-byte access per thread and stream; mode bits: 1 = read src, 2 = write dst (full size), 4 = write dst2 (9 of every 16 chunks)
template <int MODE>
__global__ __launch_bounds__(256) void k_pass(const unsigned* __restrict__ src, unsigned* __restrict__ dst, unsigned* __restrict__ dst2, long chunks) {
    const long c = blockIdx.x * 256L + threadIdx.x;
    if (c >= chunks) return;
    u32x4 v = {(unsigned)c, 1, 2, 3};
    if (MODE & 1) v ^= *((const GLOBAL u32x4*)src + c);
    if (MODE & 2) *((GLOBAL u32x4*)dst + c) = v;
    if (MODE & 4) { const long w = c >> 4, r = c & 15; if (r < 9) *((GLOBAL u32x4*)dst2 + w * 9 + r) = v; }
    if (MODE == 1 && v.x == 0x12345678u && v.y == 0x9abcdef1u) dst[0] = 1;
}

int main(int argc, char** argv) {
    const long total = 4L << 30;                       // bytes of X
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, pci bus %d; X = %.1f GB\n", prop.gcnArchName, prop.pciBusID, total / 1e9);
    unsigned *x, *z, *scratch;
    CHK(hipMalloc(&x, total)); CHK(hipMalloc(&z, total)); CHK(hipMalloc(&scratch, 1L << 30));
    CHK(hipMemset(x, 1, total)); CHK(hipMemset(z, 2, total)); CHK(hipMemset(scratch, 3, 1L << 30));
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    for (long chunk_mb : {16L, 32L, 64L, 96L, 128L, 192L, 256L, 512L}) {
        const long cb = chunk_mb << 20, nchunk = total / cb, chunks = cb / 16;
        const unsigned grid = (unsigned)((chunks + 255) / 256);
        for (int variant = 0; variant < 3; ++variant) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                CHK(hipEventRecord(a));
                for (long k = 0; k < nchunk; ++k) {
                    const unsigned* xs = x + k * (cb / 4);
                    unsigned* zs = z + k * (cb / 4);
                    if (variant == 0) {            // staged: X -> scratch, scratch -> Z
                        hipLaunchKernelGGL(k_pass<3>, dim3(grid), dim3(256), 0, 0, xs, scratch, nullptr, chunks);
                        hipLaunchKernelGGL(k_pass<5>, dim3(grid), dim3(256), 0, 0, scratch, nullptr, zs, chunks);
                    } else if (variant == 1) {     // two-pass as it is: read X, read X again -> Z
                        hipLaunchKernelGGL(k_pass<1>, dim3(grid), dim3(256), 0, 0, xs, zs, nullptr, chunks);
                        hipLaunchKernelGGL(k_pass<5>, dim3(grid), dim3(256), 0, 0, xs, nullptr, zs, chunks);
                    } else {                       // one pass (the floor): X -> Z
                        hipLaunchKernelGGL(k_pass<5>, dim3(grid), dim3(256), 0, 0, xs, nullptr, zs, chunks);
                    }
                }
                CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
                float ms; CHK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
            }
            const char* names[] = {"staged through scratch", "read twice            ", "single pass (floor)   "};
            printf("  chunks of %3ld MB: %s %8.1f us  = %.2f of 8 TB/s for (1 + 9/16) x X\n", chunk_mb, names[variant], best * 1e3, total * (25.0 / 16) / (best * 1e-3) / 8e12);
        }
    }
    return 0;
}
