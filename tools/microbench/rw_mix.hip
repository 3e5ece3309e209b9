// Micro-benchmark (not product code): what the memory system sustains for pure reads, pure writes and mixes of the two, with
// nothing else in the kernel -- one 16-byte access per thread, short workgroups in dispatch order (the fastest copy shape
// measured on this machine, tools/microbench/copy_bw.hip).  The codec's decode writes 4 bytes per element and reads b/8; if
// writes alone cannot go faster than some rate W, no decoder can beat (4 B x elements) / W whatever its arithmetic costs.
//   hipcc -O3 --offload-arch=gfx950 rw_mix.hip -o rw_mix && ./rw_mix [GB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#define GLOBAL __attribute__((address_space(1)))
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// every thread: R loads of 16 B (consecutive chunks of its own), W stores of 16 B
template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void k_rw(const unsigned* __restrict__ src, unsigned* __restrict__ dst, long nthreads) {
    const long t = blockIdx.x * 256L + threadIdx.x;
    if (t >= nthreads) return;
    u32x4 acc = {(unsigned)t, 1, 2, 3};
    const long wg = blockIdx.x;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const GLOBAL u32x4* p = (const GLOBAL u32x4*)src + (wg * R + k) * 256 + threadIdx.x;
        const u32x4 v = NT ? __builtin_nontemporal_load(p) : *p;
        acc ^= v;
    }
#pragma unroll
    for (int k = 0; k < W; ++k) {
        GLOBAL u32x4* p = (GLOBAL u32x4*)dst + (wg * W + k) * 256 + threadIdx.x;
        if (NT) __builtin_nontemporal_store(acc, p); else *p = acc;
    }
    if (W == 0 && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) dst[0] = 1;      // keep the loads
}

// the sub-cell pattern of minp: rows of `run` bytes every `stride` bytes, one workgroup per 12 KB of rows, written only
__global__ __launch_bounds__(256) void k_rows(unsigned* __restrict__ dst, long chunks, int run16, long stride16) {
    const long c = blockIdx.x * 256L + threadIdx.x;          // 16-byte chunk of the compact image
    if (c >= chunks) return;
    const long row = c / run16, col = c - row * run16;
    u32x4 v = {(unsigned)c, 1, 2, 3};
    *((GLOBAL u32x4*)dst + row * stride16 + col) = v;
}

template <typename F> float best_of(F f, int reps) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    f(); CHK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < reps; ++r) { CHK(hipEventRecord(a)); f(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b)); float ms; CHK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms); }
    return best;
}

static bool g_quiet = false;
static double g_last_total = 0;
template <int R, int W, bool NT> void run(const unsigned* src, unsigned* dst, long bytes_moved, const char* name) {
    // threads so that (R + W) * 16 B * threads = bytes_moved
    const long nthreads = bytes_moved / (16L * (R + W)) / 256 * 256;
    const long grid = nthreads / 256;
    const float ms = best_of([&] { hipLaunchKernelGGL((k_rw<R, W, NT>), dim3((unsigned)grid), dim3(256), 0, 0, src, dst, nthreads); }, 8);
    const double rd = 16.0 * R * nthreads, wr = 16.0 * W * nthreads;
    g_last_total = (rd + wr) / ms / 1e6;      // GB/s
    if (!g_quiet) printf("  %-34s %7.1f us   read %5.2f GB at %5.2f TB/s   written %5.2f GB at %5.2f TB/s   total %5.2f TB/s\n", name, ms * 1e3, rd / 1e9, rd / ms / 1e9, wr / 1e9,
           wr / ms / 1e9, (rd + wr) / ms / 1e9);
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 6.0;
    const bool json = argc > 2 && !strcmp(argv[2], "json");
    g_quiet = json;
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    if (!json) printf("device %s, pci bus %d; every kernel moves %.1f GB in all, 16 B per access, best of 8\n", prop.gcnArchName, prop.pciBusID, gb);
    const long bytes = (long)(gb * 1e9);
    unsigned *src, *dst;
    CHK(hipMalloc(&src, bytes + 4096)); CHK(hipMalloc(&dst, bytes + 4096));
    CHK(hipMemset(src, 1, bytes)); CHK(hipMemset(dst, 2, bytes));
    if (json) {
        // the float codec's mixes at b = 18 only: 4 B read + 2.25 B written per element (encode), the reverse (decode)
        run<9, 5, false>(src, dst, bytes, ""); const double e = g_last_total;
        run<5, 9, false>(src, dst, bytes, ""); const double d = g_last_total;
        run<1, 1, false>(src, dst, bytes, ""); const double c = g_last_total;
        printf("{\"encode_mix_GBps\": %.1f, \"decode_mix_GBps\": %.1f, \"copy_GBps\": %.1f, \"pci_bus_id\": %d, \"what\": \"tools/microbench/rw_mix: %g GB moved per kernel, "
               "one 16-byte access per thread in the codec's read : write ratio at 18 bits (9 : 5 and 5 : 9), no arithmetic, best of 8\"}\n", e, d, c, prop.pciBusID, gb);
        return 0;
    }
    run<1, 0, false>(src, dst, bytes, "read only");
    run<4, 0, false>(src, dst, bytes, "read only, 4 per thread");
    run<0, 1, false>(src, dst, bytes, "write only");
    run<0, 4, false>(src, dst, bytes, "write only, 4 per thread");
    run<0, 4, true>(src, dst, bytes, "write only, 4 per thread, nt");
    run<1, 1, false>(src, dst, bytes, "copy (1 : 1)");
    run<2, 1, false>(src, dst, bytes, "2 reads : 1 write (encode-like)");
    run<9, 5, false>(src, dst, bytes, "9 : 5 (float encode, b = 18)");
    run<1, 2, false>(src, dst, bytes, "1 read : 2 writes (decode-like)");
    run<5, 9, false>(src, dst, bytes, "5 : 9 (float decode, b = 18)");
    run<5, 9, true>(src, dst, bytes, "5 : 9, nt");
    run<1, 3, false>(src, dst, bytes, "1 : 3");
    {
        const float ms = best_of([&] { CHK(hipMemsetAsync(dst, 0, bytes, 0)); }, 5);
        printf("  %-34s %7.1f us   written %5.2f GB at %5.2f TB/s\n", "hipMemsetAsync", ms * 1e3, bytes / 1e9, bytes / ms / 1e9);
    }
    // rows of 1,200 B every 6,000 B (C4's sub-cells) against the same bytes written contiguously
    for (int run16 : {75, 72, 80, 375}) {
        const long stride16 = 375;                      // 6,000 B
        const long rows = bytes / 3 / (run16 * 16L);    // a third of the buffer's bytes, spread over run/stride of its span ... keep inside the buffer:
        const long max_rows = bytes / 16 / stride16 - 1;
        const long nrows = std::min(rows, max_rows);
        const long chunks = nrows * run16;
        const float ms = best_of([&] { hipLaunchKernelGGL(k_rows, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, 0, dst, chunks, run16, stride16); }, 8);
        printf("  rows of %4d B every 6,000 B       %7.1f us   written %5.2f GB at %5.2f TB/s\n", run16 * 16, ms * 1e3, chunks * 16.0 / 1e9, chunks * 16.0 / ms / 1e9);
    }
    return 0;
}
