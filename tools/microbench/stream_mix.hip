// Micro-benchmark (not product code): the machine's ceiling for the codec's traffic mix -- 4 bytes read per
// element and 18/8 bytes written (encode), or the reverse (decode) -- with nothing but the loads and stores of
// the real kernels' shapes.  What a float encode / decode could reach if arithmetic and synchronisation were free.
//   hipcc -O3 --offload-arch=gfx950 stream_mix.hip -o stream_mix && ./stream_mix [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define GLOBAL __attribute__((address_space(1)))
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int kItem = 8192;           // elements per item: 32 KB in, 18 KB out
constexpr int kOutChunks = kItem * 18 / 8 / 16;   // 1152 16-byte chunks

// MODE 0: encode mix (read 32 KB, write 18 KB per item)   MODE 1: decode mix (read 18 KB, write 32 KB)
// NTL / NTS: non-temporal loads / stores.  DEPTH: items whose loads are in flight per workgroup (1 or 2).
template <int MODE, bool NTL, bool NTS, int DEPTH>
__global__ __launch_bounds__(256) void k_mix(const unsigned* __restrict__ big, unsigned* __restrict__ small_, long items, int contiguous) {
    const int tid = threadIdx.x;
    long i0, i1, step;
    if (contiguous) { i0 = items * blockIdx.x / gridDim.x; i1 = items * (blockIdx.x + 1) / gridDim.x; step = 1; }
    else { i0 = blockIdx.x; i1 = items; step = gridDim.x; }
    auto ld = [&](const unsigned* p) -> u32x4 {
        if (NTL) return __builtin_nontemporal_load((const GLOBAL u32x4*)p);
        return *(const GLOBAL u32x4*)p;
    };
    auto st = [&](unsigned* p, u32x4 v) {
        if (NTS) __builtin_nontemporal_store(v, (GLOBAL u32x4*)p);
        else *(GLOBAL u32x4*)p = v;
    };
    if (MODE == 0) {
        u32x4 v[DEPTH][8];
        long it = i0;
        if (it < i1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[0][k] = ld(big + it * kItem + 4 * (k * 256 + tid));
        }
        if (DEPTH == 2 && it + step < i1) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[DEPTH - 1][k] = ld(big + (it + step) * kItem + 4 * (k * 256 + tid));
        }
        for (; it < i1; it += step) {
            u32x4 a = v[0][0] ^ v[0][1], b = v[0][2] ^ v[0][3], c = v[0][4] ^ v[0][5], d = v[0][6] ^ v[0][7];
            u32x4 e = a ^ b ^ c ^ d;
            if (DEPTH == 2) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[0][k] = v[DEPTH - 1][k];
            }
            const long nx = it + step * DEPTH;
            if (nx < i1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[DEPTH - 1][k] = ld(big + nx * kItem + 4 * (k * 256 + tid));
            }
            unsigned* o = small_ + it * (kOutChunks * 4);
            st(o + 4 * tid, a); st(o + 4 * (256 + tid), b); st(o + 4 * (512 + tid), c); st(o + 4 * (768 + tid), d);
            if (tid < kOutChunks - 1024) st(o + 4 * (1024 + tid), e);
        }
    } else {
        for (long it = i0; it < i1; it += step) {
            const unsigned* s = small_ + it * (kOutChunks * 4);
            u32x4 a = ld(s + 4 * tid), b = ld(s + 4 * (256 + tid)), c = ld(s + 4 * (512 + tid)), d = ld(s + 4 * (768 + tid));
            u32x4 e = a;
            if (tid < kOutChunks - 1024) e = ld(s + 4 * (1024 + tid));
            unsigned* o = (unsigned*)big + it * kItem;
            st(o + 4 * tid, a); st(o + 4 * (256 + tid), b); st(o + 4 * (512 + tid), c); st(o + 4 * (768 + tid), d);
            st(o + 4 * (1024 + tid), e); st(o + 4 * (1280 + tid), a ^ b); st(o + 4 * (1536 + tid), c ^ d); st(o + 4 * (1792 + tid), a ^ e);
        }
    }
}

template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    f(); f(); CHK(hipDeviceSynchronize());
    float tot = 0;
    for (int r = 0; r < reps; ++r) { CHK(hipEventRecord(a)); f(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b)); float ms; CHK(hipEventElapsedTime(&ms, a, b)); tot += ms; }
    return tot / reps;
}

int main(int argc, char** argv) {
    for (long n : {100000000L, 1000000000L}) {
        if (argc > 1 && atol(argv[1]) != n) continue;
        const long items = n / kItem;
        unsigned *big, *sm;
        CHK(hipMalloc(&big, items * kItem * 4)); CHK(hipMalloc(&sm, items * kOutChunks * 16));
        CHK(hipMemset(big, 0x3c, items * kItem * 4)); CHK(hipMemset(sm, 0x11, items * kOutChunks * 16));
        const double bytes = (double)items * (kItem * 4 + kOutChunks * 16);
        printf("n = %ld elements: %.0f MB per pass (4 B + 2.25 B per element)\n", n, bytes / 1e6);
        for (int contiguous : {0, 1}) {
            for (int g : {512, 1024, 2048, 4096}) {
                const int reps = n > 500000000L ? 5 : 20;
#define RUN(MODE, NTL, NTS, DEPTH, label) { float t = timeit([&] { hipLaunchKernelGGL((k_mix<MODE, NTL, NTS, DEPTH>), dim3(g), dim3(256), 0, 0, big, sm, items, contiguous); }, reps); \
    printf("  %-34s grid %4d %s: %8.1f us = %.2f TB/s = %.3f of 8 TB/s\n", label, g, contiguous ? "contiguous" : "strided   ", t * 1e3, bytes / t / 1e9, bytes / t / 1e9 / 8000.0); }
                RUN(0, false, false, 1, "encode mix")
                RUN(0, false, false, 2, "encode mix, 2 items in flight")
                RUN(0, true, false, 1, "encode mix, nt loads")
                RUN(0, false, true, 1, "encode mix, nt stores")
                RUN(0, true, true, 2, "encode mix, nt both, 2 in flight")
                RUN(1, false, false, 1, "decode mix")
                RUN(1, true, false, 1, "decode mix, nt loads")
                RUN(1, false, true, 1, "decode mix, nt stores")
                RUN(1, true, true, 1, "decode mix, nt both")
            }
        }
        CHK(hipFree(big)); CHK(hipFree(sm));
    }
    return 0;
}
