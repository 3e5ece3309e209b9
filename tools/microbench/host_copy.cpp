// Micro-benchmark (not product): the pieces of a pageable-host -> device transfer on this machine: memcpy into pinned memory
// on 1..16 threads, pinned -> device DMA by chunk size, the runtime's own pageable hipMemcpy, hipHostRegister.
//   hipcc -O2 host_copy.cpp -o host_copy -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par_copy(uint8_t* d, const uint8_t* s, size_t n, int t) {
    std::vector<std::thread> th;
    const size_t per = (n + t - 1) / t;
    for (int k = 0; k < t; ++k) th.emplace_back([=] { size_t a = per * k, b = a + per < n ? a + per : n; if (a < b) memcpy(d + a, s + a, b - a); });
    for (auto& x : th) x.join();
}
int main() {
    const size_t n = 400u << 20;
    uint8_t* page = (uint8_t*)malloc(n); memset(page, 1, n);
    uint8_t* page2 = (uint8_t*)malloc(n); memset(page2, 2, n);
    uint8_t *pin, *dev;
    CHK(hipHostMalloc((void**)&pin, n, hipHostMallocDefault)); memset(pin, 3, n);
    CHK(hipMalloc((void**)&dev, n));
    printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
    for (int t : {1, 2, 4, 8, 16}) {
        par_copy(pin, page, n, t);
        double t0 = now(); par_copy(pin, page, n, t); double a = now() - t0;
        t0 = now(); par_copy(page2, pin, n, t); double b = now() - t0;
        t0 = now(); par_copy(page2, page, n, t); double c = now() - t0;
        printf("memcpy %2d threads: pageable->pinned %.1f GB/s, pinned->pageable %.1f GB/s, pageable->pageable %.1f GB/s\n", t, n / a / 1e9, n / b / 1e9, n / c / 1e9);
    }
    hipStream_t s; CHK(hipStreamCreate(&s));
    for (size_t chunk : {1u << 20, 4u << 20, 16u << 20, 64u << 20, 400u << 20}) {
        for (int dir = 0; dir < 2; ++dir) {
            CHK(hipStreamSynchronize(s));
            double t0 = now();
            for (size_t off = 0; off < n; off += chunk) {
                if (dir == 0) CHK(hipMemcpyAsync(dev + off, pin + off, chunk < n - off ? chunk : n - off, hipMemcpyHostToDevice, s));
                else CHK(hipMemcpyAsync(pin + off, dev + off, chunk < n - off ? chunk : n - off, hipMemcpyDeviceToHost, s));
            }
            CHK(hipStreamSynchronize(s));
            printf("pinned %s in chunks of %4zu MiB: %.1f GB/s\n", dir == 0 ? "H2D" : "D2H", chunk >> 20, n / (now() - t0) / 1e9);
        }
    }
    for (int dir = 0; dir < 2; ++dir) {
        double t0 = now();
        if (dir == 0) CHK(hipMemcpy(dev, page, n, hipMemcpyHostToDevice)); else CHK(hipMemcpy(page2, dev, n, hipMemcpyDeviceToHost));
        printf("pageable hipMemcpy %s: %.1f GB/s\n", dir == 0 ? "H2D" : "D2H", n / (now() - t0) / 1e9);
    }
    { double t0 = now(); CHK(hipHostRegister(page, n, hipHostRegisterDefault)); double a = now() - t0;
      t0 = now(); CHK(hipMemcpy(dev, page, n, hipMemcpyHostToDevice)); double b = now() - t0;
      t0 = now(); CHK(hipHostUnregister(page)); double c2 = now() - t0;
      printf("hipHostRegister 400 MiB: %.1f ms, copy of the registered buffer %.1f GB/s, unregister %.1f ms\n", a * 1e3, n / b / 1e9, c2 * 1e3); }
    return 0;
}
