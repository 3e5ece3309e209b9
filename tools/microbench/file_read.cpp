// Micro-benchmark (not product): ways of getting a page-cached file of 0.87 GB into pinned host memory, the first leg of
// text.Reader.Block on the GPU path: mmap + memcpy (1 / 8 threads), mmap(MAP_POPULATE) + memcpy, pread on 1 / 8 threads.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <typename F> static void par(int t, size_t n, F f) {
    std::vector<std::thread> th;
    const size_t per = ((n + t - 1) / t + 4095) & ~(size_t)4095;
    for (int k = 0; k < t; ++k) th.emplace_back([=] { size_t a = per * k, b = a + per < n ? a + per : n; if (a < b) f(a, b - a); });
    for (auto& x : th) x.join();
}
int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "/tmp/mnw_file_read.bin";
    const size_t n = 870u << 20;
    { std::vector<char> buf(64 << 20, 'x'); FILE* f = fopen(path, "wb"); for (size_t o = 0; o < n; o += buf.size()) fwrite(buf.data(), 1, buf.size(), f); fclose(f); }
    uint8_t* pin; if (hipHostMalloc((void**)&pin, n, hipHostMallocDefault) != hipSuccess) return 1;
    memset(pin, 0, n);
    const int fd = open(path, O_RDONLY);
    for (int t : {1, 8}) {
        double t0 = now();
        uint8_t* m = (uint8_t*)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        par(t, n, [=](size_t a, size_t len) { memcpy(pin + a, m + a, len); });
        printf("mmap + memcpy, %d threads: %.1f ms (%.1f GB/s)\n", t, (now() - t0) * 1e3, n / (now() - t0) / 1e9);
        munmap(m, n);
        t0 = now();
        m = (uint8_t*)mmap(nullptr, n, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
        const double tm = now() - t0;
        par(t, n, [=](size_t a, size_t len) { memcpy(pin + a, m + a, len); });
        printf("mmap(MAP_POPULATE) %.1f ms + memcpy, %d threads: %.1f ms in all (%.1f GB/s)\n", tm * 1e3, t, (now() - t0) * 1e3, n / (now() - t0) / 1e9);
        munmap(m, n);
        t0 = now();
        par(t, n, [=](size_t a, size_t len) { size_t done = 0; while (done < len) { ssize_t r = pread(fd, pin + a + done, len - done, (off_t)(a + done)); if (r <= 0) break; done += (size_t)r; } });
        printf("pread into pinned memory, %d threads: %.1f ms (%.1f GB/s)\n", t, (now() - t0) * 1e3, n / (now() - t0) / 1e9);
    }
    { std::vector<char> buf(151 << 20, 'y'); double t0 = now(); FILE* f = fopen((std::string(path) + ".out").c_str(), "wb"); fwrite(buf.data(), 1, buf.size(), f); fclose(f);
      printf("fwrite of 151 MB into a new file: %.1f ms (%.1f GB/s)\n", (now() - t0) * 1e3, buf.size() / (now() - t0) / 1e9); unlink((std::string(path) + ".out").c_str()); }
    close(fd); unlink(path);
    return 0;
}
