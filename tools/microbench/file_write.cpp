// Diagnostic (not product): how fast does this box's scratch file system take a file of 0.77 GB -- one fwrite stream of 2 MB blocks
// (what minnow::Writer does), write(2) of the same, and pwrite from T threads to disjoint ranges.   file_write <dir>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const size_t blk = 2040000, nblk = 375, total = blk * nblk;
    std::vector<char> buf(total);
    for (size_t i = 0; i < total; i += 4096) buf[i] = (char)i;
    const std::string f = dir + "/mnw_file_write.bin";
    for (int rep = 0; rep < 2; ++rep) {
        { double t0 = now(); FILE* fp = fopen(f.c_str(), "wb"); for (size_t b = 0; b < nblk; ++b) fwrite(buf.data() + b * blk, 1, blk, fp); fclose(fp);
          printf("fwrite, 2 MB blocks:        %.3f s = %.2f GB/s\n", now() - t0, total / (now() - t0) / 1e9); }
        { double t0 = now(); int fd = open(f.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644); for (size_t b = 0; b < nblk; ++b) if (write(fd, buf.data() + b * blk, blk) != (ssize_t)blk) return 1; close(fd);
          printf("write(2), 2 MB blocks:      %.3f s = %.2f GB/s\n", now() - t0, total / (now() - t0) / 1e9); }
        for (int T : {1, 2, 4, 8, 16}) {
            double t0 = now();
            int fd = open(f.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (ftruncate(fd, (off_t)total) != 0) return 1;
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
                for (size_t b = t; b < nblk; b += T) if (pwrite(fd, buf.data() + b * blk, blk, (off_t)(b * blk)) != (ssize_t)blk) abort();
            });
            for (auto& x : th) x.join();
            close(fd);
            printf("pwrite, %2d threads:          %.3f s = %.2f GB/s\n", T, now() - t0, total / (now() - t0) / 1e9);
        }
        { double t0 = now(); FILE* fp = fopen(f.c_str(), "rb"); size_t got = 0; for (size_t b = 0; b < nblk; ++b) got += fread(buf.data() + b * blk, 1, blk, fp); fclose(fp);
          printf("fread back:                 %.3f s = %.2f GB/s (%zu)\n", now() - t0, total / (now() - t0) / 1e9, got); }
        for (int T : {4, 8}) {
            double t0 = now();
            int fd = open(f.c_str(), O_RDONLY);
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) th.emplace_back([&, t] {
                for (size_t b = t; b < nblk; b += T) if (pread(fd, buf.data() + b * blk, blk, (off_t)(b * blk)) != (ssize_t)blk) abort();
            });
            for (auto& x : th) x.join();
            close(fd);
            printf("pread, %2d threads:           %.3f s = %.2f GB/s\n", T, now() - t0, total / (now() - t0) / 1e9);
        }
    }
    remove(f.c_str());
    return 0;
}
