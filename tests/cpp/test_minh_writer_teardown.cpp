// Host-only test (no GPU): a minh::Writer -- and the file-writer thread it owns -- that goes away without Close().
// The hazard: Writer::Block posts the file writes of block b to a thread; if the Writer is destroyed while a
// task is queued or running (Block of b+1 threw, or the caller dropped the object), the thread must not touch members that
// are already gone.  Built with -fsanitize=address,undefined by tests/test_sanitizers_cpu.py; also run plain.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <memory>
#include <string>
#include <thread>

#include "minh.hpp"

static int failures = 0;
#define EXPECT(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

// the Writer's member order in small: the thread object first (destroyed LAST), what its tasks use after it (destroyed FIRST)
struct Owner {
    minh::detail::FileWriter w;
    std::unique_ptr<int> target = std::make_unique<int>(0);
    ~Owner() { w.stop(); }                      // what ~Writer does: without it the running task writes through a freed pointer
};

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    std::atomic<int> ran{0}, started{0};
    {
        Owner o;
        int* t = o.target.get();
        o.w.post([&, t] { started = 1; std::this_thread::sleep_for(std::chrono::milliseconds(80)); *t = 7; ++ran; }, [] {});
        o.w.post([&, t] { *t = 9; ++ran; }, [] {});        // queued behind the first: must be dropped, not run
        while (!started) std::this_thread::yield();
    }   // destroyed with one task running and one queued
    EXPECT(ran == 1);
    {
        // stop() is idempotent and a stopped writer's destructor is harmless
        minh::detail::FileWriter w;
        w.post([&] { ++ran; }, [] {});
        w.drain();
        w.stop();
        w.stop();
    }
    EXPECT(ran == 2);
    {
        // a real Writer: created, headers written, destroyed without Close() (no block was ever posted: no GPU here)
        minh::Writer wr(dir + "/teardown.minh");
        std::vector<minh::Column> cols(1);
        cols[0] = minh::Column{};
        wr.Header({"a"}, "text", cols);
    }
    if (failures) { std::fprintf(stderr, "%d failure(s)\n", failures); return 1; }
    std::puts("ok: minh writer teardown");
    return 0;
}
