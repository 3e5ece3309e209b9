// C++ host-side tests of include/minh.hpp, written after the reference's go/minh/minh_test.go (TestReaderWriter) so they
// read the same way, plus: a whole minh file compared byte for byte with the one the reference's Python minh.Writer
// produced (tests/golden/golden_small.json "minh_files"), and the device-array forms against the host-array forms.
// argv[1] = scratch dir, argv[2] = dir holding the golden file.  Run by tests/test_cpp_host.py on a GPU box.
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <iterator>
#include <map>
#include <string>
#include <vector>

#include "minh.hpp"

using namespace minh;
static int failures = 0;
#define EXPECT(cond, msg) do { if (!(cond)) { std::cerr << "FAIL " << __func__ << ": " << msg << "\n"; ++failures; } } while (0)

static std::string scratch, golden;

static std::vector<uint8_t> slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), {});
}
static bool float32sEq(const std::vector<float>& x, const float* y, float dx) {            // minh_test.go:430-436
    for (size_t i = 0; i < x.size(); ++i) if (x[i] + dx < y[i] || x[i] - dx > y[i]) return false;
    return true;
}
static bool log32sEq(const std::vector<float>& x, const float* y, float dx) {              // minh_test.go:438-446
    for (size_t i = 0; i < x.size(); ++i) {
        const float lx = (float)std::log10((double)x[i]), ly = (float)std::log10((double)y[i]);
        if (lx + dx < ly || lx - dx > ly) return false;
    }
    return true;
}
static Column col(int64_t type, int32_t log = 0, float low = 0, float high = 0, float dx = 0) {
    Column c; c.Type = type; c.Log = log; c.Low = low; c.High = high; c.Dx = dx; return c;
}

static const std::string text = "Cats are the best. Don't we love them?!@#$%^&*(),....[]{};':\"|\\/-=_+`~meow meow meow";
static const std::vector<std::string> names = {"int64", "float32", "int", "float", "log"};

struct Blocks {
    std::vector<std::vector<int64_t>> i64, in;
    std::vector<std::vector<float>> f32, fl, lg;
};
static Blocks testBlocks(float log2nd) {
    Blocks b;
    b.i64 = {{100, 200, 300, 400, 500}, {125, 225, 325}};
    b.f32 = {{150, 250, 350, 450, 550}, {1750, 2750, 3750}};
    b.in = {{-30, -35, -25, -10, -20}, {1000, 1000, 1000}};
    b.fl = {{100, 200, 125, 150, 100}, {100, 100, 100}};
    b.lg = {{1e10f, log2nd, log2nd, 1e14f, 3e13f}, {1e14f, 1e14f, 1e14f}};
    return b;
}
static void writeFile(const std::string& fname, const std::vector<Column>& columns, const Blocks& b, bool device) {
    auto wr = Create(fname);
    wr->Header(names, text, columns);
    wr->Geometry(100, 10, 4);
    mnw_ctx* ctx = minnow::detail::ctx();
    for (size_t k = 0; k < b.i64.size(); ++k) {
        const int N = (int)b.i64[k].size();
        std::vector<const void*> cols = {b.i64[k].data(), b.f32[k].data(), b.in[k].data(), b.fl[k].data(), b.lg[k].data()};
        if (!device) { wr->Block(cols, N); continue; }
        std::vector<void*> d(5);
        const size_t esz[5] = {8, 4, 8, 4, 4};
        for (int i = 0; i < 5; ++i) {
            minnow::detail::check(mnw_dev_alloc(ctx, (int64_t)(esz[i] * N), &d[i]));
            minnow::detail::check(mnw_copy_to_device(ctx, d[i], cols[i], (int64_t)(esz[i] * N)));
        }
        wr->BlockDevice({d[0], d[1], d[2], d[3], d[4]}, N);
        for (void* p : d) mnw_dev_free(ctx, p);
    }
    wr->Close();
}

// go/minh/minh_test.go:10-117
static void TestReaderWriter() {
    const std::string fname = scratch + "/reader_writer_minh.test";
    const std::vector<Column> columns = {col(Int64), col(Float32), col(Int), col(Float, 0, 100, 200, 1), col(Float, 1, 10, 14, 0.01f)};
    const Blocks blk = testBlocks(1e11f);
    writeFile(fname, columns, blk, false);

    auto rd = Open(fname);
    EXPECT(rd->Names == names && rd->Text == text && rd->Columns == columns && rd->Blocks == 2 && rd->Length == 8 &&
           rd->BlockLengths == std::vector<int>({5, 3}) && rd->L == 100.0f && rd->Boundary == 10.0f && rd->Cells == 4, "Could not read header");
    for (int b = 0; b < 3; ++b) {
        std::vector<int64_t> i64, in;
        std::vector<float> f32, fl, lg;
        std::vector<int64_t> wi64, win;
        std::vector<float> wf32, wfl, wlg;
        if (b < 2) {
            const size_t n = blk.i64[b].size();
            i64.resize(n); in.resize(n); f32.resize(n); fl.resize(n); lg.resize(n);
            rd->IntBlock(b, {{"int64", i64.data()}, {"int", in.data()}});
            rd->FloatBlock(b, {{"float32", f32.data()}, {"float", fl.data()}, {"log", lg.data()}});
            wi64 = blk.i64[b]; win = blk.in[b]; wf32 = blk.f32[b]; wfl = blk.fl[b]; wlg = blk.lg[b];
        } else {
            auto io = rd->Ints({"int64", "int"});
            auto fo = rd->Floats({"float32", "float", "log"});
            i64 = io["int64"]; in = io["int"]; f32 = fo["float32"]; fl = fo["float"]; lg = fo["log"];
            for (int k = 0; k < 2; ++k) {
                wi64.insert(wi64.end(), blk.i64[k].begin(), blk.i64[k].end()); win.insert(win.end(), blk.in[k].begin(), blk.in[k].end());
                wf32.insert(wf32.end(), blk.f32[k].begin(), blk.f32[k].end()); wfl.insert(wfl.end(), blk.fl[k].begin(), blk.fl[k].end());
                wlg.insert(wlg.end(), blk.lg[k].begin(), blk.lg[k].end());
            }
        }
        EXPECT(i64 == wi64, "Block " << b << ") column 'int64'");
        EXPECT(f32.size() == wf32.size() && float32sEq(wf32, f32.data(), 1e-6f), "Block " << b << ") column 'float32'");
        EXPECT(in == win, "Block " << b << ") column 'int'");
        EXPECT(fl.size() == wfl.size() && float32sEq(wfl, fl.data(), columns[3].Dx), "Block " << b << ") column 'float'");
        EXPECT(lg.size() == wlg.size() && log32sEq(wlg, lg.data(), columns[4].Dx), "Block " << b << ") column 'log'");
    }
    rd->Close();
}

// the file of the reference's Python minh.Writer, byte for byte; host arrays and device arrays give the same file
static void TestGoldenFile() {
    const std::vector<Column> columns = {col(Int64), col(Float32), col(Int), col(Float, 0, 100, 200, 1), col(Float, 1, 10, 14, 0.125f)};
    const Blocks blk = testBlocks(2e11f);
    const std::string fh = scratch + "/reader_writer_golden_host.minh", fd = scratch + "/reader_writer_golden_dev.minh";
    writeFile(fh, columns, blk, false);
    writeFile(fd, columns, blk, true);
    const auto want = slurp(golden + "/reader_writer_golden.minh");
    EXPECT(!want.empty(), "golden file missing");
    EXPECT(slurp(fh) == want, "file written from host arrays differs from the reference Python writer's");
    EXPECT(slurp(fd) == want, "file written from device arrays differs from the reference Python writer's");
    // and the reference's file reads back through the device-array reader
    auto rd = Open(golden + "/reader_writer_golden.minh");
    mnw_ctx* ctx = minnow::detail::ctx();
    for (int b = 0; b < 2; ++b) {
        const size_t n = (size_t)rd->BlockLengths[(size_t)b];
        void *di = nullptr, *dl = nullptr, *df = nullptr;
        minnow::detail::check(mnw_dev_alloc(ctx, 8 * n, &di)); minnow::detail::check(mnw_dev_alloc(ctx, 4 * n, &dl)); minnow::detail::check(mnw_dev_alloc(ctx, 4 * n, &df));
        rd->IntBlockDevice(b, {"int"}, {di});
        rd->FloatBlockDevice(b, {"log", "float32"}, {dl, df});
        std::vector<int64_t> in(n); std::vector<float> lg(n), f32(n);
        minnow::detail::check(mnw_copy_to_host(ctx, in.data(), di, 8 * n)); minnow::detail::check(mnw_copy_to_host(ctx, lg.data(), dl, 4 * n));
        minnow::detail::check(mnw_copy_to_host(ctx, f32.data(), df, 4 * n));
        EXPECT(in == blk.in[b], "device read, block " << b << " column 'int'");
        EXPECT(f32 == blk.f32[b], "device read, block " << b << " column 'float32'");
        EXPECT(log32sEq(blk.lg[b], lg.data(), 0.125f), "device read, block " << b << " column 'log'");
        mnw_dev_free(ctx, di); mnw_dev_free(ctx, dl); mnw_dev_free(ctx, df);
    }
    rd->Close();
}

// larger blocks: every kind of column, 200,000 rows, three blocks; host and device writers agree, values come back
static void TestLargeBlocks() {
    const std::vector<Column> columns = {col(Int64), col(Float32), col(Int), col(Float, 0, 0, 250, 0.001f), col(Float, 1, 8, 16, 0.01f)};
    Blocks blk;
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    for (int k = 0; k < 3; ++k) {
        const size_t n = k == 2 ? 1777 : 200000;
        std::vector<int64_t> a(n), c(n); std::vector<float> b(n), d(n), e(n);
        for (size_t i = 0; i < n; ++i) {
            a[i] = (int64_t)(rnd() * 1e15) - (int64_t)5e14; b[i] = (float)(rnd() * 1e6 - 5e5); c[i] = 7000000 + (int64_t)(rnd() * 130000);
            d[i] = (float)(rnd() * 250.0); e[i] = (float)std::pow(10.0, 7.5 + 9.0 * rnd());       // a few values outside [10^8, 10^16): clamped
        }
        blk.i64.push_back(a); blk.f32.push_back(b); blk.in.push_back(c); blk.fl.push_back(d); blk.lg.push_back(e);
    }
    const std::string fh = scratch + "/large_host.minh", fd = scratch + "/large_dev.minh";
    writeFile(fh, columns, blk, false);
    writeFile(fd, columns, blk, true);
    EXPECT(slurp(fh) == slurp(fd), "host-array and device-array writers disagree");
    auto rd = Open(fd);
    EXPECT(rd->Blocks == 3 && rd->Length == 401777, "blocks / length");
    auto io = rd->Ints({"int64", "int"});
    auto fo = rd->Floats({"float32", "float", "log"});
    size_t at = 0;
    for (int k = 0; k < 3; ++k) {
        const size_t n = blk.i64[k].size();
        EXPECT(std::equal(blk.i64[k].begin(), blk.i64[k].end(), io["int64"].begin() + at), "int64 block " << k);
        EXPECT(std::equal(blk.in[k].begin(), blk.in[k].end(), io["int"].begin() + at), "int block " << k);
        EXPECT(std::equal(blk.f32[k].begin(), blk.f32[k].end(), fo["float32"].begin() + at), "float32 block " << k);
        EXPECT(float32sEq(blk.fl[k], fo["float"].data() + at, 0.001f), "float block " << k);
        std::vector<float> clamped = blk.lg[k];
        for (float& v : clamped) v = v < 1e8f ? 1e8f : (v > 9.9e15f ? 9.9e15f : v);
        EXPECT(log32sEq(clamped, fo["log"].data() + at, 0.0101f), "log block " << k);
        at += n;
    }
    rd->Close();
}

// The reader's read-ahead (next block, same columns) must never change what a call returns: blocks asked for front to back (every
// prefetch used), backwards, and with the column set changing between calls (every prefetch missed) give the floats and
// ints of a reader that does not read ahead (same dither seed), value for value.
static void TestReadAhead() {
    const std::vector<Column> columns = {col(Int64), col(Float32), col(Int), col(Float, 0, 0, 250, 0.001f), col(Float, 1, 8, 16, 0.01f)};
    Blocks blk;
    uint64_t s = 1234567ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    const int nblk = 7;
    for (int k = 0; k < nblk; ++k) {
        const size_t n = 30000 + 1111 * (size_t)k;
        std::vector<int64_t> a(n), c(n); std::vector<float> b(n), d(n), e(n);
        for (size_t i = 0; i < n; ++i) {
            a[i] = (int64_t)(rnd() * 1e12); b[i] = (float)(rnd() * 100.0); c[i] = 500 + (int64_t)(rnd() * 90000);
            d[i] = (float)(rnd() * 250.0); e[i] = (float)std::pow(10.0, 8.5 + 7.0 * rnd());
        }
        blk.i64.push_back(a); blk.f32.push_back(b); blk.in.push_back(c); blk.fl.push_back(d); blk.lg.push_back(e);
    }
    const std::string fn = scratch + "/ahead.minh";
    writeFile(fn, columns, blk, true);
    auto plain = Open(fn), ahead = Open(fn);
    plain->ReadAhead = false; plain->DitherSeed = 77; ahead->DitherSeed = 77;
    auto same = [&](int b, const std::vector<std::string>& in, const std::vector<std::string>& fn2) {
        const size_t n = (size_t)plain->BlockLengths[(size_t)b];
        std::vector<std::vector<int64_t>> i1(in.size(), std::vector<int64_t>(n)), i2 = i1;
        std::vector<std::vector<float>> f1(fn2.size(), std::vector<float>(n)), f2 = f1;
        std::map<std::string, int64_t*> p1, p2; std::map<std::string, float*> q1, q2;
        for (size_t k = 0; k < in.size(); ++k) { p1[in[k]] = i1[k].data(); p2[in[k]] = i2[k].data(); }
        for (size_t k = 0; k < fn2.size(); ++k) { q1[fn2[k]] = f1[k].data(); q2[fn2[k]] = f2[k].data(); }
        if (!in.empty()) { plain->IntBlock(b, p1); ahead->IntBlock(b, p2); }
        if (!fn2.empty()) { plain->FloatBlock(b, q1); ahead->FloatBlock(b, q2); }
        bool ok = true;
        for (size_t k = 0; k < in.size(); ++k) ok = ok && i1[k] == i2[k];
        for (size_t k = 0; k < fn2.size(); ++k) ok = ok && std::memcmp(f1[k].data(), f2[k].data(), n * 4) == 0;
        return ok;
    };
    for (int b = 0; b < nblk; ++b) EXPECT(same(b, {"int64", "int"}, {"float32", "float", "log"}), "front to back, block " << b);
    for (int b = nblk - 1; b >= 0; --b) EXPECT(same(b, {"int64", "int"}, {"float", "log"}), "backwards, block " << b);
    for (int b = 0; b < nblk; ++b) {
        const bool odd = b & 1;
        EXPECT(same(b, odd ? std::vector<std::string>{"int"} : std::vector<std::string>{"int64", "int"},
                    odd ? std::vector<std::string>{"log"} : std::vector<std::string>{"float", "float32"}), "changing columns, block " << b);
    }
    EXPECT(same(3, {}, {"log"}) && same(4, {}, {"log"}) && same(4, {"int"}, {}), "single kinds");
    plain->Close(); ahead->Close();
}

// a NaN survives processFloatGroup (both comparisons are false), quantizes to MinInt64 on amd64 and makes the block 64 bits wide:
// the writer must then take its wide stream slots, from host and from device arrays alike
static void TestNaNBlock() {
    const std::vector<Column> columns = {col(Int64), col(Float32), col(Int), col(Float, 0, 0, 250, 0.001f), col(Float, 1, 8, 16, 0.01f)};
    Blocks blk;
    const size_t n = 5000;
    std::vector<int64_t> a(n, 7), c(n, 9);
    std::vector<float> b(n, 1.5f), d(n), e(n, 1e10f);
    for (size_t i = 0; i < n; ++i) d[i] = (float)(i % 250);
    d[1234] = std::nanf("");
    blk.i64 = {a}; blk.f32 = {b}; blk.in = {c}; blk.fl = {d}; blk.lg = {e};
    const std::string fh = scratch + "/nan_host.minh", fd = scratch + "/nan_dev.minh";
    writeFile(fh, columns, blk, false);
    writeFile(fd, columns, blk, true);
    const auto hb = slurp(fh);
    EXPECT(hb == slurp(fd) && hb.size() > 8 * n, "NaN block: host-array and device-array writers disagree or the block is not wide");
    auto rd = Open(fh);
    std::vector<int64_t> in(n);
    rd->IntBlock(0, {{"int", in.data()}});
    EXPECT(in == c, "NaN block: the other columns are intact");
    rd->Close();
}

// the reference panics (minh.go:100-103, 283-293, 329-334)
static void TestPanics() {
    const std::string fname = scratch + "/panics.minh";
    auto wr = Create(fname);
    wr->Header({"a", "b"}, "", {col(Int), col(Float, 0, 0, 1, 0.5f)});
    bool threw = false;
    const int64_t x[2] = {1, 2};
    try { wr->Block({x}, 2); } catch (const Error&) { threw = true; }
    EXPECT(threw, "Block with a missing column must throw");
    const float y[2] = {0.25f, 0.75f};
    wr->Block({x, y}, 2);
    wr->Close();
    auto rd = Open(fname);
    float out[2]; int64_t iout[2];
    threw = false;
    try { rd->FloatBlock(0, {{"nope", out}}); } catch (const Error&) { threw = true; }
    EXPECT(threw, "unknown name must throw");
    threw = false;
    try { rd->FloatBlock(0, {{"a", out}}); } catch (const Error&) { threw = true; }
    EXPECT(threw, "[]float32 for an IntGroup column must throw");
    threw = false;
    try { rd->IntBlock(0, {{"b", iout}}); } catch (const Error&) { threw = true; }
    EXPECT(threw, "[]int64 for a FloatGroup column must throw");
    rd->IntBlock(0, {{"a", iout}});
    EXPECT(iout[0] == 1 && iout[1] == 2, "int column");
    threw = false;
    try { Open(scratch + "/reader_writer_minh.test.missing"); } catch (const Error&) { threw = true; }
    EXPECT(threw, "missing file must throw");
}

int main(int argc, char** argv) {
    if (argc < 3) { std::cerr << "usage: test_minh_host <scratch dir> <golden dir>\n"; return 2; }
    scratch = argv[1]; golden = argv[2];
    try {
        TestReaderWriter();
        TestGoldenFile();
        TestLargeBlocks();
        TestReadAhead();
        TestNaNBlock();
        TestPanics();
    } catch (const std::exception& e) {
        std::cerr << "EXCEPTION: " << e.what() << "\n";
        return 1;
    }
    if (failures) { std::cerr << failures << " failure(s)\n"; return 1; }
    std::cout << "ok: 6 minh host tests\n";
    return 0;
}
