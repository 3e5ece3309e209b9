// C++ host-side tests over libminnow_hip.so, written after the reference's own Go tests
// (go/bit/bit_test.go, go/minnow_test.go) so they read the same way.  Run by tests/test_cpp_host_gpu.py
// on a GPU box: argv[1] = scratch dir (the reference uses ../test_files/), argv[2] = dir holding the
// golden .minw files written by the reference's Python Writer (tests/golden/golden_small.json).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <random>
#include <string>
#include <vector>

#include "minnow.hpp"

using namespace minnow;
static int failures = 0;
#define EXPECT(cond, msg) do { if (!(cond)) { std::cerr << "FAIL " << __func__ << ": " << msg << "\n"; ++failures; } } while (0)

static std::string scratch, golden;

// go/bit/bit_test.go:9-31
static void TestArray() {
    std::mt19937_64 rng(1);
    std::vector<uint64_t> data(123), out(123);
    for (auto& d : data) d = rng() >> 1;                       // rand.Int63()
    for (int bits = 1; bits <= 64; ++bits) {
        bit::Array arr = bit::NewArray(bits, data.data(), data.size());
        arr.Slice(out.data(), out.size());
        const uint64_t mask = bits == 64 ? ~0ull : ~(~0ull << bits);
        for (size_t i = 0; i < out.size(); ++i) EXPECT(out[i] == (data[i] & mask), "bits " << bits << " i " << i);
    }
}

// go/bit/bit_test.go:33-69
static void TestArrayBuffer() {
    const std::string fname = scratch + "/array_buffer.test";
    std::FILE* f = std::fopen(fname.c_str(), "wb");
    const int lengths[4] = {10, 5, 1, 20};
    int bits[4];
    bit::ArrayBuffer ab;
    for (int i = 0; i < 4; ++i) {
        uint64_t* data = ab.Uint64(lengths[i]);
        for (int j = 0; j < lengths[i]; ++j) data[j] = (uint64_t)j;
        bits[i] = ab.Bits(data, lengths[i]);
        ab.Write(f, data, lengths[i], bits[i]);
    }
    std::fclose(f);
    bit::ArrayBuffer ab2;
    f = std::fopen(fname.c_str(), "rb");
    for (int i = 0; i < 4; ++i) {
        uint64_t* data = ab2.Read(f, bits[i], lengths[i]);
        for (int j = 0; j < lengths[i]; ++j) EXPECT((int)data[j] == j, "array_" << i << "[" << j << "]");
    }
    std::fclose(f);
    EXPECT(bits[0] == 4 && bits[1] == 3 && bits[2] == 0 && bits[3] == 5, "Bits()");
}

// error behaviour: the reference panics (bit.go:85-92, 30-33)
static void TestBitPanics() {
    std::vector<uint64_t> x(4, 0);
    std::vector<uint8_t> b(5);
    bool threw = false;
    try { bit::BufferedArray(8, x.data(), 4, b.data(), b.size()); } catch (const Error&) { threw = true; }
    EXPECT(threw, "BufferedArray with a wrong buffer length must throw");
    threw = false;
    try { bit::NewArray(65, x.data(), 4); } catch (const Error&) { threw = true; }
    EXPECT(threw, "more than 64 bits must throw");
    bit::Array a = bit::NewArray(8, x.data(), 4);
    std::vector<uint64_t> out(3);
    threw = false;
    try { a.Slice(out.data(), out.size()); } catch (const Error&) { threw = true; }
    EXPECT(threw, "Slice into a short buffer must throw");
}

// go/minnow_test.go:191-220
struct int64RecordHead { uint64_t Magic, Blocks; };
static void TestInt64Record() {
    const std::string fname = scratch + "/int_record.test";
    std::vector<std::vector<int64_t>> xs = {{1, 2, 3, 4}, {5}, {6, 7, 8, 9}, {10, 11, 12}};
    const std::string text = "I am a cat and I like to meow.";
    {
        auto f = Create(fname);
        int64RecordHead hd{0xdeadbeef, xs.size()};
        std::vector<uint64_t> lengths;
        for (auto& x : xs) lengths.push_back(x.size());
        f->Header(hd);
        f->Header(text.data(), text.size());
        for (auto& x : xs) { f->FixedSizeGroup(Int64Group, (int)x.size()); f->Data(x); }
        f->Header(lengths.data(), lengths.size() * 8);
        f->Close();
    }
    auto f = Open(fname);
    int64RecordHead hd;
    f->Header(0, hd);
    std::string rdText(f->HeaderSize(1), ' ');
    f->Header(1, &rdText[0], rdText.size());
    std::vector<uint64_t> lengths(hd.Blocks);
    f->Header(2, lengths.data(), lengths.size() * 8);
    EXPECT(rdText == text, "text");
    EXPECT(hd.Blocks == xs.size(), "blocks");
    for (size_t i = 0; i < xs.size(); ++i) {
        std::vector<int64_t> rd(lengths[i]);
        f->Data((int)i, rd);
        EXPECT(rd == xs[i], "xs[" << i << "]");
    }
}

// go/minnow_test.go:222-240
struct groupRecordHeader { int64_t Blocks, N; };
static void TestGroupRecord() {
    const std::string fname = scratch + "/group_files.test";
    std::vector<int32_t> ix(20);
    std::vector<double> fx(10);
    for (int i = 0; i < 20; ++i) ix[i] = i;
    for (int i = 0; i < 10; ++i) fx[i] = i / 10.0;
    const std::string text = "I'm a caaaat";
    {
        auto f = Create(fname);
        const int in = 5, fn = 5;
        f->Header(groupRecordHeader{4, in});
        f->FixedSizeGroup(Int32Group, in);
        for (int i = 0; i < 4; ++i) f->Data(ix.data() + i * in, in);
        f->Header(groupRecordHeader{2, fn});
        f->FixedSizeGroup(Float64Group, fn);
        for (int i = 0; i < 2; ++i) f->Data(fx.data() + i * fn, fn);
        f->Header(text.data(), text.size());
        f->Close();
    }
    auto f = Open(fname);
    groupRecordHeader iHd, fHd;
    f->Header(0, iHd); f->Header(1, fHd);
    std::vector<int32_t> rdIx(iHd.Blocks * iHd.N);
    std::vector<double> rdFx(fHd.Blocks * fHd.N);
    for (int64_t i = 0; i < iHd.Blocks; ++i) f->Data((int)i, rdIx.data() + i * iHd.N, iHd.N);
    for (int64_t i = 0; i < fHd.Blocks; ++i) f->Data((int)(i + iHd.Blocks), rdFx.data() + i * fHd.N, fHd.N);
    EXPECT(rdIx == ix, "ix");
    EXPECT(rdFx == fx, "fx");
    bool threw = false;   // TypeMatch, group.go:43-71
    try { std::vector<float> wrong(5); f->Data(0, wrong); } catch (const Error&) { threw = true; }
    EXPECT(threw, "reading an Int32Group block into []float32 must throw");
}

static std::vector<uint8_t> slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// go/minnow_test.go:242-268 -- and the file bytes against the reference's Python Writer
static void TestBitIntRecord() {
    const std::string fname = scratch + "/bit_int_record.test";
    std::vector<int64_t> x1 = {100, 101, 102, 104};
    std::vector<std::vector<int64_t>> x2 = {{1024, 1024, 1024}, {0, 1023, 500}};
    std::vector<int64_t> x3 = {-1000000, -500000};
    {
        auto f = Create(fname);
        f->IntGroup((int)x1.size()); f->Data(x1);
        f->Header((int64_t)x2.size());
        f->IntGroup((int)x2[0].size());
        for (auto& x : x2) f->Data(x);
        f->IntGroup((int)x3.size()); f->Data(x3);
        f->Close();
    }
    {
        auto f = Open(fname);
        int64_t x2Len;
        f->Header(0, x2Len);
        std::vector<int64_t> r1(f->DataLen(0)), r3(f->DataLen((int)x2Len + 1));
        f->Data(0, r1);
        EXPECT(r1 == x1, "x1");
        for (int i = 0; i < x2Len; ++i) {
            std::vector<int64_t> r(f->DataLen(1 + i));
            f->Data(i + 1, r);
            EXPECT(r == x2[i], "x2[" << i << "]");
        }
        f->Data((int)x2Len + 1, r3);
        EXPECT(r3 == x3, "x3");
    }
    if (!golden.empty()) {
        EXPECT(slurp(fname) == slurp(golden + "/bit_int_record.minw"), "file bytes differ from the reference Writer's");
        auto f = Open(golden + "/bit_int_record.minw");       // and the reference's file reads back
        std::vector<int64_t> r(3);
        f->Data(2, r);
        EXPECT((r == std::vector<int64_t>{0, 1023, 500}), "golden file block 2");
    }
}

static bool float32sEq(const std::vector<float>& x, const std::vector<float>& y, float eps) {   // minnow_test.go:328-334
    if (x.size() != y.size()) return false;
    for (size_t i = 0; i < x.size(); ++i) if (x[i] + eps < y[i] || x[i] - eps > y[i]) return false;
    return true;
}

// go/minnow_test.go:270-310 (the golden file has no header block, like the script that wrote it)
static void TestQFloatRecord() {
    const std::string fname = scratch + "/q_float_record.test";
    const float lim0 = -50, lim1 = 100, dx1 = 1.0f, dx2 = 10.0f;
    std::vector<std::vector<float>> x1 = {{-50, 0, 50, 49}, {25, 25, 25, 25}};
    std::vector<std::vector<float>> x2 = {{-50, 0, 50, 49, 0}, {1, 2, 3, 4, 5}, {0, 20, 0, 20, 0}};
    {
        auto f = Create(fname);
        f->FloatGroup(4, lim0, lim1, dx1);
        for (auto& x : x1) f->Data(x);
        f->FloatGroup(5, lim0, lim1, dx2);
        for (auto& x : x2) f->Data(x);
        f->Close();
    }
    for (const std::string& path : {fname, golden.empty() ? fname : golden + "/q_float_record.minw"}) {
        auto f = Open(path);
        for (size_t i = 0; i < x1.size(); ++i) {
            std::vector<float> r(f->DataLen((int)i));
            f->Data((int)i, r);
            EXPECT(float32sEq(x1[i], r, dx1), "x1[" << i << "] of " << path);
        }
        for (size_t i = 0; i < x2.size(); ++i) {
            std::vector<float> r(f->DataLen((int)(i + x1.size())));
            f->Data((int)(i + x1.size()), r);
            EXPECT(float32sEq(x2[i], r, dx2), "x2[" << i << "] of " << path);
        }
    }
    if (!golden.empty())
        EXPECT(slurp(fname) == slurp(golden + "/q_float_record.minw"), "file bytes differ from the reference Writer's");
    bool threw = false;   // writer.go:91-93
    try { auto f = Create(scratch + "/nogroup.test"); std::vector<int64_t> x(3); f->Data(x); } catch (const Error&) { threw = true; }
    EXPECT(threw, "Data without a group must throw");
}

int main(int argc, char** argv) {
    scratch = argc > 1 ? argv[1] : "/tmp";
    golden = argc > 2 ? argv[2] : "";
    try {
        TestArray(); TestArrayBuffer(); TestBitPanics(); TestInt64Record(); TestGroupRecord(); TestBitIntRecord(); TestQFloatRecord();
    } catch (const std::exception& e) {
        std::cerr << "EXCEPTION: " << e.what() << "\n";
        return 2;
    }
    if (failures) { std::cerr << failures << " failure(s)\n"; return 1; }
    std::cout << "ok: 7 host tests\n";
    return 0;
}
