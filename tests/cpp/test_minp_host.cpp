// C++ host-side tests of include/minp.hpp, written after the reference's go/minp/minp_test.go (TestVecReaderWriter, TestIDs)
// so they read the same way, plus: the file minp::Writer::Vectors writes (all 3 * subCells^3 blocks produced on the device
// from the grid) compared byte for byte with the file the per-block path writes (getSubCell on the CPU, in this test, and
// one minnow::Writer::Data call per block -- the reference's own data flow), and the device-grid forms.
// argv[1] = scratch dir.  Run by tests/test_cpp_host.py on a GPU box.
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>

#include "minp.hpp"

using namespace minp;
static int failures = 0;
#define EXPECT(cond, msg) do { if (!(cond)) { std::cerr << "FAIL " << __func__ << ": " << msg << "\n"; ++failures; } } while (0)
static std::string scratch;

static std::vector<uint8_t> slurp(const std::string& p) {
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), {});
}
// minp_test.go:132-155
static std::vector<float> makeVectors(const float offset[3], float L, int nFile) {
    const int nFile3 = nFile * nFile * nFile;
    std::vector<float> vec((size_t)nFile3 * 3);
    const float dx = L / (float)nFile;
    for (int i = 0; i < nFile3; ++i) {
        const int ix = i % nFile, iy = (i / nFile) % nFile, iz = i / (nFile * nFile);
        float v[3] = {offset[0] + (float)ix * dx, offset[1] + (float)iy * dx, offset[2] + (float)iz * dx};
        for (int k = 0; k < 3; ++k) { if (v[k] > L) v[k] -= L; vec[(size_t)i * 3 + k] = v[k]; }
    }
    return vec;
}
static bool vectorsEq(const std::vector<float>& a, const std::vector<float>& b, float dx) {     // minp_test.go:120-130
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i) if (a[i] + dx < b[i] || a[i] - dx > b[i]) return false;
    return true;
}
// getSubCell, minp.go:246-266 (test side only: the reference's data flow, to compare files with)
static void getSubCell(const std::vector<float>& x, std::vector<float> (&subBuf)[3], int sc, int subCells, int nSub) {
    const int nFile = nSub * subCells;
    const int sx = sc % subCells, sy = (sc / subCells) % subCells, sz = sc / (subCells * subCells);
    const int ix0 = nSub * sx, iy0 = nSub * sy, iz0 = nSub * sz;
    size_t j = 0;
    for (int jz = 0; jz < nSub; ++jz)
        for (int jy = 0; jy < nSub; ++jy)
            for (int jx = 0; jx < nSub; ++jx) {
                const size_t i = (size_t)(jx + ix0) + (size_t)(jy + iy0) * nFile + (size_t)(jz + iz0) * nFile * nFile;
                for (int k = 0; k < 3; ++k) subBuf[k][j] = x[i * 3 + k];
                ++j;
            }
}

// minp.Writer.Vectors as the reference runs it: limits, one FloatGroup per component, getSubCell + Data per block (minp.go:86-119)
static void writeBlockByBlock(const std::string& fref, const Header& hd, const std::vector<uint8_t>& rawHd, const Cell& c, double dx, bool periodic,
                              const std::vector<float>& vec, int nFile) {
    const int subCells = (int)c.SubCells, nSub = nFile / subCells, nSub3 = nSub * nSub * nSub, subCells3 = subCells * subCells * subCells;
    float mn[3] = {0, 0, 0}, mx[3] = {(float)hd.L, (float)hd.L, (float)hd.L};
    if (!periodic) {
        for (int k = 0; k < 3; ++k) { mn[k] = vec[k]; mx[k] = vec[k]; }
        for (size_t p = 0; p < vec.size() / 3; ++p)
            for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], vec[p * 3 + k]); mx[k] = std::max(mx[k], vec[p * 3 + k]); }
        for (int k = 0; k < 3; ++k) mx[k] = std::nextafterf(mx[k], 2 * mx[k]);
    }
    auto f = minnow::Create(fref);
    idHeader id{Magic, Version, basicFileType};
    f->Header(id); f->Header(hd); f->Header(rawHd.data(), rawHd.size()); f->Header(c); f->Header(dx);
    const uint8_t p = periodic ? 1 : 0;
    f->Header(p);
    std::vector<float> subBuf[3] = {std::vector<float>((size_t)nSub3), std::vector<float>((size_t)nSub3), std::vector<float>((size_t)nSub3)};
    for (int k = 0; k < 3; ++k) {
        f->FloatGroup(nSub3, mn[k], mx[k], (float)dx);
        for (int sc = 0; sc < subCells3; ++sc) { getSubCell(vec, subBuf, sc, subCells, nSub); f->Data(subBuf[k]); }
    }
    f->Close();
}

// go/minp/minp_test.go:7-73
static void TestVecReaderWriter() {
    const double dx = 0.1;
    const int tests[][3] = {{1, 1, 1}, {2, 1, 1}, {8, 1, 1}, {10, 1, 1}, {2, 1, 2}, {10, 1, 2}, {10, 1, 5}, {48, 1, 4}, {40, 2, 5}};
    Header hd;
    hd.Z = 1.0; hd.Scale = 0.5; hd.OmegaM = 0.27; hd.OmegaL = 0.73; hd.H100 = 0.7; hd.Epsilon = 2.0; hd.UniformMp = 1e10; hd.L = 100.0;
    std::vector<uint8_t> rawHd(130);
    for (size_t i = 0; i < rawHd.size(); ++i) rawHd[i] = (uint8_t)i;
    const std::string fname = scratch + "/test.minp";
    int i = 0;
    for (auto& t : tests) {
        hd.NSide = t[0];
        hd.NTotal = hd.NSide * hd.NSide * hd.NSide;
        const int nFile = (int)hd.NSide / t[1];
        Cell c;
        c.FileIndex = 0; c.FileCells = t[1]; c.SubCells = t[2];
        for (int periodic = 0; periodic < 2; ++periodic) {
            const float off[3] = {0, 0, 0};
            const std::vector<float> vec = makeVectors(off, 100, nFile);
            auto wr = Create(fname);
            wr->Header(hd, rawHd, c, dx, periodic != 0);
            wr->Vectors(vec.data(), vec.size() / 3);
            wr->Close();
            std::vector<float> out(vec.size());
            auto rd = Open(fname);
            rd->Vectors(out.data());
            EXPECT(vectorsEq(vec, out, (float)dx), i << ") vector read failed for nSide = " << t[0] << ", subCells = " << t[2] << ", periodic = " << periodic);
            EXPECT(hd == rd->Header, i << ") header");
            EXPECT(rawHd == rd->RawHeader, i << ") raw header");
            EXPECT(rd->FileIndex == 0 && rd->FileCells == t[1] && rd->Dx == dx && rd->Periodic == (periodic != 0), i << ") Incorrect header read.");
            rd->Close();

            // the same file through the reference's data flow: one FloatGroup per component, getSubCell + Data per block
            const std::string fref = scratch + "/test_blocks.minp";
            writeBlockByBlock(fref, hd, rawHd, c, dx, periodic != 0, vec, nFile);
            EXPECT(slurp(fname) == slurp(fref), i << ") the file differs from the one written block by block, nSide = " << t[0] << " subCells = " << t[2] << " periodic = " << periodic);
        }
        ++i;
    }
}

// go/minp/minp_test.go:75-100
static void TestIDs() {
    const int nSide = 10, fileCells = 5, subCells = 2, nFile = nSide / fileCells;
    const int indices[2] = {0, 3 + 5 * 2 + 25 * 1};
    const std::vector<int64_t> ids[2] = {{0, 1, 10, 11, 100, 101, 110, 111}, {246, 247, 256, 257, 346, 347, 356, 357}};
    const std::string fname = scratch + "/test_ids.minp";
    for (int i = 0; i < 2; ++i) {
        auto wr = Create(fname);
        Header hd;
        hd.NSide = 10; hd.L = 100;
        Cell c;
        c.FileIndex = indices[i]; c.FileCells = fileCells; c.SubCells = subCells;
        wr->Header(hd, {}, c, 1.0, true);
        const std::vector<float> zero((size_t)nFile * nFile * nFile * 3, 0.0f);
        wr->Vectors(zero.data(), zero.size() / 3);
        wr->Close();
        auto rd = Open(fname);
        std::vector<int64_t> out((size_t)nFile * nFile * nFile);
        rd->IDs(out.data());
        EXPECT(out == ids[i], i << ") IDs");
        rd->Close();
    }
}

// device grids in and out; velocities (non-periodic, data-derived limits with negative values); the reference's panics
static void TestDeviceAndPanics() {
    mnw_ctx* ctx = minnow::detail::ctx();
    const int nFile = 20;
    const size_t n = (size_t)nFile * nFile * nFile;
    std::vector<float> vel(n * 3);
    uint64_t s = 1234567;
    for (float& v : vel) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (float)((double)(s >> 11) / 9007199254740992.0 * 2000.0 - 1000.0); }
    Header hd;
    hd.NSide = nFile; hd.NTotal = (int64_t)n; hd.L = 250;
    Cell c;
    c.FileIndex = 0; c.FileCells = 1; c.SubCells = 5;
    const std::string fh = scratch + "/vel_host.minp", fd = scratch + "/vel_dev.minp";
    { auto wr = Create(fh); wr->Header(hd, {}, c, 1.0, false); wr->Vectors(vel.data(), n); wr->Close(); }
    void* d = nullptr;
    minnow::detail::check(mnw_dev_alloc(ctx, (int64_t)(n * 12), &d));
    minnow::detail::check(mnw_copy_to_device(ctx, d, vel.data(), (int64_t)(n * 12)));
    { auto wr = Create(fd); wr->Header(hd, {}, c, 1.0, false); wr->VectorsDevice(static_cast<const float*>(d), n); wr->Close(); }
    EXPECT(slurp(fh) == slurp(fd), "host-grid and device-grid writers disagree");
    writeBlockByBlock(scratch + "/vel_blocks.minp", hd, {}, c, 1.0, false, vel, nFile);
    EXPECT(slurp(fh) == slurp(scratch + "/vel_blocks.minp"), "velocity file differs from the one written block by block");
    {   // ... and read block by block (minnow::Reader::Data + setSubCell on the CPU), as the reference reads it
        auto f = minnow::Open(fh);
        const int subCells = 5, nSub = nFile / subCells, nSub3 = nSub * nSub * nSub, subCells3 = 125;
        std::vector<float> ref(n * 3), buf((size_t)nSub3);
        for (int k = 0; k < 3; ++k)
            for (int sc = 0; sc < subCells3; ++sc) {
                f->Data(k * subCells3 + sc, buf.data(), buf.size());
                const int sx = sc % subCells, sy = (sc / subCells) % subCells, sz = sc / (subCells * subCells);
                size_t j = 0;
                for (int jz = 0; jz < nSub; ++jz) for (int jy = 0; jy < nSub; ++jy) for (int jx = 0; jx < nSub; ++jx) {
                    const size_t i = (size_t)(jx + nSub * sx) + (size_t)(jy + nSub * sy) * nFile + (size_t)(jz + nSub * sz) * nFile * nFile;
                    ref[i * 3 + k] = buf[j++];
                }
            }
        f->Close();
        auto rd2 = Open(fh);
        std::vector<float> got(n * 3);
        rd2->Vectors(got.data());
        rd2->Close();
        size_t diff = 0;
        for (size_t q = 0; q < got.size(); ++q) diff += got[q] != ref[q];
        EXPECT(diff == 0, "Reader::Vectors differs from the block-by-block read in " << diff << " values");
    }
    auto rd = Open(fd);
    void* dout = nullptr;
    minnow::detail::check(mnw_dev_alloc(ctx, (int64_t)(n * 12), &dout));
    rd->VectorsDevice(static_cast<float*>(dout));
    std::vector<float> out(n * 3), out2(n * 3);
    minnow::detail::check(mnw_copy_to_host(ctx, out.data(), dout, (int64_t)(n * 12)));
    rd->Vectors(out2.data());
    {   // within dv -- except a component's maximum, which the reference itself can send to the bottom of the range: the limits are
        // [min, Nextafter32(max)), and (max - min) / dx_q may round up to `pixels` in float32, which the always-periodic
        // FloatGroup wraps to pixel 0 (writer.go:74, minp.go:92-95)
        float mx[3] = {vel[0], vel[1], vel[2]}, mn[3] = {vel[0], vel[1], vel[2]};
        for (size_t p = 0; p < n; ++p) for (int k = 0; k < 3; ++k) { mx[k] = std::max(mx[k], vel[p * 3 + k]); mn[k] = std::min(mn[k], vel[p * 3 + k]); }
        size_t badv = 0, wrapped = 0;
        for (size_t q = 0; q < vel.size(); ++q) {
            const bool ok = !(vel[q] + 1.0f < out[q] || vel[q] - 1.0f > out[q]);
            if (ok) continue;
            if (vel[q] == mx[q % 3] && out[q] < mn[q % 3] + 1.0f) ++wrapped; else ++badv;
        }
        EXPECT(badv == 0 && wrapped <= 3, "velocities within dv: " << badv << " off, " << wrapped << " maxima wrapped");
    }
    EXPECT(out == out2, "device-grid and host-grid readers disagree");
    EXPECT(rd->N() == 125, "N() = blocks / 3");
    rd->Close();
    mnw_dev_free(ctx, d); mnw_dev_free(ctx, dout);
    bool threw = false;
    try { auto wr = Create(scratch + "/bad.minp"); wr->Header(hd, {}, c, 1.0, true); wr->Vectors(vel.data(), n - 1); } catch (const Error&) { threw = true; }
    EXPECT(threw, "len(vec) != NFile^3 must throw");
    threw = false;
    try { Cell bad; bad.FileCells = 3; bad.SubCells = 1; bad.NFile(20); } catch (const Error&) { threw = true; }
    EXPECT(threw, "NSide not a multiple of FileCells must throw");
    threw = false;
    try { Open(fh + ".missing"); } catch (const Error&) { threw = true; }
    EXPECT(threw, "missing file must throw");
}

int main(int argc, char** argv) {
    if (argc < 2) { std::cerr << "usage: test_minp_host <scratch dir>\n"; return 2; }
    scratch = argv[1];
    try {
        TestVecReaderWriter();
        TestIDs();
        TestDeviceAndPanics();
    } catch (const std::exception& e) {
        std::cerr << "EXCEPTION: " << e.what() << "\n";
        return 1;
    }
    if (failures) { std::cerr << failures << " failure(s)\n"; return 1; }
    std::cout << "ok: 3 minp host tests\n";
    return 0;
}
