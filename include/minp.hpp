// minp particle snapshots (go/minp/minp.go) over include/minnow.hpp: the reference's Header / Cell / Writer / Reader with
// their names, argument meaning and panics (as exceptions).  minp.Writer.Vectors gathers every sub-cell of the file's
// Lagrangian grid on the CPU -- three times over the same memory -- and encodes one block per (component, sub-cell)
// (minp.go:85-118); here the grid goes to the GPU once and mnw_vectors_encode produces all 3 * subCells^3 blocks from it
// (DESIGN.md 3.8); Reader.Vectors is the inverse (mnw_vectors_decode: decode, periodic fix-up, setSubCell).
// Vectors takes host arrays like the Go methods, VectorsDevice a grid that already lives on the device.
//
// Header-only, C++17, needs only the C ABI (include/minnow_hip.h).
#pragma once

#include <cmath>

#include "minnow.hpp"

namespace minp {

using minnow::Error;

constexpr int64_t Magic = 0xbadf00d;         // minp.go:10-15
constexpr int64_t Version = 0;
constexpr int64_t basicFileType = 2;         // the const block's iota: basicFileType is its third entry

struct idHeader { int64_t Magic, Version, FileType; };                 // minp.go:17-19

// minp.go:24-30
struct Header {
    double Z = 0, Scale = 0;                 // redshift, scale factor
    double OmegaM = 0, OmegaL = 0, H100 = 0;
    double L = 0, Epsilon = 0;               // box size, force softening
    int64_t NSide = 0, NTotal = 0;
    double UniformMp = 0;
    bool operator==(const Header& o) const { return std::memcmp(this, &o, sizeof(Header)) == 0; }
};
static_assert(sizeof(Header) == 80, "Header is ten 8-byte fields");

// minp.go:32-49
struct Cell {
    int64_t FileIndex = 0, FileCells = 0, SubCells = 0;
    int NFile(int nSide) const {
        if (nSide < 0 || FileCells <= 0 || nSide % (int)FileCells != 0)
            throw Error("NSide = " + std::to_string(nSide) + " not a valid combination with FileCells = " + std::to_string(FileCells));
        return nSide / (int)FileCells;
    }
    void FileCoord(int& fx, int& fy, int& fz) const {
        fx = (int)(FileIndex % FileCells);
        fy = (int)((FileIndex / FileCells) % FileCells);
        fz = (int)(FileIndex / (FileCells * FileCells));
    }
};
static_assert(sizeof(Cell) == 24, "Cell is three int64");

namespace detail {
struct DevBuf {                              // grow-only device buffer
    void* p = nullptr;
    size_t cap = 0;
    DevBuf() {}
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) mnw_dev_free(minnow::detail::ctx(), p); }
    uint8_t* get(size_t n) {
        if (n > cap) {
            if (p) minnow::detail::check(mnw_dev_free(minnow::detail::ctx(), p));
            p = nullptr; cap = 0;
            const size_t want = n + n / 4 + 256;
            minnow::detail::check(mnw_dev_alloc(minnow::detail::ctx(), (int64_t)want, &p));
            cap = want;
        }
        return static_cast<uint8_t*>(p);
    }
};
inline size_t align16(size_t n) { return (n + 15) & ~(size_t)15; }

// Scratch that outlives the Writer / Reader objects: one object per file is the reference's usage (minp.go:64, 141), and a
// snapshot is hundreds of files -- buffers owned by the object would be allocated, page-faulted in (0.1 s per GB of host memory)
// and freed again for every file.  Per thread, grow-only, used only inside one Vectors call; ReleaseScratch() gives it back.
struct Scratch {
    DevBuf grid, streams;
    minnow::detail::RawBuf host;
};
inline Scratch& scratch() { static thread_local Scratch s; return s; }
}  // namespace detail
inline void ReleaseScratch() {
    detail::Scratch& s = detail::scratch();
    if (s.grid.p) { mnw_dev_free(minnow::detail::ctx(), s.grid.p); s.grid.p = nullptr; s.grid.cap = 0; }
    if (s.streams.p) { mnw_dev_free(minnow::detail::ctx(), s.streams.p); s.streams.p = nullptr; s.streams.cap = 0; }
    s.host.p.reset(); s.host.cap = 0;
}

// ------------------------------------------------------------------------------------------------
// minp.Writer, minp.go:55-123
class Writer {
   public:
    explicit Writer(const std::string& fname) : f_(minnow::Create(fname)) {          // Create, minp.go:64-69
        idHeader hd{Magic, Version, basicFileType};
        f_->Header(hd);
    }

    // Header, minp.go:71-84
    void Header(const minp::Header& hd, const std::vector<uint8_t>& rawHd, const Cell& c, double dx, bool periodic) {
        f_->Header(hd);
        f_->Header(rawHd.data(), rawHd.size());
        f_->Header(c);
        f_->Header(dx);
        const uint8_t p = periodic ? 1 : 0;
        f_->Header(p);
        hd_ = hd; c_ = c; periodic_ = periodic; dx_ = (float)dx;
    }

    // Vectors, minp.go:86-119: vec = [n][3]float32 on the host
    void Vectors(const float* vec, size_t n) {
        mnw_ctx* ctx = minnow::detail::ctx();
        float* d = reinterpret_cast<float*>(detail::scratch().grid.get(n * 12 + 16));
        minnow::detail::check(mnw_copy_to_device(ctx, d, vec, (int64_t)(n * 12)));
        VectorsDevice(d, n);
    }
    // the same for a grid that is on the device already (16-byte aligned)
    void VectorsDevice(const float* dvec, size_t n) {
        mnw_ctx* ctx = minnow::detail::ctx();
        const int nFile = c_.NFile((int)hd_.NSide);
        const int subCells = (int)c_.SubCells;
        if (subCells <= 0 || nFile % subCells != 0) throw Error("SubCells = " + std::to_string(subCells) + " does not divide " + std::to_string(nFile));
        const int nSub = nFile / subCells;
        const int64_t nSub3 = (int64_t)nSub * nSub * nSub, subCells3 = (int64_t)subCells * subCells * subCells;
        if ((size_t)nFile * nFile * nFile != n)
            throw Error("len(vec) = " + std::to_string(n) + ", but NSide = " + std::to_string(hd_.NSide) + " and FileCells = " + std::to_string(c_.FileCells));
        float mn[3], mx[3];
        if (periodic_) {
            const float L = (float)hd_.L;
            for (int k = 0; k < 3; ++k) { mn[k] = 0; mx[k] = L; }
        } else {
            minnow::detail::check(mnw_vectors_bounds(ctx, nullptr, dvec, (int64_t)n, mn, mx));           // bounds(), minp.go:290-300
            for (int k = 0; k < 3; ++k) mx[k] = std::nextafterf(mx[k], 2 * mx[k]);                         // math.Nextafter32(max[i], 2*max[i])
        }
        int64_t pixels[3];
        for (int k = 0; k < 3; ++k) pixels[k] = mnw_float_group_pixels(mn[k], mx[k], dx_);                // Writer.FloatGroup, writer.go:72-75
        std::vector<mnw_block_meta> meta((size_t)(3 * subCells3));
        size_t slot = 0;
        uint8_t* dst = nullptr;
        for (int wide = 0;; ++wide) {        // 32 bits per element hold any block of a sane column; a NaN needs up to 64 (see minh.hpp)
            slot = detail::align16((size_t)(wide ? 8 : 4) * (size_t)nSub3 + 16);
            dst = detail::scratch().streams.get(slot * (size_t)(3 * subCells3));
            const int rc = mnw_vectors_encode(ctx, nullptr, dvec, nFile, subCells, mn, mx, pixels, dst, (int64_t)slot, meta.data());
            if (rc == MNW_E_CAPACITY && !wide) continue;
            minnow::detail::check(rc);
            break;
        }
        // all 3 * subCells^3 streams come back in one pipelined transfer
        const size_t nb = (size_t)(3 * subCells3);
        std::vector<size_t> at(nb + 1, 0);
        for (size_t b = 0; b < nb; ++b) at[b + 1] = at[b] + (((size_t)meta[b].nbytes + 63) & ~(size_t)63);
        uint8_t* const hraw = detail::scratch().host.get(at[nb] + 64);
        {
            std::vector<void*> d(nb);
            std::vector<const void*> sp(nb);
            std::vector<int64_t> len(nb);
            for (size_t b = 0; b < nb; ++b) { d[b] = hraw + at[b]; sp[b] = dst + b * slot; len[b] = meta[b].nbytes; }
            minnow::detail::check(mnw_copy_to_host_batch(ctx, (int)nb, d.data(), sp.data(), len.data()));
        }
        for (int k = 0; k < 3; ++k) {                                  // file order: one FloatGroup per component, a block per sub-cell
            f_->FloatGroup((int)nSub3, mn[k], mx[k], dx_);
            for (int64_t sc = 0; sc < subCells3; ++sc) {
                const size_t b = (size_t)(k * subCells3 + sc);
                const mnw_block_meta& m = meta[b];
                f_->DataEncoded(hraw + at[b], (size_t)m.nbytes, m.min, m.bits);
            }
        }
    }

    void Close() { f_->Close(); }                                       // minp.go:121-123

   private:
    std::unique_ptr<minnow::Writer> f_;
    Cell c_;
    minp::Header hd_;
    bool periodic_ = false;
    float dx_ = 0;
};
inline std::unique_ptr<Writer> Create(const std::string& fname) { return std::make_unique<Writer>(fname); }

// ------------------------------------------------------------------------------------------------
// minp.Reader, minp.go:129-241
class Reader {
   public:
    minp::Header Header;
    std::vector<uint8_t> RawHeader;
    int FileIndex = 0, FileCells = 0;
    double Dx = 0;
    bool Periodic = false;
    uint64_t DitherSeed = 0;       // replaces the unseeded global math/rand of group.go:308 (DESIGN.md "Dither")

    // Open, minp.go:141-171
    explicit Reader(const std::string& fname) : f_(minnow::Open(fname)) {
        idHeader id;
        f_->Header(0, id);
        if (id.Magic != Magic) throw Error("Not a minp file. Magic number is " + std::to_string(id.Magic) + ", not " + std::to_string(Magic));
        if (id.Version != Version) throw Error("File version = " + std::to_string(id.Version) + ", but code version = " + std::to_string(Version) + ".");
        if (id.FileType != basicFileType) throw Error("File type = " + std::to_string(id.FileType));
        f_->Header(1, Header);
        RawHeader.resize((size_t)f_->HeaderSize(2));
        f_->Header(2, RawHeader.data(), RawHeader.size());
        f_->Header(3, c_);
        f_->Header(4, Dx);
        uint8_t p = 0;
        f_->Header(5, p);
        Periodic = p != 0;
        FileIndex = (int)c_.FileIndex;
        FileCells = (int)c_.FileCells;
    }

    // Vectors, minp.go:175-207: out = [n][3]float32 on the host, n = NFile^3
    void Vectors(float* out) {
        const int nFile = c_.NFile((int)Header.NSide);
        const size_t n = (size_t)nFile * nFile * nFile;
        float* d = reinterpret_cast<float*>(detail::scratch().grid.get(n * 12 + 16));
        VectorsDevice(d);
        minnow::detail::check(mnw_copy_to_host(minnow::detail::ctx(), out, d, (int64_t)(n * 12)));
    }
    // the same into a device grid (16-byte aligned); returns when the grid is complete
    void VectorsDevice(float* dout) {
        mnw_ctx* ctx = minnow::detail::ctx();
        const int nFile = c_.NFile((int)Header.NSide);
        const int subCells = (int)c_.SubCells;
        const int nSub = nFile / subCells;
        const int64_t nSub3 = (int64_t)nSub * nSub * nSub, subCells3 = (int64_t)subCells * subCells * subCells;
        if (f_->Blocks() != 3 * subCells3)
            throw Error("Expected " + std::to_string(3 * subCells) + " sub-cells, but got " + std::to_string(f_->Blocks()));
        // the blocks' streams: file -> one staging buffer of equal slots -> device
        std::vector<mnw_block_meta> meta((size_t)(3 * subCells3));
        float low[3], high[3];
        int64_t pixels[3];
        size_t slot = 16;
        for (int k = 0; k < 3; ++k) {
            auto* fg = dynamic_cast<minnow::floatGroup*>(f_->Group(f_->GroupOf((int)(k * subCells3))));
            if (!fg) throw Error("block " + std::to_string(k * subCells3) + " is not in a FloatGroup");
            low[k] = fg->low; high[k] = fg->high; pixels[k] = fg->pixels;
            for (int64_t sc = 0; sc < subCells3; ++sc) {
                mnw_block_meta& m = meta[(size_t)(k * subCells3 + sc)];
                m.min = fg->ig->mins.at((size_t)sc);
                m.bits = fg->ig->bits.at((size_t)sc);
                m.nbytes = m.bits == 0 ? 0 : minnow::bit::ArrayBytes((int)m.bits, (int)nSub3);
                m.status = 0;
                slot = std::max(slot, detail::align16((size_t)m.nbytes + 16));
            }
        }
        // the file's ranges go to the device through the library's pinned buffers (no host copy in between), each to its slot
        const size_t hostBytes = slot * (size_t)(3 * subCells3);
        uint8_t* dsrc = detail::scratch().streams.get(hostBytes + 16);
        {
            std::vector<int64_t> foff, flen, doff;
            for (int64_t b = 0; b < 3 * subCells3; ++b) {
                foff.push_back(f_->BlockFileOffset((int)b));
                flen.push_back(meta[(size_t)b].nbytes);
                doff.push_back((int64_t)((size_t)b * slot));
            }
            minnow::detail::check(mnw_copy_file_to_device(ctx, dsrc, fileno(f_->File()), (int)foff.size(), foff.data(), flen.data(), doff.data()));
        }
        minnow::detail::check(mnw_vectors_decode(ctx, nullptr, dsrc, (int64_t)slot, meta.data(), nFile, subCells, low, high, pixels,
                                                 Periodic ? 1 : 0, (float)Header.L, DitherSeed, 0, dout));
        uint8_t one;                                                     // a copy on the context's stream = wait for the decode
        minnow::detail::check(mnw_copy_to_host(ctx, &one, dsrc, 1));
    }

    // IDs, minp.go:209-230: the Lagrangian IDs of the file's particles (index arithmetic, no data involved)
    void IDs(int64_t* out) const {
        const int64_t nFile = c_.NFile((int)Header.NSide), nSide = Header.NSide;
        int ifx, ify, ifz;
        c_.FileCoord(ifx, ify, ifz);
        const int64_t ix0 = ifx * nFile, iy0 = ify * nFile, iz0 = ifz * nFile;
        size_t j = 0;
        for (int64_t jz = 0; jz < nFile; ++jz)
            for (int64_t jy = 0; jy < nFile; ++jy)
                for (int64_t jx = 0; jx < nFile; ++jx) out[j++] = (jx + ix0) + (jy + iy0) * nSide + (jz + iz0) * nSide * nSide;
    }

    int N() const { return f_->Blocks() / 3; }                           // minp.go:232-235 (sic: blocks / 3)
    void Close() { f_->Close(); }

   private:
    std::unique_ptr<minnow::Reader> f_;
    Cell c_;
};
inline std::unique_ptr<Reader> Open(const std::string& fname) { return std::make_unique<Reader>(fname); }

}  // namespace minp
