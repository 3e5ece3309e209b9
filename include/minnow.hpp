// minnow.hpp -- C++ host side above the C ABI (include/minnow_hip.h): the reference's operator
// interface for the quantize + bit-pack path and the .minw container around it, with the same
// names, argument meaning and error behaviour as the Go code (the Go toolchain is not available in
// the build image, so the host side is C++).  Header only; link with -lminnow_hip.
//
//   minnow::bit::PrecisionNeeded / ArrayBytes / Array / NewArray / BufferedArray / ArrayBuffer
//                                                        go/bit/bit.go:13-206
//   minnow::fixedSizeGroup / intGroup / floatGroup       go/group.go:77-344
//   minnow::blockIndex                                   go/block_index.go:7-39
//   minnow::Writer  (Create/Header/FixedSizeGroup/IntGroup/FloatGroup/Data/Close)   go/writer.go:10-146
//   minnow::Reader  (Open/Header/HeaderSize/Blocks/Data/DataType/DataLen/Close)     go/reader.go:14-147
//
// Every compute step of the hot path (pack, unpack, min/max scans, quantize, dequantize) runs in the
// HIP library; this header only does what stays in Go behind the cgo shim: file I/O, the per-group
// mins/bits bookkeeping, block offsets and the tail.  Where the reference panics, minnow::Error is
// thrown.  Not thread-safe, like the reference's Writer/Reader (one os.File with an implicit cursor).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "minnow_hip.h"

namespace minnow {

constexpr uint64_t Version = 1;            // go/minnow.go:7
constexpr uint64_t Magic = 0xacedad;       // go/minnow.go:8

// group types, go/group.go:11-24
enum : int64_t {
    Int64Group = 0, Int32Group, Int16Group, Int8Group, Uint64Group, Uint32Group, Uint16Group, Uint8Group,
    Float64Group, Float32Group, IntGroup, FloatGroup
};
inline const char* GroupName(int64_t gt) {
    static const char* n[] = {"Int64Group", "Int32Group", "Int16Group", "Int8Group", "Uint64Group", "Uint32Group",
                              "Uint16Group", "Uint8Group", "Float64Group", "Float32Group", "IntGroup", "FloatGroup"};
    return (gt >= 0 && gt <= FloatGroup) ? n[gt] : "?";
}
static const int fixedSizeBytes[] = {8, 4, 2, 1, 8, 4, 2, 1, 8, 4};   // go/group.go:73-75

// the reference panics on every error (writer.go:34,91-95; reader.go:30-46; bit.go:31,86,90)
struct Error : std::runtime_error {
    int status;
    explicit Error(const std::string& what, int st = MNW_E_ARG) : std::runtime_error(what), status(st) {}
};

namespace detail {
// one library context per process, created on first use (the reference is single-goroutine); on a multi-GPU node the
// files of a conversion shard over one process per GPU: device = $MNW_DEVICE (default 0), or restrict HIP_VISIBLE_DEVICES
inline mnw_ctx* ctx() {
    static mnw_ctx* c = nullptr;
    if (!c) {
        const char* dev = std::getenv("MNW_DEVICE");
        int rc = mnw_ctx_create(dev ? std::atoi(dev) : 0, &c);
        if (rc != MNW_OK) throw Error(std::string("minnow_hip: ") + mnw_strerror(rc), rc);
    }
    return c;
}
inline void check(int rc) {
    if (rc != MNW_OK) throw Error(std::string(mnw_strerror(rc)) + ": " + mnw_last_error(ctx()), rc);
}
// grow-only host buffer whose bytes are NOT zeroed when it grows (a std::vector<uint8_t> writes every new byte once, a second
// pass over hundreds of megabytes that are about to be overwritten by a copy from the device)
struct RawBuf {
    std::unique_ptr<uint8_t[]> p;
    size_t cap = 0;
    uint8_t* get(size_t n) {
        if (n > cap) { const size_t want = n + n / 8 + 64; p.reset(new uint8_t[want]); cap = want; }
        return p.get();
    }
};
inline void fwrite_all(std::FILE* f, const void* p, size_t n) {
    if (n && std::fwrite(p, 1, n, f) != n) throw Error("write failed");
}
inline void fread_all(std::FILE* f, void* p, size_t n) {
    if (n && std::fread(p, 1, n, f) != n) throw Error("read failed");
}
template <typename T> inline void binaryWrite(std::FILE* f, const T& v) { fwrite_all(f, &v, sizeof(T)); }   // little endian host
template <typename T> inline void binaryRead(std::FILE* f, T& v) { fread_all(f, &v, sizeof(T)); }
}  // namespace detail

// ------------------------------------------------------------------------------------------------
namespace bit {

inline int PrecisionNeeded(uint64_t max) { return (int)mnw_precision_needed(max); }       // bit.go:19-21
inline int ArrayBytes(int bits, int length) { return (int)mnw_array_bytes(bits, length); } // bit.go:23-25

// bit.Array, bit.go:13-17.  Data points into a caller-owned buffer (BufferedArray) or into `own`.
struct Array {
    int Length = 0;
    uint8_t Bits = 0;
    uint8_t* Data = nullptr;
    size_t DataLen = 0;
    std::vector<uint8_t> own;

    // (*Array).Slice, bit.go:29-82: len(out) must be at least Length
    void Slice(uint64_t* out, size_t out_len) const {
        if (out_len < (size_t)Length)
            throw Error("Array has length " + std::to_string(Length) + ", but out buffer has length " +
                        std::to_string(out_len) + ".", MNW_E_BUFLEN);
        detail::check(mnw_unpack_u64(detail::ctx(), Data, (int64_t)DataLen, Bits, Length, out));
    }
};

// bit.BufferedArray, bit.go:84-134
inline Array BufferedArray(int bits, const uint64_t* x, size_t n, uint8_t* b, size_t b_len) {
    if (bits > 64) throw Error("Cannot pack more than 64 bits per element into a bit.Array");
    const int nBytes = ArrayBytes(bits, (int)n);
    if ((int)b_len != nBytes)
        throw Error("bit.BufferedArray given buffer of length " + std::to_string(b_len) + ", but length " +
                    std::to_string(nBytes) + " was required.", MNW_E_BUFLEN);
    detail::check(mnw_pack_u64(detail::ctx(), x, (int64_t)n, bits, b, (int64_t)b_len));
    Array a;
    a.Length = (int)n; a.Bits = (uint8_t)bits; a.Data = b; a.DataLen = b_len;
    return a;
}

// bit.NewArray, bit.go:138-142
inline Array NewArray(int bits, const uint64_t* x, size_t n) {
    if (bits > 64) throw Error("Cannot pack more than 64 bits per element into a bit.Array");
    Array a;
    a.own.assign((size_t)ArrayBytes(bits, (int)n), 0);
    Array t = BufferedArray(bits, x, n, a.own.data(), a.own.size());
    a.Length = t.Length; a.Bits = t.Bits; a.Data = a.own.data(); a.DataLen = a.own.size();
    return a;
}

// bit.ArrayBuffer, bit.go:146-206: reusable, grow-only scratch
struct ArrayBuffer {
    std::vector<uint8_t> byteBuf;
    std::vector<uint64_t> uint64Buf;

    int Bits(const uint64_t* x, size_t n) {                                  // bit.go:151-159
        int64_t b = 0;
        detail::check(mnw_bits_u64(detail::ctx(), x, (int64_t)n, &b));
        return (int)b;
    }
    void Write(std::FILE* f, const uint64_t* x, size_t n, int bits) {        // bit.go:161-167
        if (bits == 0) return;
        setByteSize((size_t)ArrayBytes(bits, (int)n));
        Array arr = BufferedArray(bits, x, n, byteBuf.data(), byteBuf.size());
        detail::fwrite_all(f, arr.Data, arr.DataLen);
    }
    uint64_t* Read(std::FILE* f, int bits, size_t n) {                        // bit.go:169-181
        setUint64Size(n);
        if (bits == 0) { std::fill(uint64Buf.begin(), uint64Buf.end(), 0); return uint64Buf.data(); }
        setByteSize((size_t)ArrayBytes(bits, (int)n));
        detail::fread_all(f, byteBuf.data(), byteBuf.size());
        Array arr;
        arr.Length = (int)n; arr.Bits = (uint8_t)bits; arr.Data = byteBuf.data(); arr.DataLen = byteBuf.size();
        arr.Slice(uint64Buf.data(), uint64Buf.size());
        return uint64Buf.data();
    }
    uint64_t* Uint64(size_t n) { setUint64Size(n); return uint64Buf.data(); }  // bit.go:183-186
    uint8_t* Bytes(size_t n) { setByteSize(n); return byteBuf.data(); }
    void setByteSize(size_t n) { byteBuf.resize(n); }                          // bit.go:188-196 (capacity is kept)
    void setUint64Size(size_t n) { uint64Buf.resize(n); }                      // bit.go:198-206
};

}  // namespace bit

// ------------------------------------------------------------------------------------------------
// go/block_index.go:7-39
struct blockIndex {
    int64_t startBlock = 0;
    std::vector<int64_t> offsets;
    explicit blockIndex(int64_t start = 0) : startBlock(start) {}
    void addBlock(int64_t size) { offsets.push_back(offsets.empty() ? size : size + offsets.back()); }
    int64_t blockOffset(int b) const {
        const int64_t b64 = b;
        if (b64 < startBlock || b64 >= startBlock + (int64_t)offsets.size())
            throw Error("Group contains blocks in range [" + std::to_string(startBlock) + ", " +
                        std::to_string(startBlock + (int64_t)offsets.size()) + "), but block " + std::to_string(b64) +
                        " was requested.");
        if (b64 == startBlock) return 0;
        return offsets[(size_t)(b64 - startBlock - 1)];
    }
    int64_t blocks() const { return (int64_t)offsets.size(); }
};

// the unexported `group` interface, go/group.go:77-87.  x/out are typed by the group type.
struct group {
    virtual ~group() {}
    virtual int64_t groupType() const = 0;
    virtual int length(int b) const = 0;
    virtual void writeData(std::FILE* f, const void* x, size_t n) = 0;
    virtual void writeTail(std::FILE* f) = 0;
    virtual int64_t blockOffset(int b) const = 0;
    virtual void readData(std::FILE* f, int b, void* out, size_t n) = 0;
    // a block the caller has already encoded on the device (batched mnw_encode_blocks): the stream and what
    // intGroup.writeData would have appended to g.mins / g.bits (group.go:251-254)
    virtual void writeEncoded(std::FILE*, const uint8_t*, size_t, int64_t, int64_t) { throw Error("this group type is not a codec group"); }
};

// go/group.go:109-163: raw copy, no codec involved
struct fixedSizeGroup : group {
    blockIndex idx;
    int64_t N = 0, typeSize = 0, gt = 0;
    fixedSizeGroup(int startBlock, int n, int64_t gtype) : idx(startBlock), N(n), typeSize(fixedSizeBytes[gtype]), gt(gtype) {}
    static std::unique_ptr<group> fromTail(std::FILE* f, int64_t gtype) {
        int64_t N, startBlock, blocks;
        detail::binaryRead(f, N); detail::binaryRead(f, startBlock); detail::binaryRead(f, blocks);
        auto g = std::make_unique<fixedSizeGroup>((int)startBlock, (int)N, gtype);
        for (int64_t i = 0; i < blocks; ++i) g->idx.addBlock(g->typeSize * g->N);
        return g;
    }
    int64_t groupType() const override { return gt; }
    int length(int) const override { return (int)N; }
    void writeData(std::FILE* f, const void* x, size_t n) override {
        detail::fwrite_all(f, x, n * (size_t)typeSize);
        idx.addBlock(typeSize * N);
    }
    void readData(std::FILE* f, int, void* out, size_t n) override { detail::fread_all(f, out, n * (size_t)typeSize); }
    void writeTail(std::FILE* f) override {
        detail::binaryWrite(f, N); detail::binaryWrite(f, idx.startBlock); detail::binaryWrite(f, idx.blocks());
    }
    int64_t blockOffset(int b) const override { return idx.blockOffset(b); }
};

// go/group.go:172-263
struct intGroup : group {
    blockIndex idx;
    int64_t N = 0;
    bit::ArrayBuffer ab;
    std::vector<int64_t> mins, bits;

    intGroup(int startBlock, int n) : idx(startBlock), N(n) {}

    // write(x) / read() of writeTail / newIntGroupFromTail (group.go:190-198, 216-225): a small int64
    // array stored as (min, bits, packed x-min) -- the same codec as a data block
    static void writeArray(std::FILE* f, const std::vector<int64_t>& x) {
        if (x.size() == 1) {                 // min = x[0], x - min = {0}, PrecisionNeeded(0) = 0, no bytes (group.go:190-198, bit.go:162):
            detail::binaryWrite(f, x[0]);    // minh writes one group per (block, column), so every tail array has one element
            detail::binaryWrite(f, (int64_t)0);
            return;
        }
        std::vector<uint8_t> out(8 * x.size() + 16);
        int64_t mn = 0, b = 0, n = 0;
        detail::check(mnw_int_encode(detail::ctx(), x.data(), (int64_t)x.size(), &mn, &b, out.data(), (int64_t)out.size(), &n));
        detail::binaryWrite(f, mn);
        detail::binaryWrite(f, b);
        detail::fwrite_all(f, out.data(), (size_t)n);
    }
    static std::vector<int64_t> readArray(std::FILE* f, int64_t blocks) {
        int64_t mn, b;
        detail::binaryRead(f, mn); detail::binaryRead(f, b);
        std::vector<int64_t> out((size_t)blocks);
        if (b == 0) { std::fill(out.begin(), out.end(), mn); return out; }      // no bytes: zeros (bit.go:171-174) + min (group.go:262)
        const size_t nb = (size_t)bit::ArrayBytes((int)b, (int)blocks);
        std::vector<uint8_t> buf(nb);
        detail::fread_all(f, buf.data(), nb);
        detail::check(mnw_int_decode(detail::ctx(), buf.data(), (int64_t)nb, mn, b, blocks, out.data()));
        return out;
    }
    static std::unique_ptr<intGroup> fromTail(std::FILE* f) {
        int64_t N, startBlock, blocks;
        detail::binaryRead(f, N); detail::binaryRead(f, startBlock); detail::binaryRead(f, blocks);
        auto g = std::make_unique<intGroup>((int)startBlock, (int)N);
        g->mins = readArray(f, blocks);
        g->bits = readArray(f, blocks);
        for (size_t i = 0; i < g->bits.size(); ++i) g->idx.addBlock(bit::ArrayBytes((int)g->bits[i], (int)g->N));
        return g;
    }
    void writeTail(std::FILE* f) override {
        detail::binaryWrite(f, N); detail::binaryWrite(f, idx.startBlock); detail::binaryWrite(f, idx.blocks());
        writeArray(f, mins);
        writeArray(f, bits);
    }
    int64_t groupType() const override { return IntGroup; }
    int length(int) const override { return (int)N; }
    int64_t blockOffset(int b) const override { return idx.blockOffset(b); }

    // intGroup.writeData, group.go:242-255
    void writeData(std::FILE* f, const void* x, size_t n) override {
        const int64_t* data = static_cast<const int64_t*>(x);
        uint8_t* out = ab.Bytes(8 * n + 16);
        int64_t mn = 0, b = 0, nb = 0;
        detail::check(mnw_int_encode(detail::ctx(), data, (int64_t)n, &mn, &b, out, (int64_t)(8 * n + 16), &nb));
        detail::fwrite_all(f, out, (size_t)nb);            // nothing when bits == 0 (bit.go:162)
        mins.push_back(mn);
        bits.push_back(b);
        idx.addBlock(bit::ArrayBytes((int)b, (int)N));
    }
    void writeEncoded(std::FILE* f, const uint8_t* stream, size_t nbytes, int64_t mn, int64_t b) override {
        if ((int64_t)nbytes != (b == 0 ? 0 : bit::ArrayBytes((int)b, (int)N))) throw Error("encoded block has the wrong length", MNW_E_BUFLEN);
        detail::fwrite_all(f, stream, nbytes);
        mins.push_back(mn);
        bits.push_back(b);
        idx.addBlock(bit::ArrayBytes((int)b, (int)N));
    }
    // intGroup.readData, group.go:257-263
    void readData(std::FILE* f, int b, void* out, size_t n) override {
        const size_t bIdx = (size_t)(b - (int)idx.startBlock);
        const int64_t bt = bits[bIdx], mn = mins[bIdx];
        const size_t nb = bt == 0 ? 0 : (size_t)bit::ArrayBytes((int)bt, (int)N);
        uint8_t* buf = ab.Bytes(nb);
        detail::fread_all(f, buf, nb);
        detail::check(mnw_int_decode(detail::ctx(), buf, (int64_t)nb, mn, bt, (int64_t)n, static_cast<int64_t*>(out)));
    }
};

// go/group.go:269-344
struct floatGroup : group {
    std::unique_ptr<intGroup> ig;
    float low = 0, high = 0;
    int64_t pixels = 0;
    uint8_t periodic = 0;
    uint64_t ditherSeed = 0;        // replaces the unseeded global math/rand of group.go:308 (DESIGN.md "Dither")

    floatGroup(int startBlock, int n, float lo, float hi, int64_t px, bool per)
        : ig(new intGroup(startBlock, n)), low(lo), high(hi), pixels(px), periodic(per ? 1 : 0) {}
    floatGroup() {}
    static std::unique_ptr<group> fromTail(std::FILE* f) {
        auto g = std::make_unique<floatGroup>();
        g->ig = intGroup::fromTail(f);
        detail::binaryRead(f, g->low); detail::binaryRead(f, g->high);
        detail::binaryRead(f, g->pixels); detail::binaryRead(f, g->periodic);
        return g;
    }
    int64_t groupType() const override { return FloatGroup; }
    int length(int b) const override { return ig->length(b); }
    int64_t blockOffset(int b) const override { return ig->blockOffset(b); }
    void writeTail(std::FILE* f) override {
        ig->writeTail(f);
        detail::binaryWrite(f, low); detail::binaryWrite(f, high);
        detail::binaryWrite(f, pixels); detail::binaryWrite(f, periodic);
    }
    // floatGroup.writeData, group.go:312-327
    void writeData(std::FILE* f, const void* x, size_t n) override {
        uint8_t* out = ig->ab.Bytes(8 * n + 16);
        int64_t mn = 0, b = 0, nb = 0;
        detail::check(mnw_float_encode(detail::ctx(), static_cast<const float*>(x), (int64_t)n, low, high, pixels, periodic,
                                       &mn, &b, out, (int64_t)(8 * n + 16), &nb));
        detail::fwrite_all(f, out, (size_t)nb);
        ig->mins.push_back(mn);
        ig->bits.push_back(b);
        ig->idx.addBlock(bit::ArrayBytes((int)b, (int)ig->N));
    }
    void writeEncoded(std::FILE* f, const uint8_t* stream, size_t nbytes, int64_t mn, int64_t b) override { ig->writeEncoded(f, stream, nbytes, mn, b); }
    // floatGroup.readData, group.go:299-310
    void readData(std::FILE* f, int b, void* out, size_t n) override {
        const size_t bIdx = (size_t)(b - (int)ig->idx.startBlock);
        const int64_t bt = ig->bits[bIdx], mn = ig->mins[bIdx];
        const size_t nb = bt == 0 ? 0 : (size_t)bit::ArrayBytes((int)bt, (int)ig->N);
        uint8_t* buf = ig->ab.Bytes(nb);
        detail::fread_all(f, buf, nb);
        detail::check(mnw_float_decode(detail::ctx(), buf, (int64_t)nb, mn, bt, low, high, pixels, periodic, ditherSeed,
                                       (uint64_t)b, (int64_t)n, static_cast<float*>(out)));
    }
};

// go/group.go:93-103
inline std::unique_ptr<group> groupFromTail(std::FILE* f, int64_t gt) {
    if (gt >= Int64Group && gt <= Float32Group) return fixedSizeGroup::fromTail(f, gt);
    if (gt == IntGroup) return intGroup::fromTail(f);
    if (gt == FloatGroup) return floatGroup::fromTail(f);
    throw Error("Unrecognized group type, " + std::to_string(gt) + ".");
}

// element type -> the group types it may be written to / read from (TypeMatch, go/group.go:43-71)
template <typename T> struct TypeOf;
template <> struct TypeOf<int64_t> { static bool ok(int64_t gt) { return gt == Int64Group || gt == IntGroup; } static const char* name() { return "[]int64"; } };
template <> struct TypeOf<int32_t> { static bool ok(int64_t gt) { return gt == Int32Group; } static const char* name() { return "[]int32"; } };
template <> struct TypeOf<int16_t> { static bool ok(int64_t gt) { return gt == Int16Group; } static const char* name() { return "[]int16"; } };
template <> struct TypeOf<int8_t> { static bool ok(int64_t gt) { return gt == Int8Group; } static const char* name() { return "[]int8"; } };
template <> struct TypeOf<uint64_t> { static bool ok(int64_t gt) { return gt == Uint64Group; } static const char* name() { return "[]uint64"; } };
template <> struct TypeOf<uint32_t> { static bool ok(int64_t gt) { return gt == Uint32Group; } static const char* name() { return "[]uint32"; } };
template <> struct TypeOf<uint16_t> { static bool ok(int64_t gt) { return gt == Uint16Group; } static const char* name() { return "[]uint16"; } };
template <> struct TypeOf<uint8_t> { static bool ok(int64_t gt) { return gt == Uint8Group; } static const char* name() { return "[]uint8"; } };
template <> struct TypeOf<double> { static bool ok(int64_t gt) { return gt == Float64Group; } static const char* name() { return "[]float64"; } };
template <> struct TypeOf<float> { static bool ok(int64_t gt) { return gt == Float32Group || gt == FloatGroup; } static const char* name() { return "[]float32"; } };

template <typename T> inline void TypeMatch(int64_t gt) {
    if (!TypeOf<T>::ok(gt)) throw Error(std::string("Got type ") + TypeOf<T>::name() + " for group " + GroupName(gt) + ".");
}

// minnowHeader, go/writer.go:24-29
struct minnowHeader {
    uint64_t Magic, Version, Groups, Headers, Blocks;
    int64_t TailStart;
};
static_assert(sizeof(minnowHeader) == 48, "minnowHeader is 48 bytes");

// ------------------------------------------------------------------------------------------------
// go/writer.go
class Writer {
   public:
    // Create, writer.go:32-41
    explicit Writer(const std::string& fname) {
        f_ = std::fopen(fname.c_str(), "wb+");
        if (!f_) throw Error("cannot create " + fname);
        minnowHeader hd{};
        detail::binaryWrite(f_, hd);
    }
    ~Writer() { if (f_) std::fclose(f_); }
    Writer(const Writer&) = delete;
    Writer& operator=(const Writer&) = delete;

    // Header, writer.go:44-57: returns the header index
    int Header(const void* x, size_t nbytes) {
        headerOffsets_.push_back(std::ftell(f_));
        headerSizes_.push_back((int64_t)nbytes);
        detail::fwrite_all(f_, x, nbytes);
        ++headers_;
        currGroup_ = -1;
        return headers_ - 1;
    }
    template <typename T> int Header(const T& v) { return Header(&v, sizeof(T)); }

    void FixedSizeGroup(int64_t groupType, int N) { newGroup(std::make_unique<fixedSizeGroup>(blocks_, N, groupType)); }  // writer.go:61-63
    void IntGroup(int N) { newGroup(std::make_unique<intGroup>(blocks_, N)); }                                            // writer.go:67-69
    // FloatGroup, writer.go:72-75: periodic is always true here
    void FloatGroup(int N, float lim0, float lim1, float dx) {
        const int64_t pixels = mnw_float_group_pixels(lim0, lim1, dx);
        newGroup(std::make_unique<floatGroup>(blocks_, N, lim0, lim1, pixels, true));
    }

    // Data, writer.go:90-103: returns the block index
    template <typename T> int Data(const T* x, size_t n) {
        if (currGroup_ == -1) throw Error("Data written to minnow.Writer without assigning Group first.");
        TypeMatch<T>(currGroup_);
        writers_.back()->writeData(f_, x, n);
        ++groupBlocks_.back();
        ++blocks_;
        return blocks_ - 1;
    }
    template <typename T> int Data(const std::vector<T>& x) { return Data(x.data(), x.size()); }
    // Data for a block of the current IntGroup / FloatGroup that was encoded on the device by the caller (one batched
    // mnw_encode_blocks call for many blocks): same bookkeeping as Data, the stream is written as it is
    int DataEncoded(const uint8_t* stream, size_t nbytes, int64_t min, int64_t bits) {
        if (currGroup_ == -1) throw Error("Data written to minnow.Writer without assigning Group first.");
        writers_.back()->writeEncoded(f_, stream, nbytes, min, bits);
        ++groupBlocks_.back();
        ++blocks_;
        return blocks_ - 1;
    }
    // Data for a block of the current fixed-size group given as raw bytes (n elements of the group's type)
    int DataRaw(const void* x, size_t n) {
        if (currGroup_ == -1) throw Error("Data written to minnow.Writer without assigning Group first.");
        if (!(currGroup_ >= Int64Group && currGroup_ <= Float32Group)) throw Error("DataRaw needs a fixed-size group");
        writers_.back()->writeData(f_, x, n);
        ++groupBlocks_.back();
        ++blocks_;
        return blocks_ - 1;
    }

    // Close, writer.go:107-141
    void Close() {
        const int64_t tailStart = std::ftell(f_);
        std::vector<int64_t> groupTypes;
        for (auto& g : writers_) groupTypes.push_back(g->groupType());
        for (const std::vector<int64_t>* a : {&headerOffsets_, &headerSizes_, &groupOffsets_, &groupTypes, &groupBlocks_})
            detail::fwrite_all(f_, a->data(), a->size() * 8);
        for (auto& g : writers_) g->writeTail(f_);
        std::fseek(f_, 0, SEEK_SET);
        minnowHeader hd{Magic, Version, (uint64_t)writers_.size(), (uint64_t)headers_, (uint64_t)blocks_, tailStart};
        detail::binaryWrite(f_, hd);
        std::fclose(f_);
        f_ = nullptr;
    }

   private:
    void newGroup(std::unique_ptr<group> g) {   // writer.go:78-87
        currGroup_ = g->groupType();
        writers_.push_back(std::move(g));
        groupBlocks_.push_back(0);
        groupOffsets_.push_back(std::ftell(f_));
    }
    std::FILE* f_ = nullptr;
    int headers_ = 0, blocks_ = 0;
    int64_t currGroup_ = -1;
    std::vector<std::unique_ptr<group>> writers_;
    std::vector<int64_t> headerOffsets_, headerSizes_, groupBlocks_, groupOffsets_;
};
inline std::unique_ptr<Writer> Create(const std::string& fname) { return std::make_unique<Writer>(fname); }

// ------------------------------------------------------------------------------------------------
// go/reader.go
class Reader {
   public:
    // Open, reader.go:28-88
    explicit Reader(const std::string& fname) {
        f_ = std::fopen(fname.c_str(), "rb");
        if (!f_) throw Error("cannot open " + fname);
        minnowHeader hd;
        detail::binaryRead(f_, hd);
        if (hd.Magic != Magic) throw Error(fname + " is not a minnow file.");
        if (hd.Version != Version) throw Error(fname + " was written with a different minnow version.");
        groups_ = (int)hd.Groups; headers_ = (int)hd.Headers; blocks_ = (int)hd.Blocks;
        std::fseek(f_, (long)hd.TailStart, SEEK_SET);
        headerOffsets_.resize(headers_); headerSizes_.resize(headers_);
        groupOffsets_.resize(groups_); groupTypes_.resize(groups_);
        std::vector<int64_t> groupBlocks(groups_);
        for (std::vector<int64_t>* a : {&headerOffsets_, &headerSizes_, &groupOffsets_, &groupTypes_, &groupBlocks})
            detail::fread_all(f_, a->data(), a->size() * 8);
        for (int i = 0; i < groups_; ++i) readers_.push_back(groupFromTail(f_, groupTypes_[i]));
        blockIndex_.resize(blocks_);
        int i = 0;
        for (int j = 0; j < groups_; ++j)
            for (int64_t k = 0; k < groupBlocks[j]; ++k) blockIndex_[i++] = j;
    }
    ~Reader() { if (f_) std::fclose(f_); }
    Reader(const Reader&) = delete;
    Reader& operator=(const Reader&) = delete;

    // Header, reader.go:92-101
    void Header(int i, void* out, size_t nbytes) {
        if ((int64_t)nbytes != headerSizes_[i])
            throw Error("Header buffer has size " + std::to_string(nbytes) + ", but written header has size " +
                        std::to_string(headerSizes_[i]));
        std::fseek(f_, (long)headerOffsets_[i], SEEK_SET);
        detail::fread_all(f_, out, nbytes);
    }
    template <typename T> void Header(int i, T& out) { Header(i, &out, sizeof(T)); }
    int HeaderSize(int i) const { return (int)headerSizes_[i]; }     // reader.go:104-106
    int Blocks() const { return blocks_; }                           // reader.go:109-111

    // Data, reader.go:114-127
    template <typename T> void Data(int b, T* out, size_t n) {
        const int i = blockIndex_.at((size_t)b);
        TypeMatch<T>(DataType(b));
        std::fseek(f_, (long)groupOffsets_[i], SEEK_SET);
        std::fseek(f_, (long)readers_[i]->blockOffset(b), SEEK_CUR);
        readers_[i]->readData(f_, b, out, n);
    }
    template <typename T> void Data(int b, std::vector<T>& out) { Data(b, out.data(), out.size()); }
    int64_t DataType(int b) const { return groupTypes_[blockIndex_.at((size_t)b)]; }         // reader.go:130-132
    int DataLen(int b) const { return readers_[blockIndex_.at((size_t)b)]->length(b); }      // reader.go:135-137
    void Close() { if (f_) { std::fclose(f_); f_ = nullptr; } }                              // reader.go:140-142
    group* Group(int i) { return readers_.at((size_t)i).get(); }
    // for callers that decode blocks on the device themselves: the group of block b, where its bytes start in the file
    int GroupOf(int b) const { return blockIndex_.at((size_t)b); }
    int64_t BlockFileOffset(int b) const { const int i = GroupOf(b); return groupOffsets_[i] + readers_[i]->blockOffset(b); }
    std::FILE* File() { return f_; }

   private:
    std::FILE* f_ = nullptr;
    int groups_ = 0, headers_ = 0, blocks_ = 0;
    std::vector<std::unique_ptr<group>> readers_;
    std::vector<int> blockIndex_;
    std::vector<int64_t> headerOffsets_, headerSizes_, groupOffsets_, groupTypes_;
};
inline std::unique_ptr<Reader> Open(const std::string& fname) { return std::make_unique<Reader>(fname); }

}  // namespace minnow
