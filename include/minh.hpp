// minh halo catalogues (go/minh/minh.go) over include/minnow.hpp: the reference's Writer / Reader with its names,
// argument meaning and panics (as exceptions).  What differs is where the work happens: minh.Writer.Block loops over
// the columns, copies every FloatGroup column into minh.buf, runs processFloatGroup on it and encodes it, one column
// at a time on one core (minh.go:99-149).  Here the columns of a block go to the GPU as ONE batch of codec jobs --
// processFloatGroup rides along as a job flag (mnw_enc_job.xform) -- and the reader decodes the requested columns of a
// block as one batch with the 10^x of logarithmic columns (minh.go:315-319) applied on the device.  Block / IntBlock /
// FloatBlock take host arrays like the Go methods; BlockDevice / IntBlockDevice / FloatBlockDevice take device arrays
// (e.g. the outputs of mnw_text_parse + mnw_generic_cut), which is the form a GPU-resident text_to_minh uses.
//
// Header-only, C++17, needs only the C ABI (include/minnow_hip.h).
#pragma once

#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <unistd.h>

#include "minnow.hpp"

namespace minh {

using minnow::Error;

constexpr int64_t Magic = 0xbaff1ed;       // minh.go:15-18
constexpr int64_t Version = 0;
constexpr int64_t basicFileType = 0, boundaryFileType = 1;   // minh.go:20-23

// column types = minnow's group types (minh.go:25-38)
enum : int64_t { Int64 = 0, Int32, Int16, Int8, Uint64, Uint32, Uint16, Uint8, Float64, Float32, Int, Float };

// minh.go:50-55: 256 bytes in the file
struct Column {
    int64_t Type = 0;
    int32_t Log = 0;
    float Low = 0, High = 0, Dx = 0;
    uint8_t Buffer[232] = {};
    bool operator==(const Column& o) const { return std::memcmp(this, &o, sizeof(Column)) == 0; }
};
static_assert(sizeof(Column) == 256, "Sizeof(Column{}) must be 256 (minh.go:79-82)");

struct idHeader { int64_t Magic, Version, FileType; };                 // minh.go:62-64
struct geometry { float L, Boundary; int64_t Cells; };                 // minh.go:66-69
static_assert(sizeof(idHeader) == 24 && sizeof(geometry) == 16, "header layouts");

namespace detail {
// grow-only device buffer
struct DevBuf {
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
// One thread that performs the file writes of the blocks in order, so that the writes of block b overlap the upload,
// parse and encode of block b+1 (minnow::Writer itself is single threaded: only this thread touches it between
// Header/Geometry and Close).  `done` callbacks run on the posting thread, in drain() / post().
struct FileWriter {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv, cv_idle;
    std::deque<std::pair<std::function<void()>, std::function<void()>>> q;
    std::vector<std::function<void()>> finished;
    std::exception_ptr err;
    bool quit = false, busy = false;
    void run() {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv.wait(lk, [&] { return quit || !q.empty(); });
            if (quit || q.empty()) return;                          // (on quit the queue has been dropped)
            auto t = std::move(q.front());
            q.pop_front();
            busy = true;
            lk.unlock();
            try { t.first(); } catch (...) { std::lock_guard<std::mutex> g(mu); if (!err) err = std::current_exception(); }
            lk.lock();
            busy = false;
            finished.push_back(std::move(t.second));
            cv_idle.notify_all();
        }
    }
    void reap(std::unique_lock<std::mutex>& lk) {
        std::vector<std::function<void()>> f;
        f.swap(finished);
        const std::exception_ptr e = err;
        err = nullptr;
        lk.unlock();
        for (auto& fn : f) fn();
        if (e) std::rethrow_exception(e);
        lk.lock();
    }
    void post(std::function<void()> task, std::function<void()> done) {
        std::unique_lock<std::mutex> lk(mu);
        if (!th.joinable()) th = std::thread([this] { run(); });
        cv_idle.wait(lk, [&] { return q.size() < 2; });          // at most two blocks of output in flight
        reap(lk);
        q.emplace_back(std::move(task), std::move(done));
        cv.notify_all();
    }
    void drain() {
        std::unique_lock<std::mutex> lk(mu);
        cv_idle.wait(lk, [&] { return q.empty() && !busy; });
        reap(lk);
    }
    // drops what is queued, waits for the task that is running, ends the thread (idempotent; never throws)
    void stop() noexcept {
        { std::lock_guard<std::mutex> g(mu); quit = true; q.clear(); }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
    ~FileWriter() { stop(); }
};
inline size_t align16(size_t n) { return (n + 15) & ~(size_t)15; }
inline bool isFixed(int64_t t) { return t >= Int64 && t <= Float32; }
}  // namespace detail

// ------------------------------------------------------------------------------------------------
// minh.Writer, minh.go:40-158
class Writer {
   public:
    explicit Writer(const std::string& fname, int64_t fileType = basicFileType) : f_(minnow::Create(fname)) {     // create, minh.go:78-86
        idHeader hd{Magic, Version, fileType};
        f_->Header(hd);
    }
    // A Writer that goes away without Close() -- a Block() that threw while the block before it was still being written, a
    // caller that drops the object -- must not leave the file-writer thread running tasks that use f_ and cols_: queued tasks
    // are dropped, the one that is running is waited for, and only then are the members destroyed.
    ~Writer() { writer_.stop(); }
    Writer(const Writer&) = delete;
    Writer& operator=(const Writer&) = delete;

    // Header, minh.go:88-93
    void Header(const std::vector<std::string>& names, const std::string& text, const std::vector<Column>& cols) {
        f_->Header(text.data(), text.size());
        std::string joined;
        for (size_t i = 0; i < names.size(); ++i) { if (i) joined += "$"; joined += names[i]; }
        f_->Header(joined.data(), joined.size());
        f_->Header(cols.data(), cols.size() * sizeof(Column));
        cols_ = cols;
    }

    // Geometry, minh.go:95-97
    void Geometry(float L, float boundary, int cells) { l_ = L; boundary_ = boundary; cells_ = cells; }

    // Block, minh.go:99-139: cols[i] = host array of N elements of column i's type
    void Block(const std::vector<const void*>& cols, int N) { blockImpl(cols, N, false); }
    // the same with DEVICE arrays (16-byte aligned)
    void BlockDevice(const std::vector<const void*>& dcols, int N) { blockImpl(dcols, N, true); }

    // Close, minh.go:151-156
    void Close() {
        writer_.drain();                                           // the blocks' file writes run on a thread of their own
        geometry g{l_, boundary_, (int64_t)cells_};
        f_->Header(g);
        const int64_t blocks = blocks_;
        f_->Header(blocks);
        f_->Header(blockSizes_.data(), blockSizes_.size() * 8);
        f_->Close();
    }

    int Blocks() const { return blocks_; }

   private:
    void blockImpl(const std::vector<const void*>& cols, int N, bool onDevice) {
        if (cols.size() != cols_.size())
            throw Error("Expected " + std::to_string(cols_.size()) + " columns, got " + std::to_string(cols.size()) + ".");
        for (size_t i = 0; i < cols.size(); ++i) {
            if (cols_[i].Type < Int64 || cols_[i].Type > Float) throw Error("Column " + std::to_string(i) + ": unknown type");
            if (!cols[i] && N > 0) throw Error("Column " + std::to_string(i) + ": null array");
        }
        blockSizes_.push_back(N);
        ++blocks_;
        mnw_ctx* ctx = minnow::detail::ctx();

        // ---- every codec column of the block is one job of one batch --------------------------------------
        std::vector<mnw_enc_job> jobs;
        std::vector<mnw_block_meta> meta;
        std::vector<size_t> slotAt, slotCap;
        uint8_t* dstreams = nullptr;
        for (bool forceWide = false;; forceWide = true) {
        jobs.clear(); slotAt.clear(); slotCap.clear();
        // a column's stream slot holds the widest stream it can produce: 64 bits per element for an int column; a float
        // column's pixel indices stay below `pixels`, so 32 bits when that is below 2^32 (it is for every real column)
        size_t ncodec = 0, stage = 0, streamBytes = 0;
        for (const Column& c : cols_) {
            if (detail::isFixed(c.Type)) continue;
            const bool wide = forceWide || c.Type == Int || (uint64_t)mnw_float_group_pixels(c.Low, c.High, c.Dx) >= (1ull << 32);
            slotAt.push_back(streamBytes);
            slotCap.push_back(detail::align16((size_t)(wide ? 8 : 4) * (size_t)N + 16));
            streamBytes += slotCap.back();
            ++ncodec;
            stage += detail::align16((size_t)N * (c.Type == Int ? 8 : 4));
        }
        dstreams = ncodec ? streams_.get(streamBytes) : nullptr;
        uint8_t* dstage = (ncodec && !onDevice) ? stage_.get(stage + 16) : nullptr;
        meta.assign(ncodec, mnw_block_meta());
        size_t soff = 0;
        for (size_t i = 0; i < cols_.size(); ++i) {
            const Column& c = cols_[i];
            if (detail::isFixed(c.Type)) continue;
            const size_t esz = c.Type == Int ? 8 : 4;
            const void* src = cols[i];
            if (!onDevice) {
                minnow::detail::check(mnw_copy_to_device(ctx, dstage + soff, cols[i], (int64_t)((size_t)N * esz)));
                src = dstage + soff;
                soff += detail::align16((size_t)N * esz);
            }
            mnw_enc_job j;
            std::memset(&j, 0, sizeof j);
            j.src = src;
            j.dst = dstreams + slotAt[jobs.size()];
            j.n = N;
            j.cap = (int64_t)slotCap[jobs.size()];
            if (c.Type == Int) {
                j.kind = MNW_INT_GROUP;
            } else {
                j.kind = MNW_FLOAT_GROUP;
                j.low = c.Low; j.high = c.High;
                j.pixels = mnw_float_group_pixels(c.Low, c.High, c.Dx);             // Writer.FloatGroup, writer.go:72-75
                j.periodic = 1;
                j.xform = MNW_XF_CLAMP | (c.Log != 0 ? MNW_XF_LOG10 : 0);             // processFloatGroup, minh.go:141-149
            }
            jobs.push_back(j);
        }
        if (jobs.empty()) break;
        const int rc = mnw_encode_blocks(ctx, nullptr, jobs.data(), (int)jobs.size(), meta.data());
        // a NaN in a float column quantizes to MinInt64 (amd64) and the block then needs up to 64 bits per element, as it
        // does in the reference: encode once more with slots of that size
        if (rc == MNW_E_CAPACITY && !forceWide) continue;
        minnow::detail::check(rc);
        break;
        }

        // ---- everything the file needs of this block comes back in ONE pipelined transfer (the streams of all codec columns,
        // and the fixed-size columns of a device block) into a buffer that the file-writing thread then owns
        auto hb = std::make_shared<minnow::detail::RawBuf>();
        std::vector<size_t> hostAt(cols_.size(), 0), hostLen(cols_.size(), 0);
        {
            size_t tot = 0, k = 0;
            for (size_t i = 0; i < cols_.size(); ++i) {
                const Column& c = cols_[i];
                if (detail::isFixed(c.Type)) hostLen[i] = onDevice ? (size_t)N * (size_t)minnow::fixedSizeBytes[c.Type] : 0;
                else hostLen[i] = (size_t)meta[k++].nbytes;
                hostAt[i] = tot;
                tot += (hostLen[i] + 63) & ~(size_t)63;
            }
            if (!spare_.empty()) { hb = spare_.back(); spare_.pop_back(); }       // a buffer the writer thread has finished with
            uint8_t* const hbase = hb->get(tot);
            std::vector<void*> d;
            std::vector<const void*> sptr;
            std::vector<int64_t> nb;
            k = 0;
            for (size_t i = 0; i < cols_.size(); ++i) {
                const bool fixed = detail::isFixed(cols_[i].Type);
                const void* from = fixed ? cols[i] : (const void*)(dstreams + slotAt[k]);
                if (!fixed) ++k;
                if (hostLen[i] == 0) continue;
                d.push_back(hbase + hostAt[i]); sptr.push_back(from); nb.push_back((int64_t)hostLen[i]);
            }
            if (!d.empty()) minnow::detail::check(mnw_copy_to_host_batch(ctx, (int)d.size(), d.data(), sptr.data(), nb.data()));
        }
        // ---- the file, column after column as the reference writes it: one group per (block, column).  The writes run on
        // the writer thread, in block order, while the caller prepares the next block (host arrays of a host block are
        // written before Block returns: the caller may reuse them).
        std::vector<const void*> hostCols = cols;
        std::vector<mnw_block_meta> metaCopy = meta;
        auto task = [this, hb, hostAt, hostLen, hostCols, metaCopy, N, onDevice]() {
            size_t k = 0;
            for (size_t i = 0; i < cols_.size(); ++i) {
                const Column& c = cols_[i];
                if (detail::isFixed(c.Type)) {
                    f_->FixedSizeGroup(c.Type, N);
                    f_->DataRaw(onDevice ? (const void*)(hb->p.get() + hostAt[i]) : hostCols[i], (size_t)N);
                    continue;
                }
                if (c.Type == Int) f_->IntGroup(N);
                else f_->FloatGroup(N, c.Low, c.High, c.Dx);
                const mnw_block_meta& m = metaCopy[k++];
                f_->DataEncoded(hb->p.get() + hostAt[i], (size_t)m.nbytes, m.min, m.bits);
            }
        };
        bool hasHostFixed = false;
        for (const Column& c : cols_) hasHostFixed = hasHostFixed || (detail::isFixed(c.Type) && !onDevice);
        writer_.post(task, [this, hb]() { spare_.push_back(hb); });
        if (hasHostFixed) writer_.drain();              // the task reads the caller's arrays
    }

    detail::FileWriter writer_;
    std::vector<std::shared_ptr<minnow::detail::RawBuf>> spare_;

    std::unique_ptr<minnow::Writer> f_;
    int blocks_ = 0;
    std::vector<Column> cols_;
    std::vector<int64_t> blockSizes_;
    float l_ = 0, boundary_ = 0;
    int cells_ = 0;
    detail::DevBuf streams_, stage_;
    std::vector<uint8_t> host_;
};
inline std::unique_ptr<Writer> Create(const std::string& fname) { return std::make_unique<Writer>(fname); }     // minh.go:72-76

// ------------------------------------------------------------------------------------------------
// minh.Reader, minh.go:172-327
class Reader {
   public:
    std::vector<std::string> Names;
    std::string Text;
    std::vector<Column> Columns;
    int Blocks = 0, Length = 0;
    std::vector<int> BlockLengths;
    float L = 0, Boundary = 0;
    int Cells = 0;
    uint64_t DitherSeed = 0;       // replaces the unseeded global math/rand of group.go:308 (DESIGN.md "Dither")
    bool ReadAhead = true;         // prefetch the next block's streams of the columns just asked for (see blockImpl)

    ~Reader() { stopAhead(); }
    // Open, minh.go:184-236
    explicit Reader(const std::string& fname) : f_(minnow::Open(fname)) {
        idHeader hd;
        f_->Header(0, hd);
        if (hd.Magic != Magic)
            throw Error(fname + " is not a minh file. Expected magic number " + std::to_string(Magic) + ", but got " + std::to_string(hd.Magic) + ".");
        if (hd.Version < Version) throw Error(fname + " written with an older minh version.");
        fileType_ = hd.FileType;
        Text.resize((size_t)f_->HeaderSize(1));
        f_->Header(1, &Text[0], Text.size());
        std::string names((size_t)f_->HeaderSize(2), '\0');
        f_->Header(2, &names[0], names.size());
        Columns.resize((size_t)f_->HeaderSize(3) / sizeof(Column));
        f_->Header(3, Columns.data(), Columns.size() * sizeof(Column));
        geometry g;
        f_->Header(4, g);
        int64_t blocks = 0;
        f_->Header(5, blocks);
        std::vector<int64_t> lens((size_t)f_->HeaderSize(6) / 8);
        f_->Header(6, lens.data(), lens.size() * 8);
        size_t a = 0;                                            // strings.Split(names, "$")
        for (;;) {
            const size_t b = names.find('$', a);
            Names.push_back(names.substr(a, b == std::string::npos ? std::string::npos : b - a));
            if (b == std::string::npos) break;
            a = b + 1;
        }
        Blocks = (int)blocks;
        L = g.L; Boundary = g.Boundary; Cells = (int)g.Cells;
        for (int64_t n : lens) { BlockLengths.push_back((int)n); Length += (int)n; }
    }

    // IntBlock / FloatBlock, minh.go:272-323: out[name] = host array of BlockLengths[b] elements
    void IntBlock(int b, const std::map<std::string, int64_t*>& out) {
        std::vector<std::string> names; std::vector<void*> ptr;
        for (auto& kv : out) { names.push_back(kv.first); ptr.push_back(kv.second); }
        blockImpl(b, names, ptr, true, false);
    }
    void FloatBlock(int b, const std::map<std::string, float*>& out) {
        std::vector<std::string> names; std::vector<void*> ptr;
        for (auto& kv : out) { names.push_back(kv.first); ptr.push_back(kv.second); }
        blockImpl(b, names, ptr, false, false);
    }
    // the same into DEVICE arrays (16-byte aligned)
    void IntBlockDevice(int b, const std::vector<std::string>& names, const std::vector<void*>& dptr) { blockImpl(b, names, dptr, true, true); }
    void FloatBlockDevice(int b, const std::vector<std::string>& names, const std::vector<void*>& dptr) { blockImpl(b, names, dptr, false, true); }

    // Ints / Floats, minh.go:238-270: whole columns
    std::map<std::string, std::vector<int64_t>> Ints(const std::vector<std::string>& names) {
        std::map<std::string, std::vector<int64_t>> out;
        for (auto& n : names) out[n].assign((size_t)Length, 0);
        int end = 0;
        for (int b = 0; b < Blocks; ++b) {
            const int start = end;
            end = start + BlockLengths[(size_t)b];
            std::map<std::string, int64_t*> bOut;
            for (auto& n : names) bOut[n] = out[n].data() + start;
            IntBlock(b, bOut);
        }
        return out;
    }
    std::map<std::string, std::vector<float>> Floats(const std::vector<std::string>& names) {
        std::map<std::string, std::vector<float>> out;
        for (auto& n : names) out[n].assign((size_t)Length, 0);
        int end = 0;
        for (int b = 0; b < Blocks; ++b) {
            const int start = end;
            end = start + BlockLengths[(size_t)b];
            std::map<std::string, float*> bOut;
            for (auto& n : names) bOut[n] = out[n].data() + start;
            FloatBlock(b, bOut);
        }
        return out;
    }

    void Close() { stopAhead(); f_->Close(); }                     // minh.go:325-327

   private:
    int findName(const std::string& name) const {                  // minh.go:329-334
        for (size_t i = 0; i < Names.size(); ++i) if (Names[i] == name) return (int)i;
        throw Error("Name " + name + " not in Reader.Names.");
    }

    void blockImpl(int b, const std::vector<std::string>& names, const std::vector<void*>& out, bool ints, bool onDevice) {
        if (b < 0 || b >= Blocks) throw Error("block " + std::to_string(b) + " out of range");
        if (names.size() != out.size()) throw Error("one output array per name");
        const size_t N = (size_t)BlockLengths[(size_t)b];
        const size_t esz = ints ? 8 : 4;
        mnw_ctx* ctx = minnow::detail::ctx();
        struct Item { int c, idx; size_t off, nbytes; void* out; };
        std::vector<Item> codec;
        size_t total = 0;
        for (size_t k = 0; k < names.size(); ++k) {
            const int c = findName(names[k]);
            const int idx = fileType_ == basicFileType ? c + b * (int)Columns.size() : c * Blocks + b;      // minh.go:283-288
            const int64_t t = Columns[(size_t)c].Type;
            if (ints ? !(t == Int64 || t == Int) : !(t == Float32 || t == Float))                           // minnow.TypeMatch
                throw Error("Column '" + names[k] + "': Got type " + (ints ? "[]int64" : "[]float32") + " for group " + minnow::GroupName(t) + ".");
            if (N == 0) continue;
            if (!out[k]) throw Error("Column '" + names[k] + "': null array");
            if (detail::isFixed(t)) {
                if (onDevice) {
                    host_.resize(N * esz);
                    if (ints) f_->Data(idx, reinterpret_cast<int64_t*>(host_.data()), N); else f_->Data(idx, reinterpret_cast<float*>(host_.data()), N);
                    minnow::detail::check(mnw_copy_to_device(ctx, out[k], host_.data(), (int64_t)(N * esz)));
                } else {
                    if (ints) f_->Data(idx, static_cast<int64_t*>(out[k]), N); else f_->Data(idx, static_cast<float*>(out[k]), N);
                }
                continue;
            }
            minnow::intGroup* ig = intGroupOf(idx);
            const size_t bIdx = (size_t)(idx - (int)ig->idx.startBlock);
            const int64_t bits = ig->bits.at(bIdx);
            const size_t nb = bits == 0 ? 0 : (size_t)minnow::bit::ArrayBytes((int)bits, (int)N);
            codec.push_back({c, idx, total, nb, out[k]});
            total += nb;
        }
        if (codec.empty()) return;
        // The streams of the requested columns: file -> one staging buffer -> device, then one batch of decode jobs.  A file is
        // normally read front to back with the same columns for every block (minh.go's callers loop over rd.Blocks): while this
        // block is decoded, a thread already reads the same columns of block b + 1 and uploads them through a context of its
        // own; the call for b + 1 finds its streams on the device.  Anything else simply misses the prefetch.
        std::vector<int> key;
        key.reserve(codec.size());
        for (const Item& it : codec) key.push_back(it.c);
        Ahead& ah = ahead_[ints ? 1 : 0];
        uint8_t* dstage = nullptr;
        if (ah.join() && ah.ready && ah.b == b && ah.cols == key && ah.total == total) {
            dstage = static_cast<uint8_t*>(ah.dev[ah.cur]);
        } else {
            std::vector<int64_t> foff, flen;
            for (const Item& it : codec) { foff.push_back(f_->BlockFileOffset(it.idx)); flen.push_back((int64_t)it.nbytes); }
            dstage = stage_.get(total + 64);
            minnow::detail::check(mnw_copy_file_to_device(ctx, dstage, fileno(f_->File()), (int)foff.size(), foff.data(), flen.data(), nullptr));
        }
        ah.ready = false;
        if (ReadAhead && b + 1 < Blocks) startAhead(ah, b + 1, key, ints);
        uint8_t* dout = onDevice ? nullptr : outs_.get(codec.size() * detail::align16(N * esz));
        std::vector<mnw_dec_job> jobs;
        for (size_t k = 0; k < codec.size(); ++k) {
            const Item& it = codec[k];
            const Column& col = Columns[(size_t)it.c];
            minnow::intGroup* ig = intGroupOf(it.idx);
            const size_t bIdx = (size_t)(it.idx - (int)ig->idx.startBlock);
            mnw_dec_job j;
            std::memset(&j, 0, sizeof j);
            j.src = dstage + it.off;
            j.dst = onDevice ? it.out : (void*)(dout + k * detail::align16(N * esz));
            j.n = (int64_t)N;
            j.src_len = (int64_t)it.nbytes;
            j.min = ig->mins.at(bIdx);
            j.bits = ig->bits.at(bIdx);
            if (col.Type == Int) {
                j.kind = MNW_INT_GROUP;
            } else {
                auto* fg = dynamic_cast<minnow::floatGroup*>(f_->Group(f_->GroupOf(it.idx)));
                j.kind = MNW_FLOAT_GROUP;
                j.low = fg->low; j.high = fg->high; j.pixels = fg->pixels; j.periodic = fg->periodic;
                j.seed = DitherSeed; j.block = (uint64_t)it.idx;
                j.xform = col.Log != 0 ? MNW_XF_LOG10 : 0;                                    // minh.go:315-319
            }
            jobs.push_back(j);
        }
        minnow::detail::check(mnw_decode_blocks(ctx, nullptr, jobs.data(), (int)jobs.size()));
        if (onDevice) {
            uint8_t one;                                                                      // a copy on the context's stream = wait for the decode
            minnow::detail::check(mnw_copy_to_host(ctx, &one, dstage, 1));
        } else {
            for (size_t k = 0; k < codec.size(); ++k)
                minnow::detail::check(mnw_copy_to_host(ctx, codec[k].out, dout + k * detail::align16(N * esz), (int64_t)(N * esz)));
        }
    }

    // ---- read-ahead of the next block's streams (one slot for IntBlock requests, one for FloatBlock requests) ----
    struct Ahead {
        std::thread th;
        bool ready = false;
        int b = -1;
        std::vector<int> cols;
        size_t total = 0;
        int cur = 0;                                   // which of the two device buffers the finished prefetch is in
        void* dev[2] = {nullptr, nullptr};
        size_t cap[2] = {0, 0};
        bool join() { if (th.joinable()) th.join(); return true; }
    };
    void startAhead(Ahead& ah, int b, const std::vector<int>& cols, bool ints) {
        // the request for block b, worked out here (the group tables are not touched by the thread)
        struct Piece { int64_t fileOff; size_t off, nbytes; };
        std::vector<Piece> pieces;
        size_t total = 0;
        const size_t N = (size_t)BlockLengths[(size_t)b];
        if (N == 0) return;
        for (int c : cols) {
            const int idx = fileType_ == basicFileType ? c + b * (int)Columns.size() : c * Blocks + b;
            minnow::intGroup* ig = intGroupOf(idx);
            const int64_t bits = ig->bits.at((size_t)(idx - (int)ig->idx.startBlock));
            const size_t nb = bits == 0 ? 0 : (size_t)minnow::bit::ArrayBytes((int)bits, (int)N);
            pieces.push_back({f_->BlockFileOffset(idx), total, nb});
            total += nb;
        }
        if (total == 0) return;
        ah.b = b; ah.cols = cols; ah.total = total; ah.ready = false;
        const int slot = ah.cur ^ 1;                   // the other buffer: the caller may still be decoding from dev[cur]
        const int fd = fileno(f_->File());
        (void)ints;
        ah.th = std::thread([this, &ah, pieces, total, slot, fd] {
            try {
                std::lock_guard<std::mutex> lk(upMu_);              // one upload at a time through the second context
                if (!upCtx_) {
                    const char* dev = std::getenv("MNW_DEVICE");
                    if (mnw_ctx_create(dev ? std::atoi(dev) : 0, &upCtx_) != MNW_OK) { upCtx_ = nullptr; return; }
                }
                if (ah.cap[slot] < total + 64) {
                    if (ah.dev[slot]) mnw_dev_free(upCtx_, ah.dev[slot]);
                    ah.dev[slot] = nullptr; ah.cap[slot] = 0;
                    const size_t want = total + total / 4 + 256;
                    if (mnw_dev_alloc(upCtx_, (int64_t)want, &ah.dev[slot]) != MNW_OK) return;
                    ah.cap[slot] = want;
                }
                std::vector<int64_t> foff, flen;
                for (const Piece& p : pieces) { foff.push_back(p.fileOff); flen.push_back((int64_t)p.nbytes); }
                if (mnw_copy_file_to_device(upCtx_, ah.dev[slot], fd, (int)foff.size(), foff.data(), flen.data(), nullptr) != MNW_OK) return;
                ah.cur = slot;
                ah.ready = true;                                     // (read by the owner only after join())
            } catch (...) {}
        });
    }
    void stopAhead() {
        for (Ahead& ah : ahead_) {
            ah.join();
            for (int k = 0; k < 2; ++k) if (ah.dev[k]) { mnw_dev_free(upCtx_ ? upCtx_ : minnow::detail::ctx(), ah.dev[k]); ah.dev[k] = nullptr; }
        }
        if (upCtx_) { mnw_ctx_destroy(upCtx_); upCtx_ = nullptr; }
    }
    Ahead ahead_[2];
    std::mutex upMu_;
    mnw_ctx* upCtx_ = nullptr;

    minnow::intGroup* intGroupOf(int idx) {
        minnow::group* g = f_->Group(f_->GroupOf(idx));
        if (auto* ig = dynamic_cast<minnow::intGroup*>(g)) return ig;
        if (auto* fg = dynamic_cast<minnow::floatGroup*>(g)) return fg->ig.get();
        throw Error("block " + std::to_string(idx) + " is not in a codec group");
    }

    std::unique_ptr<minnow::Reader> f_;
    int64_t fileType_ = basicFileType;
    detail::DevBuf stage_, outs_;
    std::vector<uint8_t> host_;
    minnow::detail::RawBuf rawHost_;
};
inline std::unique_ptr<Reader> Open(const std::string& fname) { return std::make_unique<Reader>(fname); }       // minh.go:184

}  // namespace minh
