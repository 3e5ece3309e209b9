// Text catalogue reader (go/text/text.go, go/text/rockstar.go) over the C ABI: the reference's Reader / Rockstar with their
// names and panics (as exceptions).  File handling -- blocks of at most MaxBlockSize bytes cut at a newline found in the
// last MaxItemSize bytes (findBlocks / nextBlock, text.go:82-131), the header lines (133-163) -- stays on the host as in
// the reference; Reader.Block hands the block's bytes to the GPU (mnw_text_rows + mnw_text_parse) instead of splitting
// and converting them on SetThreads(n) goroutines, and returns DEVICE columns, ready for mnw_generic_cut and
// minh::Writer::BlockDevice.
//
// Header-only, C++17, needs only the C ABI (include/minnow_hip.h).
#pragma once

#include <algorithm>
#include <cctype>

#include <sys/mman.h>
#include <unistd.h>

#include <memory>
#include <chrono>
#include <thread>
#include "minnow.hpp"

namespace text {

using minnow::Error;

// text.go:27-39
struct ReaderConfig {
    uint8_t Separator = ' ';
    uint8_t Comment = '#';
    int64_t MaxBlockSize = 5ll * (1ll << 30);
    int64_t MaxItemSize = 100ll * (1ll << 10);
};

// the columns of one block on the device: what Block leaves in its out []interface{} of []int64 / []float32.  Owned by the
// Reader and reused by the next Block call (the reference reuses its buffers the same way, expandGeneric, text.go:192)
struct BlockColumns {
    int64_t rows = 0;
    std::vector<void*> ptr;          // ptr[i]: device int64[rows] or float32[rows] for names[i]
};

class Reader {
   public:
    double UploadWaitSeconds = 0, UploadSeconds = 0;   // time Block spent waiting for the read-ahead thread / that thread's own time (diagnostics)
    // Open / openFromReader, text.go:42-67
    explicit Reader(const std::string& fname, ReaderConfig config = ReaderConfig()) : config_(config) {
        if (config_.MaxBlockSize / 2 < config_.MaxItemSize)
            throw Error("config.MaxBlockSize = " + std::to_string(config_.MaxBlockSize) + ", but config.MaxLineSize = " + std::to_string(config_.MaxItemSize));
        f_ = std::fopen(fname.c_str(), "rb");
        if (!f_) throw Error("open " + fname + ": no such file or directory");
        std::fseek(f_, 0, SEEK_END);
        size_ = (int64_t)std::ftell(f_);
        findBlocks();
    }
    ~Reader() {
        joinUpload();
        pending_.reset();
        for (void* p : upBuf_) if (p && upCtx_) mnw_dev_free(upCtx_, p);
        if (upCtx_) mnw_ctx_destroy(upCtx_);
        Close();
        if (dev_) mnw_dev_free(minnow::detail::ctx(), dev_);
    }
    Reader(const Reader&) = delete;
    Reader& operator=(const Reader&) = delete;

    // LineHeader, text.go:133-139: the first `lines` lines of the file
    std::string LineHeader(int lines) {
        std::vector<std::string> hd;
        headerLines(hd);
        if ((int)hd.size() <= lines) throw Error("config.MaxItemSize too small.");
        return join(hd, (size_t)lines);
    }
    // CommentHeader, text.go:141-147.  As in the reference the number of header lines is the number of comment
    // CHARACTERS in the first MaxItemSize bytes (split() counts bytes, parse.go:16-21): one per line in a Rockstar header.
    std::string CommentHeader() {
        std::vector<std::string> hd;
        const int nComm = headerLines(hd);
        if ((int)hd.size() == nComm) throw Error("config.MaxItemSize too small.");
        if (nComm > (int)hd.size()) throw Error("slice bounds out of range");      // the reference's runtime panic
        return join(hd, (size_t)nComm);
    }

    void SetNames(const std::vector<std::string>& names) { names_ = names; haveNames_ = true; }     // text.go:165-166
    int Blocks() const { return (int)blockStarts_.size(); }                                         // text.go:168-169
    int64_t BlockSize(int b) const { return blockSizes_.at((size_t)b); }
    void SetThreads(int) {}                                                                          // text.go:174: the device parses
    void Close() { if (f_) { std::fclose(f_); f_ = nullptr; } }                                     // text.go:171-172

    // Block, text.go:176-200: the columns `names` of block b; isInt[i] says whether names[i] is read as []int64
    // (strconv.Atoi) or as []float32 (float32(strconv.ParseFloat(s, 64)))
    BlockColumns Block(int b, const std::vector<std::string>& names, const std::vector<bool>& isInt) {
        if (!haveNames_) throw Error("Must call text.Reader.SetNames() before text.Reader.Blocks()");
        if (names.size() != isInt.size()) throw Error("one type per name");
        if (b < 0 || b >= Blocks()) throw Error("block index out of range");
        mnw_ctx* ctx = minnow::detail::ctx();
        // rd.buf: the block's bytes.  The reference reads them into a buffer (text.go:183-187); here the file's pages are
        // mapped and go to the device straight from the page cache.  While block b is parsed (and the caller cuts, encodes and
        // writes it), block b + 1 is already on its way: a second thread with a library context of its own maps it and copies
        // it into the other of two device buffers (the reference's callers walk the blocks in order, text_to_minh.go:205).
        const bool trace = std::getenv("MNW_TEXT_TRACE") != nullptr;
        const auto tr0 = std::chrono::steady_clock::now();
        auto lap = [&](const char* what) {
            if (trace) std::fprintf(stderr, "Block(%d) %s at %.3f ms\n", b, what, 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count());
        };
        std::unique_ptr<Upload> up = takeUpload(b);
        lap("upload taken");
        if (b + 1 < Blocks() && blockSizes_[(size_t)b + 1] > 0) startUpload(b + 1);
        lap("next upload started");
        const uint8_t* const text = up->block->data;
        const int64_t nbytes = blockSizes_[(size_t)b];
        int64_t lines = 0, rows = 0;
        if (up->dev) minnow::detail::check(mnw_text_rows_resident(ctx, text, up->dev, nbytes, config_.Separator, config_.Comment, &lines, &rows));
        else minnow::detail::check(mnw_text_rows(ctx, text, nbytes, config_.Separator, config_.Comment, &lines, &rows));
        lap("rows counted");
        // splitByType, text.go:202-221
        std::vector<int32_t> fIdx, iIdx;
        std::vector<size_t> fAt, iAt;
        for (size_t i = 0; i < names.size(); ++i) {
            const int idx = nameIndex(names[i]);
            if (isInt[i]) { iIdx.push_back(idx); iAt.push_back(i); } else { fIdx.push_back(idx); fAt.push_back(i); }
        }
        // expandGeneric, text.go:192: every output gets len(lines) elements -- one grow-only slab on the device, cut into columns
        BlockColumns out;
        out.rows = rows;
        out.ptr.resize(names.size());
        std::vector<size_t> at(names.size());
        size_t need = 0;
        for (size_t i = 0; i < names.size(); ++i) { at[i] = need; need += ((size_t)rows * (isInt[i] ? 8 : 4) + 255) & ~(size_t)255; }
        if (cap_ < need + 256) {
            if (dev_) minnow::detail::check(mnw_dev_free(ctx, dev_));
            dev_ = nullptr; cap_ = 0;
            const size_t want = need + need / 4 + 256;
            minnow::detail::check(mnw_dev_alloc(ctx, (int64_t)want, &dev_));
            cap_ = want;
        }
        for (size_t i = 0; i < names.size(); ++i) out.ptr[i] = static_cast<uint8_t*>(dev_) + at[i];
        // at most 128 columns per call of the C ABI
        size_t f0 = 0, i0 = 0;
        do {
            const size_t nf = std::min<size_t>(128, fIdx.size() - f0), ni = std::min<size_t>(128 - nf, iIdx.size() - i0);
            std::vector<float*> fo(nf);
            std::vector<int64_t*> io(ni);
            for (size_t k = 0; k < nf; ++k) fo[k] = static_cast<float*>(out.ptr[fAt[f0 + k]]);
            for (size_t k = 0; k < ni; ++k) io[k] = static_cast<int64_t*>(out.ptr[iAt[i0 + k]]);
            int64_t n = 0;
            minnow::detail::check(mnw_text_parse(ctx, text, nbytes, config_.Separator, config_.Comment,
                                                 fIdx.data() + f0, (int)nf, fo.data(), iIdx.data() + i0, (int)ni, io.data(), rows, &n));
            f0 += nf; i0 += ni;
        } while (f0 < fIdx.size() || i0 < iIdx.size());
        lap("parsed");
        return out;
    }

   private:
    struct Mapped;
    // a block on its way to the device (or there): the mapping stays alive as long as the parse may look at the host bytes
    struct Upload {
        int b = -1;
        std::unique_ptr<Mapped> block;
        const uint8_t* dev = nullptr;                   // null: not uploaded (Block uploads it itself)
        std::string err;
    };
    std::unique_ptr<Upload> takeUpload(int b) {
        joinUpload();
        std::unique_ptr<Upload> up;
        if (pending_ && pending_->b == b) up = std::move(pending_);
        pending_.reset();
        if (up && !up->err.empty()) throw Error(up->err);
        if (!up) {                                       // first block, or blocks asked for out of order: fetch it now, the same way
            startUpload(b);
            joinUpload();
            up = std::move(pending_);
            pending_.reset();
            if (up && !up->err.empty()) throw Error(up->err);
            if (!up || !up->block) {
                up.reset(new Upload());
                up->b = b;
                up->block.reset(new Mapped(fileno(f_), blockStarts_[(size_t)b], blockSizes_[(size_t)b]));
            }
        }
        return up;
    }
    void joinUpload() {
        if (!upThread_.joinable()) return;
        const auto t0 = std::chrono::steady_clock::now();
        upThread_.join();
        UploadWaitSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    void startUpload(int b) {
        joinUpload();
        pending_.reset(new Upload());
        Upload* up = pending_.get();
        up->b = b;
        const int fd = fileno(f_);
        const int64_t off = blockStarts_[(size_t)b], len = blockSizes_[(size_t)b];
        const int slot = upNext_;
        upNext_ ^= 1;
        upThread_ = std::thread([this, up, fd, off, len, slot] {
            const auto t0 = std::chrono::steady_clock::now();
            struct Fin { Reader* r; std::chrono::steady_clock::time_point t; ~Fin() { r->UploadSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); } } fin{this, t0};
            try {
                up->block.reset(new Mapped(fd, off, len));
                if (!upCtx_) {
                    const char* dev = std::getenv("MNW_DEVICE");
                    const int rc = mnw_ctx_create(dev ? std::atoi(dev) : 0, &upCtx_);
                    if (rc != MNW_OK) { upCtx_ = nullptr; return; }          // no second context: Block uploads the block itself
                }
                if (upCap_[slot] < (size_t)len + 64) {
                    if (upBuf_[slot]) mnw_dev_free(upCtx_, upBuf_[slot]);
                    upBuf_[slot] = nullptr; upCap_[slot] = 0;
                    const size_t want = (size_t)len + (size_t)len / 8 + 256;
                    if (mnw_dev_alloc(upCtx_, (int64_t)want, &upBuf_[slot]) != MNW_OK) return;
                    upCap_[slot] = want;
                }
                // (straight from the file into the library's pinned buffers: reading the mapping instead costs a page-cache -> pinned
                // memcpy of the whole block, which, not the link, then sets the converter's pace)
                if (mnw_copy_file_to_device(upCtx_, upBuf_[slot], fd, 1, &off, &len, nullptr) != MNW_OK) return;
                up->dev = static_cast<const uint8_t*>(upBuf_[slot]);
            } catch (const std::exception& e) { up->err = e.what(); }
        });
    }
    std::unique_ptr<Upload> pending_;
    std::thread upThread_;
    mnw_ctx* upCtx_ = nullptr;
    void* upBuf_[2] = {nullptr, nullptr};
    size_t upCap_[2] = {0, 0};
    int upNext_ = 0;

    struct Mapped {                                     // [offset, offset + len) of a file, read-only
        void* base = nullptr;
        size_t maplen = 0;
        const uint8_t* data = nullptr;
        Mapped(int fd, int64_t offset, int64_t len) {
            if (len <= 0) { static const uint8_t none = 0; data = &none; return; }
            const int64_t page = (int64_t)sysconf(_SC_PAGESIZE), a = offset / page * page;
            maplen = (size_t)(offset - a + len);
            // (not populated: the block's bytes reach the device by pread into pinned memory; the mapping is only looked at for the
            // few literals the host has to convert itself.  Populating it cost 32k page-table entries per 128 MiB block on the way
            // in and again on the way out: 2 ms of every Block call)
            base = mmap(nullptr, maplen, PROT_READ, MAP_PRIVATE, fd, (off_t)a);
            if (base == MAP_FAILED) { base = nullptr; throw Error("mmap of the text block failed"); }
            data = static_cast<const uint8_t*>(base) + (offset - a);
        }
        ~Mapped() { if (base) munmap(base, maplen); }
        Mapped(const Mapped&) = delete;
        Mapped& operator=(const Mapped&) = delete;
    };

    // findBlocks / nextBlock, text.go:82-131
    void findBlocks() {
        int64_t curr = 0;
        std::vector<uint8_t> buf;
        for (;;) {
            blockStarts_.push_back(curr);
            if (curr + config_.MaxBlockSize >= size_) break;                       // no more blocks in the file
            const int64_t searchStart = config_.MaxBlockSize - config_.MaxItemSize + curr;
            buf.resize((size_t)config_.MaxItemSize);
            std::fseek(f_, (long)searchStart, SEEK_SET);
            minnow::detail::fread_all(f_, buf.data(), buf.size());
            const auto it = std::find(buf.begin(), buf.end(), (uint8_t)'\n');
            if (it == buf.end()) throw Error("config.MaxItemSize too small.");
            curr = searchStart + (it - buf.begin()) + 1;
        }
        blockSizes_.resize(blockStarts_.size());
        for (size_t i = 0; i + 1 < blockStarts_.size(); ++i) blockSizes_[i] = blockStarts_[i + 1] - blockStarts_[i];
        blockSizes_.back() = size_ - blockStarts_.back();
    }

    // headerLines, text.go:149-163: the lines of the first MaxItemSize bytes and the number of comment characters in them
    int headerLines(std::vector<std::string>& lines) {
        const size_t n = (size_t)std::min(size_, config_.MaxItemSize);
        std::string buf(n, '\0');
        std::fseek(f_, 0, SEEK_SET);
        minnow::detail::fread_all(f_, &buf[0], n);
        int nComm = 0;
        size_t a = 0;
        for (size_t i = 0; i < n; ++i) {
            if ((uint8_t)buf[i] == config_.Comment) ++nComm;
            if (buf[i] == '\n') { lines.push_back(buf.substr(a, i - a)); a = i + 1; }
        }
        lines.push_back(buf.substr(a));
        return nComm;
    }
    static std::string join(const std::vector<std::string>& l, size_t n) {
        std::string s;
        for (size_t i = 0; i < n; ++i) { if (i) s += "\n"; s += l[i]; }
        return s;
    }
    int nameIndex(const std::string& name) const {                                  // text.go:223-231
        auto lower = [](std::string s) { for (char& c : s) c = (char)std::tolower((unsigned char)c); return s; };
        const std::string want = lower(name);
        for (size_t i = 0; i < names_.size(); ++i) if (lower(names_[i]) == want) return (int)i;
        throw Error("Name '" + name + "' doesn't match to any columns.");
    }

    ReaderConfig config_;
    std::FILE* f_ = nullptr;
    int64_t size_ = 0;
    std::vector<int64_t> blockStarts_, blockSizes_;
    std::vector<std::string> names_;
    bool haveNames_ = false;
    void* dev_ = nullptr;            // the columns of the last Block
    size_t cap_ = 0;
};
inline std::unique_ptr<Reader> Open(const std::string& fname, ReaderConfig config = ReaderConfig()) { return std::make_unique<Reader>(fname, config); }

// go/text/rockstar.go
class Rockstar {
   public:
    explicit Rockstar(const std::string& fname, ReaderConfig config = ReaderConfig()) : rd_(Open(fname, config)) {}
    double UploadWaitSeconds() const { return rd_->UploadWaitSeconds; }             // (diagnostics: see Reader)
    double UploadSeconds() const { return rd_->UploadSeconds; }
    // Names, rockstar.go:16-23: the column names in the first line of the file
    std::vector<std::string> Names() {
        const std::string firstLine = rd_->LineHeader(1);
        return firstLine.find("(0)") != std::string::npos ? consistentTreesNames(firstLine) : rockstarNames(firstLine);
    }
    static std::vector<std::string> rockstarNames(const std::string& line) {        // rockstar.go:25-36
        std::vector<std::string> out;
        size_t a = line.empty() ? 0 : 1;                                            // line[1:]: the comment character
        while (a <= line.size()) {
            size_t b = line.find(' ', a);
            if (b == std::string::npos) b = line.size();
            std::string tok = trim(line.substr(a, b - a));
            if (!tok.empty()) out.push_back(tok);
            a = b + 1;
        }
        return out;
    }
    static std::vector<std::string> consistentTreesNames(const std::string& line) { // rockstar.go:38-52: "name(12)" -> "name"
        std::vector<std::string> names = rockstarNames(line);
        for (std::string& n : names) {
            const size_t p = n.rfind('(');
            n = trim(p == std::string::npos ? std::string() : n.substr(0, p));
        }
        return names;                                                               // (the reference returns names, empty ones included)
    }
    void SetThreads(int n) { rd_->SetThreads(n); }
    std::string Header() { return rd_->CommentHeader(); }
    void SetNames(const std::vector<std::string>& names) { rd_->SetNames(names); }
    int Blocks() const { return rd_->Blocks(); }
    int64_t BlockSize(int b) const { return rd_->BlockSize(b); }
    void Close() { rd_->Close(); }
    BlockColumns Block(int b, const std::vector<std::string>& names, const std::vector<bool>& isInt) { return rd_->Block(b, names, isInt); }

   private:
    static std::string trim(const std::string& s) {                                 // strings.Trim(s, " \n\t")
        size_t a = 0, b = s.size();
        while (a < b && (s[a] == ' ' || s[a] == '\n' || s[a] == '\t')) ++a;
        while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\n' || s[b - 1] == '\t')) --b;
        return s.substr(a, b - a);
    }
    std::unique_ptr<Reader> rd_;
};
inline std::unique_ptr<Rockstar> OpenRockstar(const std::string& fname, ReaderConfig config = ReaderConfig()) { return std::make_unique<Rockstar>(fname, config); }

}  // namespace text
