// text_to_minh on the GPU: the conversion loop of the reference's scripts/text_to_minh.go (ConvertFile, 166-214) --
//     for every block of the text catalogue: Block -> GenericCut on the mass column -> minh.Writer.Block
// -- with every stage on the device: text::Rockstar::Block (mnw_text_rows + mnw_text_parse), mnw_generic_cut,
// minh::Writer::BlockDevice.  The reference's config-file machinery (go/config, name index) is not rebuilt: the columns
// are given on the command line.
//
//   text_to_minh <in.txt> <out.minh> <L> <mass column> <cutoff> <max block bytes> COLUMN...
//   COLUMN = name:int64 | name:float32 | name:int | name:lin:LOW:HIGH:DX | name:log:MIN:MAX:DX | name:pos:EPS
//       int64 / float32  fixed-size columns          (ParseTypeString "int64" / "float32", text_to_minh.go:225-230)
//       int              minh.Int                    ("int")
//       lin              minh.Float, Low / High / Dx ("q_float linear")
//       log              minh.Float, Log = 1, Low = log10(MIN), High = log10(MAX)   ("q_float log"; the reference tests
//                        t[0] == "log" where it means t[1], text_to_minh.go:250, so its own files never set Log)
//       pos              minh.Float, 0 / L / EPS     ("q_float position")
//   text_to_minh --dump <file.minh>      reads a file back through minh::Reader and prints its columns as JSON
//   text_to_minh --read <file.minh>      reads every column of every block back into device arrays (the reader side of the
//                                        round trip: minh.Reader.IntBlock / FloatBlock per block) and prints the time
//
// Prints one JSON object with the time spent per stage.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "minh.hpp"
#include "minnow_text.hpp"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static std::vector<std::string> splitc(const std::string& s, char c) {
    std::vector<std::string> out;
    std::stringstream ss(s);
    std::string tok;
    while (std::getline(ss, tok, c)) out.push_back(tok);
    return out;
}

static int dump(const std::string& fname) {
    auto rd = minh::Open(fname);
    std::vector<std::string> inames, fnames;
    for (size_t i = 0; i < rd->Names.size(); ++i) {
        const int64_t t = rd->Columns[i].Type;
        if (t == minh::Int64 || t == minh::Int) inames.push_back(rd->Names[i]);
        else if (t == minh::Float32 || t == minh::Float) fnames.push_back(rd->Names[i]);
    }
    auto io = rd->Ints(inames);
    auto fo = rd->Floats(fnames);
    std::printf("{\"blocks\": %d, \"length\": %d, \"L\": %.9g, \"text_bytes\": %zu, \"block_lengths\": [", rd->Blocks, rd->Length, rd->L, rd->Text.size());
    for (size_t i = 0; i < rd->BlockLengths.size(); ++i) std::printf("%s%d", i ? ", " : "", rd->BlockLengths[i]);
    std::printf("], \"columns\": {");
    bool first = true;
    for (auto& kv : io) {
        std::printf("%s\"%s\": [", first ? "" : ", ", kv.first.c_str());
        for (size_t i = 0; i < kv.second.size(); ++i) std::printf("%s%lld", i ? ", " : "", (long long)kv.second[i]);
        std::printf("]");
        first = false;
    }
    for (auto& kv : fo) {
        std::printf("%s\"%s\": [", first ? "" : ", ", kv.first.c_str());
        for (size_t i = 0; i < kv.second.size(); ++i) std::printf("%s%.9g", i ? ", " : "", (double)kv.second[i]);
        std::printf("]");
        first = false;
    }
    std::printf("}}\n");
    rd->Close();
    return 0;
}

static int readBack(const std::string& fname) {
    mnw_ctx* ctx = minnow::detail::ctx();
    mnw_timing_enable(ctx, 1);
    const double t0 = now();
    auto rd = minh::Open(fname);
    std::vector<std::string> inames, fnames;
    for (size_t i = 0; i < rd->Names.size(); ++i) {
        const int64_t t = rd->Columns[i].Type;
        if (t == minh::Int64 || t == minh::Int) inames.push_back(rd->Names[i]);
        else if (t == minh::Float32 || t == minh::Float) fnames.push_back(rd->Names[i]);
    }
    int maxLen = 0;
    for (int n : rd->BlockLengths) maxLen = std::max(maxLen, n);
    std::vector<void*> di(inames.size()), df(fnames.size());
    for (void*& p : di) minnow::detail::check(mnw_dev_alloc(ctx, 8ll * maxLen + 16, &p));
    for (void*& p : df) minnow::detail::check(mnw_dev_alloc(ctx, 4ll * maxLen + 16, &p));
    const double t1 = now();
    int64_t outBytes = 0;
    for (int b = 0; b < rd->Blocks; ++b) {
        if (!inames.empty()) rd->IntBlockDevice(b, inames, di);
        if (!fnames.empty()) rd->FloatBlockDevice(b, fnames, df);
        outBytes += (int64_t)rd->BlockLengths[(size_t)b] * (int64_t)(8 * inames.size() + 4 * fnames.size());
    }
    const double t2 = now();
    for (void* p : di) mnw_dev_free(ctx, p);
    for (void* p : df) mnw_dev_free(ctx, p);
    mnw_kernel_time kt[MNW_K_COUNT];
    mnw_timing_get(ctx, kt);
    double kms = 0;
    for (int k = 0; k < MNW_K_COUNT; ++k) kms += kt[k].total_ms;
    std::printf("{\"blocks\": %d, \"length\": %d, \"columns\": %zu, \"decoded_bytes\": %lld, \"open_s\": %.6f, \"blocks_s\": %.6f, "
                "\"decoded_GBps\": %.4f, \"kernels_total_ms\": %.4f}\n", rd->Blocks, rd->Length, inames.size() + fnames.size(), (long long)outBytes,
                t1 - t0, t2 - t1, (double)outBytes / (t2 - t1) / 1e9, kms);
    rd->Close();
    return 0;
}

int main(int argc, char** argv) {
    try {
        if (argc == 3 && std::string(argv[1]) == "--dump") return dump(argv[2]);
        if (argc == 3 && std::string(argv[1]) == "--read") return readBack(argv[2]);
        if (argc < 8) { std::cerr << "usage: text_to_minh <in.txt> <out.minh> <L> <mass column> <cutoff> <max block bytes> COLUMN...\n"; return 2; }
        const std::string in = argv[1], out = argv[2], massName = argv[4];
        const float L = (float)std::atof(argv[3]);
        const float cutoff = (float)std::atof(argv[5]);           // cutoff := float32(Mp * MinParticles), text_to_minh.go:193
        text::ReaderConfig cfg;
        cfg.MaxBlockSize = std::atoll(argv[6]);
        if (cfg.MaxBlockSize / 2 < cfg.MaxItemSize) cfg.MaxItemSize = cfg.MaxBlockSize / 2;

        std::vector<std::string> names;
        std::vector<bool> isInt;
        std::vector<minh::Column> cols;
        for (int a = 7; a < argc; ++a) {                          // ParseTypeString, text_to_minh.go:219-268
            const std::vector<std::string> t = splitc(argv[a], ':');
            if (t.size() < 2) throw minnow::Error(std::string("bad column ") + argv[a]);
            minh::Column c;
            bool asInt = false;
            if (t[1] == "int64") { c.Type = minh::Int64; asInt = true; }
            else if (t[1] == "float32") c.Type = minh::Float32;
            else if (t[1] == "int") { c.Type = minh::Int; asInt = true; }
            else if (t[1] == "pos" && t.size() == 3) { c.Type = minh::Float; c.Low = 0; c.High = L; c.Dx = (float)std::atof(t[2].c_str()); }
            else if ((t[1] == "lin" || t[1] == "log") && t.size() == 5) {
                const double mn = std::atof(t[2].c_str()), mx = std::atof(t[3].c_str());
                c.Type = minh::Float;
                if (t[1] == "log") { c.Log = 1; c.Low = (float)std::log10(mn); c.High = (float)std::log10(mx); }
                else { c.Low = (float)mn; c.High = (float)mx; }
                c.Dx = (float)std::atof(t[4].c_str());
            } else throw minnow::Error("Type " + t[1] + " not recognized.");
            names.push_back(t[0]); isInt.push_back(asInt); cols.push_back(c);
        }
        int iMass = -1;
        for (size_t i = 0; i < names.size(); ++i) if (names[i] == massName) iMass = (int)i;
        if (iMass == -1) throw minnow::Error("MassName '" + massName + "' not in name index.");
        if (isInt[(size_t)iMass]) throw minnow::Error("the mass column must be a float column");

        mnw_ctx* ctx = minnow::detail::ctx();
        mnw_timing_enable(ctx, 1);
        const double t0 = now();
        auto fR = text::OpenRockstar(in, cfg);
        const std::vector<std::string> allNames = fR->Names();
        const std::string header = fR->Header();
        fR->SetNames(allNames);
        auto fM = minh::Create(out);
        fM->Header(names, header, cols);
        fM->Geometry(L, 0, 0);
        std::vector<void*> cut(names.size(), nullptr);
        void* cutSlab = nullptr;
        size_t cutCap = 0;
        double tParse = 0, tCut = 0, tWrite = 0;
        int64_t rowsIn = 0, rowsKept = 0, bytesIn = 0;
        for (int b = 0; b < fR->Blocks(); ++b) {
            double t = now();
            const text::BlockColumns bc = fR->Block(b, names, isInt);
            tParse += now() - t; t = now();
            rowsIn += bc.rows; bytesIn += fR->BlockSize(b);
            std::vector<mnw_cut_col> cc(names.size());
            size_t need = 0;
            std::vector<size_t> at(names.size());
            for (size_t i = 0; i < names.size(); ++i) { at[i] = need; need += ((size_t)bc.rows * (isInt[i] ? 8 : 4) + 255) & ~(size_t)255; }
            if (cutCap < need + 256) {                            // one grow-only slab for the filtered columns
                if (cutSlab) mnw_dev_free(ctx, cutSlab);
                cutSlab = nullptr; cutCap = 0;
                minnow::detail::check(mnw_dev_alloc(ctx, (int64_t)(need + need / 4 + 256), &cutSlab));
                cutCap = need + need / 4 + 256;
            }
            for (size_t i = 0; i < names.size(); ++i) {
                cut[i] = static_cast<uint8_t*>(cutSlab) + at[i];
                cc[i].src = bc.ptr[i]; cc[i].dst = cut[i]; cc[i].elem_bytes = isInt[i] ? 8 : 4; cc[i].reserved = 0;
            }
            int64_t n = 0;                                        // GenericCut, text_to_minh.go:271-288
            minnow::detail::check(mnw_generic_cut(ctx, nullptr, static_cast<const float*>(bc.ptr[(size_t)iMass]), cutoff, bc.rows, cc.data(), (int)cc.size(), &n));
            tCut += now() - t; t = now();
            rowsKept += n;
            if (n > 0) fM->BlockDevice(std::vector<const void*>(cut.begin(), cut.end()), (int)n);       // text_to_minh.go:209
            tWrite += now() - t;
        }
        fM->Close();
        fR->Close();
        const double total = now() - t0;
        if (cutSlab) mnw_dev_free(ctx, cutSlab);
        mnw_kernel_time kt[MNW_K_COUNT];
        mnw_timing_get(ctx, kt);
        double kms = 0;
        std::string kjson;
        for (int k = 0; k < MNW_K_COUNT; ++k)
            if (kt[k].launches) {
                kms += kt[k].total_ms;
                char buf[160];
                std::snprintf(buf, sizeof buf, "%s\"%s\": %.4f", kjson.empty() ? "" : ", ", mnw_kernel_name(k), kt[k].total_ms);
                kjson += buf;
            }
        // how often the single-read encode kernels gave up and how many batches skipped them afterwards: not 0 means something else
        // was running on the GPU and the conversion paid 50 ms waits and the two-launch reruns (minnow_hip.h: mnw_counters_get)
        int64_t cnt[MNW_C_COUNT] = {0};
        mnw_counters_get(ctx, cnt);
        std::FILE* fo = std::fopen(out.c_str(), "rb");
        std::fseek(fo, 0, SEEK_END);
        const long outBytes = std::ftell(fo);
        std::fclose(fo);
        std::printf("{\"text_bytes\": %lld, \"blocks\": %d, \"rows\": %lld, \"rows_kept\": %lld, \"minh_bytes\": %ld, \"total_s\": %.6f, "
                    "\"block_s\": %.6f, \"upload_wait_s\": %.6f, \"upload_thread_s\": %.6f, \"cut_s\": %.6f, \"minh_block_s\": %.6f, \"text_GBps\": %.4f, \"kernel_ms\": {%s}, \"kernels_total_ms\": %.4f, "
                    "\"single_read_aborts\": %lld, \"backoff_batches\": %lld, \"exact_blocks\": %lld}\n",
                    (long long)bytesIn, fR->Blocks(), (long long)rowsIn, (long long)rowsKept, outBytes, total, tParse, fR->UploadWaitSeconds(), fR->UploadSeconds(), tCut, tWrite,
                    (double)bytesIn / total / 1e9, kjson.c_str(), kms, (long long)cnt[MNW_C_SINGLE_READ_ABORTS], (long long)cnt[MNW_C_BACKOFF_BATCHES],
                    (long long)cnt[MNW_C_EXACT_BLOCKS]);
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "panic: " << e.what() << "\n";
        return 1;
    }
}
