// BASELINE configs[3] on one GPU, end to end at the file level: one minp file cell (n_file^3 particles, sub_cells^3 sub-cells)
// written with minp::Writer and read back with minp::Reader (include/minp.hpp over the C ABI), positions (periodic box) and
// velocities (limits from the data), from a grid in HOST memory -- what the reference's Writer.Vectors is handed -- and from a
// grid that is already on the device.  Prints one JSON object with the stage times.  (C++ because the reference's caller is
// compiled code; no CPU figure here: the oracle is test infrastructure.)
//   minp_cell <scratch dir> [n_file = 500] [sub_cells = 5]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "minp.hpp"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static uint32_t hash32(uint32_t h) { h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16; return h; }
// a standard normal deviate from two hashes (Box-Muller); enough for a benchmark grid
static float gauss(uint32_t i) {
    const float u1 = ((hash32(2 * i + 1) >> 8) + 1) * (1.0f / 16777217.0f), u2 = (hash32(2 * i + 2) >> 8) * (1.0f / 16777216.0f);
    return std::sqrt(-2.0f * std::log(u1)) * std::cos(6.2831853f * u2);
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: minp_cell <scratch dir> [n_file] [sub_cells]\n"); return 2; }
    const std::string dir = argv[1];
    const int nFile = argc > 2 ? std::atoi(argv[2]) : 500, subCells = argc > 3 ? std::atoi(argv[3]) : 5;
    const size_t n = (size_t)nFile * nFile * nFile;
    const float L = 250.0f;
    const double dxp = 0.001, dxv = 1.0;
    try {
        // ---- the grids: particles on a lattice displaced by 2 Mpc/h (wrapped into the box), velocities of 300 km/s
        std::vector<float> pos(3 * n), vel(3 * n);
        const double tg0 = now();
        for (size_t i = 0; i < n; ++i) {
            const int c[3] = {(int)(i % nFile), (int)((i / nFile) % nFile), (int)(i / ((size_t)nFile * nFile))};
            for (int k = 0; k < 3; ++k) {
                float x = c[k] * (L / nFile) + 2.0f * gauss((uint32_t)(3 * i + k));
                x = std::fmod(x, L); if (x < 0) x += L; if (x >= L) x = 0;
                pos[3 * i + k] = x;
                vel[3 * i + k] = 300.0f * gauss((uint32_t)(3 * i + k) ^ 0x9e3779b9u);
            }
        }
        const double t_gen = now() - tg0;
        minp::Header hd;
        hd.Z = 0; hd.Scale = 1; hd.OmegaM = 0.27; hd.OmegaL = 0.73; hd.H100 = 0.7; hd.Epsilon = 1; hd.UniformMp = 1e9; hd.L = L;
        hd.NSide = nFile; hd.NTotal = (int64_t)n;
        minp::Cell cell; cell.FileIndex = 0; cell.FileCells = 1; cell.SubCells = subCells;
        const std::vector<uint8_t> raw(256, 7);
        mnw_ctx* ctx = minnow::detail::ctx();
        void* dgrid = nullptr;
        minnow::detail::check(mnw_dev_alloc(ctx, (int64_t)(n * 12 + 16), &dgrid));
        std::string js = "{";
        auto add = [&](const std::string& k, double v) { char b[96]; std::snprintf(b, sizeof b, "%s\"%s\": %.6g", js.size() > 1 ? ", " : "", k.c_str(), v); js += b; };
        add("n_file", nFile); add("sub_cells", subCells); add("particles", (double)n); add("grid_GB", n * 12e-9); add("generate_s", t_gen);

        for (int which = 0; which < 2; ++which) {
            const bool periodic = which == 0;
            const std::vector<float>& src = periodic ? pos : vel;
            const double dx = periodic ? dxp : dxv;
            const std::string tag = periodic ? "pos" : "vel";
            const std::string fhost = dir + "/cell_" + tag + "_host.minp", fdev = dir + "/cell_" + tag + "_dev.minp";
            // (a) from host memory, as the reference's caller has it; twice, the second timed (buffers and tables exist by then)
            double t_write_host = 0, t_write_dev = 0;
            for (int rep = 0; rep < 2; ++rep) {
                std::remove(fhost.c_str());              // (a new file each time: overwriting an existing one is 2.5x slower on this file system)
                const double t0 = now();
                minp::Writer w(fhost);
                w.Header(hd, raw, cell, dx, periodic);
                w.Vectors(src.data(), n);
                w.Close();
                t_write_host = now() - t0;
            }
            // (b) from a grid that is on the device already
            minnow::detail::check(mnw_copy_to_device(ctx, dgrid, src.data(), (int64_t)(n * 12)));
            for (int rep = 0; rep < 2; ++rep) {
                std::remove(fdev.c_str());
                const double t0 = now();
                minp::Writer w(fdev);
                w.Header(hd, raw, cell, dx, periodic);
                w.VectorsDevice((const float*)dgrid, n);
                w.Close();
                t_write_dev = now() - t0;
            }
            // the two files are the same bytes
            {
                FILE* a = std::fopen(fhost.c_str(), "rb"); FILE* b = std::fopen(fdev.c_str(), "rb");
                if (!a || !b) throw minp::Error("cannot reopen the files");
                std::vector<uint8_t> ba(1 << 20), bb(1 << 20);
                size_t total = 0;
                for (;;) {
                    const size_t ra = std::fread(ba.data(), 1, ba.size(), a), rb = std::fread(bb.data(), 1, bb.size(), b);
                    if (ra != rb || std::memcmp(ba.data(), bb.data(), ra) != 0) throw minp::Error("host-grid and device-grid files differ");
                    total += ra;
                    if (ra == 0) break;
                }
                std::fclose(a); std::fclose(b);
                add(tag + "_file_GB", total * 1e-9);
            }
            // (c) read back: into host memory, and into a device grid
            std::vector<float> back(3 * n);
            double t_read_host = 0, t_read_dev = 0;
            for (int rep = 0; rep < 2; ++rep) {
                const double t0 = now();
                minp::Reader r(fhost);
                r.DitherSeed = 11;
                r.Vectors(back.data());
                r.Close();
                t_read_host = now() - t0;
            }
            for (int rep = 0; rep < 2; ++rep) {
                const double t0 = now();
                minp::Reader r(fdev);
                r.DitherSeed = 11;
                r.VectorsDevice((float*)dgrid);
                r.Close();
                t_read_dev = now() - t0;
            }
            // the reference's own criterion (minp_test.go:120-130): every coordinate within dx (on the ring for positions)
            double worst = 0;
            for (size_t i = 0; i < 3 * n; ++i) {
                double d = std::fabs((double)back[i] - (double)src[i]);
                if (periodic && d > L / 2) d = L - d;
                if (d > worst) worst = d;
            }
            // (a velocity component's maximum may wrap to the bottom of its range: SURVEY.md 9-Q8; positions may not deviate at all)
            if (periodic && worst > dx * 1.01 + 1e-4) throw minp::Error("round trip error " + std::to_string(worst) + " exceeds dx");
            add(tag + "_write_from_host_s", t_write_host); add(tag + "_write_from_device_s", t_write_dev);
            add(tag + "_read_to_host_s", t_read_host); add(tag + "_read_to_device_s", t_read_dev);
            add(tag + "_write_from_host_GBps", n * 12e-9 / t_write_host); add(tag + "_write_from_device_GBps", n * 12e-9 / t_write_dev);
            add(tag + "_read_to_host_GBps", n * 12e-9 / t_read_host); add(tag + "_read_to_device_GBps", n * 12e-9 / t_read_dev);
            add(tag + "_max_error", worst);
            std::remove(fhost.c_str()); std::remove(fdev.c_str());
        }
        mnw_dev_free(ctx, dgrid);
        js += "}";
        std::printf("%s\n", js.c_str());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "minp_cell: %s\n", e.what());
        return 1;
    }
    return 0;
}
