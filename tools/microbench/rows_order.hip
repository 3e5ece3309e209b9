// Micro-benchmark (not product code): the grid access pattern of the minp vector kernels and nothing else.  A file cell is an
// array-of-structures grid of n_file^3 particles x 12 bytes; a sub-cell's particles are rows of n_sub x 12 bytes (1,200 B at C4's
// geometry) every n_file x 12 bytes (6,000 B).  Every workgroup writes (or reads) the bytes of `tiles` vtiles (2,048 particles each)
// of one sub-cell as 16-byte chunks, consecutive lanes consecutive chunks; what varies is WHICH workgroup takes which piece:
//   order 0: sub-cell major (workgroup w: sub-cell w / Q, piece w % Q)
//   order 1: sx fastest in dispatch order (k_unpack_vec3's vec3_x_order)
//   order 2: sx fastest among the workgroups of ONE XCD (workgroup w runs on XCD w % 8): the five segments of a grid row go through one L2
//   order 3: whole rows: workgroup w writes a contiguous piece of the grid (the ceiling)
//   order 4: like 2, but the five x-neighbours are one after the other on the same XCD *and* the y-rows of a piece interleave (piece = rows r, r + Q, ...)
//   hipcc -O3 --offload-arch=gfx950 rows_order.hip -o rows_order && ./rows_order [n_file] [sub_cells] [tiles]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#define GLOBAL __attribute__((address_space(1)))
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Geom { unsigned n_file, S, n_sub, T, tiles, Q, s3; long n; };

template <bool WRITE>
__global__ __launch_bounds__(256) void k_rows(unsigned* __restrict__ grid, Geom g, int order, unsigned* sink) {
    const unsigned w = blockIdx.x;
    unsigned sc, piece;
    const unsigned S = g.S, Q = g.Q;
    if (order == 0) { sc = w / Q; piece = w % Q; }
    else if (order == 1) { const unsigned sx = w % S, q = w / S; piece = q % Q; sc = sx + S * (q / Q); }
    else if (order == 2 || order == 4) {
        const unsigned xcd = w & 7, slot = w >> 3;                  // slot-th workgroup of its XCD
        const unsigned sx = slot % S, q = (slot / S) * 8 + xcd;     // q: (piece, syz) index, x-neighbours share q
        if (q >= Q * (g.s3 / S)) return;
        piece = q % Q; sc = sx + S * (q / Q);
    } else { sc = 0; piece = 0; }
    u32x4 acc = {w, 1, 2, 3};
    const long chunks_per_piece = (long)g.tiles * 2048 * 12 / 16;
    if (order == 3) {
        // contiguous: piece w of the whole grid
        const long c0 = (long)w * chunks_per_piece, cend = std::min(c0 + chunks_per_piece, g.n * 12 / 16);
        for (long c = c0 + threadIdx.x; c < cend; c += 256) {
            GLOBAL u32x4* p = (GLOBAL u32x4*)grid + c;
            if (WRITE) *p = acc; else acc ^= *p;
        }
    } else {
        const unsigned s2 = S * S, sz = sc / s2, sq = sc - sz * s2, sy = sq / S, sx = sq - sy * S;
        const long row_chunks = g.n_sub * 12 / 16;                   // 75
        const long sub_chunks = (long)g.n_sub * g.n_sub * row_chunks;
        const long c0 = (long)piece * chunks_per_piece, cend = std::min(c0 + chunks_per_piece, sub_chunks);
        for (long c = c0 + threadIdx.x; c < cend; c += 256) {
            const long row = c / row_chunks, col = c - row * row_chunks;
            const long jz = row / g.n_sub, jy = row - jz * g.n_sub;
            const long prow = ((long)(sz * g.n_sub + jz) * g.n_file + (sy * g.n_sub + jy)) * g.n_file + sx * g.n_sub;   // first particle of the row
            GLOBAL u32x4* p = (GLOBAL u32x4*)grid + prow * 12 / 16 + col;
            if (WRITE) *p = acc; else acc ^= *p;
        }
    }
    if (!WRITE && acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = 1;
}

template <typename F> float best_of(F f, int reps) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    f(); CHK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < reps; ++r) { CHK(hipEventRecord(a)); f(); CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b)); float ms; CHK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms); }
    return best;
}

int main(int argc, char** argv) {
    Geom g;
    g.n_file = argc > 1 ? atoi(argv[1]) : 500; g.S = argc > 2 ? atoi(argv[2]) : 5; g.tiles = argc > 3 ? atoi(argv[3]) : 4;
    g.n_sub = g.n_file / g.S; g.s3 = g.S * g.S * g.S; g.n = (long)g.n_file * g.n_file * g.n_file;
    g.T = (unsigned)(((long)g.n_sub * g.n_sub * g.n_sub + 2047) / 2048); g.Q = (g.T + g.tiles - 1) / g.tiles;
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, pci bus %d; grid of %u^3 particles x 12 B = %.2f GB, %u^3 sub-cells (rows of %u B every %u B), %u vtiles per workgroup\n", prop.gcnArchName, prop.pciBusID,
           g.n_file, g.n * 12 / 1e9, g.S, g.n_sub * 12, g.n_file * 12, g.tiles);
    unsigned *grid, *sink;
    CHK(hipMalloc(&grid, g.n * 12 + 4096)); CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(grid, 1, g.n * 12));
    const char* names[] = {"sub-cell major", "sx fastest in dispatch order", "sx fastest within an XCD", "contiguous pieces (ceiling)"};
    for (int wr = 1; wr >= 0; --wr)
        for (int order = 0; order < 4; ++order) {
            const long pieces = order == 3 ? (g.n * 12 / 16 + (long)g.tiles * 1536 - 1) / ((long)g.tiles * 1536) : (long)g.s3 * g.Q;
            const long grid_n = order == 2 ? ((pieces / g.S + 7) / 8 * g.S) * 8 : pieces;
            const float ms = best_of([&] {
                if (wr) hipLaunchKernelGGL(k_rows<true>, dim3((unsigned)grid_n), dim3(256), 0, 0, grid, g, order, sink);
                else hipLaunchKernelGGL(k_rows<false>, dim3((unsigned)grid_n), dim3(256), 0, 0, grid, g, order, sink);
            }, 8);
            printf("  %-6s %-34s %6ld workgroups  %7.1f us  %5.2f TB/s\n", wr ? "write" : "read", names[order], grid_n, ms * 1e3, g.n * 12.0 / ms / 1e9);
        }
    return 0;
}
