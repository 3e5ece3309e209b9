// GenericCut of the reference's text_to_minh (scripts/text_to_minh.go:271-312): the rows of a catalogue block whose mass
// exceeds the cutoff are kept, in order, in every column -- ok[i] = m[i] > cutoff, then x[j++] = x[i] for the kept i.
// On the device this is an order-preserving stream compaction:
//
//   k_cut_count    rows kept per tile of 2,048 rows
//   k_cut_scan     exclusive prefix over the tiles (one workgroup), total to the host
//   k_cut_scatter  every thread owns 8 consecutive rows; its first output row = tile base + rows kept by the threads in
//                  front of it (prefix sum over the workgroup); it then moves its kept rows of every column
// (This file is included by mnw_kernels.hip, inside namespace mnw.)

constexpr int kCutRows = 8;                        // rows per thread
constexpr int kCutTile = 256 * kCutRows;           // rows per workgroup
constexpr int kCutMaxCols = 64;

struct CutCols {
    const void* src[kCutMaxCols];
    void* dst[kCutMaxCols];
    uint8_t esz[kCutMaxCols];                      // 4 or 8 bytes per element
    int32_t n;
};

// bit k = mass[i0 + k] > cutoff (false for NaN, as in Go), rows behind the end are not kept
__device__ __forceinline__ uint32_t cut_mask(const float* __restrict__ mass, float cutoff, int64_t i0, int64_t n) {
    uint32_t m = 0;
    if (i0 + kCutRows <= n && ((uintptr_t)(mass + i0) & 15) == 0) {
        const f32x4 a = ldg<f32x4>(mass + i0), b = ldg<f32x4>(mass + i0 + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { m |= (a[k] > cutoff ? 1u : 0u) << k; m |= (b[k] > cutoff ? 1u : 0u) << (4 + k); }
    } else {
        for (int k = 0; k < kCutRows; ++k) if (i0 + k < n && mass[i0 + k] > cutoff) m |= 1u << k;
    }
    return m;
}

__global__ __launch_bounds__(256) void k_cut_count(const float* __restrict__ mass, float cutoff, int64_t n, int64_t* __restrict__ tile_cnt) {
    __shared__ int red[4];
    const int64_t i0 = (int64_t)blockIdx.x * kCutTile + threadIdx.x * kCutRows;
    int c = __popc(cut_mask(mass, cutoff, i0, n));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane_id() == 0) red[wave_id()] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// tile_cnt[t] -> exclusive prefix; the total behind the last tile (tile_cnt[tiles])
__global__ __launch_bounds__(1024) void k_cut_scan(int64_t* __restrict__ tile_cnt, int64_t tiles) {
    __shared__ long long ws[16];
    const int t = threadIdx.x;
    const int64_t per = (tiles + 1023) / 1024;
    const int64_t a = t * per, b = a + per < tiles ? a + per : tiles;
    long long s = 0;
    for (int64_t k = a; k < b; ++k) s += tile_cnt[k];
    long long incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const long long u = __shfl_up(incl, o); if ((t & 63) >= o) incl += u; }
    if ((t & 63) == 63) ws[t >> 6] = incl;
    __syncthreads();
    long long base = incl - s;
    for (int w = 0; w < (t >> 6); ++w) base += ws[w];
    for (int64_t k = a; k < b; ++k) { const long long c = tile_cnt[k]; tile_cnt[k] = base; base += c; }
    if (t == 1023) tile_cnt[tiles] = base;
}

__global__ __launch_bounds__(256) void k_cut_scatter(const float* __restrict__ mass, float cutoff, int64_t n,
                                                     const int64_t* __restrict__ tile_base, CutCols cols) {
    __shared__ int wsum[4];
    const int64_t i0 = (int64_t)blockIdx.x * kCutTile + threadIdx.x * kCutRows;
    const uint32_t m = cut_mask(mass, cutoff, i0, n);
    int total;
    const int64_t o0 = tile_base[blockIdx.x] + block_exclusive_scan(__popc(m), wsum, &total);
    if (!m) return;
    const bool whole = i0 + kCutRows <= n;
    for (int c = 0; c < cols.n; ++c) {
        if (cols.esz[c] == 4) {
            const uint32_t* __restrict__ s = (const uint32_t*)cols.src[c] + i0;
            uint32_t* __restrict__ d = (uint32_t*)cols.dst[c] + o0;
            uint32_t v[kCutRows];
            if (whole && ((uintptr_t)s & 15) == 0) {
                const u32x4 a = ldg<u32x4>(s), b = ldg<u32x4>(s + 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) { v[k] = a[k]; v[4 + k] = b[k]; }
            } else {
#pragma unroll
                for (int k = 0; k < kCutRows; ++k) v[k] = (m >> k & 1u) ? s[k] : 0u;
            }
            int j = 0;
#pragma unroll
            for (int k = 0; k < kCutRows; ++k) if (m >> k & 1u) d[j++] = v[k];
        } else {
            const uint64_t* __restrict__ s = (const uint64_t*)cols.src[c] + i0;
            uint64_t* __restrict__ d = (uint64_t*)cols.dst[c] + o0;
            uint64_t v[kCutRows];
            if (whole && ((uintptr_t)s & 15) == 0) {
#pragma unroll
                for (int k = 0; k < kCutRows; k += 2) { const u64x2 a = ldg<u64x2>(s + k); v[k] = a[0]; v[k + 1] = a[1]; }
            } else {
#pragma unroll
                for (int k = 0; k < kCutRows; ++k) v[k] = (m >> k & 1u) ? s[k] : 0ull;
            }
            int j = 0;
#pragma unroll
            for (int k = 0; k < kCutRows; ++k) if (m >> k & 1u) d[j++] = v[k];
        }
    }
}

void launch_cut_count(hipStream_t s, const float* mass, float cutoff, int64_t n, int64_t* tile_cnt) {
    const int64_t tiles = (n + kCutTile - 1) / kCutTile;
    hipLaunchKernelGGL(k_cut_count, dim3((unsigned)tiles), dim3(256), 0, s, mass, cutoff, n, tile_cnt);
    hipLaunchKernelGGL(k_cut_scan, dim3(1), dim3(1024), 0, s, tile_cnt, tiles);
}
void launch_cut_scatter(hipStream_t s, const float* mass, float cutoff, int64_t n, const int64_t* tile_base, const CutCols& cols) {
    const int64_t tiles = (n + kCutTile - 1) / kCutTile;
    hipLaunchKernelGGL(k_cut_scatter, dim3((unsigned)tiles), dim3(256), 0, s, mass, cutoff, n, tile_base, cols);
}
