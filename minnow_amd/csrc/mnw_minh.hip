// minh's per-column value transforms (go/minh/minh.go): processFloatGroup on the writer side (139-149) and
// 10^x on the reader side (315-319), as element-wise passes over device columns.  The reference has the same
// data flow: Writer.Block copies a column into minh.buf, processes it in place and encodes from there
// (minh.go:128-135).  Up to kMinhBatch columns per launch; the column table travels in the kernel arguments.
// HBM bound when Log == 0 (8 B/element); with Log != 0 the float64 logarithm / power (mnw_common.h, restated
// from Go's portable definitions) makes them VALU bound.
// (This file is included by mnw_kernels.hip, inside namespace mnw.)

// self-test: which = 0: log10, 1: pow10.  Compares the fast and the exact form for the float32 bit patterns
// [first, first + count) and counts mismatches (out[0]) and inputs that took the exact fallback (out[1]).
__global__ __launch_bounds__(256) void k_minh_selftest(int which, uint64_t first, uint64_t count, unsigned long long* out) {
    __shared__ FastTables tb;
    fast_tables_load(&tb);
    unsigned long long bad = 0, slow = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256) {
        const float v = __uint_as_float((uint32_t)(first + i));
        float fast, exact;
        bool fell_back;
        if (which == 0) {
            fast = go_log10_f32(v, &tb); exact = (float)go_log10((double)v);
            const uint32_t xb = __float_as_uint(v);
            fell_back = !(xb - 0x00800000u < 0x7f000000u);
        } else {
            fast = go_pow10_f32(v, &tb); exact = (float)go_pow10((double)v);
            const float ay = fabsf(v);
            fell_back = !(ay < 37.5f && ay > 1e-6f && v != 1.0f && ay != 0.5f);
        }
        const uint32_t a = __float_as_uint(fast), b = __float_as_uint(exact);
        const bool both_nan = (fast != fast) && (exact != exact);
        if (a != b && !both_nan) ++bad;
        if (fell_back) ++slow;
    }
    if (bad) atomicAdd(&out[0], bad);
    if (slow) atomicAdd(&out[1], slow);
}

constexpr int kMinhBatch = 64;
struct MinhCol {
    const float* src;
    float* dst;
    int64_t n;
    float low, high, high_below;
    int32_t xform;             // kXfLog10 | kXfClamp (writer) ; kXfLog10 (reader: 10^x)
};
struct MinhArgs {
    MinhCol col[kMinhBatch];
};

__global__ __launch_bounds__(256) void k_minh_process(MinhArgs a) {
    __shared__ FastTables tb;
    const MinhCol c = a.col[blockIdx.y];
    if (c.xform & kXfLog10) fast_tables_load(&tb);          // block-uniform
    const int64_t n4 = ((((uintptr_t)c.src | (uintptr_t)c.dst) & 15) == 0) ? c.n / 4 : 0;
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += step) {
        f32x4 v = ldg<f32x4>(c.src + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = xf_apply_fast(v[e], c.xform, c.low, c.high, c.high_below, &tb);
        stg<f32x4>(c.dst + 4 * q, v);
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < c.n; i += step)
        c.dst[i] = xf_apply_fast(c.src[i], c.xform, c.low, c.high, c.high_below, &tb);
}

// float32(math.Pow(10, float64(x))), in place
__global__ __launch_bounds__(256) void k_minh_unlog(MinhArgs a) {
    __shared__ FastTables tb;
    fast_tables_load(&tb);
    const MinhCol c = a.col[blockIdx.y];
    const int64_t n4 = (((uintptr_t)c.dst & 15) == 0) ? c.n / 4 : 0;
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += step) {
        f32x4 v = ldg<f32x4>(c.dst + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = go_pow10_f32(v[e], &tb);
        stg<f32x4>(c.dst + 4 * q, v);
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < c.n; i += step)
        c.dst[i] = go_pow10_f32(c.dst[i], &tb);
}

void launch_minh_process(hipStream_t s, int cus, const MinhArgs& a, int ncols, int64_t max_n) {
    int64_t gx = (max_n / 4 + 255) / 256;
    const int64_t cap = (int64_t)cus * 16 / (ncols < 16 ? ncols : 16) + 1;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_minh_process, dim3((unsigned)gx, (unsigned)ncols), dim3(256), 0, s, a);
}
void launch_minh_unlog(hipStream_t s, int cus, const MinhArgs& a, int ncols, int64_t max_n) {
    int64_t gx = (max_n / 4 + 255) / 256;
    const int64_t cap = (int64_t)cus * 16 / (ncols < 16 ? ncols : 16) + 1;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_minh_unlog, dim3((unsigned)gx, (unsigned)ncols), dim3(256), 0, s, a);
}
void launch_minh_selftest(hipStream_t s, int which, uint64_t first, uint64_t count, unsigned long long* out) {
    hipLaunchKernelGGL(k_minh_selftest, dim3(4096), dim3(256), 0, s, which, first, count, out);
}
