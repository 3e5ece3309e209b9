// Fast forms of minh's float32(math.Log10(float64(x))) and float32(math.Pow(10, float64(y))) for device code.
// (Included by mnw_kernels.hip inside namespace mnw; the exact restatements are in mnw_common.h.)
#pragma once
#include "mnw_fast_tables.h"

// ---- fast paths -----------------------------------------------------------------------------------------
// float32(math.Log10(float64(x))) and float32(math.Pow(10, float64(y))) only need Go's float64 result to the
// precision that decides the float32 rounding.  The fast paths compute the value to ~1e-13 with a table and a
// short polynomial (no float64 division, no frexp/ldexp loops), check that no float32 rounding boundary lies within
// that error of it, and otherwise -- and for every special input -- run the exact restatement.  That the two
// agree is not argued from the error bounds alone: tests/test_minh_gpu.py runs mnw_minh_selftest over ALL 2^32
// float32 inputs of each function and requires zero mismatches.

// true if rounding to float32 could go either way for some value within eps of A (r32 = float32(A)): a midpoint
// between two adjacent float32 values lies within eps of A.  On A's side of r32 the midpoint is half an ulp away;
// below a power of two the spacing halves, which is simply handed to the exact path, like zero, subnormals, Inf, NaN.
__device__ __forceinline__ bool f32_rounding_ambiguous(double A, float r32, double eps) {
    const uint32_t rb = __float_as_uint(r32);
    const uint32_t ex = (rb >> 23) & 0xffu;
    if (ex == 0 || ex == 0xff || (rb & 0x007fffffu) == 0) return true;
    const double half_ulp = __longlong_as_double((long long)(ex - 127 - 24 + 1023) << 52);
    const double d = fabs(A - (double)r32);                         // exact: both are within half an ulp of each other
    return half_ulp - d <= eps;
}

// (tables: the workgroup's LDS copies -- 64 lanes indexing a global table go through the texture path one
// 8-byte gather at a time, which made the fast path no faster than the exact one)
struct FastTables {
    double inv[128], lg[128], ex2[65];
};
__device__ __forceinline__ void fast_tables_load(FastTables* t) {
    for (int i = threadIdx.x; i < 128; i += blockDim.x) { t->inv[i] = kLogInv[i]; t->lg[i] = kLogLg[i]; }
    for (int i = threadIdx.x; i < 65; i += blockDim.x) t->ex2[i] = kExp2[i];
    __syncthreads();
}

// the exact forms, out of line: they run for a few inputs in a million, and inlined they would put ~400
// instructions of float64 code at every call site
__device__ __attribute__((noinline)) float go_log10_exact_f32(float x) { return (float)go_log10((double)x); }
__device__ __attribute__((noinline)) float go_pow10_exact_f32(float y) { return (float)go_pow10((double)y); }

__device__ __forceinline__ float go_log10_f32(float x, const FastTables* tb) {
    const uint32_t xb = __float_as_uint(x);
    if (xb - 0x00800000u < 0x7f000000u) {                          // positive, normal, finite
        const double xd = (double)x;
        const uint64_t u = (uint64_t)__double_as_longlong(xd);
        const int e = (int)((u >> 52) & 0x7ff) - 1023;
        const double m = __longlong_as_double((long long)((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull));   // [1, 2)
        const int idx = (int)((u >> 45) & 127);
        const double r = m * tb->inv[idx] - 1.0;                   // |r| < 2^-7.9
        // log1p(r) = r - r^2/2 + r^3/3 - r^4/4 + r^5/5 - r^6/6  (r^7/7 < 3e-18)
        const double p = r * (1.0 + r * (-0.5 + r * (1.0 / 3.0 + r * (-0.25 + r * (0.2 + r * (-1.0 / 6.0))))));
        const double A = ((double)e + tb->lg[idx]) * 0.30102999566398119521 + p * 0.43429448190325182765;
        const float r32 = (float)A;
        if (!f32_rounding_ambiguous(A, r32, 4e-13)) return r32;
    }
    return go_log10_exact_f32(x);
}

// the fast value alone; `amb` is set if the exact form has to decide (the caller redoes the value with it)
__device__ __forceinline__ float go_log10_f32_try(float x, const FastTables* tb, bool& amb) {
    const uint32_t xb = __float_as_uint(x);
    if (xb - 0x00800000u < 0x7f000000u) {
        const double xd = (double)x;
        const uint64_t u = (uint64_t)__double_as_longlong(xd);
        const int e = (int)((u >> 52) & 0x7ff) - 1023;
        const double m = __longlong_as_double((long long)((u & 0x000fffffffffffffull) | 0x3ff0000000000000ull));
        const int idx = (int)((u >> 45) & 127);
        const double r = m * tb->inv[idx] - 1.0;
        const double p = r * (1.0 + r * (-0.5 + r * (1.0 / 3.0 + r * (-0.25 + r * (0.2 + r * (-1.0 / 6.0))))));
        const double A = ((double)e + tb->lg[idx]) * 0.30102999566398119521 + p * 0.43429448190325182765;
        const float r32 = (float)A;
        amb |= f32_rounding_ambiguous(A, r32, 4e-13);
        return r32;
    }
    amb = true;
    return x;
}

__device__ __forceinline__ float go_pow10_f32(float y, const FastTables* tb) {
    const float ay = fabsf(y);
    if (ay < 37.5f && ay > 1e-6f && y != 1.0f && ay != 0.5f) {
        const double t = (double)y * 3.3219280948873623479;        // y * log2(10)
        const double n = rint(t);
        const double f = t - n;                                     // [-0.5, 0.5]
        const double j = rint(f * 64.0);
        const double z = (f - j * (1.0 / 64.0)) * 0.69314718055994530942;   // |z| <= ln2/128
        // exp(z), z^7/5040 < 3e-20
        const double q = 1.0 + z * (1.0 + z * (0.5 + z * (1.0 / 6.0 + z * (1.0 / 24.0 + z * (1.0 / 120.0 + z * (1.0 / 720.0))))));
        const double s = tb->ex2[(int)j + 32] * q;                  // 2^f, in [0.70, 1.42]
        const double A = __longlong_as_double(__double_as_longlong(s) + ((long long)n << 52));   // * 2^n: |n| <= 125, no overflow
        const float r32 = (float)A;
        if (!f32_rounding_ambiguous(A, r32, 1e-12 * A)) return r32;
    }
    return go_pow10_exact_f32(y);
}

// minh.processFloatGroup on one value (go/minh/minh.go:139-149), logarithm through the fast path
__device__ __forceinline__ float xf_apply_fast(float x, int xform, float low, float high, float high_below, const FastTables* tb) {
    if (xform & kXfLog10) x = go_log10_f32(x, tb);
    if (xform & kXfClamp) {
        if (x < low) x = low;
        if (x >= high) x = high_below;
    }
    return x;
}


// minh.processFloatGroup on four values through the exact logarithm (the rare redo of a vector in which the fast
// form could not decide); out of line
__device__ __attribute__((noinline)) u32x4 xf_apply_exact4(const float* __restrict__ src, int xform, float low, float high, float high_below) {
    u32x4 r;
    for (int c = 0; c < 4; ++c) r[c] = __float_as_uint(xf_apply(src[c], xform, low, high, high_below));
    return r;
}
