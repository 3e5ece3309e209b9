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
    double il[128][2];          // {1 / (1 + (i + 0.5) / 128), -log10 of that}: one 16-byte read
    double ex2[65];
};
__device__ __forceinline__ void fast_tables_load(FastTables* t) {
    for (int i = threadIdx.x; i < 128; i += blockDim.x) { t->il[i][0] = kLogInv[i]; t->il[i][1] = kLogLg10[i]; }
    for (int i = threadIdx.x; i < 65; i += blockDim.x) t->ex2[i] = kExp2[i];
    __syncthreads();
}

// Round 3: the fast forms written against the measured instruction costs (tools/microbench/valu_cost.hip: every float64
// operation, conversion, compare and bit-field instruction issues at half the rate of a float32 add or a shift).  What changed:
// the exponent and the table index come from the float32 bits (two shifts, a subtraction, a mask), the mantissa becomes a
// double through ONE conversion, the table value is in decimal already and the polynomial's coefficients carry log10(e) /
// ln 2, and "could the float32 rounding go either way" is read off the DISCARDED BITS of the double: the 29 mantissa bits
// that the conversion to float32 drops lie within 2^-40 (relative) of the half-way pattern -- three integer instructions
// instead of an exponent extraction, a constructed half-ulp, a subtraction and a compare in float64.  Relative windows need
// |A| away from zero: results below 1/16 in magnitude (x within 15 % of 1) go to the exact form.
__device__ __forceinline__ bool f32_rounding_ambiguous_bits(double A) {
    const uint32_t lo = (uint32_t)__double_as_longlong(A);
    return ((lo & 0x1fffffffu) - (0x10000000u - 4096u)) < 8192u;      // the dropped bits are within 2^12 units of 1000...0
}

// the exact forms, out of line: they run for a few inputs in a million, and inlined they would put ~400
// instructions of float64 code at every call site
__device__ __attribute__((noinline)) float go_log10_exact_f32(float x) { return (float)go_log10((double)x); }
__device__ __attribute__((noinline)) float go_pow10_exact_f32(float y) { return (float)go_pow10((double)y); }

// the fast value of float32(math.Log10(float64(x))); ok = false if the exact form has to decide
__device__ __forceinline__ float go_log10_f32_fast(float x, const FastTables* tb, bool& ok) {
    const uint32_t xb = __float_as_uint(x);
    const int e = (int)(xb >> 23) - 127;
    const uint32_t idx = (xb >> 16) & 127u;
    const double m = (double)__uint_as_float((xb & 0x007fffffu) | 0x3f800000u);                    // [1, 2), exact
    const double2 t = *reinterpret_cast<const double2*>(&tb->il[idx][0]);
    const double r = __fma_rn(m, t.x, -1.0);                       // |r| <= 2^-8
    // log10(1 + r) = log10(e) (r - r^2/2 + r^3/3 - r^4/4 + r^5/5 - r^6/6)   (r^7/7 < 2e-18)
    const double p = r * __fma_rn(r, __fma_rn(r, __fma_rn(r, __fma_rn(r, __fma_rn(r, -0.072382413650541971275, 0.086858896380650365530),
                                  -0.10857362047581295691), 0.14476482730108394255), -0.21714724095162591383), 0.43429448190325182765);
    const double A = __fma_rn((double)e, 0.30102999566398119521, t.y) + p;
    const uint32_t ahi = (uint32_t)((uint64_t)__double_as_longlong(A) >> 32) & 0x7fffffffu;
    ok = (xb - 0x00800000u < 0x7f000000u) && ahi >= 0x3fb00000u && !f32_rounding_ambiguous_bits(A);   // positive normal x; |A| >= 1/16
    return (float)A;
}
__device__ __forceinline__ float go_log10_f32(float x, const FastTables* tb) {
    bool ok;
    const float r32 = go_log10_f32_fast(x, tb, ok);
    if (ok) return r32;
    return go_log10_exact_f32(x);
}

// the fast value alone; `amb` is set if the exact form has to decide (the caller redoes the value with it)
__device__ __forceinline__ float go_log10_f32_try(float x, const FastTables* tb, bool& amb) {
    bool ok;
    const float r32 = go_log10_f32_fast(x, tb, ok);
    amb |= !ok;
    return ok ? r32 : x;
}

__device__ __forceinline__ float go_pow10_f32(float y, const FastTables* tb) {
    const float ay = fabsf(y);
    if (ay < 37.5f && ay > 1e-6f && y != 1.0f && ay != 0.5f) {
        const double t = (double)y * 3.3219280948873623479;        // y * log2(10)
        const double n = rint(t);
        const double f = t - n;                                     // [-0.5, 0.5]
        const double j = rint(f * 64.0);
        const double z = __fma_rn(j, -1.0 / 64.0, f);               // |z| <= 2^-7, in units of log2
        // 2^z = exp(z ln 2): coefficients ln2^k / k!   ((z ln2)^6 / 720 < 4e-17)
        const double q = __fma_rn(z, __fma_rn(z, __fma_rn(z, __fma_rn(z, __fma_rn(z, 1.3333558146428443423e-3, 9.6181291076284771619e-3),
                                  5.5504108664821579953e-2), 2.4022650695910071233e-1), 6.9314718055994530942e-1), 1.0);
        const double s = tb->ex2[(int)j + 32] * q;                  // 2^f, in [0.70, 1.42]
        const double A = __longlong_as_double(__double_as_longlong(s) + ((long long)(int)n << 52));   // * 2^n: |n| <= 125, no overflow
        if (!f32_rounding_ambiguous_bits(A)) return (float)A;       // (A is a positive normal double, far from the float32 denormals: 10^-37.5)
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
