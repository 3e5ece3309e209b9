// mnw_common.h -- types and scalar helpers shared by the HIP kernels and the C-ABI host code.
// Product code: written for gfx950 (MI355X) only; never includes anything from oracle/.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define MNW_HD __host__ __device__ __forceinline__

namespace mnw {

// Launch geometry.  A tile is the unit of work of one 256-thread workgroup iteration; it is a
// multiple of 64 elements so every tile's packed output starts on a 64-bit word of the block's
// stream (64 elements at b bits = b whole words).
constexpr int kThreads = 256;
constexpr int kTile = 8192;               // encode: elements per tile (32 per thread: one packed group each)
constexpr int kTileD = 2048;              // decode: elements per tile
constexpr int kStatSlots = 16;            // per-job stats records, one 64-B line each
constexpr int kStatWords = 8;             // uint64 per record

// device-side job record (host fills it from mnw_enc_job / mnw_dec_job)
struct DevJob {
    const void* src;
    void* dst;
    int64_t n;
    int64_t cap;         // encode: capacity of dst; decode: src_len
    int64_t pixels;
    int64_t tile0;       // first tile of this job in the launch-wide tile numbering
    int64_t min;         // decode / explicit pack: intGroup.mins[b]
    uint64_t seed;       // decode dither key
    uint64_t block;      // decode dither key
    double rdx;          // 1.0 / double(dxq): the pack pass divides by multiplying in float64
    float low, high;
    float dxq;           // (high-low)/float32(pixels), IEEE float32, computed on the host
    float xh;            // least float32 x with q(x) >= pixels/2 (splits the ring for the periodic stats)
    int32_t kind;        // MNW_INT_GROUP / MNW_FLOAT_GROUP
    int32_t periodic;
    int32_t bits;        // decode / explicit pack
    int32_t mode;        // kModeStats: width from the reduce pass; kModeExplicit: (min,bits) given
    int32_t xform;       // kXfLog10 | kXfClamp: minh's processFloatGroup on every value before it is looked at (encode);
                         // kXfLog10: 10^x on every value after the dequantize (decode)
    float high_below;    // Nextafter32(high, -Inf): what the clamp maps x >= high to
    int64_t grp0;        // single-read encode: first group record (64 items each) of this job in the launch
};

constexpr int kModeStats = 0;
constexpr int kModeExplicit = 1;
constexpr int kModeUnsigned = 2;   // reduce only: the int64 input holds uint64 values

// per-block result, same layout as mnw_block_meta
struct Meta {
    int64_t min, bits, nbytes, status;
};

// internal status: the block needs the exact sequential path (out-of-range / NaN / huge values,
// whose periodicMin result depends on element order, group.go:384-409)
constexpr int64_t kNeedsExact = 100;

// order-preserving map int64 -> uint64 so that zero-initialised memory is the identity of max
MNW_HD uint64_t bias(int64_t v) { return (uint64_t)v ^ 0x8000000000000000ull; }
MNW_HD int64_t unbias(uint64_t u) { return (int64_t)(u ^ 0x8000000000000000ull); }

// ---- Go's math.Frexp / Log / Log2 (go/bit/bit.go:20 calls math.Log2) -----------------------
// Restated from the Go standard library's published algorithm (log.go = FreeBSD e_log.c).
// Needed bit-for-bit only for max >= 2^48: below that ceil(log2(max+1)) is the bit length.
MNW_HD double go_frexp_pos(double x, int* e) {   // x finite, > 0, normal (x >= 1 here)
    union { double d; uint64_t u; } c;
    c.d = x;
    *e = (int)((c.u >> 52) & 0x7ff) - 1022;
    c.u = (c.u & ~(0x7ffull << 52)) | (1022ull << 52);
    return c.d;
}

MNW_HD double go_log_frac(double x) {            // x in [0.5, 1)
    const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10;
    const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01;
    const double L3 = 2.857142874366239149e-01, L4 = 2.222219843214978396e-01;
    const double L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01;
    const double L7 = 1.479819860511658591e-01;
    double f1 = x;
    int ki = 0;
    if (f1 < 0.70710678118654752440) { f1 *= 2; ki = -1; }
    double f = f1 - 1;
    double k = (double)ki;
    double s = f / (2 + f);
    double s2 = s * s;
    double s4 = s2 * s2;
    double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    double R = t1 + t2;
    double hfsq = 0.5 * f * f;
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}

// ---- Go's math.Log10 / math.Pow(10, y) for minh's logarithmic columns (go/minh/minh.go:141, 316) ----
// Restated from the portable Go definitions (src/math: frexp.go, log.go, log10.go, exp.go, modf.go, ldexp.go,
// pow.go).  Parity unpinned: there is no Go toolchain to compare with, and amd64 builds of Go use an assembly
// Exp whose last float64 bit can differ.  The library is built with -ffp-contract=off: no operation below
// may be fused.
MNW_HD double go_frexp(double x, int* e) {       // any x
    union { double d; uint64_t u; } c;
    *e = 0;
    if (x == 0.0 || x != x || x - x != 0.0) return x;          // 0, NaN, +-Inf
    c.d = x;
    if (((c.u >> 52) & 0x7ff) == 0) { x *= 4503599627370496.0; *e = -52; c.d = x; }   // subnormal: normalize
    *e += (int)((c.u >> 52) & 0x7ff) - 1022;
    c.u = (c.u & ~(0x7ffull << 52)) | (1022ull << 52);
    return c.d;
}

MNW_HD double go_log(double x) {
    const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10;
    const double L1 = 6.666666666666735130e-01, L2 = 3.999999999940941908e-01;
    const double L3 = 2.857142874366239149e-01, L4 = 2.222219843214978396e-01;
    const double L5 = 1.818357216161805012e-01, L6 = 1.531383769920937332e-01;
    const double L7 = 1.479819860511658591e-01;
    if (x != x || (x > 0 && x - x != 0.0)) return x;           // NaN, +Inf
    if (x < 0) return __builtin_nan("");
    if (x == 0) return -__builtin_inf();
    int ki;
    double f1 = go_frexp(x, &ki);
    if (f1 < 0.70710678118654752440) { f1 *= 2; ki--; }
    const double f = f1 - 1;
    const double k = (double)ki;
    const double s = f / (2 + f);
    const double s2 = s * s;
    const double s4 = s2 * s2;
    const double t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    const double t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    const double R = t1 + t2;
    const double hfsq = 0.5 * f * f;
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}

MNW_HD double go_log10(double x) {               // log10.go: math.log2(x) * (Ln2 / Ln10)
    int e;
    const double frac = go_frexp(x, &e);
    const double l2 = frac == 0.5 ? (double)(e - 1) : go_log(frac) * 1.44269504088896340736 + (double)e;
    return l2 * 0.30102999566398119521;          // Ln2/Ln10 as an exact constant expression, rounded once
}

MNW_HD double go_ldexp(double frac, int exp) {
    union { double d; uint64_t u; } c;
    if (frac == 0 || frac != frac || frac - frac != 0.0) return frac;
    int e = 0;
    c.d = frac;
    if (((c.u >> 52) & 0x7ff) == 0) { frac *= 4503599627370496.0; e = -52; c.d = frac; }   // normalize
    exp += e;
    exp += (int)((c.u >> 52) & 0x7ff) - 1023;
    if (exp < -1075) return frac < 0 ? -0.0 : 0.0;
    if (exp > 1023) return frac < 0 ? -__builtin_inf() : __builtin_inf();
    double m = 1;
    if (exp < -1022) { exp += 53; m = 1.0 / 9007199254740992.0; }      // denormal result: 2**-53
    c.u &= ~(0x7ffull << 52);
    c.u |= (uint64_t)(exp + 1023) << 52;
    return m * c.d;
}

MNW_HD double go_exp(double x) {
    const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10;
    const double Log2e = 1.44269504088896338700e+00;
    const double Overflow = 7.09782712893383973096e+02, Underflow = -7.45133219101941108420e+02;
    const double NearZero = 1.0 / 268435456.0;   // 2**-28
    const double P1 = 1.66666666666666657415e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05;
    const double P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x || (x > 0 && x - x != 0.0)) return x;
    if (x < 0 && x - x != 0.0) return 0;
    if (x > Overflow) return __builtin_inf();
    if (x < Underflow) return 0;
    if (-NearZero < x && x < NearZero) return 1 + x;
    int k = 0;
    if (x < 0) k = (int)(Log2e * x - 0.5);
    else if (x > 0) k = (int)(Log2e * x + 0.5);
    const double hi = x - (double)k * Ln2Hi;
    const double lo = (double)k * Ln2Lo;
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    const double y = 1 - ((lo - (r * c) / (2 - c)) - hi);
    return go_ldexp(y, k);
}

// math.Pow(10, y), pow.go with x == 10 (Frexp(10) = 0.625, 4)
MNW_HD double go_pow10(double y) {
    if (y == 0) return 1;
    if (y == 1) return 10;
    if (y != y) return y;
    if (y - y != 0.0) return y > 0 ? __builtin_inf() : 0.0;      // +-Inf with |x| > 1
    if (y == 0.5) return __builtin_sqrt(10.0);
    if (y == -0.5) return 1 / __builtin_sqrt(10.0);
    const double ay = y < 0 ? -y : y;
    if (ay >= 9223372036854775808.0) return y > 0 ? __builtin_inf() : 0.0;
    double yi = (double)(int64_t)ay;             // Modf of a non-negative value below 2^63: truncation
    double yf = ay - yi;
    double a1 = 1.0;
    int ae = 0;
    if (yf != 0) {
        if (yf > 0.5) { yf--; yi++; }
        a1 = go_exp(yf * go_log(10.0));
    }
    double x1 = 0.625;
    int xe = 4;
    for (int64_t i = (int64_t)yi; i != 0; i >>= 1) {
        if (xe < -4096 || 4096 < xe) { ae += xe; break; }        // catastrophic overflow
        if (i & 1) { a1 *= x1; ae += xe; }
        x1 *= x1;
        xe <<= 1;
        if (x1 < .5) { x1 += x1; xe--; }
    }
    if (y < 0) { a1 = 1 / a1; ae = -ae; }
    return go_ldexp(a1, ae);
}

// minh.processFloatGroup on one value (go/minh/minh.go:139-149)
constexpr int kXfLog10 = 1;
constexpr int kXfClamp = 2;
MNW_HD float xf_apply(float x, int xform, float low, float high, float high_below) {
    if (xform & kXfLog10) x = (float)go_log10((double)x);
    if (xform & kXfClamp) {
        if (x < low) x = low;
        if (x >= high) x = high_below;           // math.Nextafter32(col.High, float32(math.Inf(-1)))
    }
    return x;
}

// bit.PrecisionNeeded (bit.go:19-21).  Returns -1 for max == 2^64-1, where Go computes
// int(ceil(log2(0))) = int(-Inf): out of domain (MNW_E_DOMAIN).
MNW_HD int precision_needed(uint64_t max) {
    if (max == ~0ull) return -1;
    uint64_t m1 = max + 1;
    if (m1 <= (1ull << 48)) {                    // exact region: bit length of max
        return max == 0 ? 0 : 64 - __builtin_clzll(max);
    }
    double x = (double)m1;                       // uint64 -> float64, round to nearest even
    int e;
    double frac = go_frexp_pos(x, &e);
    double l2;
    if (frac == 0.5) l2 = (double)(e - 1);
    else l2 = go_log_frac(frac) * 1.44269504088896340736 + (double)e;
    // ceil of a value in [48, 64]
    int c = (int)l2;
    if ((double)c < l2) c++;
    return c;
}

MNW_HD int64_t array_bytes(int64_t bits, int64_t n) { return (bits * n + 7) / 8; }

// ---- dither: counter-based uniform deviate for floatGroup.readData (DESIGN.md "Dither") -----
MNW_HD uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
MNW_HD uint64_t dither_key(uint64_t seed, uint64_t block, uint64_t i_hi) {
    return mix64(seed + 0x9E3779B97F4A7C15ull * (block + 1) + 0xD1B54A32D192ED03ull * i_hi);
}
MNW_HD uint32_t dither_bits(uint32_t i_lo, uint32_t ka, uint32_t kb) {
    uint32_t x = i_lo ^ ka;
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x ^ kb;
}

}  // namespace mnw
