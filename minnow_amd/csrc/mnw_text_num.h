// Number conversions of the text catalogue ingest (go/text/parse.go:126-170: strconv.Atoi, float32(strconv.ParseFloat(s, 64))).
// Pure functions of their arguments, written so that they compile for the device (included by mnw_text.hip) and for the
// host (tests/test_text_num_cpu.py builds them with g++ and checks them against correctly rounded conversions) -- the host
// build is a test harness, not a code path of the library.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MNW_TN __host__ __device__ __forceinline__
#if defined(__HIP_DEVICE_COMPILE__)
#define MNW_DMUL(a, b) __dmul_rn(a, b)
#define MNW_DDIV(a, b) __ddiv_rn(a, b)
#else
#define MNW_DMUL(a, b) ((a) * (b))
#define MNW_DDIV(a, b) ((a) / (b))
#endif
#else
#define MNW_TN static inline
#define MNW_DMUL(a, b) ((a) * (b))      /* IEEE double, no contraction possible in a single operation */
#define MNW_DDIV(a, b) ((a) / (b))
#endif

// 10^k, 0 <= k <= 22: the powers of ten a float64 holds exactly
MNW_TN double text_pow10(int k) {
    const double t[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                          1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    return t[k];
}

#include "mnw_pow10_table.h"

MNW_TN void text_mul64(uint64_t a, uint64_t b, uint64_t* hi, uint64_t* lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    *lo = a * b; *hi = __umul64hi(a, b);
#else
    const unsigned __int128 p = (unsigned __int128)a * b;
    *lo = (uint64_t)p; *hi = (uint64_t)(p >> 64);
#endif
}

// m * 10^e10 (0 < m < 2^64) rounded to the nearest float64, ties to even -- what strconv.ParseFloat returns for a literal
// with that mantissa and exponent -- or false when this function cannot decide (the caller then leaves the literal to the
// host): exponents outside the table, results outside the normal range, and the rare products too close to a rounding
// boundary.  Interval form of the Eisel-Lemire method: with T = the 128 leading bits of 10^e10, truncated (10^e10 =
// (T + eps) 2^(k-127), 0 <= eps < 1, eps == 0 exactly for 0 <= e10 <= 55) and mn = m shifted to 64 significant bits, the
// 192-bit integer P = mn * T brackets the value: V / 2^(k-127-lz) lies in [P, P + mn), and equals P when T is exact.  The
// 53-bit result is decided where every point of that interval rounds the same way; since mn < 2^64 lies far below the 138
// or 139 bits that are rounded away, that fails only if those bits are within mn of the half-way point from below.
MNW_TN bool text_scale10(uint64_t m, int e10, double* out) {
    if (e10 < MNW_POW10_EMIN || e10 > MNW_POW10_EMAX) return false;
    const uint64_t tab[MNW_POW10_EMAX - MNW_POW10_EMIN + 1][2] = MNW_POW10_TABLE;
    const uint64_t th = tab[e10 - MNW_POW10_EMIN][0], tl = tab[e10 - MNW_POW10_EMIN][1];
    const int k = (e10 * 217706) >> 16;                    // floor(log2(10^e10)) on this range (tools/gen_pow10_table.py asserts it)
    const int lz = __builtin_clzll(m);
    const uint64_t mn = m << lz;
    uint64_t ah, al, bh, bl;
    text_mul64(mn, th, &ah, &al);
    text_mul64(mn, tl, &bh, &bl);
    const uint64_t p0 = bl, p1 = al + bh;
    const uint64_t p2 = ah + (p1 < al ? 1u : 0u);          // P = p2 2^128 + p1 2^64 + p0, 2^190 <= P < 2^192
    const int top = (int)(p2 >> 63);                       // P's leading bit is bit 190 + top
    const int sh = 10 + top;                               // bits of p2 below the 53 kept ones
    uint64_t q = p2 >> sh;
    const uint64_t rt = p2 & ((1ull << sh) - 1), hb = 1ull << (sh - 1);
    bool up;
    if (e10 >= 0 && e10 <= MNW_POW10_EXACT_MAX) {          // V == P: ordinary round to nearest even
        const bool above = rt > hb || (rt == hb && (p1 | p0) != 0), tie = rt == hb && (p1 | p0) == 0;
        up = above || (tie && (q & 1u));
    } else if (rt >= hb) up = true;                        // P is at or above half and V lies strictly above P
    else {
        // below half: stays below unless the discarded bits are within mn of it
        if (rt == hb - 1 && p1 == ~0ull && p0 + mn < p0) return false;
        up = false;
    }
    q += up ? 1u : 0u;
    int be = 63 + top + k - lz + 1023;                     // biased exponent of a 53-bit mantissa with its top bit set
    if (q >> 53) { q >>= 1; be += 1; }
    if (be < 1 || be > 2046) return false;                 // subnormal, zero, overflow: the host decides
    const uint64_t bits = ((uint64_t)be << 52) | (q & 0x000fffffffffffffull);
    double v;
    __builtin_memcpy(&v, &bits, 8);
    *out = v;
    return true;
}

// Both general conversions are single loops over the token without early exits: every character updates a few state
// words and a verdict.  (An earlier form with break / continue / return inside the loops was compiled into code that
// returned "syntax error" for every literal with an exponent when the token was read through a global pointer, while the
// same source read through an LDS pointer worked: tools/microbench/text_num_dev.hip shows it and guards this form.)

// strconv.Atoi: optional sign, decimal digits, must fit int64 (int is 64 bits on the reference's platforms)
MNW_TN bool text_atoi(const uint8_t* p, int len, int64_t* out) {
    const bool sgn = len > 0 && (p[0] == '+' || p[0] == '-');
    const bool neg = sgn && p[0] == '-';
    const int i0 = sgn ? 1 : 0;
    const uint64_t limit = 0x7fffffffffffffffull + (neg ? 1u : 0u);       // |value| may reach 2^63 for a negative literal
    uint64_t v = 0;
    bool bad = i0 >= len;
    for (int i = i0; i < len; ++i) {
        const uint32_t d = (uint32_t)p[i] - '0';
        const bool isd = d <= 9u;
        // v * 10 + d > limit  <=>  v > (limit - d) / 10 : Atoi reports a range error
        const bool over = isd && v > (limit - d) / 10;
        bad = bad || !isd || over;
        v = bad ? v : v * 10 + d;
    }
    *out = neg ? (int64_t)(0 - v) : (int64_t)v;
    return !bad;
}

// 0: converted (exactly as strconv.ParseFloat(s, 64) would), 1: leave to the host, 2: syntax error
MNW_TN int text_atof(const uint8_t* p, int len, double* out) {
    const bool sgn = len > 0 && (p[0] == '+' || p[0] == '-');
    const bool neg = sgn && p[0] == '-';
    const int i0 = sgn ? 1 : 0;
    if (i0 >= len) return 2;
    const uint32_t c0 = p[i0] | 0x20u;
    if (c0 == 'i' || c0 == 'n') return 1;                      // inf / infinity / nan: the host decides
    uint64_t m = 0;
    int nd = 0, e10 = 0, ex = 0, exp_pos = -2, verdict = 0;
    bool seen_dot = false, seen_digit = false, in_exp = false, eneg = false, exp_digit = false, inexact = false;
    for (int i = i0; i < len; ++i) {
        const uint32_t c = p[i], d = c - '0';
        const bool isd = d <= 9u;
        const bool live = verdict == 0;
        const bool mant = live && !in_exp, expo = live && in_exp;
        const bool is_e = (c | 0x20u) == 'e' && seen_digit;
        const bool is_esign = i == exp_pos + 1 && (c == '+' || c == '-');
        // mantissa: at most 19 digits are kept, further ones only move the decimal point (and make the literal inexact)
        const bool lead0 = m == 0 && d == 0;                                   // leading zeros
        const bool keep = mant && isd && !lead0 && nd < 19, drop = mant && isd && !lead0 && nd >= 19;
        m = keep ? m * 10 + d : m;
        nd += keep ? 1 : 0;
        e10 += (mant && isd && seen_dot && !drop) ? -1 : 0;
        e10 += (drop && !seen_dot) ? 1 : 0;
        inexact = inexact || (drop && d != 0);
        seen_digit = seen_digit || (mant && isd);
        // exponent (in_exp as it was before this character)
        ex = (expo && isd && ex < 100000) ? ex * 10 + (int)d : ex;
        exp_digit = exp_digit || (expo && isd);
        eneg = eneg || (expo && !isd && is_esign && c == '-');
        // everything else
        int v = 0;
        if (mant && !isd) {
            if (c == '.') v = seen_dot ? 2 : 0;
            else if (is_e) v = 0;
            else if (c == '_' || (c | 0x20u) == 'x') v = 1;                    // digit separators, hex floats: host
            else v = 2;
        }
        if (expo && !isd && !is_esign) v = c == '_' ? 1 : 2;
        if (mant && !isd && c != '.' && is_e) { in_exp = true; exp_pos = i; }
        seen_dot = seen_dot || (mant && c == '.');
        verdict = live ? v : verdict;
    }
    if (verdict) return verdict;
    if (!seen_digit || (in_exp && !exp_digit)) return 2;
    e10 += eneg ? -ex : ex;
    double v;
    if (m == 0) v = 0.0;
    else if (inexact) return 1;                            // more than 19 significant digits: the host decides
    else {
        // trailing zeros of the mantissa belong to the exponent ("2.500000000000000000e+02", numpy.savetxt's default
        // '%.18e', is 25 * 10^1): five conditional divisions by constants
#define MNW_STRIP0(P, K) { const uint64_t q_ = m / P; if (q_ * P == m) { m = q_; e10 += K; } }
        MNW_STRIP0(10000000000000000ull, 16) MNW_STRIP0(100000000ull, 8) MNW_STRIP0(10000ull, 4) MNW_STRIP0(100ull, 2) MNW_STRIP0(10ull, 1)
#undef MNW_STRIP0
        if (m < (1ull << 53) && e10 >= 0 && e10 <= 22) v = MNW_DMUL((double)m, text_pow10(e10));         // Clinger: one exact operation
        else if (m < (1ull << 53) && e10 < 0 && e10 >= -22) v = MNW_DDIV((double)m, text_pow10(-e10));
        else if (!text_scale10(m, e10, &v)) return 1;      // 17-19 digits or a larger exponent: 128-bit power-of-ten table
    }
    *out = neg ? -v : v;
    return 0;
}

// ---- a token of at most 16 bytes held in four registers (byte i = bits 8(i & 3).. of w[i >> 2]) ------------------------
// The conversions below are the straight-line forms of text_atoi / text_atof for the tokens a catalogue is made of; they
// give up (false / 1) on anything else and the caller then asks the general functions above, which decide between
// value, host list and error.  Sixteen digits cannot overflow an int64 or exceed the 19 digits text_atof keeps.
MNW_TN uint32_t tok_byte(const uint32_t (&w)[4], int i) { return (w[i >> 2] >> (8 * (i & 3))) & 0xffu; }

MNW_TN bool fast_atoi16(const uint32_t (&w)[4], int len, int64_t* out, int bound = 16) {   // bound: >= len, the same in every lane
    const uint32_t c0 = w[0] & 0xffu;
    const bool sgn = c0 == '+' || c0 == '-', neg = c0 == '-';
    uint64_t v = 0;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i < bound) {                                   // (a wave-uniform test; no break, so that the loop unrolls)
            const uint32_t d = tok_byte(w, i) - '0';
            const bool act = i < len && !(i == 0 && sgn);
            bad |= act && d > 9u;
            v = act ? v * 10 + d : v;
        }
    }
    if (bad || len - (sgn ? 1 : 0) <= 0) return false;
    *out = neg ? (int64_t)(0 - v) : (int64_t)v;
    return true;
}

// 4 bits: which bytes of w equal the byte replicated in pat / which bytes are not a decimal digit (exact for all 256 byte values)
MNW_TN uint32_t tok_eq4(uint32_t w, uint32_t pat) {
    const uint32_t x = w ^ pat;
    const uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);
    return (z * 0x00204081u) >> 28;
}
MNW_TN uint32_t tok_nondigit4(uint32_t w) {
    const uint32_t x = w ^ 0x30303030u;                                   // '0'..'9' -> 0..9
    const uint32_t t = (((x & 0x7f7f7f7fu) + 0x76767676u) | x) & 0x80808080u;   // high bit: the byte is above 9
    return (t * 0x00204081u) >> 28;
}

// The token's structure -- sign, digits with at most one dot, optionally e / E, sign, digits -- is read off five 16-bit
// masks made with byte-parallel compares; the loop over the characters then only accumulates digits: the mantissa where
// the mantissa-digit mask has a bit, the exponent where the exponent-digit mask has one.
MNW_TN int fast_atof16(const uint32_t (&w)[4], int len, const double* p10, double* out, int bound = 16) {
    uint32_t nd = 0, dot = 0, em = 0, plus = 0, minus = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        nd |= tok_nondigit4(w[q]) << (4 * q);
        dot |= tok_eq4(w[q], 0x2e2e2e2eu) << (4 * q);
        em |= tok_eq4(w[q] | 0x20202020u, 0x65656565u) << (4 * q);
        plus |= tok_eq4(w[q], 0x2b2b2b2bu) << (4 * q);
        minus |= tok_eq4(w[q], 0x2d2d2d2du) << (4 * q);
    }
    const uint32_t lenmask = len >= 16 ? 0xffffu : ((1u << len) - 1u);
    const uint32_t sign = plus | minus;
    const uint32_t s0 = sign & 1u;
    const bool neg = (minus & 1u) != 0;
    em &= lenmask & ~1u;
    const int epos = em ? __builtin_ctz(em) : len;
    const uint32_t below_e = (1u << epos) - 1u;
    const uint32_t mant = below_e & lenmask & ~s0;                        // positions of the mantissa
    const uint32_t dots = dot & mant;
    const uint32_t mdig = ~nd & mant;                                     // its digits
    const int dpos = dots ? __builtin_ctz(dots) : epos;
    bool bad = (nd & mant) != dots || (dots & (dots - 1u)) != 0 || mdig == 0;
    uint32_t edig = 0;
    bool eneg = false;
    if (epos < len) {
        const uint32_t expo = lenmask & ~below_e & ~(1u << epos);         // positions behind the e
        const uint32_t es = sign & (1u << (epos + 1)) & expo;             // a sign directly behind it
        eneg = (minus & es) != 0;
        edig = ~nd & expo;
        bad = bad || (nd & expo) != es || edig == 0;
    }
    if (bad) return 1;
    uint64_t m = 0;
    int ex = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i < bound) {                                   // (a wave-uniform test; no break, so that the loop unrolls)
            const uint32_t d = tok_byte(w, i) - '0';
            m = ((mdig >> i) & 1u) ? m * 10 + d : m;
            ex = ((edig >> i) & 1u) ? (ex > 99999 ? ex : ex * 10 + (int)d) : ex;
        }
    }
    int e10 = -(int)__builtin_popcount(mdig & ~((2u << dpos) - 1u));      // mantissa digits behind the dot
    e10 += eneg ? -ex : ex;
    double v;
    if (m == 0) v = 0.0;
    else if (m >> 53) return 1;
    else if (e10 >= 0 && e10 <= 22) v = MNW_DMUL((double)m, p10[e10]);
    else if (e10 < 0 && e10 >= -22) v = MNW_DDIV((double)m, p10[-e10]);
    else return 1;
    *out = neg ? -v : v;
    return 0;
}

