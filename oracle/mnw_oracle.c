/*
 * mnw_oracle.c -- CPU restatement of minnow's quantize + bit-pack path (see mnw_oracle.h).
 * TEST INFRASTRUCTURE ONLY: never linked into, included by or called from the product.
 *
 * Build: plain C99, no fast-math, no FP contraction (Go on amd64 never fuses a*b+c and
 * evaluates float32 expressions in float32).
 */
#include "mnw_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------
 * Go's math.Frexp / math.Log / math.Log2, restated from the published Go standard library
 * (src/math/frexp.go, log.go, log10.go; log.go is the FreeBSD e_log.c algorithm).
 * The reference calls math.Log2 in go/bit/bit.go:20.  libm's log2 is NOT the same
 * function bit-for-bit, and PrecisionNeeded's result near 2^49 and above depends on it.
 * ---------------------------------------------------------------------------------- */
static double go_frexp(double x, int *e) {
    uint64_t u;
    if (x == 0.0 || isnan(x) || isinf(x)) { *e = 0; return x; }
    *e = 0;
    memcpy(&u, &x, 8);
    if (((u >> 52) & 0x7ff) == 0) {           /* subnormal: normalize first */
        x *= 4503599627370496.0;              /* 2^52 */
        *e = -52;
        memcpy(&u, &x, 8);
    }
    *e += (int)((u >> 52) & 0x7ff) - 1022;
    u &= ~(0x7ffull << 52);
    u |= 1022ull << 52;
    memcpy(&x, &u, 8);
    return x;
}

static double go_log(double x) {
    static const double Ln2Hi = 6.93147180369123816490e-01; /* 3fe62e42 fee00000 */
    static const double Ln2Lo = 1.90821492927058770002e-10; /* 3dea39ef 35793c76 */
    static const double L1 = 6.666666666666735130e-01;      /* 3FE55555 55555593 */
    static const double L2 = 3.999999999940941908e-01;      /* 3FD99999 9997FA04 */
    static const double L3 = 2.857142874366239149e-01;      /* 3FD24924 94229359 */
    static const double L4 = 2.222219843214978396e-01;      /* 3FCC71C5 1D8E78AF */
    static const double L5 = 1.818357216161805012e-01;      /* 3FC74664 96CB03DE */
    static const double L6 = 1.531383769920937332e-01;      /* 3FC39A09 D078C69F */
    static const double L7 = 1.479819860511658591e-01;      /* 3FC2F112 DF3E5244 */
    static const double HalfSqrt2 = 0.70710678118654752440; /* Sqrt2/2 */
    int ki;
    double f1, f, k, s, s2, s4, t1, t2, R, hfsq;

    if (isnan(x) || (isinf(x) && x > 0)) return x;
    if (x < 0) return NAN;
    if (x == 0) return -INFINITY;

    f1 = go_frexp(x, &ki);
    if (f1 < HalfSqrt2) { f1 *= 2; ki--; }
    f = f1 - 1;
    k = (double)ki;

    s = f / (2 + f);
    s2 = s * s;
    s4 = s2 * s2;
    t1 = s2 * (L1 + s4 * (L3 + s4 * (L5 + s4 * L7)));
    t2 = s4 * (L2 + s4 * (L4 + s4 * L6));
    R = t1 + t2;
    hfsq = 0.5 * f * f;
    return k * Ln2Hi - ((hfsq - (s * (hfsq + R) + k * Ln2Lo)) - f);
}

double mnwo_go_log2(double x) {
    static const double InvLn2 = 1.44269504088896340736; /* 1/Ln2, exact constant rounded once */
    int e;
    double frac = go_frexp(x, &e);
    if (frac == 0.5) return (double)(e - 1);
    return go_log(frac) * InvLn2 + (double)e;
}

/* int(f) for a float64 on amd64 (CVTTSD2SQ): NaN and out-of-range give MinInt64. */
static int64_t go_int64_of_f64(double v) {
    if (!(v >= -9223372036854775808.0 && v < 9223372036854775808.0)) return INT64_MIN;
    return (int64_t)v;
}

/* go/bit/bit.go:19-21 */
int64_t mnwo_precision_needed(uint64_t max) {
    uint64_t m1 = max + 1;                     /* wraps to 0 at 2^64-1, as in Go */
    return go_int64_of_f64(ceil(mnwo_go_log2((double)m1)));
}

/* go/bit/bit.go:23-25 */
int64_t mnwo_array_bytes(int64_t bits, int64_t length) {
    return go_int64_of_f64(ceil((double)(bits * length) / 8));
}

/* ------------------------------------------------------------------------------------
 * go/bit/bit.go:84-134  BufferedArray -- literal, byte by byte.
 * ---------------------------------------------------------------------------------- */
int mnwo_pack(int bits, const uint64_t *x, int64_t n, uint8_t *b, int64_t b_len) {
    uint8_t buf[8] = {0}, tbuf[9] = {0};
    uint64_t buf_bytes, mask, j;
    int64_t i, k;

    if (bits > 64) return -1;                               /* bit.go:85-87 */
    if (b_len != mnwo_array_bytes(bits, n)) return -2;      /* bit.go:88-92 */
    for (k = 0; k < b_len; k++) b[k] = 0;                   /* bit.go:94 */

    buf_bytes = (uint64_t)(bits / 8);
    if (buf_bytes * 8 < (uint64_t)bits) buf_bytes++;
    /* Go: (^uint64(0)) >> uint64(64-bits); a shift count >= 64 yields 0 in Go */
    mask = (bits == 0) ? 0 : (~(uint64_t)0) >> (64 - bits);

    for (i = 0; i < n; i++) {
        uint64_t xi = x[i] & mask;                          /* bit.go:107 */
        uint64_t curr_bit = (uint64_t)((i * bits) % 8);
        int64_t start_byte, end_byte;

        for (j = 0; j < buf_bytes; j++) buf[j] = (uint8_t)(xi >> (8 * j));

        tbuf[buf_bytes] = 0;
        for (j = 0; j < buf_bytes; j++) tbuf[j] = (uint8_t)(buf[j] << curr_bit);
        /* Go: byte >> 8 is 0 (curr_bit == 0) */
        for (j = 0; j < buf_bytes; j++)
            tbuf[j + 1] |= (curr_bit == 0) ? 0 : (uint8_t)(buf[j] >> (8 - curr_bit));

        start_byte = i * bits / 8;
        end_byte = ((i + 1) * bits - 1) / 8;
        for (k = 0; k < (end_byte - start_byte) + 1; k++) b[start_byte + k] |= tbuf[k];
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * go/bit/bit.go:29-82  (*Array).Slice -- literal, byte by byte.
 * ---------------------------------------------------------------------------------- */
int mnwo_unpack(int bits, const uint8_t *data, int64_t n, uint64_t *out) {
    uint8_t buf[8] = {0}, tbuf[10] = {0};
    uint64_t buf_bytes, j;
    int64_t i, k;

    if (bits > 64 || bits < 0) return -1;
    buf_bytes = (uint64_t)(bits / 8);
    if (buf_bytes * 8 < (uint64_t)bits) buf_bytes++;

    for (i = 0; i < n; i++) {
        uint64_t start_bit = (uint64_t)(i * bits % 8);
        uint64_t next_start_bit = (start_bit + (uint64_t)bits) % 8;
        int64_t start_byte = i * bits / 8;
        int64_t end_byte = ((i + 1) * bits - 1) / 8;
        int64_t tbuf_bytes = end_byte - start_byte + 1;
        uint8_t start_mask, end_mask;
        uint64_t xi;

        for (k = 0; k < tbuf_bytes; k++) tbuf[k] = data[start_byte + k];

        start_mask = (uint8_t)(0xffu << start_bit);
        end_mask = (next_start_bit == 0) ? 0xff : (uint8_t)(0xffu >> (8 - next_start_bit));
        tbuf[0] &= start_mask;
        tbuf[tbuf_bytes - 1] &= end_mask;

        for (j = 0; j < buf_bytes; j++) buf[j] = (uint8_t)(tbuf[j] >> start_bit);
        for (j = 0; j < buf_bytes; j++)
            buf[j] |= (start_bit == 0) ? 0 : (uint8_t)(tbuf[j + 1] << (8 - start_bit));

        for (k = 0; k < tbuf_bytes; k++) tbuf[k] = 0;

        xi = 0;
        for (j = 0; j < buf_bytes; j++) xi |= (uint64_t)buf[j] << (8 * j);
        out[i] = xi;
    }
    return 0;
}

/* Word-wise forms with the same LSB-first stream layout (element i occupies stream bits
 * [i*bits, (i+1)*bits); stream bit k is bit k%8 of byte k/8; pad bits zero). */
int mnwo_pack_fast(int bits, const uint64_t *x, int64_t n, uint8_t *b, int64_t b_len) {
    uint64_t mask, acc = 0;
    int fill = 0;
    int64_t i, pos = 0;
    if (bits > 64) return -1;
    if (b_len != mnwo_array_bytes(bits, n)) return -2;
    if (bits == 0) return 0;
    mask = (~(uint64_t)0) >> (64 - bits);
    for (i = 0; i < n; i++) {
        uint64_t v = x[i] & mask;
        acc |= v << fill;
        if (fill + bits >= 64) {
            memcpy(b + pos, &acc, 8);
            pos += 8;
            acc = (fill == 0) ? 0 : (v >> (64 - fill));
            fill = fill + bits - 64;
        } else {
            fill += bits;
        }
    }
    while (fill > 0) { b[pos++] = (uint8_t)acc; acc >>= 8; fill -= 8; }
    return 0;
}

int mnwo_unpack_fast(int bits, const uint8_t *data, int64_t n, uint64_t *out) {
    uint64_t mask;
    int64_t i, nbytes;
    if (bits > 64 || bits < 0) return -1;
    if (bits == 0) { for (i = 0; i < n; i++) out[i] = 0; return 0; }
    mask = (~(uint64_t)0) >> (64 - bits);
    nbytes = mnwo_array_bytes(bits, n);
    for (i = 0; i < n; i++) {
        int64_t bitpos = i * (int64_t)bits;
        int64_t byte = bitpos >> 3;
        int sh = (int)(bitpos & 7);
        uint64_t lo = 0;
        uint8_t hi = 0;
        int64_t avail = nbytes - byte;
        if (avail >= 9) { memcpy(&lo, data + byte, 8); hi = data[byte + 8]; }
        else { memcpy(&lo, data + byte, (size_t)(avail > 8 ? 8 : avail)); }
        lo >>= sh;
        if (sh) lo |= (uint64_t)hi << (64 - sh);
        out[i] = lo & mask;
    }
    return 0;
}

/* go/bit/bit.go:151-159 */
int64_t mnwo_bits(const uint64_t *x, int64_t n) {
    uint64_t max;
    int64_t i;
    if (n == 0) return 0;
    max = x[0];
    for (i = 0; i < n; i++) if (x[i] > max) max = x[i];
    return mnwo_precision_needed(max);
}

/* go/group.go:350-357 */
int64_t mnwo_int64_min(const int64_t *x, int64_t n) {
    int64_t min, i;
    if (n == 0) return 0;
    min = x[0];
    for (i = 0; i < n; i++) if (x[i] < min) min = x[i];
    return min;
}

/* go/group.go:374-382: a single wrap, not a modulo */
void mnwo_bound(int64_t *x, int64_t n, int64_t min, int64_t pixels) {
    int64_t i;
    for (i = 0; i < n; i++) {
        if (x[i] < min) x[i] = (int64_t)((uint64_t)x[i] + (uint64_t)pixels);
        else if (x[i] >= (int64_t)((uint64_t)min + (uint64_t)pixels))
            x[i] = (int64_t)((uint64_t)x[i] - (uint64_t)pixels);
    }
}

/* go/group.go:412-420 (int64 arithmetic wraps in Go; done in uint64 here) */
int64_t mnwo_periodic_distance(int64_t x, int64_t x0, int64_t pixels) {
    int64_t d = (int64_t)((uint64_t)x - (uint64_t)x0);
    if (d >= 0) {
        if (d > (int64_t)((uint64_t)pixels - (uint64_t)d))
            return (int64_t)((uint64_t)d - (uint64_t)pixels);
    } else {
        if (d < (int64_t)(0 - ((uint64_t)d + (uint64_t)pixels)))
            return (int64_t)((uint64_t)pixels + (uint64_t)d);
    }
    return d;
}

/* go/group.go:384-409 */
int64_t mnwo_periodic_min(const int64_t *x, int64_t n, int64_t pixels) {
    int64_t x0, width, i;
    if (n == 0) return 0;
    x0 = x[0];
    width = 1;
    for (i = 0; i < n; i++) {
        int64_t xi = x[i], d0, d1;
        int64_t x1 = (int64_t)((uint64_t)x0 + (uint64_t)width - 1);
        if (x1 >= pixels) x1 = (int64_t)((uint64_t)x1 - (uint64_t)pixels);

        d0 = mnwo_periodic_distance(xi, x0, pixels);
        d1 = mnwo_periodic_distance(xi, x1, pixels);

        if (d0 > 0 && d1 < 0) continue;

        if (d1 > (int64_t)(0 - (uint64_t)d0)) {
            width = (int64_t)((uint64_t)width + (uint64_t)d1);
        } else {
            x0 = (int64_t)((uint64_t)x0 + (uint64_t)d0);
            if (x0 < 0) x0 = (int64_t)((uint64_t)x0 + (uint64_t)pixels);
            width = (int64_t)((uint64_t)width - (uint64_t)d0);
        }
        if (width > pixels / 2) return 0;
    }
    return x0;
}

/* go/writer.go:72-75 */
int64_t mnwo_float_group_pixels(float low, float high, float dx) {
    volatile float span = high - low;
    volatile float r = span / dx;
    return go_int64_of_f64(ceil((double)r));
}

/* go/group.go:242-255 */
int mnwo_int_encode(const int64_t *x, int64_t n, int64_t *min, int64_t *bits,
                    uint8_t *out, int64_t *out_len, uint64_t *scratch, int literal) {
    int64_t m = mnwo_int64_min(x, n), i, b, nbytes;
    for (i = 0; i < n; i++) scratch[i] = (uint64_t)x[i] - (uint64_t)m;   /* group.go:247 */
    b = mnwo_bits(scratch, n);                                           /* group.go:248 */
    *min = m;
    *bits = b;
    if (b < 0 || b > 64) { *out_len = 0; return -1; }                    /* Go would panic */
    nbytes = mnwo_array_bytes(b, n);
    *out_len = (b == 0) ? 0 : nbytes;                                    /* bit.go:162 */
    if (b == 0) return 0;
    return literal ? mnwo_pack((int)b, scratch, n, out, nbytes)
                   : mnwo_pack_fast((int)b, scratch, n, out, nbytes);
}

/* go/group.go:257-263 (+ bit.go:169-181) */
int mnwo_int_decode(const uint8_t *in, int64_t min, int64_t bits, int64_t n,
                    int64_t *out, uint64_t *scratch, int literal) {
    int64_t i;
    int rc = 0;
    if (bits == 0) { for (i = 0; i < n; i++) scratch[i] = 0; }           /* bit.go:171-174 */
    else rc = literal ? mnwo_unpack((int)bits, in, n, scratch)
                      : mnwo_unpack_fast((int)bits, in, n, scratch);
    if (rc) return rc;
    for (i = 0; i < n; i++) out[i] = (int64_t)((uint64_t)min + scratch[i]);
    return 0;
}

/* go/group.go:316-320 */
void mnwo_quantize(const float *x, int64_t n, float low, float high, int64_t pixels,
                   int64_t *q) {
    volatile float span = high - low;
    volatile float dx = span / (float)pixels;
    int64_t i;
    for (i = 0; i < n; i++) {
        volatile float t = x[i] - low;
        volatile float r = t / dx;
        q[i] = go_int64_of_f64(floor((double)r));
    }
}

/* go/group.go:312-327 */
int mnwo_float_encode(const float *x, int64_t n, float low, float high, int64_t pixels,
                      int periodic, int64_t *min, int64_t *bits, uint8_t *out,
                      int64_t *out_len, int64_t *qbuf, uint64_t *scratch, int literal) {
    mnwo_quantize(x, n, low, high, pixels, qbuf);
    if (periodic == 1) {
        int64_t pm = mnwo_periodic_min(qbuf, n, pixels);
        mnwo_bound(qbuf, n, pm, pixels);
    }
    return mnwo_int_encode(qbuf, n, min, bits, out, out_len, scratch, literal);
}

int mnwo_float_decode_q(const uint8_t *in, int64_t min, int64_t bits, int64_t pixels,
                        int periodic, int64_t n, int64_t *q, uint64_t *scratch, int literal) {
    int rc = mnwo_int_decode(in, min, bits, n, q, scratch, literal);
    if (rc) return rc;
    if (periodic == 1) mnwo_bound(q, n, 0, pixels);                      /* group.go:303 */
    return 0;
}

/* go/group.go:299-310 */
int mnwo_float_decode(const uint8_t *in, int64_t min, int64_t bits, float low, float high,
                      int64_t pixels, int periodic, const double *u, int64_t n, float *out,
                      int64_t *qbuf, uint64_t *scratch, int literal) {
    volatile float L = high - low;
    volatile float dx = L / (float)pixels;
    int64_t i;
    int rc = mnwo_float_decode_q(in, min, bits, pixels, periodic, n, qbuf, scratch, literal);
    if (rc) return rc;
    for (i = 0; i < n; i++) {
        double ui = u ? u[i] : 0.0;
        volatile float t = (float)((double)qbuf[i] + ui);
        volatile float p = dx * t;
        out[i] = p + low;
    }
    return 0;
}

/* The build's dither generator (DESIGN.md "Dither"): a per-(seed, block, i>>32) key from
 * one splitmix64 finalizer, then a 32-bit xorshift-multiply hash of the low index word. */
static uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

double mnwo_dither_u01(uint64_t seed, uint64_t block, uint64_t i) {
    uint64_t key = mix64(seed + 0x9E3779B97F4A7C15ull * (block + 1) +
                         0xD1B54A32D192ED03ull * (i >> 32));
    uint32_t ka = (uint32_t)key, kb = (uint32_t)(key >> 32);
    uint32_t x = (uint32_t)i ^ ka;
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    x ^= kb;
    return (double)x * (1.0 / 4294967296.0);
}

void mnwo_dither_fill(uint64_t seed, uint64_t block, int64_t n, double *u) {
    int64_t i;
    for (i = 0; i < n; i++) u[i] = mnwo_dither_u01(seed, block, (uint64_t)i);
}

/* go/group.go:216-225 write(x) inside intGroup.writeTail */
int64_t mnwo_tail_array_encode(const int64_t *x, int64_t n, uint8_t *out) {
    int64_t min = mnwo_int64_min(x, n), bits, i, nbytes, rc = -1;
    uint64_t *tmp = (uint64_t *)malloc((size_t)(n > 0 ? n : 1) * 8);
    if (!tmp) return -1;
    for (i = 0; i < n; i++) tmp[i] = (uint64_t)x[i] - (uint64_t)min;
    bits = mnwo_bits(tmp, n);
    memcpy(out, &min, 8);                                   /* binaryWrite(f, min) */
    memcpy(out + 8, &bits, 8);                              /* binaryWrite(f, int64(bits)) */
    if (bits == 0) { rc = 16; }                             /* bit.go:162 */
    else if (bits > 0 && bits <= 64) {
        nbytes = mnwo_array_bytes(bits, n);
        if (mnwo_pack((int)bits, tmp, n, out + 16, nbytes) == 0) rc = 16 + nbytes;
    }
    free(tmp);
    return rc;
}

/* go/group.go:190-198 read() inside newIntGroupFromTail */
int64_t mnwo_tail_array_decode(const uint8_t *in, int64_t blocks, int64_t *out) {
    int64_t min, bits, i, rc = -1;
    uint64_t *tmp;
    memcpy(&min, in, 8);
    memcpy(&bits, in + 8, 8);
    if (bits == 0) { for (i = 0; i < blocks; i++) out[i] = min; return 16; }
    if (bits < 0 || bits > 64) return -1;
    tmp = (uint64_t *)malloc((size_t)(blocks > 0 ? blocks : 1) * 8);
    if (!tmp) return -1;
    if (mnwo_unpack((int)bits, in + 16, blocks, tmp) == 0) {
        for (i = 0; i < blocks; i++) out[i] = (int64_t)((uint64_t)min + tmp[i]);
        rc = 16 + mnwo_array_bytes(bits, blocks);
    }
    free(tmp);
    return rc;
}

/* ---- minp sub-cells ------------------------------------------------------------------------ */
/* go/minp/minp.go:246-266 */
void mnwo_get_sub_cell(const float *x, float *buf0, float *buf1, float *buf2, int64_t sc,
                       int64_t sub_cells, int64_t n_sub) {
    float *buf[3];
    int64_t n_file = n_sub * sub_cells;
    int64_t sx = sc % sub_cells, sy = (sc / sub_cells) % sub_cells, sz = sc / (sub_cells * sub_cells);
    int64_t ix0 = n_sub * sx, iy0 = n_sub * sy, iz0 = n_sub * sz;
    int64_t j = 0, jx, jy, jz;
    int k;
    buf[0] = buf0; buf[1] = buf1; buf[2] = buf2;
    for (jz = 0; jz < n_sub; jz++)
        for (jy = 0; jy < n_sub; jy++)
            for (jx = 0; jx < n_sub; jx++) {
                int64_t ix = jx + ix0, iy = jy + iy0, iz = jz + iz0;
                int64_t i = ix + iy * n_file + iz * n_file * n_file;
                for (k = 0; k < 3; k++) buf[k][j] = x[3 * i + k];
                j++;
            }
}

/* go/minp/minp.go:272-288 */
void mnwo_set_sub_cell(float *x, const float *buf0, const float *buf1, const float *buf2, int64_t sc,
                       int64_t sub_cells, int64_t n_sub) {
    const float *buf[3];
    int64_t n_file = n_sub * sub_cells;
    int64_t sx = sc % sub_cells, sy = (sc / sub_cells) % sub_cells, sz = sc / (sub_cells * sub_cells);
    int64_t ix0 = n_sub * sx, iy0 = n_sub * sy, iz0 = n_sub * sz;
    int64_t j = 0, jx, jy, jz;
    int k;
    buf[0] = buf0; buf[1] = buf1; buf[2] = buf2;
    for (jz = 0; jz < n_sub; jz++)
        for (jy = 0; jy < n_sub; jy++)
            for (jx = 0; jx < n_sub; jx++) {
                int64_t ix = jx + ix0, iy = jy + iy0, iz = jz + iz0;
                int64_t i = ix + iy * n_file + iz * n_file * n_file;
                for (k = 0; k < 3; k++) x[3 * i + k] = buf[k][j];
                j++;
            }
}

/* go/minp/minp.go:290-300 */
void mnwo_bounds3(const float *vec, int64_t n, float *min3, float *max3) {
    int64_t i;
    int k;
    for (k = 0; k < 3; k++) { min3[k] = vec[k]; max3[k] = vec[k]; }
    for (i = 0; i < n; i++)
        for (k = 0; k < 3; k++) {
            if (vec[3 * i + k] < min3[k]) min3[k] = vec[3 * i + k];
            if (vec[3 * i + k] > max3[k]) max3[k] = vec[3 * i + k];
        }
}

/* go/minp/minp.go:188-196 */
void mnwo_periodic_fix(float *x, int64_t n, float L) {
    int64_t i;
    for (i = 0; i < n; i++) {
        if (x[i] < 0) x[i] = x[i] + L;
        else if (x[i] >= L) x[i] = x[i] - L;
    }
}

/* Go src/math/nextafter.go Nextafter32: NaN if either is NaN; x if x == y; smallest subnormal with
 * y's sign if x == 0; else step the bit pattern towards y. */
float mnwo_nextafter32(float x, float y) {
    uint32_t u;
    if (x != x || y != y) return x != x ? x : y;
    if (x == y) return x;
    if (x == 0) { u = 1; memcpy(&x, &u, 4); return y > 0 ? x : -x; }
    memcpy(&u, &x, 4);
    if ((y > x) == (x > 0)) u++; else u--;
    memcpy(&x, &u, 4);
    return x;
}

/* ---- minh column transforms ------------------------------------------------------------------- */
/* src/math/log10.go: log10(x) = math.Log2(x) * (Ln2 / Ln10); the constant expression is exact, rounded once */
double mnwo_go_log10(double x) {
    static const double Ln2OverLn10 = 0.30102999566398119521;
    return mnwo_go_log2(x) * Ln2OverLn10;
}

/* src/math/ldexp.go */
static double go_ldexp(double frac, int exp) {
    uint64_t u;
    int e = 0;
    double m = 1;
    if (frac == 0 || isinf(frac) || isnan(frac)) return frac;
    memcpy(&u, &frac, 8);
    if (((u >> 52) & 0x7ff) == 0) { frac *= 4503599627370496.0; e = -52; memcpy(&u, &frac, 8); }   /* normalize */
    exp += e;
    exp += (int)((u >> 52) & 0x7ff) - 1023;
    if (exp < -1075) return copysign(0.0, frac);
    if (exp > 1023) return frac < 0 ? -INFINITY : INFINITY;
    if (exp < -1022) { exp += 53; m = 1.0 / 9007199254740992.0; }
    u &= ~(0x7ffull << 52);
    u |= (uint64_t)(exp + 1023) << 52;
    memcpy(&frac, &u, 8);
    return m * frac;
}

/* src/math/exp.go (pure Go; FreeBSD e_exp.c) */
static double go_exp(double x) {
    static const double Ln2Hi = 6.93147180369123816490e-01, Ln2Lo = 1.90821492927058770002e-10;
    static const double Log2e = 1.44269504088896338700e+00;
    static const double Overflow = 7.09782712893383973096e+02, Underflow = -7.45133219101941108420e+02;
    static const double NearZero = 1.0 / (1 << 28);
    static const double P1 = 1.66666666666666657415e-01, P2 = -2.77777777770155933842e-03;
    static const double P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06;
    static const double P5 = 4.13813679705723846039e-08;
    int k = 0;
    double hi, lo, r, t, c, y;
    if (isnan(x) || (isinf(x) && x > 0)) return x;
    if (isinf(x)) return 0;
    if (x > Overflow) return INFINITY;
    if (x < Underflow) return 0;
    if (-NearZero < x && x < NearZero) return 1 + x;
    if (x < 0) k = (int)(Log2e * x - 0.5);
    else if (x > 0) k = (int)(Log2e * x + 0.5);
    hi = x - (double)k * Ln2Hi;
    lo = (double)k * Ln2Lo;
    r = hi - lo;
    t = r * r;
    c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    y = 1 - ((lo - (r * c) / (2 - c)) - hi);
    return go_ldexp(y, k);
}

/* src/math/modf.go */
static double go_modf(double f, double *frac) {
    uint64_t x;
    unsigned e;
    double ip;
    if (f < 1) {
        if (f < 0) { ip = go_modf(-f, frac); *frac = -*frac; return -ip; }
        if (f == 0) { *frac = f; return f; }
        *frac = f;
        return 0;
    }
    memcpy(&x, &f, 8);
    e = (unsigned)((x >> 52) & 0x7ff) - 1023;
    if (e < 64 - 12) x &= ~((1ull << (64 - 12 - e)) - 1);
    memcpy(&ip, &x, 8);
    *frac = f - ip;
    return ip;
}

/* src/math/pow.go */
double mnwo_go_pow(double x, double y) {
    double yi, yf, a1 = 1.0, x1;
    int ae = 0, xe;
    int64_t i;
    if (y == 0 || x == 1) return 1;
    if (y == 1) return x;
    if (isnan(x) || isnan(y)) return NAN;
    if (x == 0) {
        if (y < 0) { double t; if (go_modf(y, &t), t == 0 && ((int64_t)y & 1)) return copysign(INFINITY, x); return INFINITY; }
        if (y > 0) { double t; if (go_modf(y, &t), t == 0 && ((int64_t)y & 1)) return x; return 0; }
    }
    if (isinf(y)) {
        if (x == -1) return 1;
        if ((fabs(x) < 1) == (y > 0)) return 0;
        return INFINITY;
    }
    if (isinf(x)) {
        if (x < 0) { double t; int odd = (go_modf(y, &t), t == 0 && ((int64_t)y & 1)); if (y < 0) return odd ? -0.0 : 0.0; return odd ? -INFINITY : INFINITY; }
        return y < 0 ? 0 : INFINITY;
    }
    if (y == 0.5) return sqrt(x);
    if (y == -0.5) return 1 / sqrt(x);
    yi = go_modf(fabs(y), &yf);
    if (yf != 0 && x < 0) return NAN;
    if (yi >= 9223372036854775808.0) {
        if (x == -1) return 1;
        if ((fabs(x) < 1) == (y > 0)) return 0;
        return INFINITY;
    }
    if (yf != 0) {
        if (yf > 0.5) { yf--; yi++; }
        a1 = go_exp(yf * go_log(x));
    }
    x1 = go_frexp(x, &xe);
    for (i = (int64_t)yi; i != 0; i >>= 1) {
        if (xe < -(1 << 12) || (1 << 12) < xe) { ae += xe; break; }
        if (i & 1) { a1 *= x1; ae += xe; }
        x1 *= x1;
        xe <<= 1;
        if (x1 < .5) { x1 += x1; xe--; }
    }
    if (y < 0) { a1 = 1 / a1; ae = -ae; }
    return go_ldexp(a1, ae);
}

/* go/minh/minh.go:139-149 */
void mnwo_process_float_group(float *buf, int64_t n, int log, float low, float high) {
    int64_t j;
    for (j = 0; j < n; j++) {
        if (log != 0) buf[j] = (float)mnwo_go_log10((double)buf[j]);
        if (buf[j] < low) buf[j] = low;
        if (buf[j] >= high) buf[j] = mnwo_nextafter32(high, -INFINITY);
    }
}

/* go/minh/minh.go:315-319 */
void mnwo_unlog(float *buf, int64_t n) {
    int64_t i;
    for (i = 0; i < n; i++) buf[i] = (float)mnwo_go_pow(10, (double)buf[i]);
}
