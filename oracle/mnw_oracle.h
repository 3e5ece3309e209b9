/*
 * mnw_oracle.h -- CPU restatement of minnow's column quantize + bit-pack path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / reported CPU baseline.  The product path
 * (minnow_amd/csrc, include/minnow_hip.h) never links, includes or calls it.
 *
 * Every function follows the Go text of the reference and cites it
 * (paths relative to the reference checkout):
 *   go/bit/bit.go     -- PrecisionNeeded, ArrayBytes, BufferedArray, Slice, ArrayBuffer.Bits
 *   go/group.go       -- intGroup / floatGroup writeData, readData, writeTail,
 *                        int64Min, bound, periodicMin, periodicDistance
 *   go/writer.go      -- Writer.FloatGroup parameter derivation
 *
 * Pinning: see oracle/README.md.  The byte-level codec (pack, unpack, periodic_min,
 * int-group streams, group tails) is pinned against outputs of the reference's own
 * Python/Cython mirror (python/cy_bit.pyx, python/minnow.py) committed as
 * tests/golden/ fixtures, and against the literal vectors of the reference's Go tests.
 * Two things are "parity unpinned" because the Go toolchain is absent here and the
 * Python mirror differs from the Go text: (1) PrecisionNeeded for max >= 2^49, which
 * depends on Go's math.Log2 bit-for-bit (restated below from the published Go/FreeBSD
 * algorithm), (2) int64(math.Floor(x)) for NaN / |x| >= 2^63 (amd64 behaviour assumed).
 */
#ifndef MNW_ORACLE_H
#define MNW_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* go/bit/bit.go:19-21.  Go's math.Log2/Log/Frexp restated (no libm log2). */
int64_t mnwo_precision_needed(uint64_t max);
/* the same through Go's math.Log2 restatement only (exposed for tests) */
double mnwo_go_log2(double x);
/* go/bit/bit.go:23-25 */
int64_t mnwo_array_bytes(int64_t bits, int64_t length);

/* go/bit/bit.go:84-134 BufferedArray.  Literal byte-wise algorithm.
 * Returns 0, or -1 if bits > 64, -2 if b_len != ArrayBytes(bits, n) (the Go panics). */
int mnwo_pack(int bits, const uint64_t *x, int64_t n, uint8_t *b, int64_t b_len);
/* go/bit/bit.go:29-82 (*Array).Slice.  Literal byte-wise algorithm. */
int mnwo_unpack(int bits, const uint8_t *data, int64_t n, uint64_t *out);
/* Word-wise pack/unpack with identical output; used to check big inputs quickly and
 * cross-checked against the literal versions in tests. */
int mnwo_pack_fast(int bits, const uint64_t *x, int64_t n, uint8_t *b, int64_t b_len);
int mnwo_unpack_fast(int bits, const uint8_t *data, int64_t n, uint64_t *out);

/* go/bit/bit.go:151-159 ArrayBuffer.Bits */
int64_t mnwo_bits(const uint64_t *x, int64_t n);

/* go/group.go:350-357, 374-382, 384-409, 412-420 */
int64_t mnwo_int64_min(const int64_t *x, int64_t n);
void mnwo_bound(int64_t *x, int64_t n, int64_t min, int64_t pixels);
int64_t mnwo_periodic_min(const int64_t *x, int64_t n, int64_t pixels);
int64_t mnwo_periodic_distance(int64_t x, int64_t x0, int64_t pixels);

/* go/writer.go:72-75: pixels = int64(ceil(float64((lim1-lim0)/dx))) in float32 */
int64_t mnwo_float_group_pixels(float low, float high, float dx);

/* go/group.go:242-255 intGroup.writeData up to (not including) f.Write.
 * out must hold 8*n bytes (worst case); *out_len = ArrayBytes(bits, n)
 * (0 bytes when bits == 0, bit.go:162).  scratch: n uint64 (the ArrayBuffer). */
int mnwo_int_encode(const int64_t *x, int64_t n, int64_t *min, int64_t *bits,
                    uint8_t *out, int64_t *out_len, uint64_t *scratch, int literal);
/* go/group.go:257-263 intGroup.readData after f.Read */
int mnwo_int_decode(const uint8_t *in, int64_t min, int64_t bits, int64_t n,
                    int64_t *out, uint64_t *scratch, int literal);

/* go/group.go:312-327 floatGroup.writeData up to f.Write.
 * qbuf: n int64 (g.buf), scratch: n uint64. */
int mnwo_float_encode(const float *x, int64_t n, float low, float high, int64_t pixels,
                      int periodic, int64_t *min, int64_t *bits, uint8_t *out,
                      int64_t *out_len, int64_t *qbuf, uint64_t *scratch, int literal);
/* quantize step alone, go/group.go:316-320 */
void mnwo_quantize(const float *x, int64_t n, float low, float high, int64_t pixels,
                   int64_t *q);

/* go/group.go:299-310 floatGroup.readData after f.Read.  The reference draws the
 * dither from Go's global math/rand (group.go:308), which no other program can
 * reproduce; the oracle takes the uniform deviates as an explicit array u[i] in [0,1)
 * (u == NULL means 0.0 everywhere), or generates them with mnwo_dither_u01. */
int mnwo_float_decode(const uint8_t *in, int64_t min, int64_t bits, float low, float high,
                      int64_t pixels, int periodic, const double *u, int64_t n, float *out,
                      int64_t *qbuf, uint64_t *scratch, int literal);
/* decoded integer stream (before dequantize), group.go:301-303 */
int mnwo_float_decode_q(const uint8_t *in, int64_t min, int64_t bits, int64_t pixels,
                        int periodic, int64_t n, int64_t *q, uint64_t *scratch, int literal);

/* The build's counter-based dither (NOT the reference's math/rand): restated here
 * independently from DESIGN.md's definition so the float outputs of the HIP decode
 * can be compared bit-for-bit. */
double mnwo_dither_u01(uint64_t seed, uint64_t block, uint64_t i);
void mnwo_dither_fill(uint64_t seed, uint64_t block, int64_t n, double *u);

/* go/group.go:215-232 intGroup.writeTail's inner write(x): min, bits, packed stream.
 * out must hold 16 + 8*n bytes. Returns bytes written. */
int64_t mnwo_tail_array_encode(const int64_t *x, int64_t n, uint8_t *out);
/* go/group.go:190-198 read(): returns bytes consumed */
int64_t mnwo_tail_array_decode(const uint8_t *in, int64_t blocks, int64_t *out);

/* ---- minp sub-cells (go/minp/minp.go) ---------------------------------------------------
 * getSubCell / setSubCell (minp.go:246-288): x is [n_file^3][3] floats, buf k holds n_sub^3. */
void mnwo_get_sub_cell(const float *x, float *buf0, float *buf1, float *buf2, int64_t sc,
                       int64_t sub_cells, int64_t n_sub);
void mnwo_set_sub_cell(float *x, const float *buf0, const float *buf1, const float *buf2, int64_t sc,
                       int64_t sub_cells, int64_t n_sub);
/* bounds (minp.go:290-300) */
void mnwo_bounds3(const float *vec, int64_t n, float *min3, float *max3);
/* the reader's periodic fix-up (minp.go:188-196), in place */
void mnwo_periodic_fix(float *x, int64_t n, float L);
/* math.Nextafter32(x, y) (Go src/math/nextafter.go), used by minp.go:94 */
float mnwo_nextafter32(float x, float y);

/* ---- minh column transforms (go/minh/minh.go) ----------------------------------------------
 * processFloatGroup (minh.go:139-149), in place; Go's math.Log10 restated (log10.go, log.go, frexp.go). */
void mnwo_process_float_group(float *buf, int64_t n, int log, float low, float high);
/* the reader's float32(math.Pow(10, float64(x))) (minh.go:315-319), in place; Go's math.Pow / Exp / Ldexp /
 * Modf restated from their portable definitions (pow.go, exp.go, ldexp.go, modf.go). */
void mnwo_unlog(float *buf, int64_t n);
double mnwo_go_log10(double x);
double mnwo_go_pow(double x, double y);

#ifdef __cplusplus
}
#endif
#endif
