/*
 * minnow_hip.h -- C ABI of libminnow_hip.so: minnow's column quantize + bit-pack hot path
 * on MI355X (gfx950), hand-written HIP kernels behind plain pointers and sizes.
 *
 * This is the drop-in boundary for the reference's Go code: a cgo shim replaces the BODIES of
 *   intGroup.writeData / readData      go/group.go:242-255, 257-263
 *   floatGroup.writeData / readData    go/group.go:312-327, 299-310
 *   bit.BufferedArray / (*Array).Slice go/bit/bit.go:84-134, 29-82
 * up to (not including) f.Write / f.Read; file I/O, the per-group mins/bits slices and
 * blockIndex.addBlock stay in Go.  INTEGRATION.md shows the binding.  Every entry point names the
 * reference function it replaces.  All sizes are in elements unless they end in _len/_bytes/cap.
 *
 * Error convention: the reference panics on every error (go/writer.go:34,91-95; bit.go:31,86,90);
 * a C ABI cannot, so every call returns an int status (0 = MNW_OK) and the shim re-panics.
 * There is NO CPU fallback: without a usable HIP device every compute entry point fails with
 * MNW_E_NOGPU.
 */
#ifndef MINNOW_HIP_H
#define MINNOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MNW_ABI_VERSION 5   /* 2: xform field in mnw_enc_job / mnw_dec_job; 3: mnw_counters_get; 4: MNW_K_COUNT 21 (k_encode_reg_i64, k_encode_reg_vec3); 5: mnw_copy_file_to_device */

/* ---- status codes ---------------------------------------------------------------------- */
#define MNW_OK            0
#define MNW_E_ARG        (-1)  /* null pointer, negative length, bits > 64 (bit.go:85-87) */
#define MNW_E_BUFLEN     (-2)  /* buffer length != ArrayBytes(bits, n) (bit.go:88-92) / out too short (bit.go:30-33) */
#define MNW_E_NOGPU      (-3)  /* no HIP device: the product path has no CPU fallback */
#define MNW_E_HIP        (-4)  /* HIP runtime error; see mnw_last_error() */
#define MNW_E_CAPACITY   (-5)  /* packed stream does not fit the caller's out capacity */
#define MNW_E_DOMAIN     (-6)  /* width outside 0..64: max-min == 2^64-1 makes Go's PrecisionNeeded
                                  return int(-Inf) (bit.go:19-21) and the reference panic later */
#define MNW_E_NOMEM      (-7)
#define MNW_E_IO         (-8)  /* mnw_copy_file_to_device: pread failed or the file ends inside a requested range; errno text in mnw_last_error() */

/* group-type constants of the reference (go/group.go:11-24); only these two are codec paths */
#define MNW_INT_GROUP    10
#define MNW_FLOAT_GROUP  11

typedef struct mnw_ctx mnw_ctx;       /* one per Writer/Reader; not thread-safe, like the reference */
typedef void *mnw_stream_t;           /* a hipStream_t; NULL = the context's own stream, a non-blocking stream
                                         that is NOT ordered with the legacy default stream: buffers produced
                                         on other streams must be complete before they are passed in */

/* ---- context --------------------------------------------------------------------------- */
int mnw_ctx_create(int device, mnw_ctx **ctx);
int mnw_ctx_destroy(mnw_ctx *ctx);
const char *mnw_strerror(int status);
const char *mnw_last_error(const mnw_ctx *ctx);
int mnw_abi_version(void);
/* number of HIP devices visible (0 when none / no driver); never initialises a context */
int mnw_device_count(void);

/* ---- scalars (host arithmetic, no GPU needed) -------------------------------------------- */
/* bit.PrecisionNeeded, go/bit/bit.go:19-21: int(ceil(log2(float64(max+1)))) with Go's math.Log2 */
int64_t mnw_precision_needed(uint64_t max);
/* bit.ArrayBytes, go/bit/bit.go:23-25 */
int64_t mnw_array_bytes(int64_t bits, int64_t length);
/* pixels of Writer.FloatGroup, go/writer.go:72-75 (float32 subtract and divide, float64 ceil) */
int64_t mnw_float_group_pixels(float low, float high, float dx);

/* ---- host-pointer, synchronous forms: what the cgo shim binds ---------------------------------
 * Inputs/outputs are ordinary host memory (Go slices); the call stages them through the context's
 * pinned buffers, runs the kernels and returns when the result is in the caller's memory.  No
 * pointer is retained after return (cgo rule). */

/* bit.BufferedArray(bits, x, b), bit.go:84-134: packs x[i] & mask at `bits` bits, LSB first.
 * out_len must equal ArrayBytes(bits, n). */
int mnw_pack_u64(mnw_ctx *ctx, const uint64_t *x, int64_t n, int bits, uint8_t *out, int64_t out_len);
/* (*bit.Array).Slice(out), bit.go:29-82.  in_len must be >= ArrayBytes(bits, n). */
int mnw_unpack_u64(mnw_ctx *ctx, const uint8_t *in, int64_t in_len, int bits, int64_t n, uint64_t *out);

/* bit.ArrayBuffer.Bits(x), bit.go:151-159: PrecisionNeeded(max(x)); 0 when x is empty. */
int mnw_bits_u64(mnw_ctx *ctx, const uint64_t *x, int64_t n, int64_t *bits);

/* intGroup.writeData, group.go:242-250: min, buf = x-min, bits = Bits(buf), pack.
 * *out_len = ArrayBytes(bits, n), 0 when bits == 0 (bit.go:162).  cap = capacity of out. */
int mnw_int_encode(mnw_ctx *ctx, const int64_t *x, int64_t n, int64_t *min, int64_t *bits,
                   uint8_t *out, int64_t cap, int64_t *out_len);
/* intGroup.readData, group.go:257-263 (bits == 0 -> all min, bit.go:171-174) */
int mnw_int_decode(mnw_ctx *ctx, const uint8_t *in, int64_t in_len, int64_t min, int64_t bits,
                   int64_t n, int64_t *out);

/* floatGroup.writeData, group.go:312-327: quantize, periodicMin + bound when periodic == 1,
 * then the intGroup path. */
int mnw_float_encode(mnw_ctx *ctx, const float *x, int64_t n, float low, float high, int64_t pixels,
                     int periodic, int64_t *min, int64_t *bits, uint8_t *out, int64_t cap,
                     int64_t *out_len);
/* floatGroup.readData, group.go:299-310.  The reference dithers with the global math/rand stream
 * (group.go:308), which no other program can reproduce; here the uniform deviate of element i is a
 * counter-based function of (seed, block, i) (DESIGN.md "Dither"), so decodes are reproducible. */
int mnw_float_decode(mnw_ctx *ctx, const uint8_t *in, int64_t in_len, int64_t min, int64_t bits,
                     float low, float high, int64_t pixels, int periodic, uint64_t seed,
                     uint64_t block, int64_t n, float *out);

/* ---- device-pointer, batched forms: per-column blocks as the unit of GPU dispatch --------------
 * src/dst are DEVICE pointers (hipMalloc / torch tensors).  One call enqueues every block of the
 * batch on `stream`; blocks are independent (group.go:242-255 depends on one block only).  The
 * packed stream of block k is written at dst (16-byte aligned, capacity cap bytes); its exact
 * length comes back in meta[k].nbytes, which is what blockIndex.addBlock receives
 * (block_index.go:16-23). */
#define MNW_XF_LOG10 1   /* Column.Log != 0 */
#define MNW_XF_CLAMP 2   /* x < Low -> Low, x >= High -> Nextafter32(High, -Inf): always set by minh */
typedef struct mnw_enc_job {
    const void *src;     /* device: float32[n] (MNW_FLOAT_GROUP) or int64[n] (MNW_INT_GROUP) */
    void *dst;           /* device: packed stream, 16-byte aligned */
    int64_t n;           /* elements in the block */
    int64_t cap;         /* bytes available at dst */
    int64_t pixels;      /* float groups: floatGroup.pixels */
    float low, high;     /* float groups: floatGroup.low / high */
    int32_t kind;        /* MNW_INT_GROUP or MNW_FLOAT_GROUP */
    int32_t periodic;    /* float groups: floatGroup.periodic (1 from Writer.FloatGroup, writer.go:74) */
    int32_t xform;       /* float groups: MNW_XF_* -- minh.processFloatGroup (go/minh/minh.go:139-149) applied to every
                          * value before anything else looks at it, limits = low / high; 0 for plain minnow groups */
    int32_t reserved;
} mnw_enc_job;

typedef struct mnw_dec_job {
    const void *src;     /* device: packed stream, ANY byte alignment (file offsets are byte-granular) */
    void *dst;           /* device: float32[n] or int64[n] */
    int64_t n;
    int64_t src_len;     /* bytes available at src; must be >= ArrayBytes(bits, n) */
    int64_t min;         /* intGroup.mins[b] */
    int64_t bits;        /* intGroup.bits[b] */
    int64_t pixels;
    float low, high;
    int32_t kind;
    int32_t periodic;
    uint64_t seed;       /* dither key (float groups) */
    uint64_t block;      /* dither key: block index */
    int32_t xform;       /* float groups: MNW_XF_LOG10 = float32(math.Pow(10, float64(x))) on every decoded value
                          * (minh.Reader.FloatBlock, go/minh/minh.go:315-319); 0 otherwise */
    int32_t reserved;
} mnw_dec_job;

typedef struct mnw_block_meta {
    int64_t min;         /* what intGroup.writeData appends to g.mins (group.go:251) */
    int64_t bits;        /* ... and to g.bits (group.go:252) */
    int64_t nbytes;      /* bytes written at dst = ArrayBytes(bits, n), 0 when bits == 0 */
    int64_t status;      /* MNW_OK, MNW_E_CAPACITY or MNW_E_DOMAIN for this block */
} mnw_block_meta;

/* Encode njobs blocks.  jobs and meta are HOST arrays.  Synchronous with respect to meta: returns
 * after the kernels have finished and meta[] is filled (the widths are data dependent, and the
 * caller needs them for its block index).  Returns the first non-OK block status, else MNW_OK. */
int mnw_encode_blocks(mnw_ctx *ctx, mnw_stream_t stream, const mnw_enc_job *jobs, int njobs,
                      mnw_block_meta *meta);
/* Split form for pipelining: enqueue only (no host synchronisation) ... */
int mnw_encode_blocks_async(mnw_ctx *ctx, mnw_stream_t stream, const mnw_enc_job *jobs, int njobs);
/* ... then collect the metadata of the most recent async batch (synchronises the stream). */
int mnw_encode_blocks_finish(mnw_ctx *ctx, mnw_stream_t stream, mnw_block_meta *meta);

/* Decode njobs blocks; enqueue only (widths are known from the group tail, group.go:186-213).
 * The caller synchronises `stream`. */
int mnw_decode_blocks(mnw_ctx *ctx, mnw_stream_t stream, const mnw_dec_job *jobs, int njobs);

/* ---- minp vector blocks (go/minp/minp.go) ------------------------------------------------------
 * A minp file stores the particles of one file cell, an array-of-structures grid vec[n_file^3][3], as
 * 3 * sub_cells^3 FloatGroup blocks: for each component k and sub-cell sc the n_sub^3 values of that
 * sub-cube (n_sub = n_file / sub_cells), k-major.  Element j = jx + jy*n_sub + jz*n_sub^2 of block
 * (k, sc) is vec[i][k] with i = (jx + n_sub*sx) + (jy + n_sub*sy)*n_file + (jz + n_sub*sz)*n_file^2 and
 * sc = sx + sy*sub_cells + sz*sub_cells^2 (getSubCell / setSubCell, minp.go:246-288).
 * n_file must be a multiple of sub_cells and n_file^3 < 2^31.  All pointers except meta/low/high/pixels are
 * device pointers; vec and dst 16-byte aligned. */

/* minp.Writer.Vectors up to the file writes (minp.go:85-118): gather every sub-cell and encode block
 * (k, sc) -- a FloatGroup with limits [low[k], high[k]), pixels[k] and periodic = 1 (writer.go:74) -- to
 * dst + (k*sub_cells^3 + sc)*slot_bytes.  meta (host, 3*sub_cells^3 entries, k-major like the file) is
 * filled as by mnw_encode_blocks; synchronous in the same sense. */
int mnw_vectors_encode(mnw_ctx *ctx, mnw_stream_t stream, const float *vec, int64_t n_file, int64_t sub_cells,
                       const float low[3], const float high[3], const int64_t pixels[3],
                       uint8_t *dst, int64_t slot_bytes, mnw_block_meta *meta);
/* minp.Reader.Vectors after the file reads (minp.go:170-198): decode block (k, sc) from
 * src + (k*sub_cells^3 + sc)*slot_bytes with meta[k*sub_cells^3 + sc].{min,bits,nbytes} (dither key:
 * seed, block index first_block + k*sub_cells^3 + sc), wrap once into [0, L) if periodic (188-196),
 * scatter into vec_out (setSubCell).  Enqueue only. */
int mnw_vectors_decode(mnw_ctx *ctx, mnw_stream_t stream, const uint8_t *src, int64_t slot_bytes,
                       const mnw_block_meta *meta, int64_t n_file, int64_t sub_cells,
                       const float low[3], const float high[3], const int64_t pixels[3],
                       int periodic, float L, uint64_t seed, uint64_t first_block, float *vec_out);
/* bounds() of minp.go:290-300 over vec[n][3] (device): per-component minimum and maximum (host outputs).
 * The caller applies Nextafter32(max, 2*max) as minp.Writer.Vectors does (minp.go:92-95).  Synchronous. */
int mnw_vectors_bounds(mnw_ctx *ctx, mnw_stream_t stream, const float *vec, int64_t n, float min[3], float max[3]);
/* Host-pointer forms of the two calls above: vec / vec_out are host arrays ([][3]float32 slices), the blocks live in
 * a host buffer of 3*sub_cells^3 slots of slot_bytes (a multiple of 16) each.  Synchronous; nothing retained. */
int mnw_vectors_encode_host(mnw_ctx *ctx, const float *vec, int64_t n_file, int64_t sub_cells,
                            const float low[3], const float high[3], const int64_t pixels[3],
                            uint8_t *out, int64_t slot_bytes, mnw_block_meta *meta);
int mnw_vectors_decode_host(mnw_ctx *ctx, const uint8_t *in, int64_t slot_bytes, const mnw_block_meta *meta,
                            int64_t n_file, int64_t sub_cells, const float low[3], const float high[3],
                            const int64_t pixels[3], int periodic, float L, uint64_t seed, uint64_t first_block,
                            float *vec_out);
/* The two layout passes alone (device buffers of 3*n_file^3 floats; soa is [k][sc][j]); enqueue only. */
int mnw_subcell_gather(mnw_ctx *ctx, mnw_stream_t stream, const float *vec, int64_t n_file, int64_t sub_cells, float *soa);
int mnw_subcell_scatter(mnw_ctx *ctx, mnw_stream_t stream, const float *soa, int64_t n_file, int64_t sub_cells,
                        int periodic, float L, float *vec);

/* ---- minh column transforms (go/minh/minh.go) ---------------------------------------------------
 * minh.Writer.Block copies every FloatGroup column into a buffer, runs processFloatGroup on it and encodes
 * the buffer (minh.go:128-135).  mnw_minh_process is that copy + processFloatGroup (139-149) for up to 64
 * device columns per call: dst[i] = src[i], replaced by float32(math.Log10(float64(.))) if log != 0, then
 * clamped: below low -> low, at or above high -> Nextafter32(high, -Inf).  src == dst is allowed.  The
 * blocks are then encoded from dst with mnw_encode_blocks.  mnw_minh_unlog is the reader's
 * float32(math.Pow(10, float64(x))) (minh.go:315-319) in place, after mnw_decode_blocks.  Enqueue only.
 * Log10 / Pow are Go's portable algorithms restated (parity unpinned). */
typedef struct mnw_minh_col {
    const float *src;    /* device */
    float *dst;          /* device (process: output; unlog: the column, in place; src is ignored) */
    int64_t n;
    float low, high;     /* Column.Low / Column.High (process only) */
    int32_t log;         /* Column.Log */
    int32_t reserved;
} mnw_minh_col;
int mnw_minh_process(mnw_ctx *ctx, mnw_stream_t stream, const mnw_minh_col *cols, int ncols);
int mnw_minh_unlog(mnw_ctx *ctx, mnw_stream_t stream, const mnw_minh_col *cols, int ncols);
/* Diagnostic: compares the kernels' fast forms of float32(Log10(float64(x))) (which = 0) and float32(Pow(10, float64(y)))
 * (which = 1) with the exact restatement on the float32 bit patterns [first, first + count): out[0] = mismatches,
 * out[1] = inputs outside the fast path's domain.  Synchronous.  The GPU tests run it over all 2^32 patterns. */
int mnw_minh_selftest(mnw_ctx *ctx, int which, uint64_t first, uint64_t count, uint64_t out[2]);

/* ---- text catalogue ingest (go/text) ---------------------------------------------------------------
 * text.Reader.Block after the file read (go/text/text.go:176-197, go/text/parse.go): the block of text is split at
 * '\n', comments ("data # comment") are cut, lines holding nothing but separators are dropped, every remaining line
 * is broken into fields at runs of `sep`, and the requested fields are converted: int columns with strconv.Atoi,
 * float columns with float32(strconv.ParseFloat(s, 64)).  text is a HOST buffer (the reader's rd.buf); the outputs
 * are DEVICE arrays of `capacity_rows` elements each, ready to be handed to mnw_encode_blocks: float_out[k] receives
 * field float_cols[k], int_out[k] field int_cols[k] (0-based field numbers, at most 128 columns in all, fields < 256).
 * *n_rows = lines kept.  Every kept line must have as many fields as the first kept line has whitespace-separated
 * words (parse.go:96-103); a line that does not, or a field that Atoi / ParseFloat would reject, gives MNW_E_DOMAIN
 * (the reference panics) with the line in mnw_last_error; with no requested column nothing is checked, as in the reference
 * (parse.go:83, 132).  A field may be requested more than once, also as both types: every output is filled, as by the
 * reference (text.go:202-221).  Decimal literals of up to 19 significant digits are converted on the device (exactly: Clinger's
 * fast path, else a 128-bit power-of-ten table); what is left -- more digits, subnormal or overflowing values, inf / nan,
 * hexadecimal floats, digit separators -- is converted on the host from `text` with ParseFloat's rules (any number of them).
 * Synchronous. */
int mnw_text_parse(mnw_ctx *ctx, const uint8_t *text, int64_t nbytes, uint8_t sep, uint8_t comment,
                   const int32_t *float_cols, int n_float, float *const *float_out,
                   const int32_t *int_cols, int n_int, int64_t *const *int_out,
                   int64_t capacity_rows, int64_t *n_rows);
/* The first half of mnw_text_parse on its own: copies the block to the device, counts its lines and the lines that will
 * be kept (what the reference knows after split / uncomment / trim and sizes its outputs with: expandGeneric(out[i],
 * len(lines)), text.go:192), and leaves the copy and the counts on the device.  The mnw_text_parse call that directly
 * follows with the same text, nbytes, sep and comment (the buffer unchanged) skips the upload and the counting.
 * Synchronous. */
int mnw_text_rows(mnw_ctx *ctx, const uint8_t *text, int64_t nbytes, uint8_t sep, uint8_t comment,
                  int64_t *n_lines, int64_t *n_rows);
/* mnw_text_rows for a block whose bytes the caller has ALREADY copied to the device (text_dev; e.g. by mnw_copy_to_device on a
 * second context and thread while the previous block was being parsed): no upload here.  text is the same block in host memory
 * (the library looks at its first kept line and at the few literals the device leaves to the host); text_dev must stay
 * untouched until the mnw_text_parse calls that follow have returned. */
int mnw_text_rows_resident(mnw_ctx *ctx, const uint8_t *text, const uint8_t *text_dev, int64_t nbytes, uint8_t sep, uint8_t comment,
                           int64_t *n_lines, int64_t *n_rows);
/* Of the last mnw_text_parse: out[0] = lines (split()'s tokens), out[1] = lines kept, out[2] = 12 KB tiles of text that
 * took the general one-thread-per-line path (a comment character or an overlong line in reach) instead of the
 * one-thread-per-field path. */
int mnw_text_stats(mnw_ctx *ctx, int64_t out[3]);

/* ---- device memory for callers without a HIP binding (a cgo shim, include/minh.hpp) -------------------------
 * Plain hipMalloc / hipFree / hipMemcpy on the context's device; the copies are synchronous (they return when the
 * bytes have arrived) and ordered behind the work already enqueued on the context's own stream. */
int mnw_dev_alloc(mnw_ctx *ctx, int64_t nbytes, void **dev_ptr);
int mnw_dev_free(mnw_ctx *ctx, void *dev_ptr);
int mnw_copy_to_device(mnw_ctx *ctx, void *dev_dst, const void *host_src, int64_t nbytes);
/* n byte ranges of an open file (descriptor fd; pread, the descriptor's position is not touched) into a device buffer: range i goes to
 * dev_dst + dst_offsets[i], or, with dst_offsets NULL, the ranges are packed back to back in the order given.  The block streams a
 * reader fetches (reader.go:121-126 `f.Read` per block), here for many blocks at once: the library's host threads read straight into
 * its pinned buffers while earlier chunks are on the link.  Synchronous. */
int mnw_copy_file_to_device(mnw_ctx *ctx, void *dev_dst, int fd, int n, const int64_t *file_offsets, const int64_t *nbytes,
                            const int64_t *dst_offsets);
int mnw_copy_to_host(mnw_ctx *ctx, void *host_dst, const void *dev_src, int64_t nbytes);
/* n device -> host copies as one pipelined transfer through the library's pinned buffers (the streams of all columns of a
 * minh block, the blocks of a minp file): host_dst[i] receives nbytes[i] bytes from dev_src[i].  Synchronous. */
int mnw_copy_to_host_batch(mnw_ctx *ctx, int n, void *const *host_dst, const void *const *dev_src, const int64_t *nbytes);

/* ---- GenericCut of the reference's text_to_minh (scripts/text_to_minh.go:271-312) ----------------------------
 * Keeps, in order and in every column, the rows i with mass[i] > cutoff (ok[i] = m[i] > cutoff; filterFloat32 /
 * filterInt64).  All pointers are DEVICE pointers; mass may be the src of one of the columns; dst != src (the device
 * compacts out of place); elem_bytes is 4 ([]float32) or 8 ([]int64).  *n_kept = rows kept (GenericCut's return
 * value); synchronous with respect to it. */
typedef struct mnw_cut_col {
    const void *src;
    void *dst;
    int32_t elem_bytes;
    int32_t reserved;
} mnw_cut_col;
int mnw_generic_cut(mnw_ctx *ctx, mnw_stream_t stream, const float *mass, float cutoff, int64_t n,
                    const mnw_cut_col *cols, int ncols, int64_t *n_kept);

/* Raw bit.Array pack / unpack on device buffers (bit.go:84-134, 29-82); enqueue only.
 * out of pack must be 16-byte aligned. */
int mnw_pack_u64_dev(mnw_ctx *ctx, mnw_stream_t stream, const uint64_t *x, int64_t n, int bits,
                     uint8_t *out, int64_t cap);
int mnw_unpack_u64_dev(mnw_ctx *ctx, mnw_stream_t stream, const uint8_t *in, int64_t in_len, int bits,
                       int64_t n, uint64_t *out);

/* ---- instrumentation ------------------------------------------------------------------- */
/* Per-kernel device time of the launches made through this context since the last reset, measured
 * with HIP events on the stream each kernel was launched on (bench.py's roofline leg).  Enabling
 * timing inserts event records around every launch. */
#define MNW_K_REDUCE_F32  0
#define MNW_K_PACK_F32    1
#define MNW_K_REDUCE_I64  2
#define MNW_K_PACK_I64    3
#define MNW_K_UNPACK_F32  4
#define MNW_K_UNPACK_I64  5
#define MNW_K_ENCODE_F32  6   /* fused float encode: reduce + pack of a whole batch in one persistent launch */
#define MNW_K_ENCODE_REG_F32 7 /* single-read float encode: the block stays in registers between reduce and pack */
#define MNW_K_SUBCELL_GATHER 8  /* minp getSubCell over a whole file cell */
#define MNW_K_SUBCELL_SCATTER 9 /* minp setSubCell (+ periodic fix-up) */
#define MNW_K_BOUNDS3     10
#define MNW_K_REDUCE_VEC3 11  /* minp vectors without staging: statistics of the three blocks of every sub-cell */
#define MNW_K_PACK_VEC3   12  /* ... and their pack, straight from the array-of-structures grid */
#define MNW_K_UNPACK_VEC3 13  /* minp vectors without staging: the three streams of a sub-cell into the grid */
#define MNW_K_MINH_PROCESS 14 /* minh processFloatGroup over device columns */
#define MNW_K_MINH_UNLOG  15  /* minh reader: 10^x */
#define MNW_K_TEXT_LINES  16  /* text ingest: lines and kept lines per tile + prefix (k_text_count, k_text_scan) */
#define MNW_K_TEXT_PARSE  17  /* text ingest: fields and numbers (k_text_parse) */
#define MNW_K_CUT         18  /* GenericCut: count + prefix, scatter */
#define MNW_K_ENCODE_REG_I64 19 /* single-read int64 encode: the block is read once (k_encode_reg_i64) */
#define MNW_K_ENCODE_REG_VEC3 20 /* minp vectors with the grid read once (k_encode_reg_vec3) */
#define MNW_K_COUNT       21
typedef struct mnw_kernel_time {
    double total_ms;     /* sum of event-timed durations */
    int64_t launches;
} mnw_kernel_time;
int mnw_timing_enable(mnw_ctx *ctx, int on);
int mnw_timing_reset(mnw_ctx *ctx);
int mnw_timing_get(mnw_ctx *ctx, mnw_kernel_time *out /* [MNW_K_COUNT] */);
const char *mnw_kernel_name(int kernel_id);

/* Event counters of the context since it was created (or since mnw_counters_reset): which encode structure the
 * float batches took, and how often the single-read kernel gave up.  That kernel needs its whole grid co-resident
 * (mnw_set_option: single_read); when something else holds part of the GPU its bounded wait (50 ms) expires, the
 * host reruns the batch through the reduce + pack launches -- correct, but 50 ms and a 2x slower path, so a caller
 * that cares about throughput watches MNW_C_SINGLE_READ_ABORTS and MNW_C_BACKOFF_BATCHES (bench.py reports both per rank and
 * marks its JSON line "degraded" when either is not 0; tools/text_to_minh prints them in its summary). */
#define MNW_C_SINGLE_READ_BATCHES 0  /* float batches encoded by k_encode_reg_f32 */
#define MNW_C_SINGLE_READ_ABORTS  1  /* ... of which gave up and were rerun through reduce + pack (also counts fused launches) */
#define MNW_C_TWO_PASS_BATCHES    2  /* float batches that went to the reduce + pack launches directly (not eligible / option) */
#define MNW_C_EXACT_BLOCKS        3  /* float blocks sent to the exact sequential path (out-of-domain data) */
#define MNW_C_FUSED_BATCHES       4  /* float batches encoded by the opt-in fused launch k_encode_f32 */
#define MNW_C_INT_SINGLE_READ_BATCHES 5 /* batches whose int64 blocks were encoded by k_encode_reg_i64 (aborts count in MNW_C_SINGLE_READ_ABORTS) */
#define MNW_C_BACKOFF_BATCHES      7  /* batches that skipped the single-read kernels because a recent attempt had given up (single_read_backoff) */
#define MNW_C_VEC3_SINGLE_READ_CALLS 6 /* mnw_vectors_encode calls that went through k_encode_reg_vec3 (aborts as above) */
#define MNW_C_COUNT               8
int mnw_counters_get(mnw_ctx *ctx, int64_t out[MNW_C_COUNT]);
int mnw_counters_reset(mnw_ctx *ctx);

/* Tunables (0 = default).  group_bytes: blocks are processed in groups whose input is at most this
 * many bytes so that the pack pass re-reads its group from the Infinity Cache, not HBM.
 * fused: 1 = float batches run as one persistent launch (k_encode_f32); 0 (default) = see single_read.
 * single_read: 1 (default) = a float batch whose blocks can all be held on chip -- at most 2^24 elements each on a
 *        whole MI355X (twice as many 8,192-element items as the launch has workgroups) -- is encoded by
 *        k_encode_reg_f32, which reads each block once; 0 = always the reduce + pack launches.
 * single_read_int: 1 (default) = the int64 blocks of a batch, if each has at most 2^21 elements (512 items of 4,096), are
 *        encoded by k_encode_reg_i64, which reads each block once; 0 = always k_reduce_i64 + k_pack_i64.
 * big_blocks: 1 (default) = that includes blocks of more than 2^22 elements; 0 = those take reduce + pack.
 * reg_weighted: 1 (default) = the single-read kernel hands its work out in weighted rounds (the workgroups that the
 *        hardware schedules first get 14 items where the last get 10); 0 = every workgroup the same share.
 * single_read_backoff: 1 (default) = after a single-read launch has given up (something else holds part of the GPU), the next batch
 *        goes to the two-launch path without trying; a give-up right after the pause doubles it (1, 2, 4 ... 64 batches), a
 *        launch that runs to its end resets it.  0 = try every time (50 ms per give-up).
 * reg_round_table: which built-in table of round sizes the weighted hand-out uses (0, default: the quarters of the grid take
 *        14 / 12 / 11 / 10 items per cycle; 1..4: other ratios, kept for measurements -- none was faster).
 * reg_test_drop_item: test hook (default -1): the single-read kernel never publishes the statistics of that item, so
 *        the launch must give up after its 50 ms bound and the host must rerun the batch through reduce + pack.
 * vec3_fused: 1 (default) = mnw_vectors_encode reads the particle grid directly in its reduce and pack kernels
 *        when the geometry allows (n_sub and n_file multiples of four); 0 = always gather into the staging
 *        buffer first and run the block encoder on that.
 * vec3_single_read: 1 = mnw_vectors_encode reads the grid once (k_encode_reg_vec3) when vec3_fused applies, 8 <= n_sub <= 128 and a
 *        block has at most as many 2,048-particle vtiles as the launch has workgroups (and at most 1,024); 0 (default) =
 *        k_reduce_vec3 + k_pack_vec3, which measured as fast on configs[3]'s geometry and need no co-resident grid.
 * unlog_fused: 1 (default) = minh's reader transform 10^x (mnw_dec_job.xform & MNW_XF_LOG10) is applied inside the unpack kernel;
 *        0 = by a pass over the decoded column afterwards (k_minh_unlog).  Same floats either way.
 * vec3_tiles_per_wg: minp vector kernels: 2,048-particle tiles per workgroup (default 4; 0 = persistent grids).
 * host_threads: threads that copy between the caller's memory and the library's pinned buffers in the host-pointer entry
 *        points (default min(8, cores / 2)); set before the first such call.  stage_min_bytes: transfers below this size
 *        use a plain hipMemcpy (default 4 MiB).
 * grid_cap: upper bound on workgroups per launch (decode launches: 0 = about four tiles per workgroup). */
int mnw_set_option(mnw_ctx *ctx, const char *name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* MINNOW_HIP_H */
