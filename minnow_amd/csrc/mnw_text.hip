// Text catalogue ingest (go/text/text.go:176-197, go/text/parse.go): a block of ASCII text -> the requested columns
// as int64 / float32 device arrays, ready for the encoder.  The reference splits the block at '\n', cuts comments
// ("data # comment"), drops lines that hold nothing but separators, breaks every line into fields at runs of the
// separator byte and runs strconv.Atoi / float32(strconv.ParseFloat(s, 64)) on the requested fields.
//
//   k_text_count   newlines per 16 KB tile                        (host: exclusive prefix over the tiles)
//   k_text_lines   start offset of every line                     line i = [start[i], start[i+1] - 1)
//   k_text_valid   which lines survive uncomment + trim, count per 256 lines   (host: prefix again)
//   k_text_parse   one thread per line: fields, numbers, written at the line's compacted row
//
// Numbers: a decimal literal whose digits fit 2^53 and whose power of ten is within 10^+-22 is converted with ONE
// correctly rounded float64 multiplication or division of two exactly representable values (Clinger's fast path), so
// the result is the correctly rounded float64 -- the unique value strconv.ParseFloat returns -- and the cast to
// float32 reproduces the reference's double rounding.  Every other literal (more digits, larger exponents, hex floats)
// is put on a list and converted on the host with strtod, which is correctly rounded as well.
// (This file is included by mnw_kernels.hip, inside namespace mnw.)

constexpr int kTextTile = 16384;          // bytes per workgroup in the line passes: 256 threads x 64 bytes
constexpr int kTextMaxCols = 256;         // fields per line that can be addressed
constexpr int kTextMaxOut = 48;           // requested columns of each type

struct TextCols {
    int16_t slot[kTextMaxCols];           // field -> -1 (not wanted) | float slot | 0x4000 + int slot
    float* fout[kTextMaxOut];
    int64_t* iout[kTextMaxOut];
    int32_t fields_expected;              // every line must have this many fields (parse.go:96-103)
    int32_t max_field;                    // highest wanted field: the rest of the line is only counted
};

struct TextSlow {                         // a literal the device did not convert
    int64_t row;
    int64_t offset;
    int32_t len;
    int32_t slot;
};

struct TextState {                        // device words the host reads back
    unsigned long long err_line;          // lowest (raw line index + 1) << 8 | code, ~0 = none
    unsigned int n_slow;
    unsigned int pad;
};
constexpr int kTextErrFields = 1;         // wrong number of fields
constexpr int kTextErrInt = 2;            // strconv.Atoi would fail
constexpr int kTextErrFloat = 3;          // strconv.ParseFloat would fail

__device__ __forceinline__ int count_byte16(const u32x4 v, uint32_t pat) {
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t x = v[k] ^ pat;                                  // zero bytes where the byte matches
        const uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);   // 0x80 in every zero byte
        c += __popc(z);
    }
    return c;
}

__global__ __launch_bounds__(256) void k_text_count(const uint8_t* __restrict__ text, int64_t n, uint32_t* __restrict__ tile_nl) {
    __shared__ int red[4];
    const int64_t p0 = (int64_t)blockIdx.x * kTextTile + threadIdx.x * 64;
    int c = 0;
    if (p0 + 64 <= n) {
#pragma unroll
        for (int k = 0; k < 4; ++k) c += count_byte16(ldg<u32x4>(text + p0 + 16 * k), 0x0a0a0a0au);
    } else {
        for (int64_t p = p0; p < n; ++p) c += text[p] == '\n';
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane_id() == 0) red[wave_id()] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_nl[blockIdx.x] = (uint32_t)(red[0] + red[1] + red[2] + red[3]);
}

// exclusive prefix of a per-thread count over the 256 threads of a workgroup
__device__ __forceinline__ int block_exclusive_scan(int v, int* wsum) {
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane_id() >= o) incl += t; }
    if (lane_id() == 63) wsum[wave_id()] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave_id(); ++w) base += wsum[w];
    __syncthreads();
    return base + incl - v;
}

__global__ __launch_bounds__(256) void k_text_lines(const uint8_t* __restrict__ text, int64_t n, const int64_t* __restrict__ tile_base,
                                                    int64_t* __restrict__ line_start) {
    __shared__ int wsum[4];
    const int64_t p0 = (int64_t)blockIdx.x * kTextTile + threadIdx.x * 64;
    const int64_t p1 = p0 + 64 < n ? p0 + 64 : n;
    // the thread's 64 bytes, once, as four 16-byte vectors (zero past the end: never a newline)
    u32x4 w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (p0 + 16 * k + 16 <= n) w[k] = ldg<u32x4>(text + p0 + 16 * k);
        else {
            w[k] = u32x4{0, 0, 0, 0};
            for (int64_t p = p0 + 16 * k; p < p1; ++p) w[k][(int)(p - p0 - 16 * k) >> 2] |= (uint32_t)text[p] << (8 * ((p - p0) & 3));
        }
    }
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) c += count_byte16(w[k], 0x0a0a0a0au);
    int64_t idx = tile_base[blockIdx.x] + block_exclusive_scan(c, wsum);
    if (c) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t word = w[k][d];
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (((word >> (8 * b)) & 0xffu) == '\n') line_start[++idx] = p0 + 16 * k + 4 * d + b + 1;   // line idx starts behind the idx-th newline
            }
        }
    }
}

// [s, e) of raw line i with its comment cut off (parse.go:43-63)
__device__ __forceinline__ void text_line_span(const uint8_t* __restrict__ text, const int64_t* __restrict__ line_start, int64_t i,
                                               uint8_t comm, int64_t* s, int64_t* e) {
    const int64_t a = line_start[i], b = line_start[i + 1] - 1;
    int64_t p = a;
    while (p < b && text[p] != comm) ++p;
    *s = a; *e = p;
}

__global__ __launch_bounds__(256) void k_text_valid(const uint8_t* __restrict__ text, const int64_t* __restrict__ line_start, int64_t nlines,
                                                    uint8_t sep, uint8_t comm, uint8_t* __restrict__ valid, uint32_t* __restrict__ tile_valid) {
    __shared__ int red[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int v = 0;
    if (i < nlines) {
        int64_t s, e;
        text_line_span(text, line_start, i, comm, &s, &e);
        for (int64_t p = s; p < e; ++p) if (text[p] != sep) { v = 1; break; }       // trim(), parse.go:66-78
        valid[i] = (uint8_t)v;
    }
    int c = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if (lane_id() == 0) red[wave_id()] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_valid[blockIdx.x] = (uint32_t)(red[0] + red[1] + red[2] + red[3]);
}

__device__ __forceinline__ void text_error(TextState* st, int64_t line, int code) {
    atomicMin(&st->err_line, ((unsigned long long)(line + 1) << 8) | (unsigned)code);
}

// strconv.Atoi: optional sign, decimal digits, must fit int64 (int is 64 bits on the reference's platforms)
__device__ __forceinline__ bool text_atoi(const uint8_t* __restrict__ p, int len, int64_t* out) {
    int i = 0;
    bool neg = false;
    if (len > 0 && (p[0] == '+' || p[0] == '-')) { neg = p[0] == '-'; i = 1; }
    if (i >= len) return false;
    uint64_t v = 0;
    for (; i < len; ++i) {
        const uint32_t d = (uint32_t)p[i] - '0';
        if (d > 9) return false;
        if (v > (0x7fffffffffffffffull + (neg ? 1 : 0) - d) / 10) return false;       // overflow: Atoi reports a range error
        v = v * 10 + d;
    }
    *out = neg ? (int64_t)(0 - v) : (int64_t)v;
    return true;
}

__device__ const double kPow10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15,
                                      1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// 0: converted (exactly as strconv.ParseFloat(s, 64) would), 1: leave to the host, 2: syntax error
__device__ __forceinline__ int text_atof(const uint8_t* __restrict__ p, int len, double* out) {
    int i = 0;
    bool neg = false;
    if (len > 0 && (p[0] == '+' || p[0] == '-')) { neg = p[0] == '-'; i = 1; }
    if (i >= len) return 2;
    const uint32_t c0 = p[i] | 0x20;
    if (c0 == 'i' || c0 == 'n') return 1;                      // inf / infinity / nan: the host decides
    uint64_t m = 0;
    int nd = 0, dropped = 0, e10 = 0;
    bool seen_digit = false, seen_dot = false, inexact = false;
    for (; i < len; ++i) {
        const uint8_t c = p[i];
        if (c == '.') { if (seen_dot) return 2; seen_dot = true; continue; }
        const uint32_t d = (uint32_t)c - '0';
        if (d > 9) break;
        seen_digit = true;
        if (m == 0 && d == 0) { if (seen_dot) --e10; continue; }           // leading zeros
        if (nd < 19) { m = m * 10 + d; ++nd; if (seen_dot) --e10; }
        else { if (d) inexact = true; ++dropped; if (!seen_dot) ++e10; }
    }
    (void)dropped;
    if (!seen_digit) return (i < len && (p[i] | 0x20) == 'x') ? 1 : 2;
    if (i < len) {
        const uint32_t c = p[i] | 0x20;
        if (c == 'x' || p[i] == '_') return 1;                               // hex floats, digit separators: host
        if (c != 'e') return 2;
        ++i;
        bool eneg = false;
        if (i < len && (p[i] == '+' || p[i] == '-')) { eneg = p[i] == '-'; ++i; }
        if (i >= len) return 2;
        int ex = 0;
        for (; i < len; ++i) {
            const uint32_t d = (uint32_t)p[i] - '0';
            if (d > 9) return p[i] == '_' ? 1 : 2;
            if (ex < 100000) ex = ex * 10 + (int)d;
        }
        e10 += eneg ? -ex : ex;
    }
    double v;
    if (m == 0) v = 0.0;
    else if (inexact || m >= (1ull << 53)) return 1;
    else if (e10 >= 0 && e10 <= 22) v = __dmul_rn((double)m, kPow10[e10]);
    else if (e10 < 0 && e10 >= -22) v = __ddiv_rn((double)m, kPow10[-e10]);
    else return 1;
    *out = neg ? -v : v;
    return 0;
}

// Lines per workgroup (one wave) and the LDS budget for their text.  Sixty-four lines walked straight out of global
// memory touch 64 different cache lines per load instruction and evict each other from the L1 long before a lane has
// used up its line (measured: 20 GB/s of text); so the wave first copies the contiguous span of its 64 lines into
// LDS with coalesced 16-byte loads and the lanes parse from there.  A span that does not fit (very long lines) is
// parsed from global memory.
constexpr int kTextLinesPerWG = 64;
constexpr int kTextSpanMax = 60 * 1024;

__global__ __launch_bounds__(kTextLinesPerWG) void k_text_parse(const uint8_t* __restrict__ text, const int64_t* __restrict__ line_start,
                                                                int64_t nlines, uint8_t sep, uint8_t comm, const uint8_t* __restrict__ valid,
                                                                const int64_t* __restrict__ tile_vbase, TextCols cols,
                                                                TextSlow* __restrict__ slow, unsigned int slow_cap, TextState* __restrict__ st) {
    __shared__ __attribute__((aligned(16))) uint8_t span[kTextSpanMax];
    const int64_t i0 = (int64_t)blockIdx.x * kTextLinesPerWG;
    const int64_t i = i0 + threadIdx.x;
    const int v = i < nlines ? (int)valid[i] : 0;
    // the row of the line among the kept ones: tile_vbase is per 256 lines
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if ((int)threadIdx.x >= o) incl += t; }
    int64_t row = tile_vbase[i0 / 256] + incl - v;
    for (int64_t q = i0 / 256 * 256; q < i0; q += 64) {          // kept lines of the earlier waves' worth of this 256-line tile
        const int64_t j = q + threadIdx.x;
        row += __popcll(__ballot(j < nlines && valid[j]));
    }
    const int64_t last = i0 + kTextLinesPerWG < nlines ? i0 + kTextLinesPerWG : nlines;
    const int64_t g0 = line_start[i0] & ~(int64_t)15, g1 = line_start[last] - 1;         // span of the wave's lines, 16-byte aligned start
    const bool staged = g1 - g0 <= kTextSpanMax;
    if (staged) {
        for (int64_t q = g0 + 16 * threadIdx.x; q < g1; q += 16 * kTextLinesPerWG)
            *reinterpret_cast<u32x4*>(span + (q - g0)) = ldg<u32x4>(text + q);           // (the text buffer is padded by 64 bytes)
        __syncthreads();
    }
    if (!v) return;
    const uint8_t* const tb = staged ? span - g0 : text;          // tb[p] is byte p of the text either way (flat addressing)
    const int64_t a = line_start[i], b = line_start[i + 1] - 1;
    int64_t e = a;
    while (e < b && tb[e] != comm) ++e;                           // uncomment(), parse.go:43-63
    int field = 0;
    int64_t p = a;
    while (p < e) {
        while (p < e && tb[p] == sep) ++p;                        // fields(), parse.go:172-211
        if (p >= e) break;
        const int64_t t0 = p;
        while (p < e && tb[p] != sep) ++p;
        if (field <= cols.max_field && field < kTextMaxCols) {
            const int sl = cols.slot[field];
            if (sl >= 0x4000) {
                int64_t x;
                if (text_atoi(tb + t0, (int)(p - t0), &x)) cols.iout[sl - 0x4000][row] = x;
                else text_error(st, i, kTextErrInt);
            } else if (sl >= 0) {
                double x;
                const int rc = text_atof(tb + t0, (int)(p - t0), &x);
                if (rc == 0) cols.fout[sl][row] = (float)x;                          // float32(x), parse.go:165
                else if (rc == 1) {
                    const unsigned k = atomicAdd(&st->n_slow, 1u);
                    if (k < slow_cap) { TextSlow w = {row, t0, (int32_t)(p - t0), sl}; slow[k] = w; }
                } else text_error(st, i, kTextErrFloat);
            }
        }
        ++field;
    }
    if (field != cols.fields_expected) text_error(st, i, kTextErrFields);
}

void launch_text_count(hipStream_t s, const uint8_t* text, int64_t n, uint32_t* tile_nl) {
    const int64_t tiles = (n + kTextTile - 1) / kTextTile;
    hipLaunchKernelGGL(k_text_count, dim3((unsigned)tiles), dim3(256), 0, s, text, n, tile_nl);
}
void launch_text_lines(hipStream_t s, const uint8_t* text, int64_t n, const int64_t* tile_base, int64_t* line_start) {
    const int64_t tiles = (n + kTextTile - 1) / kTextTile;
    hipLaunchKernelGGL(k_text_lines, dim3((unsigned)tiles), dim3(256), 0, s, text, n, tile_base, line_start);
}
void launch_text_valid(hipStream_t s, const uint8_t* text, const int64_t* line_start, int64_t nlines, uint8_t sep, uint8_t comm,
                       uint8_t* valid, uint32_t* tile_valid) {
    hipLaunchKernelGGL(k_text_valid, dim3((unsigned)((nlines + 255) / 256)), dim3(256), 0, s, text, line_start, nlines, sep, comm, valid, tile_valid);
}
void launch_text_parse(hipStream_t s, const uint8_t* text, const int64_t* line_start, int64_t nlines, uint8_t sep, uint8_t comm,
                       const uint8_t* valid, const int64_t* tile_vbase, const TextCols& cols, TextSlow* slow, unsigned slow_cap, TextState* st) {
    hipLaunchKernelGGL(k_text_parse, dim3((unsigned)((nlines + kTextLinesPerWG - 1) / kTextLinesPerWG)), dim3(kTextLinesPerWG), 0, s, text,
                       line_start, nlines, sep, comm, valid, tile_vbase, cols, slow, slow_cap, st);
}
