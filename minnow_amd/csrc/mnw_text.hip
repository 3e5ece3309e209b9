// Text catalogue ingest (go/text/text.go:176-197, go/text/parse.go): a block of ASCII text -> the requested columns
// as int64 / float32 device arrays, ready for the encoder.  The reference splits the block at '\n', cuts comments
// ("data # comment"), drops lines that hold nothing but separators, breaks every line into fields at runs of the
// separator byte and runs strconv.Atoi / float32(strconv.ParseFloat(s, 64)) on the requested fields.
//
// The text is cut into tiles of 12,288 bytes; a tile OWNS the lines that start inside it (wherever they end).
//
//   k_text_count   per tile: lines owned, owned lines that survive uncomment + trim
//   k_text_scan1/2 exclusive prefix over the tiles (chunk totals, then every chunk scans itself): first line number and first output row of every tile
//   k_text_parse   per tile: 16 KB of text (16 bytes before the tile, the tile, 4,080 bytes after it for the end of its
//                  last line) go to LDS; every thread turns its 64 bytes into three 64-bit masks (newline, separator,
//                  comment) and from then on lines and fields are bit arithmetic: field starts = content bits whose
//                  predecessor is no content; the line of a field = newlines before it (prefix sum over the threads);
//                  its number inside the line = field starts since the last newline (a segmented prefix sum); its output
//                  row = owned non-empty lines that ended before it (prefix sum).  The start of every wanted field goes
//                  into a table in LDS, [column][line], and the fields are converted ONE FIELD PER THREAD in table order:
//                  the lanes stay busy whatever the lines look like, neighbouring lanes convert neighbouring rows of one
//                  column -- the same code path, stores into the same lines -- and a token is converted out of four
//                  registers holding its first 16 bytes, in straight-line code as long as the wave's longest token.
//                  A tile that holds a comment character, a line that does not end inside the staged text or more wanted
//                  fields than the table holds (6,144) takes the general path: one thread per line, walking it byte by
//                  byte as the reference does.
//
// Numbers: a decimal literal whose digits fit 2^53 and whose power of ten is within 10^+-22 is converted with ONE
// correctly rounded float64 multiplication or division of two exactly representable values (Clinger's fast path), so
// the result is the correctly rounded float64 -- the unique value strconv.ParseFloat returns -- and the cast to
// float32 reproduces the reference's double rounding.  Every other literal (more digits, larger exponents, hex floats)
// is put on a list and converted on the host with strtod, which is correctly rounded as well.
// (This file is included by mnw_kernels.hip, inside namespace mnw.)

constexpr int kTextTile = 12288;          // bytes of text whose line starts one workgroup owns
constexpr int kTextRegion = 16384;        // bytes a workgroup stages: 256 threads x 64
constexpr int kTextLead = 16;             // staged bytes in front of the tile (the byte before the tile decides whether a line starts at its first byte)
constexpr int kTextRecCap = 6144;         // wanted fields per tile on the fast path (kept lines x wanted columns)
constexpr int kTextMaxCols = 256;         // fields per line that can be addressed
constexpr int kTextMaxOut = 128;          // requested columns per call, float32 and int64 together

struct TextCols {
    int16_t slot[kTextMaxCols];           // field -> -1 (not wanted) | output column: 0 .. n_float-1 float32, n_float .. n_float+n_int-1 int64
    void* out[kTextMaxOut];               // output column -> device array
    int32_t n_float, n_int;
    int32_t fields_expected;              // every line must have this many fields (parse.go:96-118); -1: no column is wanted, nothing is checked (parse.go:83)
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
    unsigned int flags;                   // 1: a row beyond capacity_rows was dropped
    long long n_rows;                     // lines kept
    long long n_lines;                    // lines
    unsigned int tiles_general;           // tiles that took the general path (diagnostic)
    unsigned int pad;
};
constexpr int kTextErrFields = 1;         // wrong number of fields
constexpr int kTextErrInt = 2;            // strconv.Atoi would fail
constexpr int kTextErrFloat = 3;          // strconv.ParseFloat would fail

// 4 bits: which bytes of w equal the byte replicated in pat
__device__ __forceinline__ uint32_t text_eq4(uint32_t w, uint32_t pat) {
    const uint32_t x = w ^ pat;                                                      // zero bytes where the byte matches
    const uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);       // 0x80 in every zero byte, exactly
    return (z * 0x00204081u) >> 28;                                                  // bits 7,15,23,31 -> 0,1,2,3 (no carries: the partial products are disjoint)
}
__device__ __forceinline__ uint64_t text_eq_mask(const u32x4 (&w)[4], uint8_t c) {
    const uint32_t pat = 0x01010101u * c;
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            lo |= text_eq4(w[k][d], pat) << (16 * k + 4 * d);
            hi |= text_eq4(w[k + 2][d], pat) << (16 * k + 4 * d);
        }
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t bits_from(int lo) { return lo <= 0 ? ~0ull : lo > 63 ? 0ull : ~0ull << lo; }          // bits >= lo
__device__ __forceinline__ uint64_t bits_upto(int hi) { return hi < 0 ? 0ull : hi >= 63 ? ~0ull : ~0ull >> (63 - hi); }    // bits <= hi
__device__ __forceinline__ uint64_t bits_below(int b) { return (1ull << b) - 1; }                                          // bits < b, 0 <= b <= 63

// the 64 bytes at text[a, a + 64), a a multiple of 16; bytes in front of the text and behind its end read as '\n'
// (the buffer is padded, so a 16-byte load that starts inside the text is always legal)
__device__ __forceinline__ void text_load_chunk(const uint8_t* __restrict__ text, int64_t n, int64_t a, u32x4 (&w)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t q = a + 16 * k;
        if (q >= 0 && q + 16 <= n) w[k] = ldg<u32x4>(text + q);
        else {
            u32x4 v = {0x0a0a0a0au, 0x0a0a0a0au, 0x0a0a0a0au, 0x0a0a0a0au};
            if (q >= 0 && q < n) {
                const u32x4 g = ldg<u32x4>(text + q);
                const int keep = (int)(n - q);                                          // 1..15 bytes of text
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int kd = keep - 4 * d;
                    const uint32_t m = kd >= 4 ? ~0u : kd <= 0 ? 0u : (1u << (8 * kd)) - 1u;
                    v[d] = (g[d] & m) | (0x0a0a0a0au & ~m);
                }
            }
            w[k] = v;
        }
    }
}

// does the line that starts at s survive uncomment() and trim() (parse.go:43-78)?
__device__ __forceinline__ bool text_line_valid(const uint8_t* __restrict__ text, int64_t s, int64_t n, uint8_t sep, uint8_t comm) {
    for (int64_t p = s; p < n; ++p) {
        const uint8_t c = text[p];
        if (c == '\n' || c == comm) return false;
        if (c != sep) return true;
    }
    return false;
}

// exclusive prefix of a per-thread count over the 256 threads of a workgroup, and the total
__device__ __forceinline__ int block_exclusive_scan(int v, int* wsum, int* total) {
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane_id() >= o) incl += t; }
    __syncthreads();                                   // the previous use of wsum is over
    if (lane_id() == 63) wsum[wave_id()] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave_id(); ++w) base += wsum[w];
    *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return base + incl - v;
}

// which newlines of the thread's chunk start a line the tile owns: line starts in [T0, T0 + kTextTile), i.e. newlines at
// region offsets 15 .. kTextLead + kTextTile - 2, and inside the text (the fill behind the end is not a line start)
__device__ __forceinline__ uint64_t text_own_mask(uint64_t nl, int t, int64_t R, int64_t n) {
    const int64_t hi_abs = n - 1 - R;
    const int hi = (int)(hi_abs < kTextLead + kTextTile - 2 ? hi_abs : kTextLead + kTextTile - 2);
    return nl & bits_from(kTextLead - 1 - 64 * t) & bits_upto(hi - 64 * t);
}

__global__ __launch_bounds__(256) void k_text_count(const uint8_t* __restrict__ text, int64_t n, uint8_t sep, uint8_t comm,
                                                    int64_t* __restrict__ tile_cnt) {
    __shared__ int red[8];
    const int t = threadIdx.x;
    const int64_t R = (int64_t)blockIdx.x * kTextTile - kTextLead;
    int c_nl = 0, c_valid = 0;
    if (64 * t <= kTextLead + kTextTile - 2) {
        u32x4 w[4];
        text_load_chunk(text, n, R + 64 * t, w);
        const uint64_t nl = text_eq_mask(w, '\n'), sp = text_eq_mask(w, sep), cm = text_eq_mask(w, comm);
        const uint64_t own = text_own_mask(nl, t, R, n);
        c_nl = __popcll(own);
        const uint64_t next = own << 1;                               // the first byte of each owned line (bit 63's is in the next chunk)
        c_valid = __popcll(next & ~(nl | cm | sp));                   // content: kept
        uint64_t walk = next & sp & ~(nl | cm);                       // a separator: look further
        while (walk) {
            const int b = __builtin_ctzll(walk);
            walk &= walk - 1;
            c_valid += text_line_valid(text, R + 64 * t + b, n, sep, comm);
        }
        if (own >> 63) c_valid += text_line_valid(text, R + 64 * t + 64, n, sep, comm);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { c_nl += __shfl_xor(c_nl, o); c_valid += __shfl_xor(c_valid, o); }
    if (lane_id() == 0) { red[wave_id()] = c_nl; red[4 + wave_id()] = c_valid; }
    __syncthreads();
    if (t == 0) {
        tile_cnt[2 * (int64_t)blockIdx.x] = red[0] + red[1] + red[2] + red[3];
        tile_cnt[2 * (int64_t)blockIdx.x + 1] = red[4] + red[5] + red[6] + red[7];
    }
}

// tile_cnt[2t], [2t+1] -> exclusive prefixes (first line number, first output row of tile t); totals to the state.
// Two small launches of many workgroups (round 3; one workgroup of 1,024 threads walking 70 tiles each with strided loads took
// 153 us for the 71,000 tiles of a 0.87 GB block): k_text_scan1 sums chunks of 1,024 tiles (one tile per thread, coalesced 16-byte
// loads) into chunk totals, k_text_scan2 adds the totals of the chunks before its own and scans its chunk.
constexpr int kScanChunk = 1024;
__device__ __forceinline__ void scan_pair_1024(long long& il, long long& ir, long long* wl, long long* wr) {   // inclusive scan over the workgroup
    const int t = threadIdx.x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const long long tl = __shfl_up(il, o), tr = __shfl_up(ir, o);
        if ((t & 63) >= o) { il += tl; ir += tr; }
    }
    if ((t & 63) == 63) { wl[t >> 6] = il; wr[t >> 6] = ir; }
    __syncthreads();
    long long bl = 0, br = 0;
    for (int w = 0; w < (t >> 6); ++w) { bl += wl[w]; br += wr[w]; }
    il += bl; ir += br;
}
__global__ __launch_bounds__(1024) void k_text_scan1(const int64_t* __restrict__ tile_cnt, int64_t tiles, long long* __restrict__ totals) {
    __shared__ long long wl[16], wr[16];
    const int64_t k = (int64_t)blockIdx.x * kScanChunk + threadIdx.x;
    long long il = 0, ir = 0;
    if (k < tiles) { const i64x2 c = ldg<i64x2>(tile_cnt + 2 * k); il = c.x; ir = c.y; }
    scan_pair_1024(il, ir, wl, wr);
    if (threadIdx.x == kScanChunk - 1) { totals[2 * blockIdx.x] = il; totals[2 * blockIdx.x + 1] = ir; }
}
__global__ __launch_bounds__(1024) void k_text_scan2(int64_t* __restrict__ tile_cnt, int64_t tiles, const long long* __restrict__ totals,
                                                      TextState* __restrict__ st) {
    __shared__ long long wl[16], wr[16], base_s[2];
    __shared__ long long pl[16], pr[16];
    const int t = threadIdx.x;
    // the totals of the chunks in front of this one
    long long ol = 0, orr = 0;
    for (int64_t j = t; j < (int64_t)blockIdx.x; j += kScanChunk) { ol += totals[2 * j]; orr += totals[2 * j + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { ol += __shfl_xor(ol, o); orr += __shfl_xor(orr, o); }
    if ((t & 63) == 0) { pl[t >> 6] = ol; pr[t >> 6] = orr; }
    __syncthreads();
    if (t == 0) { long long a = 0, b2 = 0; for (int w = 0; w < 16; ++w) { a += pl[w]; b2 += pr[w]; } base_s[0] = a; base_s[1] = b2; }
    const int64_t k = (int64_t)blockIdx.x * kScanChunk + t;
    long long cl = 0, cr = 0;
    if (k < tiles) { const i64x2 c = ldg<i64x2>(tile_cnt + 2 * k); cl = c.x; cr = c.y; }
    long long il = cl, ir = cr;
    scan_pair_1024(il, ir, wl, wr);             // (its barrier also publishes base_s)
    const long long bl = base_s[0], br = base_s[1];
    if (k < tiles) { i64x2 o2; o2.x = bl + il - cl; o2.y = br + ir - cr; stg<i64x2>(tile_cnt + 2 * k, o2); }
    if (k == tiles - 1) { st->n_lines = bl + il; st->n_rows = br + ir; }
    if (tiles == 0 && blockIdx.x == 0 && t == 0) { st->n_lines = 0; st->n_rows = 0; }
}

__device__ __forceinline__ void text_error(TextState* st, int64_t line, int code) {
    atomicMin(&st->err_line, ((unsigned long long)(line + 1) << 8) | (unsigned)code);
}

#include "mnw_text_num.h"

// one wanted field -> its output (general functions); tok[0, len) through a flat pointer (LDS or global)
__device__ __forceinline__ void text_convert(const uint8_t* tok, int len, int col, int n_float, int64_t row, int64_t line, int64_t offset,
                                             void* const* outp, int64_t capacity,
                                             TextSlow* __restrict__ slow, unsigned int slow_cap, TextState* __restrict__ st) {
    if (row >= capacity) { atomicOr(&st->flags, 1u); return; }
    if (col >= n_float) {
        int64_t x;
        if (text_atoi(tok, len, &x)) static_cast<int64_t*>(outp[col])[row] = x;
        else text_error(st, line, kTextErrInt);
    } else {
        double x;
        const int rc = text_atof(tok, len, &x);
        if (rc == 0) static_cast<float*>(outp[col])[row] = (float)x;                 // float32(x), parse.go:165
        else if (rc == 1) {
            const unsigned k = atomicAdd(&st->n_slow, 1u);
            if (k < slow_cap) { TextSlow w = {row, offset, (int32_t)len, col}; slow[k] = w; }
        } else text_error(st, line, kTextErrFloat);
    }
}

__global__ __launch_bounds__(256) void k_text_parse(const uint8_t* __restrict__ text, int64_t n, uint8_t sep, uint8_t comm,
                                                    const int64_t* __restrict__ tile_base, TextCols cols, int64_t capacity,
                                                    TextSlow* __restrict__ slow, unsigned int slow_cap, TextState* __restrict__ st) {
    __shared__ __attribute__((aligned(16))) uint8_t region[kTextRegion + 32];      // (+32: a token's 16-byte window may reach behind the end)
    __shared__ __attribute__((aligned(16))) uint16_t rec[kTextRecCap];      // rec[column * kept lines + line] = where the field starts
    __shared__ int16_t s_slot[kTextMaxCols];
    __shared__ void* s_out[kTextMaxOut];
    __shared__ double s_p10[23];
    __shared__ int wsum[4];
    __shared__ uint32_t wseg[4];
    __shared__ unsigned int s_flags, s_nown;

    const int t = threadIdx.x;
    const int64_t R = (int64_t)blockIdx.x * kTextTile - kTextLead;
    const int64_t line_base = tile_base[2 * (int64_t)blockIdx.x], row_base = tile_base[2 * (int64_t)blockIdx.x + 1];
    if (t == 0) { s_flags = 0; s_nown = 0; }
    s_slot[t] = cols.slot[t];
    if (t < kTextMaxOut) s_out[t] = cols.out[t];
    if (t < 23) s_p10[t] = text_pow10(t);
    if (t < 8) reinterpret_cast<uint32_t*>(region + kTextRegion)[t] = 0x0a0a0a0au;
    // every table entry starts out as "no such field": a line with too few fields (an error the host reports) leaves holes
    static_assert(kTextRecCap * 2 == 256 * 48, "three 16-byte stores per thread cover the table");
#pragma unroll
    for (int k = 0; k < 3; ++k) reinterpret_cast<u32x4*>(rec)[3 * t + k] = u32x4{~0u, ~0u, ~0u, ~0u};
    u32x4 w[4];
    text_load_chunk(text, n, R + 64 * t, w);
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(region + 64 * t + 16 * k) = w[k];
    const uint64_t nl = text_eq_mask(w, '\n'), sp = text_eq_mask(w, sep), cm = text_eq_mask(w, comm);
    const uint64_t nlm = nl & bits_from(kTextLead - 1 - 64 * t);       // newlines that start a line at or behind the tile's first byte
    const uint64_t own = text_own_mask(nl, t, R, n);
    __syncthreads();
    if (cm) atomicOr(&s_flags, 1u);
    if (own) atomicAdd(&s_nown, (unsigned)__popcll(own));
    int nl_total;
    const int nl_before = block_exclusive_scan(__popcll(nlm), wsum, &nl_total);      // (its barriers publish s_flags / s_nown)
    const int nown = (int)s_nown;
    const int n_float = cols.n_float, n_want = cols.n_float + cols.n_int;
    // general path: a comment somewhere in the staged text, or the tile's last line does not end inside it
    bool general = (s_flags & 1u) || (nown > 0 && nl_total < nown + 1);
    int vtotal = 0;

    if (!general) {
        const uint8_t prevb = t ? region[64 * t - 1] : (uint8_t)'\n';
        const uint64_t content = ~(nl | sp);
        const uint64_t start = content & ~((content << 1) | ((prevb != sep && prevb != '\n') ? 1ull : 0ull));
        // fields since the last newline: segmented sum, value = flag "a newline seen" << 31 | count
        const int lastnl = nlm ? 63 - __builtin_clzll(nlm) : -1;
        const uint32_t seg = nlm ? (0x80000000u | (uint32_t)__popcll(start & bits_from(lastnl + 1))) : (uint32_t)__popcll(start);
        uint32_t incl = seg;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t a = __shfl_up(incl, o);
            if (lane_id() >= o) incl = (incl >> 31) ? incl : ((a & 0x80000000u) | ((a + incl) & 0x7fffffffu));
        }
        uint32_t excl = __shfl_up(incl, 1);
        if (lane_id() == 0) excl = 0;
        if (lane_id() == 63) wseg[wave_id()] = incl;
        __syncthreads();
        uint32_t pre = 0;
        for (int k = 0; k < wave_id(); ++k) { const uint32_t b = wseg[k]; pre = (b >> 31) ? b : ((pre & 0x80000000u) | ((pre + b) & 0x7fffffffu)); }
        const uint32_t carry = (excl >> 31) ? excl : ((pre & 0x80000000u) | ((pre + excl) & 0x7fffffffu));
        const int carry_cnt = (int)(carry & 0x7fffffffu);
        // the newlines of the chunk: which of them end an owned line that has fields (a kept line)
        uint64_t validend = 0;
        {
            uint64_t m = nlm;
            int prev = -1, cnt_in = carry_cnt, j = 0;
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                const int cnt = cnt_in + __popcll(start & bits_below(b) & bits_from(prev + 1));
                const int l = nl_before + j;                       // the line this newline ends was started by counted newline l - 1
                if (l >= 1 && l <= nown && cnt > 0) {
                    validend |= 1ull << b;
                    if (cols.fields_expected >= 0 && cnt != cols.fields_expected) text_error(st, line_base + l - 1, kTextErrFields);     // parse.go:114-118
                }
                cnt_in = 0; prev = b; ++j;
            }
        }
        const int vbase = block_exclusive_scan(__popcll(validend), wsum, &vtotal);
        if (vtotal * n_want > kTextRecCap) general = true;
        else {
            uint64_t m = start;
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                const uint64_t below = bits_below(b);
                const uint64_t nlb = nlm & below;
                const int l = nl_before + __popcll(nlb);           // started by counted newline l - 1
                if (l < 1 || l > nown) continue;
                const int fidx = nlb ? __popcll(start & below & bits_from(64 - __builtin_clzll(nlb))) : carry_cnt + __popcll(start & below);
                if (fidx > cols.max_field) continue;
                const int col = s_slot[fidx];
                if (col < 0) continue;
                const int rowoff = vbase + __popcll(validend & below);
                rec[col * vtotal + rowoff] = (uint16_t)(64 * t + b);
            }
            __syncthreads();
        }
    }

    if (!general) {
        // one field per thread, column after column (float columns first): neighbouring lanes convert neighbouring rows of the
        // same column, so they run the same code and their stores fall into the same lines
        const int total = vtotal * n_want, nfrec = vtotal * n_float;
        const float inv = 1.0f / (float)(vtotal > 0 ? vtotal : 1);
        const uint32_t sep4 = 0x01010101u * sep;
        for (int kb = 0; kb < total; kb += 256) {                             // (every lane makes every trip: the wave-wide maximum below)
            const int k = kb + t;
            const bool have = k < total;
            const int col = have ? (int)(((float)k + 0.5f) * inv) : 0;       // k / vtotal, exact for k < 2^21
            const int64_t row = row_base + (k - col * vtotal);
            const int pos = have ? rec[k] : 0xffff;
            const bool hole = pos == 0xffff;                                  // no such field: the line's field count is wrong, already flagged
            // the token's first 16 bytes
            const uint32_t* d = reinterpret_cast<const uint32_t*>(region) + ((pos & 0x3fff) >> 2);
            const uint32_t d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3], d4 = d[4];
            const uint32_t sh = (uint32_t)pos & 3u;
            uint32_t tw[4];
            tw[0] = __builtin_amdgcn_alignbyte(d1, d0, sh); tw[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
            tw[2] = __builtin_amdgcn_alignbyte(d3, d2, sh); tw[3] = __builtin_amdgcn_alignbyte(d4, d3, sh);
            uint32_t ends = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) ends |= (text_eq4(tw[q], sep4) | text_eq4(tw[q], 0x0a0a0a0au)) << (4 * q);
            int len = ends ? __builtin_ctz(ends) : 17;
            // the longest token of the wave bounds the conversion loops (a column's tokens have much the same length)
            int bound = (hole || len > 16) ? 0 : len;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int u = __shfl_xor(bound, o); bound = u > bound ? u : bound; }
            bound = __builtin_amdgcn_readfirstlane(bound);
            if (hole) continue;
            if (row >= capacity) { atomicOr(&st->flags, 1u); continue; }
            int bad = 0;
            if (k >= nfrec) {
                int64_t x;
                bool ok = len <= 16 && fast_atoi16(tw, len, &x, bound);
                if (!ok) {
                    if (len > 16) { len = 16; while (pos + len < kTextRegion && region[pos + len] != sep && region[pos + len] != '\n') ++len; }
                    ok = text_atoi(region + pos, len, &x);
                }
                if (ok) static_cast<int64_t*>(s_out[col])[row] = x;
                else bad = kTextErrInt;
            } else {
                double x;
                int rc = len <= 16 ? fast_atof16(tw, len, s_p10, &x, bound) : 1;
                if (rc != 0) {
                    if (len > 16) { len = 16; while (pos + len < kTextRegion && region[pos + len] != sep && region[pos + len] != '\n') ++len; }
                    rc = text_atof(region + pos, len, &x);
                }
                if (rc == 0) static_cast<float*>(s_out[col])[row] = (float)x;                 // float32(x), parse.go:165
                else if (rc == 1) {
                    const unsigned q = atomicAdd(&st->n_slow, 1u);
                    if (q < slow_cap) { TextSlow ws = {row, R + pos, (int32_t)len, (int32_t)col}; slow[q] = ws; }
                } else bad = kTextErrFloat;
            }
            if (bad) {                                          // the line number is only needed here: the newlines in front of the field
                int c = 0;
                for (int q = kTextLead - 1; q < pos; ++q) c += region[q] == '\n';
                text_error(st, line_base + c - 1, bad);
            }
        }
        return;
    }

    // ---- general path: the lines behind the thread's owned newlines, one after the other, as the reference walks them ----
    if (t == 0) atomicAdd(&st->tiles_general, 1u);
    uint64_t vmask = 0;
    {
        uint64_t m = own;
        while (m) {
            const int b = __builtin_ctzll(m);
            m &= m - 1;
            if (text_line_valid(text, R + 64 * t + b + 1, n, sep, comm)) vmask |= 1ull << b;
        }
    }
    int vtot;
    int64_t row = row_base + block_exclusive_scan(__popcll(vmask), wsum, &vtot);
    while (vmask) {
        const int b = __builtin_ctzll(vmask);
        vmask &= vmask - 1;
        const int64_t a = R + 64 * t + b + 1;
        const int64_t line = line_base + nl_before + __popcll(nlm & bits_below(b));
        int64_t e = a;
        while (e < n && text[e] != '\n' && text[e] != comm) ++e;                        // uncomment(), parse.go:43-63
        // the line through a pointer to its first byte: the staged copy if the whole line is in it, else the text itself
        const uint8_t* const lb = e <= R + kTextRegion ? region + (a - R) : text + a;
        const int64_t ll = e - a;
        int field = 0;
        int64_t p = 0;
        while (p < ll) {
            while (p < ll && lb[p] == sep) ++p;                                         // fields(), parse.go:172-211
            if (p >= ll) break;
            const int64_t t0 = p;
            while (p < ll && lb[p] != sep) ++p;
            if (field <= cols.max_field && field < kTextMaxCols) {
                const int col = s_slot[field];
                if (col >= 0) text_convert(lb + t0, (int)(p - t0), col, n_float, row, line, a + t0, s_out, capacity, slow, slow_cap, st);
            }
            ++field;
        }
        if (cols.fields_expected >= 0 && field != cols.fields_expected) text_error(st, line, kTextErrFields);
        ++row;
    }
}

void launch_text_count(hipStream_t s, const uint8_t* text, int64_t n, uint8_t sep, uint8_t comm, int64_t* tile_cnt) {
    const int64_t tiles = (n + kTextTile - 1) / kTextTile;
    hipLaunchKernelGGL(k_text_count, dim3((unsigned)tiles), dim3(256), 0, s, text, n, sep, comm, tile_cnt);
}
// totals: scratch of 2 * ceil(tiles / 1024) int64 (the caller keeps it behind the tile counters)
void launch_text_scan(hipStream_t s, int64_t* tile_cnt, int64_t tiles, TextState* st) {
    const int64_t chunks = tiles > 0 ? (tiles + kScanChunk - 1) / kScanChunk : 1;
    long long* totals = reinterpret_cast<long long*>(tile_cnt + 2 * tiles + 2);
    hipLaunchKernelGGL(k_text_scan1, dim3((unsigned)chunks), dim3(kScanChunk), 0, s, tile_cnt, tiles, totals);
    hipLaunchKernelGGL(k_text_scan2, dim3((unsigned)chunks), dim3(kScanChunk), 0, s, tile_cnt, tiles, totals, st);
}
// the host's conversions of the literals the device left alone, written into their columns: slow[i].offset holds the
// float32 bit pattern (low 32 bits), row and slot as the parse kernel recorded them
__global__ __launch_bounds__(256) void k_text_patch(const TextSlow* __restrict__ slow, unsigned int n, TextCols cols) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) static_cast<uint32_t*>(cols.out[slow[i].slot])[slow[i].row] = (uint32_t)slow[i].offset;
}
void launch_text_patch(hipStream_t s, const TextSlow* slow, unsigned n, const TextCols& cols) {
    hipLaunchKernelGGL(k_text_patch, dim3((n + 255) / 256), dim3(256), 0, s, slow, n, cols);
}
void launch_text_parse(hipStream_t s, const uint8_t* text, int64_t n, uint8_t sep, uint8_t comm, const int64_t* tile_base, const TextCols& cols,
                       int64_t capacity, TextSlow* slow, unsigned slow_cap, TextState* st) {
    const int64_t tiles = (n + kTextTile - 1) / kTextTile;
    hipLaunchKernelGGL(k_text_parse, dim3((unsigned)tiles), dim3(256), 0, s, text, n, sep, comm, tile_base, cols, capacity, slow, slow_cap, st);
}
