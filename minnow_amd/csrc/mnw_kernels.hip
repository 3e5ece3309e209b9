// mnw_kernels.hip -- hand-written HIP kernels for minnow's quantize + bit-pack path, gfx950 only.
//
// Reference algorithm (citations into the reference checkout):
//   quantize            go/group.go:316-320     q = int64(floor(float64((x-low)/dx)))  [f32 sub, f32 div]
//   periodicMin, bound  go/group.go:384-409, 374-382
//   intGroup.writeData  go/group.go:242-255     min, x-min, Bits = PrecisionNeeded(max), pack
//   bit.BufferedArray   go/bit/bit.go:84-134    LSB-first stream, element i at bits [i*b, (i+1)*b)
//   (*Array).Slice      go/bit/bit.go:29-82
//   floatGroup.readData go/group.go:299-310
//
// Shape of the GPU path (DESIGN.md has the full account):
//   encode = reduce pass (one read of the block: min/max statistics, atomics into 64 per-job stat
//   lines) + pack pass (re-reads the block -- from the Infinity Cache when blocks are launched in
//   groups that fit it -- quantizes again, stages the re-based values in LDS and assembles whole
//   64-bit words of the output stream, so global stores are coalesced and never need atomics:
//   64 elements at b bits are exactly b words).
//   decode = one pass: stream words -> LDS -> per-element window extract -> re-base/bound/dequantize.
// The path is integer/bandwidth bound: no MFMA.
#include "mnw_common.h"

namespace mnw {

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int64_t wave_min_i64(int64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int64_t t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}
__device__ __forceinline__ int64_t wave_max_i64(int64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int64_t t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { uint64_t t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}

// job of tile t: last j with jobs[j].tile0 <= t (jobs with n == 0 own no tile and are never hit
// because the next job has the same tile0 and the search takes the last one)
__device__ __forceinline__ int find_job(const DevJob* jobs, int njobs, int64_t t) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].tile0 <= t) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ int64_t tiles_of(int64_t n) { return (n + kTile - 1) / kTile; }

// go/group.go:318-319: float32 subtract, IEEE float32 divide, then floor.  floor() of the float32
// quotient is exact in float32, and float64() of it is the same number, so the float64 detour of
// the Go text changes nothing.  (hipcc: correctly rounded f32 division, no contraction, denormals on.)
__device__ __forceinline__ float quant_floor(float x, float low, float dx) {
    return floorf(__fdiv_rn(__fsub_rn(x, low), dx));
}

// ------------------------------------------------------------------------------------------------
// reduce pass, float32 blocks: statistics of floor((x-low)/dx) kept as float32 (integers, exact).
//   stat[0] = bias(max)   stat[1] = ~bias(min)   stat[2] = bias(max of values <  pixels/2)
//   stat[3] = ~bias(min of values >= pixels/2)   stat[4] = 1 if NaN or |v| >= 2^63 seen
// all combined with atomic unsigned max, so zero-filled memory is the identity.
// ------------------------------------------------------------------------------------------------
struct FStat {
    float mn, mx, mxlo, mnhi;
    bool bad;
};

__device__ __forceinline__ void fstat_add(FStat& s, float v, float hf) {
    s.bad |= !(fabsf(v) < 9223372036854775808.0f);   // NaN, Inf, |v| >= 2^63
    s.mn = fminf(s.mn, v);
    s.mx = fmaxf(s.mx, v);
    bool lo = v < hf;
    s.mxlo = fmaxf(s.mxlo, lo ? v : -INFINITY);
    s.mnhi = fminf(s.mnhi, lo ? INFINITY : v);
}

__global__ __launch_bounds__(kThreads) void k_reduce_f32(const DevJob* __restrict__ jobs, int njobs,
                                                         int64_t total_tiles,
                                                         uint64_t* __restrict__ stats) {
    __shared__ float red[4][kThreads / 64];
    __shared__ int redbad[kThreads / 64];
    const int tid = threadIdx.x;
    int64_t t0 = total_tiles * blockIdx.x / gridDim.x;
    const int64_t t1 = total_tiles * (blockIdx.x + 1) / gridDim.x;
    if (t0 >= t1) return;
    int j = find_job(jobs, njobs, t0);

    while (t0 < t1) {
        const DevJob jb = jobs[j];
        const int64_t jend = jb.tile0 + tiles_of(jb.n);
        const int64_t tend = t1 < jend ? t1 : jend;
        const float* __restrict__ x = (const float*)jb.src;
        const float low = jb.low, dx = jb.dxq, hf = jb.hf;
        FStat s = {INFINITY, -INFINITY, -INFINITY, INFINITY, false};

        for (int64_t t = t0; t < tend; ++t) {
            const int64_t base = (t - jb.tile0) * kTile;
            if (base + kTile <= jb.n) {
                float4 v[kTile / (4 * kThreads)];
#pragma unroll
                for (int k = 0; k < kTile / (4 * kThreads); ++k)
                    v[k] = *reinterpret_cast<const float4*>(x + base + 4 * (k * kThreads + tid));
#pragma unroll
                for (int k = 0; k < kTile / (4 * kThreads); ++k) {
                    fstat_add(s, quant_floor(v[k].x, low, dx), hf);
                    fstat_add(s, quant_floor(v[k].y, low, dx), hf);
                    fstat_add(s, quant_floor(v[k].z, low, dx), hf);
                    fstat_add(s, quant_floor(v[k].w, low, dx), hf);
                }
            } else {
                for (int64_t i = base + tid; i < jb.n; i += kThreads)
                    fstat_add(s, quant_floor(x[i], low, dx), hf);
            }
        }

        // workgroup reduction, then one 5-lane atomic instruction into this workgroup's stat line
        float mn = wave_min(s.mn), mx = wave_max(s.mx), mxlo = wave_max(s.mxlo), mnhi = wave_min(s.mnhi);
        int bad = __any(s.bad);
        __syncthreads();
        if (lane_id() == 0) {
            red[0][wave_id()] = mn; red[1][wave_id()] = mx;
            red[2][wave_id()] = mxlo; red[3][wave_id()] = mnhi;
            redbad[wave_id()] = bad;
        }
        __syncthreads();
        if (tid < 5) {
            float a = red[0][0], b = red[1][0], c = red[2][0], d = red[3][0];
            int bd = redbad[0];
#pragma unroll
            for (int w = 1; w < kThreads / 64; ++w) {
                a = fminf(a, red[0][w]); b = fmaxf(b, red[1][w]);
                c = fmaxf(c, red[2][w]); d = fminf(d, red[3][w]);
                bd |= redbad[w];
            }
            // -inf / +inf mean "no element in that half"; they map to the identity (0)
            uint64_t val;
            if (tid == 0) val = bd ? 0 : bias((int64_t)b);
            else if (tid == 1) val = bd ? 0 : ~bias((int64_t)a);
            else if (tid == 2) val = (bd || c == -INFINITY) ? 0 : bias((int64_t)c);
            else if (tid == 3) val = (bd || d == INFINITY) ? 0 : ~bias((int64_t)d);
            else val = bd ? 1 : 0;
            uint64_t* rec = stats + ((int64_t)j * kStatSlots + (blockIdx.x % kStatSlots)) * kStatWords;
            if (val != 0) atomicMax((unsigned long long*)(rec + tid), (unsigned long long)val);
        }
        t0 = tend;
        ++j;
    }
}

// ------------------------------------------------------------------------------------------------
// reduce pass, int64 blocks (go/group.go:350-357 int64Min, bit.go:151-159 max of x-min)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_reduce_i64(const DevJob* __restrict__ jobs, int njobs,
                                                         int64_t total_tiles,
                                                         uint64_t* __restrict__ stats) {
    __shared__ int64_t red[2][kThreads / 64];
    const int tid = threadIdx.x;
    int64_t t0 = total_tiles * blockIdx.x / gridDim.x;
    const int64_t t1 = total_tiles * (blockIdx.x + 1) / gridDim.x;
    if (t0 >= t1) return;
    int j = find_job(jobs, njobs, t0);

    while (t0 < t1) {
        const DevJob jb = jobs[j];
        const int64_t jend = jb.tile0 + tiles_of(jb.n);
        const int64_t tend = t1 < jend ? t1 : jend;
        const int64_t* __restrict__ x = (const int64_t*)jb.src;
        int64_t mn = INT64_MAX, mx = INT64_MIN;

        for (int64_t t = t0; t < tend; ++t) {
            const int64_t base = (t - jb.tile0) * kTile;
            if (base + kTile <= jb.n) {
                longlong2 v[kTile / (2 * kThreads)];
#pragma unroll
                for (int k = 0; k < kTile / (2 * kThreads); ++k)
                    v[k] = *reinterpret_cast<const longlong2*>(x + base + 2 * (k * kThreads + tid));
#pragma unroll
                for (int k = 0; k < kTile / (2 * kThreads); ++k) {
                    mn = v[k].x < mn ? v[k].x : mn; mx = v[k].x > mx ? v[k].x : mx;
                    mn = v[k].y < mn ? v[k].y : mn; mx = v[k].y > mx ? v[k].y : mx;
                }
            } else {
                for (int64_t i = base + tid; i < jb.n; i += kThreads) {
                    int64_t v = x[i];
                    mn = v < mn ? v : mn; mx = v > mx ? v : mx;
                }
            }
        }
        mn = wave_min_i64(mn);
        mx = wave_max_i64(mx);
        __syncthreads();
        if (lane_id() == 0) { red[0][wave_id()] = mn; red[1][wave_id()] = mx; }
        __syncthreads();
        if (tid < 2) {
            int64_t a = red[0][0], b = red[1][0];
#pragma unroll
            for (int w = 1; w < kThreads / 64; ++w) {
                a = red[0][w] < a ? red[0][w] : a;
                b = red[1][w] > b ? red[1][w] : b;
            }
            uint64_t val = tid == 0 ? bias(b) : ~bias(a);
            uint64_t* rec = stats + ((int64_t)j * kStatSlots + (blockIdx.x % kStatSlots)) * kStatWords;
            if (val != 0) atomicMax((unsigned long long*)(rec + tid), (unsigned long long)val);
        }
        t0 = tend;
        ++j;
    }
}

// ------------------------------------------------------------------------------------------------
// block descriptor: what intGroup.writeData derives from the whole block (group.go:243-254)
// ------------------------------------------------------------------------------------------------
struct Desc {
    int64_t imin;      // intGroup min (after periodic re-basing)
    int64_t pmin;      // periodicMin result (float groups)
    int64_t nbytes;
    int bits;
    int status;        // MNW_OK(0), -5 capacity, -6 domain, kNeedsExact
};

// Every lane reads one stat line; wave-wide unsigned max gives the job's statistics.
__device__ __forceinline__ void load_stats(const uint64_t* __restrict__ stats, int j, uint64_t out[5]) {
    const uint64_t* rec = stats + ((int64_t)j * kStatSlots + lane_id()) * kStatWords;
#pragma unroll
    for (int k = 0; k < 5; ++k) out[k] = wave_max_u64(rec[k]);
}

// float blocks.  For q in [0,pixels) periodicMin (group.go:384-409) is order independent: it is the
// start of the smallest ring arc covering every q if that arc is <= pixels/2 wide, else 0
// (DESIGN.md "periodicMin as reductions"; tests/test_oracle_golden.py::test_periodic_reduction_identity).
// Anything else (q out of range, NaN, huge) is order dependent -> exact sequential path.
__device__ __forceinline__ Desc make_desc_f32(const DevJob& jb, const uint64_t st[5]) {
    Desc d = {0, 0, 0, 0, 0};
    if (st[4] != 0) { d.status = (int)kNeedsExact; return d; }
    const int64_t mx0 = unbias(st[0]), mn0 = unbias(~st[1]);
    const bool has_lo = st[2] != 0, has_hi = st[3] != 0;
    const int64_t mxlo = unbias(st[2]), mnhi = unbias(~st[3]);
    uint64_t maxdelta;
    if (jb.periodic == 1) {
        // 32-bit pack path below needs q and q+pixels to fit 32 bits
        if (mn0 < 0 || mx0 >= jb.pixels || jb.pixels > 0x7fffffffll) { d.status = (int)kNeedsExact; return d; }
        const int64_t h = jb.pixels / 2;
        if (mx0 - mn0 + 1 <= h) { d.pmin = mn0; d.imin = mn0; maxdelta = (uint64_t)(mx0 - mn0); }
        else if (has_lo && has_hi && (mxlo + jb.pixels) - mnhi + 1 <= h) {
            d.pmin = mnhi; d.imin = mnhi; maxdelta = (uint64_t)(mxlo + jb.pixels - mnhi);
        } else { d.pmin = 0; d.imin = mn0; maxdelta = (uint64_t)(mx0 - mn0); }
    } else {
        if (mn0 < -0x80000000ll || mx0 > 0x7fffffffll) { d.status = (int)kNeedsExact; return d; }
        d.imin = mn0; maxdelta = (uint64_t)(mx0 - mn0);
    }
    d.bits = precision_needed(maxdelta);
    d.nbytes = d.bits == 0 ? 0 : array_bytes(d.bits, jb.n);
    if (d.nbytes > jb.cap) d.status = -5;
    return d;
}

__device__ __forceinline__ Desc make_desc_i64(const DevJob& jb, const uint64_t st[5]) {
    Desc d = {0, 0, 0, 0, 0};
    if (jb.mode == kModeExplicit) { d.imin = jb.min; d.bits = jb.bits; }
    else {
        const int64_t mx0 = unbias(st[0]), mn0 = unbias(~st[1]);
        d.imin = mn0;
        d.bits = precision_needed((uint64_t)mx0 - (uint64_t)mn0);
        if (d.bits < 0) { d.status = -6; d.bits = 0; return d; }
    }
    d.nbytes = d.bits == 0 ? 0 : array_bytes(d.bits, jb.n);
    if (d.nbytes > jb.cap) d.status = -5;
    return d;
}

// ------------------------------------------------------------------------------------------------
// word assembly: output word w of a tile from the staged re-based values (LDS), b bits each.
// Word w covers stream bits [64w, 64w+64): it starts inside element e0 = 64w/b at bit o.
// ------------------------------------------------------------------------------------------------
template <typename T>   // T = uint32_t (b <= 32) or uint64_t (b <= 64); values already < 2^b
__device__ __forceinline__ uint64_t assemble_word(const T* __restrict__ d, uint32_t w, uint32_t b,
                                                  uint32_t magic) {
    const uint32_t bitpos = w * 64u;
    // floor(bitpos / b): magic = ceil(2^32/b) is exact for bitpos*b < 2^32; b == 1 has no 32-bit magic
    uint32_t e = b == 1 ? bitpos : __umulhi(bitpos, magic);
    const uint32_t o = bitpos - e * b;
    uint64_t acc = (uint64_t)d[e] >> o;
    uint32_t pos = b - o;
    ++e;
    while (pos < 64u) {
        acc |= (uint64_t)d[e] << pos;
        pos += b;
        ++e;
    }
    return acc;
}

// store the words of one tile: full 64-bit words coalesced, the last partial word of a block
// byte-wise (ArrayBytes is a byte count, bit.go:23-25; pad bits are zero, bit.go:94)
template <typename T>
__device__ __forceinline__ void store_tile_words(const T* __restrict__ d, uint8_t* __restrict__ out,
                                                 int64_t tile_byte0, int64_t tile_bytes, uint32_t b,
                                                 uint32_t magic) {
    const uint32_t full = (uint32_t)(tile_bytes >> 3);
    uint64_t* o64 = reinterpret_cast<uint64_t*>(out + tile_byte0);
    for (uint32_t w = threadIdx.x; w < full; w += kThreads) o64[w] = assemble_word<T>(d, w, b, magic);
    const uint32_t rem = (uint32_t)(tile_bytes & 7);
    if (rem != 0 && threadIdx.x == 0) {
        uint64_t v = assemble_word<T>(d, full, b, magic);
        uint8_t* p = out + tile_byte0 + (int64_t)full * 8;
        for (uint32_t k = 0; k < rem; ++k) p[k] = (uint8_t)(v >> (8 * k));
    }
}

constexpr int kLdsBytes = 2 * kTile * 4 + 64;      // two 32-bit staging buffers (or one 64-bit) + pad

// ------------------------------------------------------------------------------------------------
// pack pass, float32 blocks (32-bit path: q in [0,pixels), pixels < 2^31, or int32-range q when
// not periodic).  One barrier per tile: the 32-bit staging buffer is double buffered.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_pack_f32(const DevJob* __restrict__ jobs, int njobs,
                                                       int64_t total_tiles,
                                                       const uint64_t* __restrict__ stats,
                                                       Meta* __restrict__ meta) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[kLdsBytes / 4];
    const int tid = threadIdx.x;
    int64_t t0 = total_tiles * blockIdx.x / gridDim.x;
    const int64_t t1 = total_tiles * (blockIdx.x + 1) / gridDim.x;
    if (t0 >= t1) return;
    int j = find_job(jobs, njobs, t0);
    int buf = 0;
    if (tid < 16) lds[2 * kTile + tid] = 0;         // pad read by the last word's look-ahead

    while (t0 < t1) {
        const DevJob jb = jobs[j];
        const int64_t jend = jb.tile0 + tiles_of(jb.n);
        const int64_t tend = t1 < jend ? t1 : jend;
        uint64_t st[5];
        load_stats(stats, j, st);
        const Desc d = make_desc_f32(jb, st);
        if (t0 == jb.tile0 && tid == 0) {
            Meta m = {d.imin, (int64_t)d.bits, d.nbytes, (int64_t)d.status};
            meta[j] = m;
        }
        if (d.status == 0 && d.bits != 0) {
            const float* __restrict__ x = (const float*)jb.src;
            const float low = jb.low, dx = jb.dxq;
            const uint32_t b = (uint32_t)d.bits;
            const uint32_t magic = (uint32_t)((0x100000000ull + b - 1) / b);
            const int32_t pmin = (int32_t)d.pmin;
            const uint32_t imin = (uint32_t)(int32_t)d.imin, pix = (uint32_t)jb.pixels;
            const bool per = jb.periodic == 1;
            // walk the range backwards: the tiles this workgroup reduced last are the likeliest to
            // still sit in its XCD's L2
            for (int64_t t = tend - 1; t >= t0; --t) {
                const int64_t base = (t - jb.tile0) * kTile;
                uint32_t* stage = lds + buf * kTile;
                if (base + kTile <= jb.n) {
                    float4 v[kTile / (4 * kThreads)];
#pragma unroll
                    for (int k = 0; k < kTile / (4 * kThreads); ++k)
                        v[k] = *reinterpret_cast<const float4*>(x + base + 4 * (k * kThreads + tid));
#pragma unroll
                    for (int k = 0; k < kTile / (4 * kThreads); ++k) {
                        const float f[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
                        uint4 r;
                        uint32_t* rp = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const int32_t q = (int32_t)quant_floor(f[c], low, dx);
                            uint32_t uq = (uint32_t)q;
                            if (per && q < pmin) uq += pix;         // bound(), group.go:374-382
                            rp[c] = uq - imin;                      // group.go:247
                        }
                        *reinterpret_cast<uint4*>(stage + 4 * (k * kThreads + tid)) = r;
                    }
                } else {
                    for (int i = tid; i < kTile; i += kThreads) {
                        uint32_t r = 0;
                        if (base + i < jb.n) {
                            const int32_t q = (int32_t)quant_floor(x[base + i], low, dx);
                            uint32_t uq = (uint32_t)q;
                            if (per && q < pmin) uq += pix;
                            r = uq - imin;
                        }
                        stage[i] = r;
                    }
                }
                __syncthreads();
                const int64_t tile_byte0 = base / 8 * b;           // base is a multiple of 64
                int64_t tile_bytes = (int64_t)kTile / 8 * b;
                if (tile_byte0 + tile_bytes > d.nbytes) tile_bytes = d.nbytes - tile_byte0;
                store_tile_words<uint32_t>(stage, (uint8_t*)jb.dst, tile_byte0, tile_bytes, b, magic);
                buf ^= 1;
            }
        }
        t0 = tend;
        ++j;
    }
}

// ------------------------------------------------------------------------------------------------
// pack pass, int64 blocks (intGroup.writeData, and raw bit.BufferedArray in explicit mode).
// b <= 32: 32-bit staging, double buffered; b > 32: 64-bit staging, single buffer, two barriers.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_pack_i64(const DevJob* __restrict__ jobs, int njobs,
                                                       int64_t total_tiles,
                                                       const uint64_t* __restrict__ stats,
                                                       Meta* __restrict__ meta) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[kLdsBytes / 4];
    const int tid = threadIdx.x;
    int64_t t0 = total_tiles * blockIdx.x / gridDim.x;
    const int64_t t1 = total_tiles * (blockIdx.x + 1) / gridDim.x;
    if (t0 >= t1) return;
    int j = find_job(jobs, njobs, t0);
    int buf = 0;
    if (tid < 16) lds[2 * kTile + tid] = 0;

    while (t0 < t1) {
        const DevJob jb = jobs[j];
        const int64_t jend = jb.tile0 + tiles_of(jb.n);
        const int64_t tend = t1 < jend ? t1 : jend;
        uint64_t st[5] = {0, 0, 0, 0, 0};
        if (jb.mode == kModeStats) load_stats(stats, j, st);
        const Desc d = make_desc_i64(jb, st);
        if (t0 == jb.tile0 && tid == 0) {
            Meta m = {d.imin, (int64_t)d.bits, d.nbytes, (int64_t)d.status};
            meta[j] = m;
        }
        if (d.status == 0 && d.bits != 0) {
            const int64_t* __restrict__ x = (const int64_t*)jb.src;
            const uint32_t b = (uint32_t)d.bits;
            const uint32_t magic = (uint32_t)((0x100000000ull + b - 1) / b);
            const uint64_t imin = (uint64_t)d.imin;
            const uint64_t mask = b == 64 ? ~0ull : ((1ull << b) - 1);   // bit.go:104-107
            const bool narrow = b <= 32;
            for (int64_t t = tend - 1; t >= t0; --t) {
                const int64_t base = (t - jb.tile0) * kTile;
                uint32_t* stage32 = lds + buf * kTile;
                uint64_t* stage64 = reinterpret_cast<uint64_t*>(lds);
                if (!narrow) __syncthreads();                      // single buffer: readers done
                if (base + kTile <= jb.n) {
                    longlong2 v[kTile / (2 * kThreads)];
#pragma unroll
                    for (int k = 0; k < kTile / (2 * kThreads); ++k)
                        v[k] = *reinterpret_cast<const longlong2*>(x + base + 2 * (k * kThreads + tid));
#pragma unroll
                    for (int k = 0; k < kTile / (2 * kThreads); ++k) {
                        const uint64_t a = ((uint64_t)v[k].x - imin) & mask;
                        const uint64_t c = ((uint64_t)v[k].y - imin) & mask;
                        const int e = 2 * (k * kThreads + tid);
                        if (narrow) *reinterpret_cast<uint2*>(stage32 + e) = make_uint2((uint32_t)a, (uint32_t)c);
                        else *reinterpret_cast<ulonglong2*>(stage64 + e) = make_ulonglong2(a, c);
                    }
                } else {
                    for (int i = tid; i < kTile; i += kThreads) {
                        uint64_t r = 0;
                        if (base + i < jb.n) r = ((uint64_t)x[base + i] - imin) & mask;
                        if (narrow) stage32[i] = (uint32_t)r; else stage64[i] = r;
                    }
                }
                __syncthreads();
                const int64_t tile_byte0 = base / 8 * b;
                int64_t tile_bytes = (int64_t)kTile / 8 * b;
                if (tile_byte0 + tile_bytes > d.nbytes) tile_bytes = d.nbytes - tile_byte0;
                if (narrow) {
                    store_tile_words<uint32_t>(stage32, (uint8_t*)jb.dst, tile_byte0, tile_bytes, b, magic);
                    buf ^= 1;
                } else {
                    store_tile_words<uint64_t>(stage64, (uint8_t*)jb.dst, tile_byte0, tile_bytes, b, magic);
                }
            }
            if (!narrow) __syncthreads();
        }
        t0 = tend;
        ++j;
    }
}

// ------------------------------------------------------------------------------------------------
// decode pass.  The stream of a block may start at any byte (file offsets are byte granular,
// block_index.go:16-23): words are read from the enclosing aligned 64-bit words and funnel-shifted.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t stream_word(const uint64_t* __restrict__ a, int64_t k, int64_t last,
                                                uint32_t sh) {
    // a = aligned base, k = word index in the (shifted) stream, last = last aligned word holding data
    uint64_t lo = k <= last ? a[k] : 0;
    if (sh == 0) return lo;
    uint64_t hi = (k + 1) <= last ? a[k + 1] : 0;
    return (lo >> sh) | (hi << (64 - sh));
}

// stage the words of one tile (tile = kTile elements = kTile/64*b words) plus one look-ahead word
__device__ __forceinline__ void stage_tile_words(uint64_t* __restrict__ lw, const DevJob& jb, int64_t base,
                                                 uint32_t b) {
    const uintptr_t p = (uintptr_t)jb.src;
    const uint64_t* a = reinterpret_cast<const uint64_t*>(p & ~(uintptr_t)7);
    const uint32_t sh = (uint32_t)(p & 7) * 8;
    const int64_t nbytes = array_bytes(b, jb.n);
    const int64_t last = (int64_t)(((p + nbytes - 1) & ~(uintptr_t)7) - (p & ~(uintptr_t)7)) >> 3;
    const int64_t w0 = base / 64 * b;
    const int nw = kTile / 64 * (int)b + 1;
    for (int w = threadIdx.x; w < nw; w += kThreads) lw[w] = stream_word(a, w0 + w, last, sh);
}

__device__ __forceinline__ uint32_t extract32(const uint32_t* __restrict__ l32, uint32_t i, uint32_t b,
                                              uint32_t mask) {
    const uint32_t bitpos = i * b;
    const uint32_t di = bitpos >> 5, sh = bitpos & 31;
    const uint64_t win = (uint64_t)l32[di] | ((uint64_t)l32[di + 1] << 32);
    return (uint32_t)(win >> sh) & mask;
}
__device__ __forceinline__ uint64_t extract64(const uint64_t* __restrict__ l64, uint32_t i, uint32_t b,
                                              uint64_t mask) {
    const uint32_t bitpos = i * b;
    const uint32_t wi = bitpos >> 6, sh = bitpos & 63;
    uint64_t v = l64[wi] >> sh;
    if (sh != 0) v |= l64[wi + 1] << (64 - sh);
    return v & mask;
}

// floatGroup.readData, group.go:299-310.  NARROW: bits <= 32 and min, min+2^bits, pixels fit int32.
template <bool NARROW>
__global__ __launch_bounds__(kThreads) void k_unpack_f32(const DevJob* __restrict__ jobs, int njobs,
                                                         int64_t total_tiles) {
    __shared__ __attribute__((aligned(16))) uint64_t lw[kTile / 64 * (NARROW ? 32 : 64) + 2];
    const int tid = threadIdx.x;
    int64_t t0 = total_tiles * blockIdx.x / gridDim.x;
    const int64_t t1 = total_tiles * (blockIdx.x + 1) / gridDim.x;
    if (t0 >= t1) return;
    int j = find_job(jobs, njobs, t0);

    while (t0 < t1) {
        const DevJob jb = jobs[j];
        const int64_t jend = jb.tile0 + tiles_of(jb.n);
        const int64_t tend = t1 < jend ? t1 : jend;
        float* __restrict__ out = (float*)jb.dst;
        const uint32_t b = (uint32_t)jb.bits;
        const float low = jb.low, dx = jb.dxq;
        const bool per = jb.periodic == 1;
        const uint64_t mask64 = b == 0 ? 0 : (b == 64 ? ~0ull : ((1ull << b) - 1));

        for (int64_t t = t0; t < tend; ++t) {
            const int64_t base = (t - jb.tile0) * kTile;
            if (b != 0) {
                __syncthreads();
                stage_tile_words(lw, jb, base, b);
                __syncthreads();
            }
            const uint64_t key = dither_key(jb.seed, jb.block, (uint64_t)base >> 32);
            const uint32_t ka = (uint32_t)key, kb = (uint32_t)(key >> 32);
#pragma unroll
            for (int k = 0; k < kTile / (4 * kThreads); ++k) {
                const uint32_t i0 = 4 * (k * kThreads + tid);
                float r[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t i = i0 + c;
                    double qd;
                    if (NARROW) {
                        const uint32_t v = b == 0 ? 0 : extract32(reinterpret_cast<const uint32_t*>(lw), i, b, (uint32_t)mask64);
                        int32_t q = (int32_t)jb.min + (int32_t)v;            // group.go:262
                        if (per) {                                          // bound(buf, 0, pixels)
                            if (q < 0) q += (int32_t)jb.pixels;
                            else if (q >= (int32_t)jb.pixels) q -= (int32_t)jb.pixels;
                        }
                        qd = (double)q;
                    } else {
                        const uint64_t v = b == 0 ? 0 : extract64(lw, i, b, mask64);
                        int64_t q = (int64_t)((uint64_t)jb.min + v);
                        if (per) {
                            if (q < 0) q = (int64_t)((uint64_t)q + (uint64_t)jb.pixels);
                            else if (q >= jb.pixels) q = (int64_t)((uint64_t)q - (uint64_t)jb.pixels);
                        }
                        qd = (double)q;
                    }
                    const uint32_t z = dither_bits((uint32_t)(base + i), ka, kb);
                    const double u = (double)z * (1.0 / 4294967296.0);
                    // group.go:308: dx*float32(float64(q) + u) + low, float32 multiply then add
                    r[c] = __fadd_rn(__fmul_rn(dx, (float)__dadd_rn(qd, u)), low);
                }
                if (base + i0 + 3 < jb.n) {
                    *reinterpret_cast<float4*>(out + base + i0) = make_float4(r[0], r[1], r[2], r[3]);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (base + i0 + c < jb.n) out[base + i0 + c] = r[c];
                }
            }
        }
        t0 = tend;
        ++j;
    }
}

// intGroup.readData, group.go:257-263; with kind == 0 (raw) it is (*bit.Array).Slice, bit.go:29-82
__global__ __launch_bounds__(kThreads) void k_unpack_i64(const DevJob* __restrict__ jobs, int njobs,
                                                         int64_t total_tiles) {
    __shared__ __attribute__((aligned(16))) uint64_t lw[kTile / 64 * 64 + 2];
    const int tid = threadIdx.x;
    int64_t t0 = total_tiles * blockIdx.x / gridDim.x;
    const int64_t t1 = total_tiles * (blockIdx.x + 1) / gridDim.x;
    if (t0 >= t1) return;
    int j = find_job(jobs, njobs, t0);

    while (t0 < t1) {
        const DevJob jb = jobs[j];
        const int64_t jend = jb.tile0 + tiles_of(jb.n);
        const int64_t tend = t1 < jend ? t1 : jend;
        int64_t* __restrict__ out = (int64_t*)jb.dst;
        const uint32_t b = (uint32_t)jb.bits;
        const uint64_t mask64 = b == 0 ? 0 : (b == 64 ? ~0ull : ((1ull << b) - 1));
        const uint64_t mn = (uint64_t)jb.min;

        for (int64_t t = t0; t < tend; ++t) {
            const int64_t base = (t - jb.tile0) * kTile;
            if (b != 0) {
                __syncthreads();
                stage_tile_words(lw, jb, base, b);
                __syncthreads();
            }
#pragma unroll
            for (int k = 0; k < kTile / (2 * kThreads); ++k) {
                const uint32_t i0 = 2 * (k * kThreads + tid);
                const uint64_t v0 = b == 0 ? 0 : extract64(lw, i0, b, mask64);
                const uint64_t v1 = b == 0 ? 0 : extract64(lw, i0 + 1, b, mask64);
                const int64_t r0 = (int64_t)(mn + v0), r1 = (int64_t)(mn + v1);
                if (base + i0 + 1 < jb.n) {
                    *reinterpret_cast<longlong2*>(out + base + i0) = make_longlong2(r0, r1);
                } else if (base + i0 < jb.n) {
                    out[base + i0] = r0;
                }
            }
        }
        t0 = tend;
        ++j;
    }
}

// ------------------------------------------------------------------------------------------------
// exact path for blocks whose periodicMin depends on element order (q outside [0,pixels), NaN,
// |q| >= 2^63, pixels >= 2^31).  Literal structure of floatGroup.writeData: quantize to int64
// (group.go:318-320), sequential periodicMin (384-409), bound (374-382), then the intGroup path.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_quantize_i64(const float* __restrict__ x, int64_t n, float low,
                                                           float dx, int64_t* __restrict__ q) {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        const float v = quant_floor(x[i], low, dx);
        // int64(float64) on amd64 (CVTTSD2SQ): NaN and out-of-range give MinInt64
        q[i] = (fabsf(v) < 9223372036854775808.0f) ? (int64_t)v : INT64_MIN;
    }
}

__device__ __forceinline__ int64_t pdist(int64_t x, int64_t x0, int64_t pixels) {  // group.go:412-420
    const int64_t d = (int64_t)((uint64_t)x - (uint64_t)x0);
    if (d >= 0) {
        if (d > (int64_t)((uint64_t)pixels - (uint64_t)d)) return (int64_t)((uint64_t)d - (uint64_t)pixels);
    } else {
        if (d < (int64_t)(0 - ((uint64_t)d + (uint64_t)pixels))) return (int64_t)((uint64_t)pixels + (uint64_t)d);
    }
    return d;
}

// One wavefront walks the block in order.  64 elements are tested at once against the current arc;
// a chunk with no element that changes the state is skipped, otherwise lane order is replayed
// sequentially from the first such element.  state[0]=x0, state[1]=width, state[2]=done(1: returned 0),
// state[3]=next element; a launch processes at most `budget` elements so no launch runs unbounded.
__global__ __launch_bounds__(64) void k_periodic_min_seq(const int64_t* __restrict__ q, int64_t n, int64_t pixels,
                                                         int64_t budget, int64_t* __restrict__ state) {
    int64_t x0 = state[0], width = state[1], done = state[2], i = state[3];
    if (i == 0 && n > 0 && done == 0) { x0 = q[0]; width = 1; }
    const int64_t stop = (i + budget < n) ? i + budget : n;
    const int lane = threadIdx.x;
    while (i < stop && done == 0) {
        const int64_t idx = i + lane;
        const bool valid = idx < stop;
        const int64_t xi = valid ? q[idx] : 0;
        int64_t x1 = (int64_t)((uint64_t)x0 + (uint64_t)width - 1);
        if (x1 >= pixels) x1 = (int64_t)((uint64_t)x1 - (uint64_t)pixels);
        const int64_t d0 = pdist(xi, x0, pixels), d1 = pdist(xi, x1, pixels);
        const bool inside = !valid || (d0 > 0 && d1 < 0);
        const unsigned long long changers = __ballot(!inside);
        if (changers == 0) { i += 64; continue; }
        const int first = __ffsll((long long)changers) - 1;
        // replay sequentially from `first` to the end of this chunk (all lanes compute the same)
        for (int l = first; l < 64 && done == 0; ++l) {
            const int64_t id2 = i + l;
            if (id2 >= stop) break;
            const int64_t xv = __shfl(xi, l);
            int64_t y1 = (int64_t)((uint64_t)x0 + (uint64_t)width - 1);
            if (y1 >= pixels) y1 = (int64_t)((uint64_t)y1 - (uint64_t)pixels);
            const int64_t e0 = pdist(xv, x0, pixels), e1 = pdist(xv, y1, pixels);
            if (e0 > 0 && e1 < 0) continue;
            if (e1 > (int64_t)(0 - (uint64_t)e0)) width = (int64_t)((uint64_t)width + (uint64_t)e1);
            else {
                x0 = (int64_t)((uint64_t)x0 + (uint64_t)e0);
                if (x0 < 0) x0 = (int64_t)((uint64_t)x0 + (uint64_t)pixels);
                width = (int64_t)((uint64_t)width - (uint64_t)e0);
            }
            if (width > pixels / 2) done = 1;
        }
        i += 64;
    }
    if (i > stop) i = stop;
    if (lane == 0) { state[0] = x0; state[1] = width; state[2] = done; state[3] = i; }
}

__global__ __launch_bounds__(kThreads) void k_bound_i64(int64_t* __restrict__ q, int64_t n, int64_t min,
                                                        int64_t pixels) {   // group.go:374-382
    const int64_t top = (int64_t)((uint64_t)min + (uint64_t)pixels);
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        int64_t v = q[i];
        if (v < min) v = (int64_t)((uint64_t)v + (uint64_t)pixels);
        else if (v >= top) v = (int64_t)((uint64_t)v - (uint64_t)pixels);
        q[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// launch wrappers (called from mnw_api.cpp)
// ------------------------------------------------------------------------------------------------
void launch_reduce_f32(hipStream_t s, int grid, const DevJob* jobs, int njobs, int64_t tiles, uint64_t* stats) {
    hipLaunchKernelGGL(k_reduce_f32, dim3(grid), dim3(kThreads), 0, s, jobs, njobs, tiles, stats);
}
void launch_reduce_i64(hipStream_t s, int grid, const DevJob* jobs, int njobs, int64_t tiles, uint64_t* stats) {
    hipLaunchKernelGGL(k_reduce_i64, dim3(grid), dim3(kThreads), 0, s, jobs, njobs, tiles, stats);
}
void launch_pack_f32(hipStream_t s, int grid, const DevJob* jobs, int njobs, int64_t tiles, const uint64_t* stats, Meta* meta) {
    hipLaunchKernelGGL(k_pack_f32, dim3(grid), dim3(kThreads), 0, s, jobs, njobs, tiles, stats, meta);
}
void launch_pack_i64(hipStream_t s, int grid, const DevJob* jobs, int njobs, int64_t tiles, const uint64_t* stats, Meta* meta) {
    hipLaunchKernelGGL(k_pack_i64, dim3(grid), dim3(kThreads), 0, s, jobs, njobs, tiles, stats, meta);
}
void launch_unpack_f32(hipStream_t s, int grid, bool narrow, const DevJob* jobs, int njobs, int64_t tiles) {
    if (narrow) hipLaunchKernelGGL(k_unpack_f32<true>, dim3(grid), dim3(kThreads), 0, s, jobs, njobs, tiles);
    else hipLaunchKernelGGL(k_unpack_f32<false>, dim3(grid), dim3(kThreads), 0, s, jobs, njobs, tiles);
}
void launch_unpack_i64(hipStream_t s, int grid, const DevJob* jobs, int njobs, int64_t tiles) {
    hipLaunchKernelGGL(k_unpack_i64, dim3(grid), dim3(kThreads), 0, s, jobs, njobs, tiles);
}
void launch_quantize_i64(hipStream_t s, int grid, const float* x, int64_t n, float low, float dx, int64_t* q) {
    hipLaunchKernelGGL(k_quantize_i64, dim3(grid), dim3(kThreads), 0, s, x, n, low, dx, q);
}
void launch_periodic_min_seq(hipStream_t s, const int64_t* q, int64_t n, int64_t pixels, int64_t budget, int64_t* state) {
    hipLaunchKernelGGL(k_periodic_min_seq, dim3(1), dim3(64), 0, s, q, n, pixels, budget, state);
}
void launch_bound_i64(hipStream_t s, int grid, int64_t* q, int64_t n, int64_t min, int64_t pixels) {
    hipLaunchKernelGGL(k_bound_i64, dim3(grid), dim3(kThreads), 0, s, q, n, min, pixels);
}

}  // namespace mnw
