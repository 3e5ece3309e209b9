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
// Block pointers come out of the job records, so the compiler only knows them as generic (flat)
// addresses; casting to the global address space turns flat_load/flat_store into global_load/
// global_store (no LDS-aperture check, not counted in lgkmcnt).
#define MNW_GLOBAL __attribute__((address_space(1)))
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
typedef int64_t i64x2 __attribute__((ext_vector_type(2)));

template <typename V, typename T>
__device__ __forceinline__ V ldg(const T* p) { return *(const MNW_GLOBAL V*)(p); }
template <typename V, typename T>
__device__ __forceinline__ void stg(T* p, V v) { *(MNW_GLOBAL V*)(p) = v; }

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() would also drain
// vmcnt, i.e. the prefetched global loads of the next tile that are meant to stay in flight.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

// Wave reductions on the DPP crossbar (no LDS traffic, unlike __shfl_xor's ds_bpermute): butterfly inside
// each row of 16 lanes (two quad permutes, half-row mirror, row mirror), then the four row results via
// v_readlane.  The result is wave-uniform.
template <typename T, typename Op>
__device__ __forceinline__ T wave_reduce_dpp(T v, Op op) {
    static_assert(sizeof(T) == 4, "32-bit values");
    auto dpp = [](T x, auto ctrl) {
        return __builtin_bit_cast(T, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v = op(v, dpp(v, std::integral_constant<int, 0xB1>{}));      // quad_perm [1,0,3,2]
    v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}));      // quad_perm [2,3,0,1]
    v = op(v, dpp(v, std::integral_constant<int, 0x141>{}));     // row_half_mirror
    v = op(v, dpp(v, std::integral_constant<int, 0x140>{}));     // row_mirror
    const T r0 = __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const T r1 = __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const T r2 = __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const T r3 = __builtin_bit_cast(T, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return op(op(r0, r1), op(r2, r3));
}
// the same over each half of the wave: lo = lanes 0..31, hi = lanes 32..63 (both wave-uniform)
__device__ __forceinline__ void half_wave_max_dpp(uint32_t v, uint32_t& lo, uint32_t& hi) {
    auto dpp = [](uint32_t x, auto ctrl) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, decltype(ctrl)::value, 0xf, 0xf, true); };
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, dpp(v, std::integral_constant<int, 0xB1>{}));
    v = mx(v, dpp(v, std::integral_constant<int, 0x4E>{}));
    v = mx(v, dpp(v, std::integral_constant<int, 0x141>{}));
    v = mx(v, dpp(v, std::integral_constant<int, 0x140>{}));
    lo = mx((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16));
    hi = mx((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48));
}
__device__ __forceinline__ float wave_min_dpp(float v) { return wave_reduce_dpp(v, [](float a, float b) { return fminf(a, b); }); }
__device__ __forceinline__ float wave_max_dpp(float v) { return wave_reduce_dpp(v, [](float a, float b) { return fmaxf(a, b); }); }
__device__ __forceinline__ uint32_t wave_max_dpp(uint32_t v) { return wave_reduce_dpp(v, [](uint32_t a, uint32_t b) { return a > b ? a : b; }); }

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

// The same, tried first at the position the tile would have if all jobs had as many tiles as the first (the usual case:
// a column cut into equal blocks): two or three scalar loads instead of a binary search.  The decode kernels run tens of
// thousands of short workgroups, each of which starts with this.
template <int TILE>
__device__ __forceinline__ int find_job_guess(const DevJob* jobs, int njobs, int64_t t) {
    const int64_t per = (jobs[0].n + TILE - 1) / TILE;
    const int64_t g = per > 0 ? t / per : 0;
    const int j = (int)(g < njobs - 1 ? g : njobs - 1);
    if (jobs[j].tile0 <= t && (j + 1 == njobs || t < jobs[j + 1].tile0)) return j;
    return find_job(jobs, njobs, t);
}

__device__ __forceinline__ int64_t tiles_of(int64_t n) { return (n + kTile - 1) / kTile; }
__device__ __forceinline__ int64_t tiles_of_d(int64_t n) { return (n + kTileD - 1) / kTileD; }

// go/group.go:318-319: float32 subtract, IEEE float32 divide, then floor.  floor() of the float32
// quotient is exact in float32, and float64() of it is the same number, so the float64 detour of
// the Go text changes nothing.  (hipcc: correctly rounded f32 division, no contraction, denormals on.)
__device__ __forceinline__ float quant_floor(float x, float low, float dx) {
    return floorf(__fdiv_rn(__fsub_rn(x, low), dx));
}

#include "mnw_gomath_fast.h"

// ------------------------------------------------------------------------------------------------
// reduce pass, float32 blocks.
// q(x) = floor(fl(fl(x-low)/dx)) is monotone non-decreasing in x for dx > 0 (every step -- rounded
// subtract, rounded divide by a positive number, floor -- is monotone), so the extreme values of q
// over a block are q of the extreme values of x.  The pass therefore reduces the raw floats (no
// division per element) and the pack pass quantizes just the four survivors:
//   stat[0] = key(max x)   stat[1] = ~key(min x)
//   stat[2] = key(max of x <  xh)   stat[3] = ~key(min of x >= xh)     xh = least float with q >= pixels/2
//   stat[4] = 1 if a NaN or an infinity was seen
// key() is the order-preserving map float32 -> uint32; everything is combined with atomic unsigned
// max, so zero-filled memory is the identity (key 0 is a NaN pattern, never a real value).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fkey(float f) {
    const uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float funkey(uint32_t k) {
    return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xffffffffu));
}

struct FStat {
    float mn, mx, mxlo, mnhi;
    bool bad;
};

__device__ __forceinline__ void fstat_add(FStat& s, float x, float xh) {
    s.bad |= !(fabsf(x) < INFINITY);                 // NaN, +-Inf
    s.mn = fminf(s.mn, x);
    s.mx = fmaxf(s.mx, x);
    const bool lo = x < xh;
    s.mxlo = fmaxf(s.mxlo, lo ? x : -INFINITY);
    s.mnhi = fminf(s.mnhi, lo ? INFINITY : x);
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

// The job's statistics = unsigned max over its stat lines; lane l reads line l % kStatSlots.
// ATOMIC: agent-scope (sc1) loads, for words that other workgroups of the same launch have just
// updated with agent-scope atomics; otherwise plain loads (after a kernel boundary).
template <bool ATOMIC>
__device__ __forceinline__ void load_stats(const uint64_t* __restrict__ stats, int j, uint64_t out[5]) {
    const int l = lane_id();
    const uint64_t* rec = stats + ((int64_t)j * kStatSlots + (l % kStatSlots)) * kStatWords;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        uint64_t v;
        if (ATOMIC) v = __hip_atomic_load(rec + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else v = rec[k];
        out[k] = wave_max_u64(v);
    }
}

__device__ __forceinline__ bool quant_i64(float x, float low, float dx, int64_t* q) {
    const float v = quant_floor(x, low, dx);
    if (!(fabsf(v) < 9223372036854775808.0f)) return false;     // int64(NaN/huge) is platform defined in Go
    *q = (int64_t)v;
    return true;
}

// float blocks.  For q in [0,pixels) periodicMin (group.go:384-409) is order independent: it is the
// start of the smallest ring arc covering every q if that arc is <= pixels/2 wide, else 0
// (DESIGN.md "periodicMin as reductions"; tests/test_oracle_golden.py::test_periodic_reduction_identity).
// Anything else (q out of range, NaN, huge) is order dependent -> exact sequential path.
__device__ __forceinline__ Desc make_desc_f32(const DevJob& jb, const uint64_t st[5]) {
    Desc d = {0, 0, 0, 0, 0};
    if (st[4] != 0) { d.status = (int)kNeedsExact; return d; }
    const bool has_lo = st[2] != 0, has_hi = st[3] != 0;
    int64_t mx0, mn0, mxlo = 0, mnhi = 0;
    bool ok = quant_i64(funkey((uint32_t)st[0]), jb.low, jb.dxq, &mx0);
    ok &= quant_i64(funkey(~(uint32_t)st[1]), jb.low, jb.dxq, &mn0);
    if (has_lo) ok &= quant_i64(funkey((uint32_t)st[2]), jb.low, jb.dxq, &mxlo);
    if (has_hi) ok &= quant_i64(funkey(~(uint32_t)st[3]), jb.low, jb.dxq, &mnhi);
    if (!ok) { d.status = (int)kNeedsExact; return d; }
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

// A float block's descriptor fits one naturally aligned 8-byte word, written by ONE store and read
// by ONE load, so it needs no ordering against anything else (it is its own flag):
//   bit 63 ready | bits 62..55 status code | bit 54 pmin == imin (else pmin == 0) | bits 53..47 width
//   | bits 31..0 imin (int32: q in [0,pixels) or int32 range on this path)
constexpr uint64_t kGranReady = 1ull << 63;
__device__ __forceinline__ uint64_t gran_encode(const Desc& d) {
    const uint32_t code = d.status == 0 ? 0u : d.status == (int)kNeedsExact ? 1u : d.status == -5 ? 2u : 3u;
    return kGranReady | ((uint64_t)code << 55) | ((uint64_t)(d.pmin == d.imin) << 54) |
           ((uint64_t)(uint32_t)d.bits << 47) | (uint64_t)(uint32_t)(int32_t)d.imin;
}
__device__ __forceinline__ Desc gran_decode(uint64_t g, int64_t n) {
    Desc d;
    const uint32_t code = (uint32_t)(g >> 55) & 0xff;
    d.status = code == 0 ? 0 : code == 1 ? (int)kNeedsExact : code == 2 ? -5 : -6;
    d.bits = (int)((g >> 47) & 0x7f);
    d.imin = (int64_t)(int32_t)(uint32_t)g;
    d.pmin = ((g >> 54) & 1) ? d.imin : 0;
    d.nbytes = d.bits == 0 ? 0 : array_bytes(d.bits, n);
    return d;
}

// Per-job synchronisation record (one 64-byte line): tiles whose statistics have been folded in, and
// the descriptor word published by the workgroup whose arrival completed the block.
struct JobSync {
    uint32_t count;
    uint32_t pad0;
    uint64_t granule;
    uint64_t pad1[6];
};

constexpr int kChunkF = 4 * 4 * kThreads;        // reduce pass: 4 x 16-byte loads per thread in flight

// Arrivals whose statistics atomics have been issued but not yet counted in.  Counting in needs the
// atomics drained first (s_waitcnt vmcnt(0) in the wave that issued them: wave 0); doing that right
// after the reduce would expose a full memory round trip per stage, so in the fused launch the count
// is deferred to a point of the following pack phase where wave 0 has just drained its loads anyway.
constexpr int kMaxPend = 4;
struct Pending {
    int cnt;
    int j[kMaxPend];
    uint32_t n[kMaxPend];
};

// wave 0 only.  The block is complete with this arrival: derive its descriptor once, publish it as
// one 8-byte word and write the block's metadata.  Every other producer drained its statistics
// atomics before counting in, so agent-scope loads see all of them.
// (not inlined: it runs once per block, and inlined its float64 temporaries would raise the register
// count of every kernel that can complete a block)
__device__ __attribute__((noinline)) void finalize_job_f32(const DevJob* __restrict__ jobs, int j, const uint64_t* __restrict__ stats,
                                                 JobSync* __restrict__ sync, Meta* __restrict__ meta) {
    const DevJob jb = jobs[j];
    uint64_t st[5];
    load_stats<true>(stats, j, st);
    const Desc d = make_desc_f32(jb, st);
    if (lane_id() == 0) {
        Meta m = {d.imin, (int64_t)d.bits, d.nbytes, (int64_t)d.status};
        meta[j] = m;
        __hip_atomic_store(&sync[j].granule, gran_encode(d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// wave 0 only: count the pending arrivals in (returning adds; `ret` is meaningful in lane 0)
__device__ __forceinline__ void pend_issue(const Pending* pend, JobSync* __restrict__ sync, uint32_t (&ret)[kMaxPend]) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int cnt = pend->cnt;
#pragma unroll
    for (int k = 0; k < kMaxPend; ++k) {
        ret[k] = 0;
        if (k < cnt && lane_id() == 0)
            ret[k] = __hip_atomic_fetch_add(&sync[pend->j[k]].count, pend->n[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// wave 0 only: finalize the jobs this workgroup's arrival completed, then clear the list
__device__ __forceinline__ void pend_finish(Pending* pend, const DevJob* __restrict__ jobs, const uint64_t* __restrict__ stats,
                                            JobSync* __restrict__ sync, Meta* __restrict__ meta, const uint32_t (&ret)[kMaxPend]) {
    const int cnt = pend->cnt;
#pragma unroll
    for (int k = 0; k < kMaxPend; ++k) {
        if (k < cnt) {
            const int j = pend->j[k];
            const uint32_t before = __builtin_amdgcn_readfirstlane(ret[k]);
            if (before + pend->n[k] == (uint32_t)tiles_of(jobs[j].n)) finalize_job_f32(jobs, j, stats, sync, meta);
        }
    }
    if (lane_id() == 0) pend->cnt = 0;
}

// Reduce the tiles [t0, t1) of a launch-wide tile numbering (jobs [j0, j1) cover them).  Each
// workgroup folds its share of a block into the block's stat lines with one 5-lane atomic instruction
// and then counts its tiles in; the workgroup whose count completes the block derives the block's
// descriptor (min, width, ...) once, publishes it as one 8-byte word and writes the block's metadata.
// All cross-workgroup traffic goes through agent-scope atomics / sc1 accesses on words nothing else
// touches, so no cache can hand out a stale copy and no fence (L2 write-back) is needed.
__device__ __forceinline__ void reduce_range_f32(const DevJob* __restrict__ jobs, int j0, int j1, int64_t t0, int64_t t1,
                                                 uint64_t* __restrict__ stats, JobSync* __restrict__ sync,
                                                 Meta* __restrict__ meta, int slot,
                                                 float (*red)[kThreads / 64], int* redbad, Pending* pend, bool defer) {
    const int tid = threadIdx.x;
    if (t0 >= t1) return;
    int j = find_job(jobs + j0, j1 - j0, t0) + j0;
    while (t0 < t1) {
        const DevJob jb = jobs[j];
        const int64_t jend = jb.tile0 + tiles_of(jb.n);
        const int64_t tend = t1 < jend ? t1 : jend;
        const float* __restrict__ x = (const float*)jb.src;
        const float xh = jb.xh;
        FStat s = {INFINITY, -INFINITY, -INFINITY, INFINITY, false};

        // this workgroup's element range of the block; tile boundaries do not matter to a reduction
        int64_t c = (t0 - jb.tile0) * kTile;
        int64_t e1 = (tend - jb.tile0) * kTile;
        if (e1 > jb.n) e1 = jb.n;
        // No software double buffering: a rotating register pair makes hipcc wait for the prefetch at
        // the loop back-edge.  Bytes in flight come from occupancy and 8 independent 16-byte loads.
        for (; c + 2 * kChunkF <= e1; c += 2 * kChunkF) {
            f32x4 va[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) va[k] = ldg<f32x4>(x + c + 4 * (k * kThreads + tid));
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                fstat_add(s, va[k].x, xh); fstat_add(s, va[k].y, xh);
                fstat_add(s, va[k].z, xh); fstat_add(s, va[k].w, xh);
            }
        }
        for (; c + kChunkF <= e1; c += kChunkF) {
            f32x4 va[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) va[k] = ldg<f32x4>(x + c + 4 * (k * kThreads + tid));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                fstat_add(s, va[k].x, xh); fstat_add(s, va[k].y, xh);
                fstat_add(s, va[k].z, xh); fstat_add(s, va[k].w, xh);
            }
        }
        for (int64_t i = c + tid; i < e1; i += kThreads) fstat_add(s, ldg<float>(x + i), xh);

        float mn = wave_min(s.mn), mx = wave_max(s.mx), mxlo = wave_max(s.mxlo), mnhi = wave_min(s.mnhi);
        int bad = __any(s.bad);
        __syncthreads();
        if (lane_id() == 0) {
            red[0][wave_id()] = mn; red[1][wave_id()] = mx;
            red[2][wave_id()] = mxlo; red[3][wave_id()] = mnhi;
            redbad[wave_id()] = bad;
        }
        __syncthreads();
        if (tid < 64) {                               // wave 0 publishes
            if (tid < 5) {
                float a = red[0][0], b = red[1][0], c2 = red[2][0], d = red[3][0];
                int bd = redbad[0];
#pragma unroll
                for (int w = 1; w < kThreads / 64; ++w) {
                    a = fminf(a, red[0][w]); b = fmaxf(b, red[1][w]);
                    c2 = fmaxf(c2, red[2][w]); d = fminf(d, red[3][w]);
                    bd |= redbad[w];
                }
                // -inf / +inf mean "no element in that half"; they map to the identity (0)
                uint64_t val;
                if (tid == 0) val = bd ? 0 : fkey(b);
                else if (tid == 1) val = bd ? 0 : (uint32_t)~fkey(a);
                else if (tid == 2) val = (bd || c2 == -INFINITY) ? 0 : fkey(c2);
                else if (tid == 3) val = (bd || d == INFINITY) ? 0 : (uint32_t)~fkey(d);
                else val = bd ? 1 : 0;
                uint64_t* rec = stats + ((int64_t)j * kStatSlots + slot) * kStatWords;
                if (val != 0) atomicMax((unsigned long long*)(rec + tid), (unsigned long long)val);
            }
            int npend = 0;
            if (tid == 0) {
                const int k = pend->cnt;
                pend->j[k] = j;
                pend->n[k] = (uint32_t)(tend - t0);
                pend->cnt = npend = k + 1;
            }
            npend = __builtin_amdgcn_readfirstlane(npend);
            // count in now (full round trips) unless the caller hides that in the pack phase that follows
            if (!defer || npend == kMaxPend) {
                uint32_t ret[kMaxPend];
                pend_issue(pend, sync, ret);
                pend_finish(pend, jobs, stats, sync, meta, ret);
            }
        }
        t0 = tend;
        ++j;
    }
}

__global__ __launch_bounds__(kThreads, 8) void k_reduce_f32(const DevJob* __restrict__ jobs, int njobs,
                                                         int64_t total_tiles,
                                                         uint64_t* __restrict__ stats, JobSync* __restrict__ sync,
                                                         Meta* __restrict__ meta) {
    __shared__ float red[4][kThreads / 64];
    __shared__ int redbad[kThreads / 64];
    __shared__ Pending pend;
    if (threadIdx.x == 0) pend.cnt = 0;
    __syncthreads();
    const int64_t t0 = total_tiles * blockIdx.x / gridDim.x;
    const int64_t t1 = total_tiles * (blockIdx.x + 1) / gridDim.x;
    reduce_range_f32(jobs, 0, njobs, t0, t1, stats, sync, meta, blockIdx.x % kStatSlots, red, redbad, &pend, false);
}

// ------------------------------------------------------------------------------------------------
// reduce pass, int64 blocks (go/group.go:350-357 int64Min, bit.go:151-159 max of x-min)
// ------------------------------------------------------------------------------------------------
constexpr int kChunkI = 4 * 2 * kThreads;        // 4 x 16-byte loads per thread in flight

__global__ __launch_bounds__(kThreads, 8) void k_reduce_i64(const DevJob* __restrict__ jobs, int njobs,
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
        // kModeUnsigned (bit.ArrayBuffer.Bits on raw uint64 values, bit.go:151-159): flipping the sign
        // bit maps unsigned order onto signed order
        const int64_t flip = (jb.mode == kModeUnsigned) ? INT64_MIN : 0;

        int64_t c = (t0 - jb.tile0) * kTile;
        int64_t e1 = (tend - jb.tile0) * kTile;
        if (e1 > jb.n) e1 = jb.n;
        for (; c + kChunkI <= e1; c += kChunkI) {
            i64x2 va[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) va[k] = ldg<i64x2>(x + c + 2 * (k * kThreads + tid));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t p = va[k].x ^ flip, q = va[k].y ^ flip;
                mn = p < mn ? p : mn; mx = p > mx ? p : mx;
                mn = q < mn ? q : mn; mx = q > mx ? q : mx;
            }
        }
        for (int64_t i = c + tid; i < e1; i += kThreads) {
            const int64_t v = ldg<int64_t>(x + i) ^ flip;
            mn = v < mn ? v : mn; mx = v > mx ? v : mx;
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

// int64 blocks
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
// pack core.  32 consecutive elements at b bits are exactly b 32-bit words of the stream, so one
// lane packs 32 consecutive re-based values into b words with no cross-lane traffic, and because b
// is the same for the whole block, every lane is at the same bit position at the same time: the
// flush test is wave-uniform (scalar branch) and an element costs one v_lshl_or_b32.
//
// Tile = 8192 elements = 256 lanes x 32.  Data path of one tile:
//   coalesced 16-B global loads -> quantize/re-base -> LDS stage (natural order, 32-element groups
//   at a 36-dword stride so the transposed reads below are bank-conflict free)
//   -> each lane reads ITS 32 values (8 x ds_read_b128) -> packs -> writes its b words to LDS in
//   stream order -> coalesced 16-B LDS reads -> coalesced 16-B global stores.
// ------------------------------------------------------------------------------------------------
constexpr int kGroup = 32;                         // elements packed by one lane
constexpr int kStride32 = kGroup + 4;              // dwords between groups of 32-bit values
constexpr int kStride64 = 2 * kGroup + 4;          // dwords between groups of 64-bit values
constexpr int kLds32 = kThreads * kStride32 * 4;   // 36,864 B
static_assert(kTile == kThreads * kGroup, "one lane packs one group of the tile");

// position of element e of the tile in the padded 32-bit stage (dword index)
__device__ __forceinline__ int stage32_idx(int e) { return e + 4 * (e >> 5); }
__device__ __forceinline__ int stage64_idx(int e) { return 2 * e + 4 * (e >> 5); }   // dword index

// Pack this lane's 32 values (already < 2^b) and write its b words to out[0..b).
// b must be wave-uniform and live in an SGPR (readfirstlane) so the branches are scalar.
__device__ __forceinline__ void pack_group32(const uint32_t (&v)[kGroup], uint32_t b, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    uint32_t pos = 0;
#pragma unroll
    for (int i = 0; i < kGroup; ++i) {
        acc |= v[i] << pos;                        // bits above 32 fall off; picked up below
        pos += b;
        if (pos >= 32) {                           // wave-uniform
            *out++ = acc;
            pos -= 32;
            acc = pos ? v[i] >> (b - pos) : 0;     // the part of v[i] that did not fit
        }
    }
}

// pack_group32 with the width as a compile-time constant per case: straight-line code, immediate shifts,
// no scalar compare/branch per element (b is wave-uniform, 1..32).
__device__ __forceinline__ void pack_group32_switch(const uint32_t (&v)[kGroup], uint32_t b, uint32_t* __restrict__ out) {
#define MNW_PACK_CASE(B) case B: pack_group32(v, B, out); break;
    switch (b) {
        MNW_PACK_CASE(1) MNW_PACK_CASE(2) MNW_PACK_CASE(3) MNW_PACK_CASE(4) MNW_PACK_CASE(5) MNW_PACK_CASE(6) MNW_PACK_CASE(7)
        MNW_PACK_CASE(8) MNW_PACK_CASE(9) MNW_PACK_CASE(10) MNW_PACK_CASE(11) MNW_PACK_CASE(12) MNW_PACK_CASE(13)
        MNW_PACK_CASE(14) MNW_PACK_CASE(15) MNW_PACK_CASE(16) MNW_PACK_CASE(17) MNW_PACK_CASE(18) MNW_PACK_CASE(19)
        MNW_PACK_CASE(20) MNW_PACK_CASE(21) MNW_PACK_CASE(22) MNW_PACK_CASE(23) MNW_PACK_CASE(24) MNW_PACK_CASE(25)
        MNW_PACK_CASE(26) MNW_PACK_CASE(27) MNW_PACK_CASE(28) MNW_PACK_CASE(29) MNW_PACK_CASE(30) MNW_PACK_CASE(31)
        MNW_PACK_CASE(32)
        default: break;
    }
#undef MNW_PACK_CASE
}

// three-input float min / max without the compiler's canonicalising v_max_f32 x,x in front (a NaN among
// the inputs is flagged separately, so what the instruction does with it does not matter)
__device__ __forceinline__ float min3_raw(float a, float b, float c) {
    float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r;
}
// int32(floor(x)) in one instruction
__device__ __forceinline__ int32_t cvt_floor_i32(float x) {
    int32_t r; asm("v_cvt_flr_i32_f32_e32 %0, %1" : "=v"(r) : "v"(x)); return r;
}

// 33..64-bit values: each is the field pair (low 32 bits, high b-32 bits) of the same LSB-first stream.
// Register-light form for the rare wide path: the lane's 32 values are read from LDS one at a time
// (`in`, 8-byte aligned) and its b words go to a separate LDS region (`out`).
__device__ __forceinline__ void pack_group64_lds(const uint32_t* __restrict__ in, uint32_t b, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    uint32_t pos = 0;
    const uint32_t bh = b - 32;                    // 1..32
#pragma unroll 1
    for (int i = 0; i < kGroup; ++i) {
        const u32x2 w = *reinterpret_cast<const u32x2*>(in + 2 * i);
        const uint32_t lo = w.x, hi = w.y;
        // low field, width 32: always completes a word
        acc |= lo << pos;
        *out++ = acc;
        acc = pos ? lo >> (32 - pos) : 0;
        // high field, width bh
        acc |= hi << pos;
        pos += bh;
        if (pos >= 32) {
            *out++ = acc;
            pos -= 32;
            acc = pos ? hi >> (bh - pos) : 0;
        }
    }
}

// coalesced write-out of one tile's packed words from LDS: 16-byte chunks, then the ragged tail of
// the block's last tile byte-wise (ArrayBytes is a byte count, bit.go:23-25; pad bits are zero)
// units_per_lane > 0: every lane's words are followed by one empty 16-byte unit in LDS (pack_stride), unit c of the stream then
// sits at unit c + c / units_per_lane (units_per_lane = b / 4 is wave-uniform; c < 2^16)
template <int NT = kThreads>
__device__ __forceinline__ void store_tile(const uint32_t* __restrict__ lds_out, uint8_t* __restrict__ dst,
                                           int64_t tile_byte0, int64_t tile_bytes, int tid = threadIdx.x, uint32_t units_per_lane = 0) {
    const int full = (int)(tile_bytes >> 4);
    const uint32_t m = units_per_lane ? 0xffffffffu / units_per_lane + 1u : 0u;
    for (int c = tid; c < full; c += NT)
        stg<u32x4>(dst + tile_byte0 + 16 * (int64_t)c, *reinterpret_cast<const u32x4*>(lds_out + 4 * ((uint32_t)c + __umulhi((uint32_t)c, m))));
    const int rem = (int)(tile_bytes & 15);
    if (tid < rem) {
        const int k = tid;
        const uint32_t w = lds_out[4 * ((uint32_t)full + __umulhi((uint32_t)full, m)) + (k >> 2)];
        stg<uint8_t>(dst + tile_byte0 + 16 * (int64_t)full + k, (uint8_t)(w >> (8 * (k & 3))));
    }
}
// Dwords between the packed outputs of neighbouring lanes: b, but for b = 8, 16, 24, 32 one 16-byte unit more -- with a stride of 8,
// 16, 24 or 32 dwords the 16-byte writes of eight neighbouring lanes fall on two or four LDS bank groups instead of eight (measured
// on k_encode_reg_vec3: bank conflicts were 56 % of its LDS cycles with 16-bit blocks); b + 4 is an odd number of units.
__device__ __forceinline__ uint32_t pack_stride(uint32_t b) { return b + ((b & 7u) == 0 ? 4u : 0u); }

// the lane's index in the workgroup as a value the compiler cannot see through: address arithmetic built
// on it inside a loop is redone there (two or three instructions) instead of being hoisted out of the loop
// and kept -- or, once registers run out, spilled -- across the whole trip
__device__ __forceinline__ int opaque_tid() {
    int t;
    asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"((int)threadIdx.x));
    return t;
}

// true for a NONZERO quotient below the float32 normal range, where the float64-multiply argument does
// not apply.  Zero needs no redo: t == 0 gives 0 either way, and a float32 divisor is never within 2^-52
// (relative) of the even integer a product rounding to zero against a quotient of 2^-149 would need.
__device__ __forceinline__ bool quant_tiny(float r) {
    return ((__float_as_uint(r) & 0x7fffffffu) - 1u) < 0x007fffffu;
}

// go/group.go:318-319 with the division done as a float64 multiply by 1/dx.  For float32 t, dx the
// value RN32(RN64(double(t) * RN64(1/double(dx)))) equals the correctly rounded float32 quotient:
// the double product is within 2^-52 (relative) of t/dx, and a quotient of two 24-bit significands is
// never closer than 2^-49 (relative) to a float32 rounding boundary (DESIGN.md "Quantize").  The
// argument needs a normal float32 result; tiny quotients take the IEEE division.
__device__ __forceinline__ float quant_floor_fast(float x, float low, float dx, double rdx) {
    const float t = __fsub_rn(x, low);
    float r = (float)__dmul_rn((double)t, rdx);
    if (__builtin_expect(quant_tiny(r), 0)) r = __fdiv_rn(t, dx);
    return floorf(r);
}

// the float64-multiply quotient alone (no guard); see quant_floor_fast
__device__ __forceinline__ float quant_mul(float x, float low, double rdx) {
    return (float)__dmul_rn((double)__fsub_rn(x, low), rdx);
}


// ------------------------------------------------------------------------------------------------
// pack pass, float32 blocks (32-bit path: q in [0,pixels), pixels < 2^31, or int32-range q when
// not periodic)
// ------------------------------------------------------------------------------------------------
#include "mnw_diag.h"     // cycle stamps and counters of the diagnostic builds (tools/microbench); expands to nothing in the library

// Descriptor of job j for the pack phase.  After a kernel boundary it is simply there; inside the
// fused launch one lane polls the 8-byte word (relaxed agent-scope load, bounded spin: a timeout or
// another workgroup's timeout aborts the launch and the host reruns the batch through the two-kernel
// path).  `pref` is a value loaded earlier (0 if none).  Returns false when aborted.
template <bool FUSED>
__device__ __forceinline__ bool get_granule(JobSync* __restrict__ sync, uint32_t* __restrict__ abort_flag, int j,
                                            uint64_t pref, uint64_t* shared_slot, uint64_t* out) {
    if (!FUSED) { *out = sync[j].granule; return true; }
    if (threadIdx.x == 0) {
        uint64_t g = pref;
        uint32_t spins = 0;
        while (!(g & kGranReady)) {
            g = __hip_atomic_load(&sync[j].granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (g & kGranReady) break;
            __builtin_amdgcn_s_sleep(8);
            ++spins;
            if ((spins & 15) == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { g = 0; break; }
            if (spins > (1u << 18)) {            // ~0.3 s: something keeps the producers from running
                __hip_atomic_fetch_add(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                g = 0;
                break;
            }
        }
        *shared_slot = g;
    }
    __syncthreads();
    const uint64_t g = *shared_slot;
    __syncthreads();
    *out = g;
    return (g & kGranReady) != 0;
}

// Pack the tiles [t0, t1) of a launch-wide tile numbering.  Returns false if the fused launch aborted.
template <bool FUSED>
__device__ __forceinline__ bool pack_range_f32(uint32_t* __restrict__ lds, uint64_t* shared_slot,
                                               const DevJob* __restrict__ jobs, int j0, int j1, int64_t t0, int64_t t1,
                                               JobSync* __restrict__ sync, uint32_t* __restrict__ abort_flag,
                                               uint64_t pref, Pending* pend, const uint64_t* __restrict__ stats,
                                               Meta* __restrict__ meta) {
    constexpr int kV = kTile / (4 * kThreads);       // 16-byte loads per thread per tile (8)
    const int tid = threadIdx.x;
    // deferred arrivals of the preceding reduce phase (fused launch, wave 0): 0 = none,
    // 1 = to be counted in at the next point where wave 0 has drained its loads, 2 = counted in,
    // results to be looked at one tile later (they are older than that tile's loads)
    int pstate = (FUSED && wave_id() == 0 && pend->cnt > 0) ? 1 : 0;
    uint32_t pret[kMaxPend] = {0, 0, 0, 0};
    if (t0 >= t1) {
        if (FUSED && pstate == 1) { pend_issue(pend, sync, pret); pend_finish(pend, jobs, stats, sync, meta, pret); }
        return true;
    }
    MNW_STAMP_DECL
    MNW_STAMP_START
    int j = find_job(jobs + j0, j1 - j0, t0) + j0;

    while (t0 < t1) {
        const DevJob jb = jobs[j];
        const int64_t jend = jb.tile0 + tiles_of(jb.n);
        const int64_t tend = t1 < jend ? t1 : jend;
        const float* __restrict__ x = (const float*)jb.src;
        // The range is walked backwards (the tiles this workgroup reduced last are the likeliest to
        // still be cached).  Only the block's last tile can be ragged and it comes first, so every
        // prefetched tile is full.  The first tile's loads are issued before the descriptor is waited
        // for so that the latencies overlap.
        int64_t t = tend - 1;
        f32x4 v[kV];
        bool cur_full = (t - jb.tile0) * kTile + kTile <= jb.n;
        if (cur_full) {
#pragma unroll
            for (int k = 0; k < kV; ++k) v[k] = ldg<f32x4>(x + (t - jb.tile0) * kTile + 4 * (k * kThreads + tid));
        }
        uint64_t gran;
        if (!get_granule<FUSED>(sync, abort_flag, j, pref, shared_slot, &gran)) return false;
        pref = 0;
        const Desc d = gran_decode(gran, jb.n);
        const int ok = __builtin_amdgcn_readfirstlane(d.status == 0 && d.bits != 0);
        MNW_STAMP(0)      // preamble: job record, first loads issued, descriptor
        if (ok) {
            const float low = jb.low, dx = jb.dxq;
            const double rdx = jb.rdx;
            const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane(d.bits);
            const int32_t pmin = (int32_t)d.pmin;
            const uint32_t imin = (uint32_t)(int32_t)d.imin, pix = (uint32_t)jb.pixels;
            const bool per = jb.periodic == 1;
            for (; t >= t0; --t) {
                const int64_t base = (t - jb.tile0) * kTile;
                if (cur_full) {
                    // quantize through the float64 multiply; quotients too small for that argument
                    // (|r| below the float32 normal range) are redone with the IEEE division, for the
                    // whole tile at once so that the common path has no per-element branch
                    bool tiny = false;
#pragma unroll
                    for (int k = 0; k < kV; ++k) {
                        u32x4 r;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float rq = quant_mul(v[k][c], low, rdx);
                            tiny |= quant_tiny(rq);
                            const int32_t q = (int32_t)floorf(rq);
                            uint32_t uq = (uint32_t)q;
                            if (per && q < pmin) uq += pix;         // bound(), group.go:374-382
                            r[c] = uq - imin;                       // group.go:247
                        }
                        *reinterpret_cast<u32x4*>(lds + stage32_idx(4 * (k * kThreads + tid))) = r;
                    }
                    if (__builtin_expect(__any(tiny), 0)) {
#pragma unroll 1
                        for (int k = 0; k < kV; ++k) {
                            u32x4 r;
                            for (int c = 0; c < 4; ++c) {
                                const int32_t q = (int32_t)quant_floor(v[k][c], low, dx);
                                uint32_t uq = (uint32_t)q;
                                if (per && q < pmin) uq += pix;
                                r[c] = uq - imin;
                            }
                            *reinterpret_cast<u32x4*>(lds + stage32_idx(4 * (k * kThreads + tid))) = r;
                        }
                    }
                } else {
                    for (int i = tid; i < kTile; i += kThreads) {
                        uint32_t r = 0;
                        if (base + i < jb.n) {
                            const int32_t q = (int32_t)quant_floor(ldg<float>(x + base + i), low, dx);
                            uint32_t uq = (uint32_t)q;
                            if (per && q < pmin) uq += pix;
                            r = uq - imin;
                        }
                        lds[stage32_idx(i)] = r;
                    }
                }
                MNW_STAMP(1)  // wait for the tile's loads + quantize + stage
                if (FUSED) {
                    // wave 0 has just consumed every load it had in flight: anything older -- the
                    // statistics atomics of the reduce phase -- has completed too
                    if (pstate == 2) { pend_finish(pend, jobs, stats, sync, meta, pret); pstate = 0; }
                    else if (pstate == 1) { pend_issue(pend, sync, pret); pstate = 2; }
                }
                if (t - 1 >= t0) {                   // next tile: loads stay in flight across the barriers
#pragma unroll
                    for (int k = 0; k < kV; ++k) v[k] = ldg<f32x4>(x + base - kTile + 4 * (k * kThreads + tid));
                }
                cur_full = true;
                lds_barrier();
                MNW_STAMP(2)  // prefetch issue + barrier
                uint32_t g[kGroup];
#pragma unroll
                for (int k = 0; k < kGroup / 4; ++k) {
                    const u32x4 r = *reinterpret_cast<const u32x4*>(lds + tid * kStride32 + 4 * k);
                    g[4 * k] = r.x; g[4 * k + 1] = r.y; g[4 * k + 2] = r.z; g[4 * k + 3] = r.w;
                }
                lds_barrier();                      // everyone holds its group: the stage can be overwritten
                MNW_STAMP(3)  // transposed LDS read + barrier
                pack_group32(g, b, lds + tid * pack_stride(b));
                lds_barrier();
                MNW_STAMP(4)  // pack + barrier
                const int64_t tile_byte0 = base / 8 * b;
                int64_t tile_bytes = (int64_t)kTile / 8 * b;
                if (tile_byte0 + tile_bytes > d.nbytes) tile_bytes = d.nbytes - tile_byte0;
                store_tile(lds, (uint8_t*)jb.dst, tile_byte0, tile_bytes, threadIdx.x, (b & 7u) == 0 ? b >> 2 : 0u);
                lds_barrier();                      // write-out done before the next tile is staged
                MNW_STAMP(5)  // LDS read-out + global stores issued + barrier
            }
        }
        t0 = tend;
        ++j;
    }
    if (FUSED) {
        if (pstate == 1) { pend_issue(pend, sync, pret); pstate = 2; }
        if (pstate == 2) pend_finish(pend, jobs, stats, sync, meta, pret);
    }
    MNW_STAMP_FLUSH
    return true;
}

__global__ __launch_bounds__(kThreads) void k_pack_f32(const DevJob* __restrict__ jobs, int njobs,
                                                       int64_t total_tiles, JobSync* __restrict__ sync) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[kLds32 / 4];
    __shared__ uint64_t shared_slot;
    const int64_t t0 = total_tiles * blockIdx.x / gridDim.x;
    const int64_t t1 = total_tiles * (blockIdx.x + 1) / gridDim.x;
    pack_range_f32<false>(lds, &shared_slot, jobs, 0, njobs, t0, t1, sync, nullptr, 0, nullptr, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------
// fused float32 encode: ONE persistent launch for a whole batch.  The batch is cut into groups of
// consecutive blocks (tiles gtile[g] .. gtile[g+1] of the batch-wide numbering, jobs gjob[g] ..
// gjob[g+1]).  Every workgroup owns the same slice of every group and runs
//     R(0) | R(1) P(0) | R(2) P(1) | ... | P(ng-1)
// so the reduce of group g runs ahead of the pack of group g-1: by the time a workgroup asks for a
// block's descriptor its producers have normally finished.  There is no grid barrier, only per-block
// arrival counters and one published word per block.  HBM reads (next group), Infinity-Cache
// re-reads (previous group) and HBM writes overlap inside one launch; launch, ramp and the final L2
// write-back are paid once per batch.
// The grid must be co-resident (the host sizes it from the occupancy query); every spin is bounded.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_encode_f32(const DevJob* __restrict__ jobs, int njobs,
                                                         const int64_t* __restrict__ gtile, const int* __restrict__ gjob,
                                                         int ngroups, uint64_t* __restrict__ stats,
                                                         JobSync* __restrict__ sync, uint32_t* __restrict__ abort_flag,
                                                         Meta* __restrict__ meta) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[kLds32 / 4];
    __shared__ float red[4][kThreads / 64];
    __shared__ int redbad[kThreads / 64];
    __shared__ uint64_t shared_slot;
    __shared__ Pending pend;
    const int slot = blockIdx.x % kStatSlots;
    if (threadIdx.x == 0) pend.cnt = 0;
    __syncthreads();
    for (int g = 0; g <= ngroups; ++g) {
        // the pack phase of this stage starts with the first block of this workgroup's slice of group
        // g-1: ask for that block's descriptor now, so the answer is in a register after the reduce
        uint64_t pref = 0;
        int64_t pa = 0, pb = 0;
        if (g >= 1) {
            const int64_t a = gtile[g - 1], n = gtile[g] - a;
            pa = a + n * blockIdx.x / gridDim.x;
            pb = a + n * (blockIdx.x + 1) / gridDim.x;
            if (pa < pb && threadIdx.x == 0) {
                const int jl = find_job(jobs + gjob[g - 1], gjob[g] - gjob[g - 1], pa) + gjob[g - 1];
                pref = __hip_atomic_load(&sync[jl].granule, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        const bool packs = g >= 1 && pa < pb;
        if (g < ngroups) {
            const int64_t a = gtile[g], n = gtile[g + 1] - a;
            reduce_range_f32(jobs, gjob[g], gjob[g + 1], a + n * blockIdx.x / gridDim.x,
                             a + n * (blockIdx.x + 1) / gridDim.x, stats, sync, meta, slot, red, redbad, &pend, packs);
        }
        if (g >= 1) {
            if (!pack_range_f32<true>(lds, &shared_slot, jobs, gjob[g - 1], gjob[g], pa, pb, sync, abort_flag, pref,
                                      &pend, stats, meta)) return;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// single-read float32 encode (k_encode_reg_f32): the block is read ONCE.
//
// An item = 8,192 consecutive elements of one block = 32 values per lane of a 256-lane workgroup.
// The items of the batch are dealt out in rounds (round k: workgroups [0, size_k) take items base_k + w;
// all rounds are the whole grid G, or -- weighted hand-out, see the kernel -- some are a prefix of it), so
// every workgroup takes its items in increasing order.  A workgroup always has two of them on chip: item i
// in the registers e[] (loaded, or still in flight from HBM) and item i-1 parked as raw floats in LDS, its
// statistics PUBLISHED and waiting for the rest of its block.  Per loop trip:
//
//   1. start the loads of the records of item i-1's block (see 3)
//   2. fold item i's floats into the four statistics of k_reduce_f32 and publish them as one 32-byte
//      record of four self-validating 8-byte words {value, tag} (tag = this batch's generation; one sc1
//      store each, no atomics, nothing to zero)
//   3. wave 0 derives the descriptor of item i-1's block from the records of all its items -- published
//      a whole trip ago, so they are normally all there; every workgroup of the block computes the same
//      descriptor: no counter, no "last arrival", one memory hop -- while the other waves turn i-1's
//      floats into pixel indices; re-base, pack, store
//   4. park item i in LDS and start the loads of item i+1
//
// The skew between the workgroups of a block hides behind a whole trip; the HBM latency of step 4's
// loads is covered the ordinary way, by the other three workgroups of the CU (36 KB of LDS and <= 128
// registers: four workgroups per CU).  A deeper pipeline (item i+1 in flight while i-1 is packed) needs
// a second LDS buffer, halves the occupancy and measured slower: a trip is latency bound (its duration
// barely depends on how many workgroups share the CU), so resident workgroups are what count.
// This is the all-gather of tagged granules of MI355X_MICROARCH.md ("allgather", "Valid forms": R2's
// granule needs no ordering).
//
// Memory operations of a wave complete in issue order, hence: the record loads are issued before
// anything else of the trip, every barrier is the LDS-only one, and the kernel must not spill.
//
// No deadlock if every block has at most G items and all G workgroups are resident.  A workgroup
// publishes item i before it waits for item i-1's block and never waits before publishing.  Suppose
// nothing moves any more and let u be the lowest item whose record is not out.  Its owner X has not
// reached it, so X is blocked: it holds a parked item p and a published current item c, u is X's next
// item or a later one, and X waits for an unpublished item w of p's block; w >= u as u is the lowest.
// So p's block contains p and w >= u, i.e. it has more than u - p items.  But X's items p, c, u are taken
// in three different rounds, so u - p is at least the size of two consecutive rounds; no two partial
// rounds are adjacent, hence u - p >= G: a block of at most G items cannot contain both.  The host only
// sends blocks of at most min(G, kMaxSweep) items here and sizes G from the occupancy query.  Should the
// premise fail (another kernel holding part of the GPU), a wait is bounded (50 ms), after which the launch
// gives up and the host reruns the batch through the two-kernel path (tests: reg_test_drop_item).
// HBM traffic = algorithmic bytes: 4 B/element read + b/8 B/element written.
// ------------------------------------------------------------------------------------------------
constexpr int kThreadsR = 256;
constexpr int kItem = kThreadsR * kGroup;           // 8,192 elements: one lane packs one group of the item
constexpr int kBufR = kThreadsR * kStride32;        // dwords of one parked item (padded stage layout)
static_assert(kBufR * 4 == 36864, "36 KB of LDS for the parked item: four workgroups per CU");
constexpr int kMaxSweep = 512;                      // items per block the flat sweep handles (2^22 elements)
// Blocks of more items (up to kMaxBig, and up to twice the grid: see "large blocks" below) exchange their statistics in
// two levels: the records of kGroupItems consecutive items of a block are folded into one GROUP record, and a block's
// descriptor comes from its group records -- 64 + 64 records per workgroup and trip instead of 2,048.
constexpr int kGroupItems = 64;                     // one record per lane of wave 0
constexpr int kMaxBig = kGroupItems * 64;           // 4,096 items = 2^25 elements: at most 64 group records per block
__device__ __forceinline__ int64_t items_of(int64_t n) { return (n + kItem - 1) / kItem; }

struct ItemRec {                                     // 32 bytes: four {value: low 32, tag: high 32} words
    uint64_t g[4];
};
constexpr uint32_t kRecBad = 0x80000000u;            // tag bit 31: the item held a NaN / infinity

// wave 0 only.  Descriptor word of job j from the swept statistics; the owner of the block's first item
// records the metadata.  (Inlined: a call would save and reload a register through scratch, and that
// reload would queue behind the wave's streaming loads.)  Lanes 0..3 quantize one survivor each, so the
// IEEE division sequence runs once, not four times.
__device__ __forceinline__ uint64_t desc_from_sweep_f32(const DevJob& jb, int j, uint32_t v0, uint32_t v1,
                                                        uint32_t v2, uint32_t v3, bool bad, Meta* __restrict__ meta,
                                                        bool write_meta) {
    Desc d = {0, 0, 0, 0, 0};
    const int l = lane_id();
    const uint32_t key = l == 0 ? v0 : l == 1 ? ~v1 : l == 2 ? v2 : ~v3;
    int64_t ql = 0;
    const bool okl = quant_i64(funkey(key), jb.low, jb.dxq, &ql);
    const bool has_lo = v2 != 0, has_hi = v3 != 0;
    const uint64_t okmask = __ballot(okl);
    const bool ok = (okmask & 3) == 3 && (!has_lo || (okmask & 4)) && (!has_hi || (okmask & 8));
    const int64_t mx0 = __shfl(ql, 0), mn0 = __shfl(ql, 1), mxlo = __shfl(ql, 2), mnhi = __shfl(ql, 3);
    uint64_t maxdelta = 0;
    if (bad || !ok) {
        d.status = (int)kNeedsExact;
    } else if (jb.periodic == 1) {
        // same decisions as make_desc_f32
        if (mn0 < 0 || mx0 >= jb.pixels || jb.pixels > 0x7fffffffll) d.status = (int)kNeedsExact;
        else {
            const int64_t h = jb.pixels / 2;
            if (mx0 - mn0 + 1 <= h) { d.pmin = mn0; d.imin = mn0; maxdelta = (uint64_t)(mx0 - mn0); }
            else if (has_lo && has_hi && (mxlo + jb.pixels) - mnhi + 1 <= h) {
                d.pmin = mnhi; d.imin = mnhi; maxdelta = (uint64_t)(mxlo + jb.pixels - mnhi);
            } else { d.pmin = 0; d.imin = mn0; maxdelta = (uint64_t)(mx0 - mn0); }
        }
    } else {
        if (mn0 < -0x80000000ll || mx0 > 0x7fffffffll) d.status = (int)kNeedsExact;
        else { d.imin = mn0; maxdelta = (uint64_t)(mx0 - mn0); }
    }
    if (d.status == 0) {
        d.bits = precision_needed(maxdelta);
        d.nbytes = d.bits == 0 ? 0 : array_bytes(d.bits, jb.n);
        if (d.nbytes > jb.cap) d.status = -5;
    }
    if (write_meta && l == 0) {
        Meta m = {d.imin, (int64_t)d.bits, d.nbytes, (int64_t)d.status};
        meta[j] = m;
    }
    return gran_encode(d);
}

// One published record by two 16-byte agent-scope loads (each 8-byte word validates itself, so the halves need no
// ordering); returns when the data is there.  Half the requests of four 8-byte loads: what a poll costs is the number of
// requests that land on one line (they are served one after the other, about 2.5 ns each with a thousand pollers).
__device__ __forceinline__ void load_rec_sc1(const ItemRec* p, uint64_t (&w)[4]) {
    u32x4 a, b;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "v"((const MNW_GLOBAL ItemRec*)p) : "memory");
    w[0] = (uint64_t)a.x | ((uint64_t)a.y << 32); w[1] = (uint64_t)a.z | ((uint64_t)a.w << 32);
    w[2] = (uint64_t)b.x | ((uint64_t)b.y << 32); w[3] = (uint64_t)b.z | ((uint64_t)b.w << 32);
}

// The records of a block, one per lane, for use LATER in the trip: two 16-byte sc1 buffer loads the compiler can see (it
// places the wait where the values are first used, and keeps its hands off the destination registers until then -- an
// inline-asm load with a separate inline-asm wait does not give that: the compiler may copy a destination register in
// between, before the data has arrived).  `base` must be wave-uniform.
__device__ __forceinline__ void rec_prefetch(const ItemRec* base, int nrec, int idx, u32x4& a, u32x4& b) {
    const uint64_t p = (uint64_t)base;
    const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(p >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)p);
    // (a lane at or beyond `nrec` is out of the buffer's range: zeros, no request.  Callers still branch around the call for
    // such lanes -- a wave with no record then skips the instructions, 3 % of the headline launch -- and must leave the
    // destinations UNASSIGNED on the other path: a value to merge with would make the compiler wait for the loads at the join.)
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)pu, 0, __builtin_amdgcn_readfirstlane(nrec) * (int)sizeof(ItemRec), 0x00020000);
    a = __builtin_amdgcn_raw_buffer_load_b128(r, idx * (int)sizeof(ItemRec), 0, 16);          // aux bit 4: sc1
    b = __builtin_amdgcn_raw_buffer_load_b128(r, idx * (int)sizeof(ItemRec) + 16, 0, 16);
}

// wave 0 only, slow path: some record of the block was not there yet when the workgroup looked.  Polls
// until all are present, then returns the descriptor word (0 on abort).
__device__ __attribute__((noinline)) uint64_t sweep_block_f32(const DevJob* __restrict__ jobs, int j,
                                                              const ItemRec* __restrict__ recs, uint32_t gen,
                                                              uint32_t* __restrict__ abort_flag, uint32_t* __restrict__ abort_host,
                                                              Meta* __restrict__ meta, bool write_meta) {
    const int nitems = (int)items_of(jobs[j].n);
    const ItemRec* r0 = recs + jobs[j].tile0;
    uint32_t spins = 0;
    const uint64_t t_start = __builtin_amdgcn_s_memrealtime();      // 100 MHz
    for (;;) {
        uint32_t v[4] = {0, 0, 0, 0};
        uint32_t bad = 0, missing = 0;
        for (int r = lane_id(); r < nitems; r += 64) {
            uint64_t w4[4];
            load_rec_sc1(r0 + r, w4);                    // two 16-byte requests per record, not four 8-byte ones
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint64_t w = w4[k];
                const uint32_t tag = (uint32_t)(w >> 32);
                missing |= (tag & ~kRecBad) != gen;
                bad |= tag & kRecBad;
                const uint32_t val = (uint32_t)w;
                v[k] = val > v[k] ? val : v[k];
            }
        }
        if (!__any(missing)) {
            const DevJob jb = jobs[j];
            return desc_from_sweep_f32(jb, j, wave_max_dpp(v[0]), wave_max_dpp(v[1]), wave_max_dpp(v[2]), wave_max_dpp(v[3]),
                                       __any(bad), meta, write_meta);
        }
        __builtin_amdgcn_s_sleep(4);
        ++spins;
        if ((spins & 15) == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return 0;
        // 50 ms without the block completing: its other workgroups are not running (something else holds part
        // of the GPU), give the batch back to the host
        if ((spins & 15) == 0 && __builtin_amdgcn_s_memrealtime() - t_start > 5000000ull) {
            if (lane_id() == 0) {
                __hip_atomic_fetch_add(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(abort_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // pinned host word
            }
            return 0;
        }
    }
}

// ---- large blocks (more than kMaxSweep items; at most min(kMaxBig, 2 G) for a grid of G workgroups, rounds unweighted) ----
// A block of more than G items has two items in some workgroups, and such a workgroup can only pack the older one once
// the block is complete, i.e. once EVERY workgroup has published both of its items: the chip then holds the whole block
// (2 G items x 32 KB = 64 MB for G = 1,024: registers + LDS of all 256 CUs) and the launch advances block by block.
// Deadlock freedom, rounds unweighted, blocks of at most 2 G items: let everything be stuck and X be the stuck workgroup
// whose parked item p is lowest.  X's items are G apart and it has published p and c = p + G.  Every other stuck
// workgroup Z has a parked item >= p, so it has published its first item q* >= p and q* + G; Z's items inside p's block
// (which lies within [p - 2G + 1, p + 2G - 1]) are at most q* + G: all published (workgroups in their first trip never
// wait; finished ones have published everything).  So every item record of p's block is out, every group of it is swept
// by an owner Y of one of its items i -- in the trip where i is Y's parked item, or already in the trip where i is Y's
// current item and Y's parked item belongs to the same block -- and Y cannot be stuck before that trip (its parked item
// would be below p) nor in it before the group record is out (the sweep only needs item records).  Hence all group
// records of p's block appear and X's wait ends: contradiction.
//
// wave 0 only.  Makes sure the group records of group `grp` of job jb exist: polls the item records of the group until
// all are there, folds them and publishes the group record (every workgroup that needs the group does this; they all
// store the same four words).  Returns false on abort.
__device__ __forceinline__ bool poll_abort(uint32_t& spins, uint64_t t_start, uint32_t* __restrict__ abort_flag,
                                           uint32_t* __restrict__ abort_host) {
    __builtin_amdgcn_s_sleep(2);
    ++spins;
    if ((spins & 15) != 0) return false;
    if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
    if (__builtin_amdgcn_s_memrealtime() - t_start > 5000000ull) {      // 50 ms (100 MHz counter)
        if (lane_id() == 0) {
            __hip_atomic_fetch_add(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(abort_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // pinned host word
        }
        return true;
    }
    return false;
}

// wave 0 only.  Large blocks, the two steps of the two-level sweep:
//   group_duty_f32: poll the item records of the group of item `it` of job j until all are there, fold them and publish
//      the group record.  Every member of a group does this (no leader; it waits for item records only), in the trip in
//      which `it` is its parked item -- the records are a trip old then and the kernel has loaded them at the start of
//      the trip like the flat sweep does: this function only runs if one was missing -- or, when the workgroup's parked
//      item belongs to the same block, already in the trip in which `it` is its current item (the block cannot complete
//      without it): those records are fresh and this is where the chip waits for its slowest workgroup.
//   block_desc_big_f32: the descriptor word of the parked item's block from its group records (0 on abort).
// What a poll costs is set by how many requests hit one line: a record that all 1,024 workgroups poll takes ~10 us per
// round.  Hence the group records exist in R copies (R = groups of the block): copy d is read only by the workgroups whose
// parked item is in group d (64 readers per line), and member m of group g writes the copies d = m, m + cnt_g, ... of
// record g (cnt_g = items of group g: one copy each, more only in a block's short last group).
__device__ __forceinline__ void publish_group_rec(ItemRec* __restrict__ copies, int R, int grp, int member, int cnt,
                                                  uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, bool anybad, uint32_t gen) {
    const int l = lane_id();
    const uint32_t val = l == 0 ? v0 : l == 1 ? v1 : l == 2 ? v2 : v3;
    const uint64_t w64 = (uint64_t)val | ((uint64_t)(gen | (anybad ? kRecBad : 0u)) << 32);
    for (int d = member; d < R; d += cnt)
        if (l < 4) __hip_atomic_store(&copies[(size_t)d * R + grp].g[l], w64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __attribute__((noinline)) bool group_duty_f32(const DevJob* __restrict__ jobs, int j, int64_t it,
                                                         ItemRec* __restrict__ recs, ItemRec* __restrict__ grecs, uint32_t gen,
                                                         uint32_t* __restrict__ abort_flag, uint32_t* __restrict__ abort_host) {
    const int64_t tile0 = jobs[j].tile0, grp0 = jobs[j].grp0;
    const int nitems = (int)items_of(jobs[j].n);
    const int R = (nitems + kGroupItems - 1) / kGroupItems;
    const int l = lane_id();
    uint32_t spins = 0;
    const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
    const int grp = (int)(it - tile0) / kGroupItems;
    const int first = grp * kGroupItems;
    const int cnt = nitems - first < kGroupItems ? nitems - first : kGroupItems;
    const int member = (int)(it - tile0) - first;
    const ItemRec* r0 = recs + tile0 + first;
    ItemRec* const copies = grecs + grp0;                    // [R reader groups][R group records]
    MNW_BIG_DIAG(0, 1);
    for (;;) {
        uint64_t w[4] = {0, 0, 0, 0};
        uint32_t bad = 0, missing = 0;
        if (l < cnt) {
            load_rec_sc1(r0 + l, w);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t tag = (uint32_t)(w[k] >> 32);
                missing |= (tag & ~kRecBad) != gen;
                bad |= tag & kRecBad;
            }
        }
        if (!__any(missing)) {
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = wave_max_dpp(l < cnt ? (uint32_t)w[k] : 0u);
            publish_group_rec(copies, R, grp, member, cnt, v[0], v[1], v[2], v[3], __any(bad), gen);
            return true;
        }
        MNW_BIG_DIAG(1, 1);
        if (poll_abort(spins, t_start, abort_flag, abort_host)) return false;
    }
}

__device__ __attribute__((noinline)) uint64_t block_desc_big_f32(const DevJob* __restrict__ jobs, int j, int64_t pit,
                                                                 ItemRec* __restrict__ grecs, uint32_t gen,
                                                                 uint32_t* __restrict__ abort_flag, uint32_t* __restrict__ abort_host,
                                                                 Meta* __restrict__ meta) {
    const DevJob jb = jobs[j];
    const int nitems = (int)items_of(jb.n);
    const int R = (nitems + kGroupItems - 1) / kGroupItems;
    const int l = lane_id();
    uint32_t spins = 0;
    const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
    const int g_prev = (int)(pit - jb.tile0) / kGroupItems;
    const ItemRec* gr = grecs + jb.grp0 + (size_t)g_prev * R;     // copy g_prev of the block's R group records
    MNW_BIG_DIAG(3, 1);
    for (;;) {
        uint64_t w[4] = {0, 0, 0, 0};
        uint32_t bad = 0, missing = 0;
        if (l < R) {
            load_rec_sc1(gr + l, w);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t tag = (uint32_t)(w[k] >> 32);
                missing |= (tag & ~kRecBad) != gen;
                bad |= tag & kRecBad;
            }
        }
        if (!__any(missing)) {
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = wave_max_dpp(l < R ? (uint32_t)w[k] : 0u);
            return desc_from_sweep_f32(jb, j, v[0], v[1], v[2], v[3], __any(bad), meta, pit == jb.tile0);
        }
        MNW_BIG_DIAG(2, 1);
        if (poll_abort(spins, t_start, abort_flag, abort_host)) return 0;
    }
}

// round sizes of the weighted item hand-out, in quarters of the grid (see k_encode_reg_f32)
// (table 0 is the default; the others are there to be measured against it: option reg_round_table.  Every table starts with a
// full round and has no two partial rounds adjacent, also around its end.)
constexpr int kRoundTables = 5, kMaxRounds = 20;
__constant__ int kRoundCount[kRoundTables] = {14, 16, 17, 15, 18};
__constant__ int kRoundTable[kRoundTables][kMaxRounds] = {
    {4, 4, 4, 1, 4, 4, 3, 4, 4, 1, 4, 4, 2, 4},                          // 14 / 12 / 11 / 10 items per cycle
    {4, 1, 4, 3, 4, 2, 4, 1, 4, 3, 4, 2, 4, 1, 4, 4},                    // 16 / 13 / 11 / 9
    {4, 1, 4, 3, 4, 2, 4, 1, 4, 3, 4, 2, 4, 1, 4, 4, 4},                 // 17 / 14 / 12 / 10
    {4, 4, 1, 4, 4, 3, 4, 4, 1, 4, 4, 2, 4, 4, 4},                       // 15 / 13 / 12 / 11
    {4, 1, 4, 2, 4, 3, 4, 1, 4, 2, 4, 3, 4, 1, 4, 2, 4, 1},              // 18 / 14 / 11 / 9
};
// (measured against 14/13/12/11, 14/14/12/11 and 14/11/10/9: within 1 % at 1e8 elements, best at 1e9)


// XF: the batch has jobs with a minh value transform (DevJob.xform).  A separate instantiation: the plain kernel has
// no scalar or vector register to spare, and carrying the transform's code and state cost it 6 %.  (The XF one
// spills ~30 registers at four workgroups per CU and is still faster that way than at three without spills:
// 5.8 vs 6.8 ms for the 24 float columns of tools/bench_c3.py.)
// BIG: the batch has blocks of more than kMaxSweep items (two-level sweep: group_duty_f32, block_desc_big_f32).  Also an instantiation of its
// own: the extra call site costs the plain kernel scratch traffic on its critical path.
template <bool XF, bool BIG>
#ifndef MNW_XF_WGS
#define MNW_XF_WGS 3      // workgroups per CU of the instantiations with the minh value transform: at four they spill 38 registers (6.8 against 6.1 ms for the 24 float columns of tools/bench_c3.py --minh)
#endif
__global__ __launch_bounds__(kThreadsR, XF ? MNW_XF_WGS : 4) void k_encode_reg_f32(const DevJob* __restrict__ jobs, int njobs,
                                                                 int64_t total_items, const int* __restrict__ item_job,
                                                                 ItemRec* __restrict__ recs, ItemRec* __restrict__ grecs,
                                                                 uint32_t gen, uint32_t* __restrict__ abort_flag,
                                                                 uint32_t* __restrict__ abort_host,
                                                                 Meta* __restrict__ meta, int weighted, int64_t test_drop_item,
                                                                 ItemRec* __restrict__ drecs, int n_sweepers, int flat_lim) {
    // (test_drop_item: test hook, normally -1 -- the record of that item is never published, so its block can
    // never complete and the launch has to take its bounded-wait exit)
    // BIG, n_sweepers > 0 (round 3): blocks of more than flat_lim items -- up to 2 G items, G = the workgroups that hold items --
    // get their descriptor from one of the last n_sweepers workgroups of the grid, which hold no items: sweeper s takes the large
    // jobs j = s (mod n_sweepers) in order, waits for the block's last record, folds all of them and publishes ONE descriptor record
    // (drecs[j]: the two halves of the descriptor word, tagged); the item workgroups poll that instead of sweeping or doing group
    // duty.  k_encode_reg_i64's header has the deadlock argument.  n_sweepers == 0: the two-level sweep below, which a block of
    // 2 x gridDim.x items (2^24 elements on a whole MI355X) needs because it leaves no workgroup over.
    __shared__ __attribute__((aligned(16))) uint32_t lds[kBufR];
    __shared__ float red[4][kThreadsR / 64];
    __shared__ int redbad[kThreadsR / 64];
    __shared__ uint32_t swv[4][kThreadsR / 64];
    __shared__ uint32_t swf[kThreadsR / 64];
    __shared__ uint64_t shared_slot;
    // tables of the fast log10 (minh transforms): 2.5 KB of LDS in the XF instantiation, 8 bytes in the plain one
    __shared__ typename std::conditional<XF, FastTables, double>::type ftab_s;
    FastTables* const ftab_p = XF ? reinterpret_cast<FastTables*>(&ftab_s) : nullptr;
    if (XF) fast_tables_load(ftab_p);
    constexpr int kV = kItem / (4 * kThreadsR);        // 16-byte loads per lane per item (8)
    constexpr int kW = kThreadsR / 64;
    const int tid = threadIdx.x;
    MNW_STAMP_DECL
    MNW_STAMP_START

    // the item in e[] (all wave-uniform)
    u32x4 e[kV];
    int cj = 0;
    int64_t cbase = 0, cn = 0;
    const float* cx = nullptr;
    float cxh = 0.0f;
    int cxf = 0;                // minh value transform of the item's job (0: none)
    auto locate = [&](int64_t t) {
        cj = item_job[t];
        cbase = (t - jobs[cj].tile0) * kItem;
        cn = jobs[cj].n;
        cx = (const float*)jobs[cj].src;
        cxh = jobs[cj].xh;
        if (XF) cxf = jobs[cj].xform;
    };
    auto issue = [&]() {
        // (a block's last, partial item is not held in registers: its statistics and its parked copy are
        // made by plain loops over global memory, see below)
        if (cbase + kItem <= cn) {
            const int t = opaque_tid();
#pragma unroll
            for (int k = 0; k < kV; ++k) {
                const float* pk = cx + cbase + k * (4 * kThreadsR);     // wave-uniform part of the address
                e[k] = ldg<u32x4>(pk + 4 * t);
            }
        }
    };

    // Item hand-out.  Rounds: in round k the workgroups [0, size_k) take the items base_k + w.  With `weighted`
    // the round sizes cycle through kRoundQuarters (in quarters of the grid), so that per cycle the four
    // quarters of the grid take 14 / 12 / 11 / 10 items: the hardware favours the oldest waves of a CU and a
    // CU's four workgroups come from the four quarters of the grid in dispatch order, which makes the last
    // quarter run at 70 % of the speed of the first (measured: 109 / 125 / 142 / 154 us for twelve items
    // each).  No two partial rounds are adjacent, so two consecutive items of a workgroup are at least a
    // whole grid apart and the header's argument holds unchanged.  (Handing the items out with a counter
    // instead puts a returning atomic on wave 0's path every trip and measured 15 % slower overall.)
    const int G = (int)gridDim.x - (BIG ? n_sweepers : 0);          // workgroups that hold items
    if (BIG && (int)blockIdx.x >= G) {
        // ---- sweeper: the descriptors of the large blocks j = s (mod n_sweepers), in job order
        const int s_id = (int)blockIdx.x - G;
        int seen = 0;                                    // large jobs before j0 (uniform)
        if (tid == 0) shared_slot = 0;                   // here: "gave up"
        lds_barrier();
        for (int j0 = 0; j0 < njobs; j0 += kThreadsR) {
            const int jt = j0 + tid;
            reinterpret_cast<uint8_t*>(lds)[tid] = (jt < njobs && items_of(jobs[jt].n) > flat_lim) ? 1 : 0;
            lds_barrier();
            for (int q = 0; q < kThreadsR && j0 + q < njobs; ++q) {
                if (!reinterpret_cast<const uint8_t*>(lds)[q]) continue;
                const bool mine = seen % n_sweepers == s_id;
                ++seen;
                if (!mine) continue;
                const int j = j0 + q;
                const int T = (int)items_of(jobs[j].n);
                const ItemRec* r0 = recs + jobs[j].tile0;
                uint32_t spins = 0;
                const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
                // a hint that the block is (nearly) complete: its last record
                for (;;) {
                    uint64_t w[4];
                    load_rec_sc1(r0 + (T - 1), w);
                    if (((uint32_t)(w[3] >> 32) & ~kRecBad) == gen) break;
                    if (poll_abort(spins, t_start, abort_flag, abort_host)) { if (lane_id() == 0) shared_slot = 1; break; }
                }
                lds_barrier();
                if (shared_slot) return;
                for (;;) {
                    uint32_t v[4] = {0, 0, 0, 0};
                    uint32_t bad = 0, missing = 0;
                    for (int r = tid; r < T; r += kThreadsR) {
                        uint64_t w4[4];
                        load_rec_sc1(r0 + r, w4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const uint32_t tag = (uint32_t)(w4[k] >> 32);
                            missing |= (tag & ~kRecBad) != gen;
                            bad |= tag & kRecBad;
                            const uint32_t val = (uint32_t)w4[k];
                            v[k] = val > v[k] ? val : v[k];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = wave_max_dpp(v[k]);
                    const uint32_t fl = (__any(missing) ? 1u : 0u) | (__any(bad) ? 2u : 0u);
                    if (lane_id() == 0) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) swv[k][wave_id()] = v[k];
                        swf[wave_id()] = fl;
                    }
                    lds_barrier();
                    uint32_t v0 = swv[0][0], v1 = swv[1][0], v2 = swv[2][0], v3 = swv[3][0], flw = swf[0];
#pragma unroll
                    for (int w = 1; w < kW; ++w) {
                        v0 = max(v0, swv[0][w]); v1 = max(v1, swv[1][w]); v2 = max(v2, swv[2][w]); v3 = max(v3, swv[3][w]);
                        flw |= swf[w];
                    }
                    if (!(flw & 1u)) {
                        if (wave_id() == 0) {
                            const DevJob jb = jobs[j];
                            const uint64_t g = desc_from_sweep_f32(jb, j, v0, v1, v2, v3, (flw & 2u) != 0, meta, true);
                            // the record: {descriptor word low, high, 0, 0}, each word tagged
                            const int l = lane_id();
                            const uint32_t val = l == 0 ? (uint32_t)g : l == 1 ? (uint32_t)(g >> 32) : 0u;
                            if (l < 4) __hip_atomic_store(&drecs[j].g[l], (uint64_t)val | ((uint64_t)gen << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        lds_barrier();                   // (swv / swf are free again)
                        break;
                    }
                    if (wave_id() == 0 && poll_abort(spins, t_start, abort_flag, abort_host) && lane_id() == 0) shared_slot = 1;
                    lds_barrier();
                    if (shared_slot) return;
                }
            }
            lds_barrier();                               // (the flags are rewritten by the next stretch)
        }
        return;
    }
    const int unit = weighted ? G / 4 : G;
    int round = 0;
    int64_t round_base = 0;
    auto next_item = [&]() -> int64_t {           // the workgroup's item after the one in (round, round_base)
        for (;;) {
            round_base += (int64_t)(weighted ? kRoundTable[weighted - 1][round] : 1) * unit;
            round = round + 1 == (weighted ? kRoundCount[weighted - 1] : 1) ? 0 : round + 1;
            if (round_base >= total_items) return total_items;
            if ((int)blockIdx.x < (weighted ? kRoundTable[weighted - 1][round] : 1) * unit) return round_base + blockIdx.x;
        }
    };
    int64_t it = blockIdx.x;                      // round 0 is a full one
    bool have_cur = it < total_items;
    if (have_cur) { locate(it); issue(); }
    bool have_prev = false;
    int desc_job = -1;          // BIG, wave 0: the job whose descriptor word is desc_word
    uint64_t desc_word = 0;
    int pj = 0;                 // the parked item
    int64_t pbase = 0, pit = 0;

    while (have_cur || have_prev) {
        const int sj = cj;
        const int64_t sbase = cbase, sit = it;
        const float xh = cxh;
        const int sxf = XF ? cxf : 0;
        // (the transform's limits are fetched from the job record where they are used: nothing more is carried
        // across the trip -- the kernel has no scalar or vector register to spare)
        const float* const sx = cx + cbase;
        const int64_t srem = cn - cbase;                 // elements of item i (>= kItem: a full one)
        const int64_t nit = have_cur ? next_item() : total_items;
        const bool have_next = have_cur && nit < total_items;
        // where the next item lives: scalar loads, started long before they are needed
        if (have_next) locate(nit);

        // ---- 1. the records of item i-1's block, one per lane (blocks of more than 256 items: the second
        // half follows once the first is folded).  Issued ahead of everything else of the trip that touches
        // memory, used after the statistics.
        DevJob pjb;
        int pitems = 0;
        uint64_t sw[4];
        u32x4 swa, swb;
        int nrec = 0;
        if (have_prev) {
            pjb = jobs[pj];
            pitems = (int)items_of(pjb.n);
            const ItemRec* r0 = recs + pjb.tile0;
            // (a large block: the records of the parked item's GROUP, lanes 0..63 -- see group_duty_f32)
            int nload = pitems, first = 0;
            if (BIG && pitems > flat_lim) {
                if (n_sweepers > 0) nload = 0;                       // (its descriptor comes from a sweeper: nothing to fold)
                else {
                    first = (int)(pit - pjb.tile0) / kGroupItems * kGroupItems;
                    nload = pitems - first < kGroupItems ? pitems - first : kGroupItems;
                }
            }
            // Two 16-byte agent-scope loads per record, not four 8-byte ones: half as many requests on the record lines, which
            // every workgroup of a block reads, and requests on one line are served one after the other (measured in one run,
            // tools/ab_libs.sh: 1e9 elements in 2^20-element blocks 1,174 -> 1,121 us, 2^22-element blocks 178 -> 170 us per
            // 1e8; one 16-byte load per lane -- lanes 2r, 2r+1 the halves of record r, one request per line -- was no better:
            // 1,168 and 168 us, its parity-wise reduction costs what the requests save).  Issued here, used by the fold below.
            nrec = nload;
            if (tid < nload) rec_prefetch(r0 + first, nload, tid, swa, swb);     // (waves without a record skip the instructions)
        }

        if (have_cur) {
            // ---- 2. statistics of item i -----------------------------------------------------------------
            float mn = INFINITY, mx = -INFINITY, mxlo = -INFINITY, mnhi = INFINITY;
            bool nonfinite = false;
            if (srem >= kItem) {
                if (sxf) {
                    const float sxlow = jobs[sj].low, sxhigh = jobs[sj].high, sxhb = jobs[sj].high_below;
                    // minh.processFloatGroup (minh.go:139-149) on the values as they arrive: what is parked, reduced
                    // and packed below is the processed column, as if it had been encoded from minh.buf.  The
                    // logarithm goes through its fast form; if that could not decide some value's float32 rounding
                    // (a few in a million) the wave redoes its vectors from memory with the exact form.
                    bool amb = false;
#pragma unroll
                    for (int k = 0; k < kV; ++k) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float xv = __uint_as_float(e[k][c]);
                            if (sxf & kXfLog10) xv = go_log10_f32_try(xv, ftab_p, amb);
                            if (sxf & kXfClamp) { if (xv < sxlow) xv = sxlow; if (xv >= sxhigh) xv = sxhb; }
                            e[k][c] = __float_as_uint(xv);
                        }
                        __builtin_amdgcn_sched_barrier(0);           // one 16-byte vector at a time: bounds the float64 temporaries
                    }
                    if (__builtin_expect(__any(amb), 0)) {
                        const int t = opaque_tid();
#pragma unroll
                        for (int k = 0; k < kV; ++k) e[k] = xf_apply_exact4(sx + k * (4 * kThreadsR) + 4 * t, sxf, sxlow, sxhigh, sxhb);
                    }
                }
#pragma unroll
                for (int k = 0; k < kV; ++k) {
#pragma unroll
                    for (int c = 0; c < 4; c += 2) {
                        const float x0 = __uint_as_float(e[k][c]), x1 = __uint_as_float(e[k][c + 1]);
                        nonfinite |= __builtin_amdgcn_classf(x0, 0x207) | __builtin_amdgcn_classf(x1, 0x207);   // NaN, +-Inf
                        mn = min3_raw(mn, x0, x1);
                        mx = max3_raw(mx, x0, x1);
                        const bool lo0 = x0 < xh, lo1 = x1 < xh;
                        mxlo = max3_raw(mxlo, lo0 ? x0 : -INFINITY, lo1 ? x1 : -INFINITY);
                        mnhi = min3_raw(mnhi, lo0 ? INFINITY : x0, lo1 ? INFINITY : x1);
                    }
                }
            } else {
#pragma unroll 1
                for (int i = tid; i < (int)srem; i += kThreadsR) {
                    float x0 = ldg<float>(sx + i);
                    if (sxf) x0 = xf_apply_fast(x0, sxf, jobs[sj].low, jobs[sj].high, jobs[sj].high_below, ftab_p);
                    nonfinite |= __builtin_amdgcn_classf(x0, 0x207);
                    mn = fminf(mn, x0); mx = fmaxf(mx, x0);
                    if (x0 < xh) mxlo = fmaxf(mxlo, x0); else mnhi = fminf(mnhi, x0);
                }
            }
            MNW_STAMP(0)      // wait for the loads + per-lane statistics
            mn = wave_min_dpp(mn); mx = wave_max_dpp(mx); mxlo = wave_max_dpp(mxlo); mnhi = wave_min_dpp(mnhi);
            const int bad = __any(nonfinite);
            if (lane_id() == 0) {
                red[0][wave_id()] = mn; red[1][wave_id()] = mx;
                red[2][wave_id()] = mxlo; red[3][wave_id()] = mnhi;
                redbad[wave_id()] = bad;
            }
        }
        if (have_prev) {
            // fold what the record loads brought
            if (tid >= nrec) { swa = u32x4{0u, gen, 0u, gen}; swb = swa; }        // no record for this lane: neutral words
            sw[0] = (uint64_t)swa.x | ((uint64_t)swa.y << 32); sw[1] = (uint64_t)swa.z | ((uint64_t)swa.w << 32);
            sw[2] = (uint64_t)swb.x | ((uint64_t)swb.y << 32); sw[3] = (uint64_t)swb.z | ((uint64_t)swb.w << 32);
            uint32_t v[4] = {0, 0, 0, 0};
            uint32_t bad = 0, missing = 0;
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t tag = (uint32_t)(sw[k] >> 32);
                    missing |= (tag & ~kRecBad) != gen;
                    bad |= tag & kRecBad;
                    const uint32_t val = (uint32_t)sw[k];
                    v[k] = val > v[k] ? val : v[k];
                }
                if (h == 1 || pitems <= kThreadsR || (BIG && pitems > flat_lim)) break;
                const ItemRec* r1 = recs + pjb.tile0 + kThreadsR;
                const uint64_t neutral = (uint64_t)gen << 32;
                if (tid + kThreadsR < pitems) load_rec_sc1(r1 + tid, sw);
                else { sw[0] = sw[1] = sw[2] = sw[3] = neutral; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = wave_max_dpp(v[k]);
            const uint32_t fl = (__any(missing) ? 1u : 0u) | (__any(bad) ? 2u : 0u);
            if (lane_id() == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) swv[k][wave_id()] = v[k];
                swf[wave_id()] = fl;
            }
        }
        lds_barrier();
        if (have_cur && tid < 4 && sit != test_drop_item) {
            // publish the item's record (wave 0, lanes 0..3: one 8-byte sc1 store each)
            float a = red[0][0], b2 = red[1][0], c2 = red[2][0], dd = red[3][0];
            int bd = redbad[0];
#pragma unroll
            for (int w = 1; w < kW; ++w) {
                a = fminf(a, red[0][w]); b2 = fmaxf(b2, red[1][w]);
                c2 = fmaxf(c2, red[2][w]); dd = fminf(dd, red[3][w]);
                bd |= redbad[w];
            }
            uint32_t val;
            if (tid == 0) val = bd ? 0 : fkey(b2);
            else if (tid == 1) val = bd ? 0 : (uint32_t)~fkey(a);
            else if (tid == 2) val = (bd || c2 == -INFINITY) ? 0 : fkey(c2);
            else val = (bd || dd == INFINITY) ? 0 : (uint32_t)~fkey(dd);
            const uint64_t w64 = (uint64_t)val | ((uint64_t)(gen | (bd ? kRecBad : 0u)) << 32);
            __hip_atomic_store(&recs[sit].g[tid], w64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // (first trip only: nothing below separates these reads of red[] from the next trip's statistics of the waves that skip
        // step 3 -- later trips have step 3's barriers in between)
        if (!have_prev) lds_barrier();
        MNW_STAMP(1)      // reductions + records folded + barrier + publish

        if (have_prev) {
            // ---- 3. item i-1: descriptor of its block (wave 0) while the others quantize ----------------
            if (wave_id() == 0) {
                uint32_t v0 = swv[0][0], v1 = swv[1][0], v2 = swv[2][0], v3 = swv[3][0], fl = swf[0];
#pragma unroll
                for (int w = 1; w < kW; ++w) {
                    v0 = max(v0, swv[0][w]); v1 = max(v1, swv[1][w]); v2 = max(v2, swv[2][w]); v3 = max(v3, swv[3][w]);
                    fl |= swf[w];
                }
                uint64_t g;
                if (BIG && pitems > flat_lim) {
                    // a large block.  Its descriptor may be known from the trip before (the workgroup's previous item was in
                    // the same block); else: from a sweeper's record, or (no sweepers) the current item's group first if that is
                    // part of the block, then the sweep
                    if (pj == desc_job) g = desc_word;
                    else if (n_sweepers > 0) {
                        uint32_t spins = 0;
                        const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
                        g = 0;
                        for (;;) {
                            uint64_t w[4];
                            load_rec_sc1(drecs + pj, w);
                            if ((uint32_t)(w[0] >> 32) == gen && (uint32_t)(w[1] >> 32) == gen) { g = (uint64_t)(uint32_t)w[0] | ((uint64_t)(uint32_t)w[1] << 32); break; }
                            if (poll_abort(spins, t_start, abort_flag, abort_host)) break;
                        }
                        desc_job = pj; desc_word = g;
                    } else {
                        // the group of the parked item: from the records loaded at the start of the trip, if they were all there
                        bool ok2 = true;
                        if (fl & 1u) ok2 = group_duty_f32(jobs, pj, pit, recs, grecs, gen, abort_flag, abort_host);
                        else {
                            const int first = (int)(pit - pjb.tile0) / kGroupItems * kGroupItems;
                            const int cnt = pitems - first < kGroupItems ? pitems - first : kGroupItems;
                            publish_group_rec(grecs + pjb.grp0, (pitems + kGroupItems - 1) / kGroupItems, first / kGroupItems,
                                              (int)(pit - pjb.tile0) - first, cnt, v0, v1, v2, v3, (fl & 2u) != 0, gen);
                        }
                        if (ok2 && have_cur && sj == pj) ok2 = group_duty_f32(jobs, sj, sit, recs, grecs, gen, abort_flag, abort_host);
                        g = ok2 ? block_desc_big_f32(jobs, pj, pit, grecs, gen, abort_flag, abort_host, meta) : 0;
                        desc_job = pj; desc_word = g;
                    }
                } else if (__builtin_expect(fl & 1u, 0)) {
                    MNW_DIAG_SLOW_SWEEP
                    g = sweep_block_f32(jobs, pj, recs, gen, abort_flag, abort_host, meta, pit == pjb.tile0);
                } else {
                    g = desc_from_sweep_f32(pjb, pj, v0, v1, v2, v3, (fl & 2u) != 0, meta, pit == pjb.tile0);
                }
                if (lane_id() == 0) shared_slot = g;
            }
            const float low = pjb.low, dx = pjb.dxq;
            const double rdx = pjb.rdx;
            uint32_t q[kGroup];
            bool tiny = false;
            if (low == 0.0f) {
                // x - 0 == x exactly (go/group.go:318 with low == 0): one instruction less per element
#pragma unroll
                for (int k = 0; k < kGroup / 4; ++k) {
                    const u32x4 r = *reinterpret_cast<const u32x4*>(lds + tid * kStride32 + 4 * k);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float rq = (float)__dmul_rn((double)__uint_as_float(r[c]), rdx);
                        tiny |= __builtin_amdgcn_classf(rq, 0x90);         // a subnormal quotient (quant_tiny)
                        q[4 * k + c] = (uint32_t)cvt_floor_i32(rq);
                    }
                    if (k & 1) __builtin_amdgcn_sched_barrier(0);      // at most two LDS reads ahead: bounds the live registers
                }
            } else {
#pragma unroll
                for (int k = 0; k < kGroup / 4; ++k) {
                    const u32x4 r = *reinterpret_cast<const u32x4*>(lds + tid * kStride32 + 4 * k);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float rq = quant_mul(__uint_as_float(r[c]), low, rdx);
                        tiny |= __builtin_amdgcn_classf(rq, 0x90);
                        q[4 * k + c] = (uint32_t)cvt_floor_i32(rq);
                    }
                    if (k & 1) __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (__builtin_expect(tiny, 0)) {
                // a quotient below the float32 normal range: this lane redoes its group with the IEEE division
#pragma unroll 1
                for (int i = 0; i < kGroup; ++i)
                    q[i] = (uint32_t)(int32_t)quant_floor(__uint_as_float(lds[tid * kStride32 + i]), low, dx);
            }
            lds_barrier();                                     // floats consumed; descriptor visible
            const uint64_t gran = shared_slot;
            MNW_STAMP(2)      // descriptor (wave 0) / quantize + barrier
            if (!(gran & kGranReady)) break;                   // aborted (uniform: every wave sees the same word)
            const Desc d = gran_decode(gran, pjb.n);
            const int ok = __builtin_amdgcn_readfirstlane(d.status == 0 && d.bits != 0);
            if (ok) {
                const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane(d.bits);
                const uint32_t imin = (uint32_t)(int32_t)d.imin;
                // bound() (group.go:374-382) then the subtraction of the block minimum (group.go:247):
                // with r = q - imin (mod 2^32) the values below the ring start are exactly those whose r
                // "went negative", and r + pixels brings precisely them back under every other r
                const uint32_t wrap = (pjb.periodic == 1 && d.pmin == d.imin) ? (uint32_t)pjb.pixels : 0u;
                if (wrap == 0) {
#pragma unroll
                    for (int i = 0; i < kGroup; ++i) q[i] -= imin;
                } else {
#pragma unroll
                    for (int i = 0; i < kGroup; ++i) {
                        const uint32_t r = q[i] - imin;
                        q[i] = min(r, r + wrap);
                    }
                }
                if (pbase + kItem > pjb.n) {
                    const int64_t e0 = pbase + kGroup * opaque_tid();
#pragma unroll
                    for (int i = 0; i < kGroup; ++i)
                        if (e0 + i >= pjb.n) q[i] = 0;
                }
                pack_group32_switch(q, b, lds + tid * pack_stride(b));
                lds_barrier();
                MNW_STAMP(3)      // re-base + pack + barrier
                const int64_t tile_byte0 = pbase / 8 * b;
                int64_t tile_bytes = (int64_t)kItem / 8 * b;
                if (tile_byte0 + tile_bytes > d.nbytes) tile_bytes = d.nbytes - tile_byte0;
                store_tile<kThreadsR>(lds, (uint8_t*)pjb.dst, tile_byte0, tile_bytes, opaque_tid(), (b & 7u) == 0 ? b >> 2 : 0u);
            }
            lds_barrier();                                     // the LDS buffer is free again
            MNW_STAMP(4)      // stores issued + barrier
        }
        // ---- 4. park item i, start item i+1 ------------------------------------------------------------
        if (have_cur) {
            if (srem >= kItem) {
                uint32_t* const park = lds + stage32_idx(4 * opaque_tid());   // + k * 1152 dwords: 4*(k*256+tid) -> stage index
#pragma unroll
                for (int k = 0; k < kV; ++k)
                    *reinterpret_cast<u32x4*>(park + k * (4 * kThreadsR / kGroup * kStride32)) = e[k];
            } else {
#pragma unroll 1
                for (int i = tid; i < kItem; i += kThreadsR)
                {
                    float xv = i < (int)srem ? ldg<float>(sx + i) : 0.0f;
                    if (sxf && i < (int)srem) xv = xf_apply_fast(xv, sxf, jobs[sj].low, jobs[sj].high, jobs[sj].high_below, ftab_p);
                    lds[stage32_idx(i)] = __float_as_uint(xv);
                }
            }
            if (have_next) issue();
            MNW_STAMP(5)      // park + next loads issued

        }
        have_prev = have_cur;
        pj = sj; pbase = sbase; pit = sit;
        it = nit;
        have_cur = have_next;
    }
    MNW_STAMP_FLUSH
}

// ------------------------------------------------------------------------------------------------
// single-read int64 encode (k_encode_reg_i64): intGroup.writeData (go/group.go:242-255) with the block read ONCE, the
// same resident two-stage pipeline and tagged-record exchange as k_encode_reg_f32 (its header has the protocol and the
// deadlock argument).  An item = 4,096 consecutive int64 = 32 KB = 16 values per lane; the parked item sits in LDS as raw
// int64 in the 64-bit stage layout (32-element groups at a 68-dword stride), and lanes 0..127 pack one group of 32 each.
// A record = the item's max and min as four self-validating words {high half, tag} {low half, tag} (max), (min); a block's
// statistics are the 64-bit max / min over its records.  Widths above 32 bits (a block whose values span more than 2^32)
// are packed in place in the LDS stage (pack_group64_inplace).  Blocks of at most min(G, 512) items (2^21 elements): flat
// sweep -- every workgroup folds the records of its parked item's block itself.  Larger blocks, up to 2 G items (2^23 elements
// = 64 MB on a whole MI355X: what two items per workgroup hold), round 3: the last `n_sweepers` workgroups of the grid hold no
// items; sweeper s takes the large jobs j = s (mod n_sweepers) in order, waits for the block's last record, folds all of them and
// publishes ONE descriptor record per block (k_encode_reg_vec3's arrangement), which the item workgroups read instead of
// sweeping.  No deadlock: rounds are unweighted (a workgroup's items are G apart, at most two of them in a block), a workgroup
// publishes its current item before it waits for its parked item's descriptor, so whoever waits has published everything it
// holds of that block; a sweeper waits for item records only; induction over the blocks in order, as in k_encode_reg_f32.
// Anything larger, and explicit-mode jobs (raw bit arrays), stay on k_reduce_i64 + k_pack_i64.
// ------------------------------------------------------------------------------------------------
constexpr int kItemI = 4096;                        // int64 elements per item (32 KB, like a float item)
constexpr int kBufI = (kItemI / kGroup) * kStride64;      // dwords of the parked item: 128 groups x 68 dwords = 34,816 B
__device__ __forceinline__ int64_t items_of_i(int64_t n) { return (n + kItemI - 1) / kItemI; }

struct DescI { int64_t imin; int bits; int status; int64_t nbytes; };

// 33..64-bit widths: the lane re-bases and packs its group of 32 raw int64 IN PLACE.  After value i the stream has
// (i + 1) b / 32 <= 2 (i + 1) complete words, which overwrite dwords of values <= i only, all of them read already.
// Values at and beyond `nvalid` (past the block's end) pack as zero.
__device__ __forceinline__ void pack_group64_inplace(uint32_t* __restrict__ grp, uint32_t b, uint64_t imin, uint64_t mask, int nvalid) {
    uint32_t acc = 0, pos = 0;
    const uint32_t bh = b - 32;                    // 1..32
    uint32_t* out = grp;
#pragma unroll 1
    for (int i = 0; i < kGroup; ++i) {
        const u32x2 w = *reinterpret_cast<const u32x2*>(grp + 2 * i);
        const uint64_t v = i < nvalid ? (((((uint64_t)w.y << 32) | w.x) - imin) & mask) : 0ull;      // group.go:247, bit.go:104-107
        const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
        acc |= lo << pos;
        *out++ = acc;
        acc = pos ? lo >> (32 - pos) : 0;
        acc |= hi << pos;
        pos += bh;
        if (pos >= 32) {
            *out++ = acc;
            pos -= 32;
            acc = pos ? hi >> (bh - pos) : 0;
        }
    }
}

// write-out after pack_group64_inplace: stream word w of the item sits at group w / b, dword w % b of that group
template <int NT>
__device__ __forceinline__ void store_tile_groups(const uint32_t* __restrict__ lds_in, uint32_t b, uint8_t* __restrict__ dst,
                                                  int64_t tile_byte0, int64_t tile_bytes, int tid) {
    const int full = (int)(tile_bytes >> 4);
    for (int c = tid; c < full; c += NT) {
        uint32_t g = (uint32_t)(4 * c) / b, off = (uint32_t)(4 * c) - g * b;
        uint32_t r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r[k] = lds_in[g * kStride64 + off];
            if (++off == b) { off = 0; ++g; }
        }
        stg<u32x4>(dst + tile_byte0 + 16 * (int64_t)c, u32x4{r[0], r[1], r[2], r[3]});
    }
    const int rem = (int)(tile_bytes & 15);
    if (tid < rem) {
        const uint32_t w = 4 * (uint32_t)full + ((uint32_t)tid >> 2);
        const uint32_t g = w / b, off = w - g * b;
        stg<uint8_t>(dst + tile_byte0 + 16 * (int64_t)full + tid, (uint8_t)(lds_in[g * kStride64 + off] >> (8 * (tid & 3))));
    }
}

// wave 0.  Folds records (64-bit max / min from the four words of every record), returns false if one is missing.
__device__ __forceinline__ bool fold_records_i64(const ItemRec* __restrict__ r0, int nitems, uint32_t gen, int64_t* mx, int64_t* mn) {
    int64_t vmax = INT64_MIN, vmin = INT64_MAX;
    uint32_t missing = 0;
    for (int r = lane_id(); r < nitems; r += 64) {
        uint64_t w[4];
        load_rec_sc1(r0 + r, w);
#pragma unroll
        for (int k = 0; k < 4; ++k) missing |= (uint32_t)(w[k] >> 32) != gen;
        const int64_t a = (int64_t)(((uint64_t)(uint32_t)w[0] << 32) | (uint32_t)w[1]);
        const int64_t b = (int64_t)(((uint64_t)(uint32_t)w[2] << 32) | (uint32_t)w[3]);
        vmax = a > vmax ? a : vmax;
        vmin = b < vmin ? b : vmin;
    }
    if (__any(missing)) return false;
    *mx = wave_max_i64(vmax);
    *mn = wave_min_i64(vmin);
    return true;
}

__global__ __launch_bounds__(kThreadsR, 4) void k_encode_reg_i64(const DevJob* __restrict__ jobs, int njobs, int64_t total_items,
                                                                 const int* __restrict__ item_job, ItemRec* __restrict__ recs,
                                                                 uint32_t gen, uint32_t* __restrict__ abort_flag,
                                                                 uint32_t* __restrict__ abort_host, Meta* __restrict__ meta, int weighted,
                                                                 int64_t drop_item, ItemRec* __restrict__ drecs, int n_sweepers, int flat_lim) {
    // (drecs: one descriptor record per job, for blocks of more than flat_lim items; n_sweepers = 0 if the launch has none)
    __shared__ __attribute__((aligned(16))) uint32_t lds[kBufI];
    __shared__ int64_t red[2][kThreadsR / 64];
    __shared__ int64_t swr[2][kThreadsR / 64];
    __shared__ int swm[kThreadsR / 64];
    __shared__ DescI desc_s;
    __shared__ int abort_s;
    constexpr int kV = kItemI / (2 * kThreadsR);       // 16-byte loads per lane per item (8)
    constexpr int kW = kThreadsR / 64;
    const int tid = threadIdx.x;
    u32x4 e[kV];                                        // the current item: 16 int64 per lane, as (lo, hi) pairs
    int cj = 0;
    int64_t cbase = 0, cn = 0;
    const int64_t* cx = nullptr;
    auto locate = [&](int64_t t) {
        cj = item_job[t];
        cbase = (t - jobs[cj].tile0) * kItemI;
        cn = jobs[cj].n;
        cx = (const int64_t*)jobs[cj].src;
    };
    auto issue = [&]() {
        if (cbase + kItemI <= cn) {
            const int t = opaque_tid();
#pragma unroll
            for (int k = 0; k < kV; ++k) e[k] = ldg<u32x4>(cx + cbase + k * (2 * kThreadsR) + 2 * t);
        }
    };
    const int G = (int)gridDim.x - n_sweepers;          // workgroups that hold items
    if (tid == 0) abort_s = 0;
    if ((int)blockIdx.x >= G) {
        // ---- sweeper: the descriptors of the large blocks j = s (mod n_sweepers), in job order
        const int s_id = (int)blockIdx.x - G;
        int seen = 0;                                    // large jobs before j0 (uniform)
        lds_barrier();
        for (int j0 = 0; j0 < njobs; j0 += kThreadsR) {
            const int jt = j0 + tid;
            const bool big = jt < njobs && items_of_i(jobs[jt].n) > flat_lim;
            // the large jobs of this stretch, in order, through LDS (one flag per thread)
            reinterpret_cast<uint8_t*>(lds)[tid] = big ? 1 : 0;
            lds_barrier();
            for (int q = 0; q < kThreadsR && j0 + q < njobs; ++q) {
                if (!reinterpret_cast<const uint8_t*>(lds)[q]) continue;
                const bool mine = seen % n_sweepers == s_id;
                ++seen;
                if (!mine) continue;
                const int j = j0 + q;
                const DevJob jb = jobs[j];
                const int T = (int)items_of_i(jb.n);
                const ItemRec* r0 = recs + jb.tile0;
                uint32_t spins = 0;
                const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
                // a hint that the block is (nearly) complete: its last record
                for (;;) {
                    uint64_t w[4];
                    load_rec_sc1(r0 + (T - 1), w);
                    if ((uint32_t)(w[3] >> 32) == gen) break;
                    if (poll_abort(spins, t_start, abort_flag, abort_host)) { if (lane_id() == 0) abort_s = 1; break; }
                }
                lds_barrier();
                if (abort_s) return;
                for (;;) {
                    int64_t vmax = INT64_MIN, vmin = INT64_MAX;
                    uint32_t missing = 0;
                    for (int r = tid; r < T; r += kThreadsR) {
                        uint64_t w[4];
                        load_rec_sc1(r0 + r, w);
#pragma unroll
                        for (int k = 0; k < 4; ++k) missing |= (uint32_t)(w[k] >> 32) != gen;
                        const int64_t a = (int64_t)(((uint64_t)(uint32_t)w[0] << 32) | (uint32_t)w[1]);
                        const int64_t b = (int64_t)(((uint64_t)(uint32_t)w[2] << 32) | (uint32_t)w[3]);
                        vmax = a > vmax ? a : vmax; vmin = b < vmin ? b : vmin;
                    }
                    vmax = wave_max_i64(vmax); vmin = wave_min_i64(vmin);
                    const int miss = __any(missing);
                    if (lane_id() == 0) { swr[0][wave_id()] = vmax; swr[1][wave_id()] = vmin; swm[wave_id()] = miss; }
                    lds_barrier();
                    int64_t bmx = swr[0][0], bmn = swr[1][0];
                    int anymiss = swm[0];
#pragma unroll
                    for (int w = 1; w < kW; ++w) { bmx = swr[0][w] > bmx ? swr[0][w] : bmx; bmn = swr[1][w] < bmn ? swr[1][w] : bmn; anymiss |= swm[w]; }
                    if (!anymiss) {
                        if (tid < 4) {
                            DescI d = {bmn, 0, 0, 0};
                            d.bits = precision_needed((uint64_t)bmx - (uint64_t)bmn);           // group.go:246-248, bit.go:151-159
                            if (d.bits < 0) { d.status = -6; d.bits = 0; }
                            else {
                                d.nbytes = d.bits == 0 ? 0 : array_bytes(d.bits, jb.n);
                                if (d.nbytes > jb.cap) d.status = -5;
                            }
                            if (tid == 0) { Meta m = {d.imin, (int64_t)d.bits, d.nbytes, (int64_t)d.status}; meta[j] = m; }
                            // the record: {min high, min low, width | status code << 8, 0}, each word tagged
                            const uint32_t code = d.status == 0 ? 0u : d.status == -5 ? 2u : 3u;
                            const uint32_t val = tid == 0 ? (uint32_t)((uint64_t)d.imin >> 32) : tid == 1 ? (uint32_t)(uint64_t)d.imin
                                               : tid == 2 ? ((uint32_t)d.bits | (code << 8)) : 0u;
                            __hip_atomic_store(&drecs[j].g[tid], (uint64_t)val | ((uint64_t)gen << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        lds_barrier();                   // (swr / swm are free again)
                        break;
                    }
                    if (wave_id() == 0 && poll_abort(spins, t_start, abort_flag, abort_host) && lane_id() == 0) abort_s = 1;
                    lds_barrier();
                    if (abort_s) return;
                }
            }
            lds_barrier();                               // (the flags are rewritten by the next stretch)
        }
        return;
    }
    const int unit = weighted ? G / 4 : G;
    int round = 0;
    int64_t round_base = 0;
    auto next_item = [&]() -> int64_t {
        for (;;) {
            round_base += (int64_t)(weighted ? kRoundTable[weighted - 1][round] : 1) * unit;
            round = round + 1 == (weighted ? kRoundCount[weighted - 1] : 1) ? 0 : round + 1;
            if (round_base >= total_items) return total_items;
            if ((int)blockIdx.x < (weighted ? kRoundTable[weighted - 1][round] : 1) * unit) return round_base + blockIdx.x;
        }
    };
    int64_t it = blockIdx.x;
    bool have_cur = it < total_items;
    if (have_cur) { locate(it); issue(); }
    bool have_prev = false;
    int pj = 0;
    int64_t pbase = 0, pit = 0;

    while (have_cur || have_prev) {
        const int sj = cj;
        const int64_t sbase = cbase, sit = it;
        const int64_t* const sx = cx + cbase;
        const int64_t srem = cn - cbase;
        const int64_t nit = have_cur ? next_item() : total_items;
        const bool have_next = have_cur && nit < total_items;
        if (have_next) locate(nit);

        // ---- 1. the records of item i-1's block, one per lane, issued ahead of everything else of the trip (they were
        // published a trip ago); folded after the statistics of item i
        DevJob pjb;
        int pitems = 0;
        bool pbig = false;                               // the parked item's block is a large one: its descriptor comes from a sweeper
        u32x4 swa, swb;
        if (have_prev) {
            pjb = jobs[pj];
            pitems = (int)items_of_i(pjb.n);
            pbig = pitems > flat_lim;
            if (!pbig && tid < pitems) rec_prefetch(recs + pjb.tile0, pitems, tid, swa, swb);     // (waves without a record skip the instructions)
        }

        if (have_cur) {
            // ---- 2. statistics of item i: int64Min (group.go:350-357) and the max that ArrayBuffer.Bits needs (bit.go:151-159)
            int64_t mn = INT64_MAX, mx = INT64_MIN;
            if (srem >= kItemI) {
#pragma unroll
                for (int k = 0; k < kV; ++k) {
                    const int64_t a = (int64_t)(((uint64_t)e[k].y << 32) | e[k].x), b = (int64_t)(((uint64_t)e[k].w << 32) | e[k].z);
                    mn = a < mn ? a : mn; mx = a > mx ? a : mx;
                    mn = b < mn ? b : mn; mx = b > mx ? b : mx;
                }
            } else {
#pragma unroll 1
                for (int i = tid; i < (int)srem; i += kThreadsR) {
                    const int64_t a = ldg<int64_t>(sx + i);
                    mn = a < mn ? a : mn; mx = a > mx ? a : mx;
                }
            }
            mn = wave_min_i64(mn); mx = wave_max_i64(mx);
            if (lane_id() == 0) { red[0][wave_id()] = mx; red[1][wave_id()] = mn; }
        }
        if (have_prev && !pbig) {
            // fold what the record loads brought (blocks of more than 256 items: the second half follows)
            const bool mine = tid < pitems;
            uint32_t missing = mine && ((swa.y != gen) | (swa.w != gen) | (swb.y != gen) | (swb.w != gen));
            int64_t vmax = mine ? (int64_t)(((uint64_t)swa.x << 32) | swa.z) : INT64_MIN;
            int64_t vmin = mine ? (int64_t)(((uint64_t)swb.x << 32) | swb.z) : INT64_MAX;
            if (pitems > kThreadsR && tid + kThreadsR < pitems) {
                uint64_t w[4];
                load_rec_sc1(recs + pjb.tile0 + kThreadsR + tid, w);
#pragma unroll
                for (int k = 0; k < 4; ++k) missing |= (uint32_t)(w[k] >> 32) != gen;
                const int64_t a = (int64_t)(((uint64_t)(uint32_t)w[0] << 32) | (uint32_t)w[1]);
                const int64_t b = (int64_t)(((uint64_t)(uint32_t)w[2] << 32) | (uint32_t)w[3]);
                vmax = a > vmax ? a : vmax; vmin = b < vmin ? b : vmin;
            }
            vmax = wave_max_i64(vmax); vmin = wave_min_i64(vmin);
            const int miss = __any(missing);
            if (lane_id() == 0) { swr[0][wave_id()] = vmax; swr[1][wave_id()] = vmin; swm[wave_id()] = miss; }
        }
        lds_barrier();
        if (have_cur && tid < 4 && sit != drop_item) {           // (drop_item: test hook, see k_encode_reg_f32)
            int64_t a = red[0][0], b = red[1][0];
#pragma unroll
            for (int w = 1; w < kW; ++w) { a = red[0][w] > a ? red[0][w] : a; b = red[1][w] < b ? red[1][w] : b; }
            const uint64_t v64 = (uint64_t)(tid < 2 ? a : b);
            const uint32_t half = (tid & 1) ? (uint32_t)v64 : (uint32_t)(v64 >> 32);
            __hip_atomic_store(&recs[sit].g[tid], (uint64_t)half | ((uint64_t)gen << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!have_prev) lds_barrier();      // (first trip only: see k_encode_reg_f32 -- lanes 0..3 have read red[] before any wave writes it again)

        if (have_prev) {
            // ---- 3. item i-1: the descriptor of its block (wave 0), then re-base and pack (lanes 0..127)
            if (wave_id() == 0 && pbig) {
                // the block's descriptor record, published by a sweeper once the block's last record is out
                uint32_t spins = 0;
                const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    uint64_t w[4];
                    load_rec_sc1(drecs + pj, w);
                    if ((uint32_t)(w[0] >> 32) == gen && (uint32_t)(w[1] >> 32) == gen && (uint32_t)(w[2] >> 32) == gen && (uint32_t)(w[3] >> 32) == gen) {
                        if (lane_id() == 0) {
                            DescI d;
                            d.imin = (int64_t)(((uint64_t)(uint32_t)w[0] << 32) | (uint32_t)w[1]);
                            const uint32_t info = (uint32_t)w[2];
                            d.bits = (int)(info & 0xffu);
                            const uint32_t code = info >> 8;
                            d.status = code == 0 ? 0 : code == 2 ? -5 : -6;
                            d.nbytes = d.bits == 0 ? 0 : array_bytes(d.bits, pjb.n);
                            desc_s = d;
                        }
                        break;
                    }
                    if (poll_abort(spins, t_start, abort_flag, abort_host)) { if (lane_id() == 0) abort_s = 1; break; }
                }
            }
            if (wave_id() == 0 && !pbig) {
                int64_t bmx = swr[0][0], bmn = swr[1][0];
                int miss = swm[0];
#pragma unroll
                for (int w = 1; w < kW; ++w) { bmx = swr[0][w] > bmx ? swr[0][w] : bmx; bmn = swr[1][w] < bmn ? swr[1][w] : bmn; miss |= swm[w]; }
                bool ok = true;
                if (__builtin_expect(miss, 0)) {
                    // some record was not there yet: poll until the block is complete (bounded: see poll_abort)
                    uint32_t spins = 0;
                    const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
                    while (!fold_records_i64(recs + pjb.tile0, pitems, gen, &bmx, &bmn)) {
                        if (poll_abort(spins, t_start, abort_flag, abort_host)) { ok = false; break; }
                    }
                }
                if (lane_id() == 0) {
                    DescI d = {bmn, 0, 0, 0};
                    if (!ok) abort_s = 1;
                    else {
                        d.bits = precision_needed((uint64_t)bmx - (uint64_t)bmn);           // group.go:246-248, bit.go:151-159
                        if (d.bits < 0) { d.status = -6; d.bits = 0; }
                        else {
                            d.nbytes = d.bits == 0 ? 0 : array_bytes(d.bits, pjb.n);
                            if (d.nbytes > pjb.cap) d.status = -5;
                        }
                        if (pit == pjb.tile0) { Meta m = {d.imin, (int64_t)d.bits, d.nbytes, (int64_t)d.status}; meta[pj] = m; }
                    }
                    desc_s = d;
                }
            }
            lds_barrier();
            if (abort_s) break;                                    // (uniform: every wave reads the same word)
            const DescI d = desc_s;
            const int ok = __builtin_amdgcn_readfirstlane(d.status == 0 && d.bits != 0 && d.bits <= 32);
            if (__builtin_amdgcn_readfirstlane(d.status == 0 && d.bits > 32)) {
                // widths 33..64: in place, one group per lane of waves 0 and 1
                const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane(d.bits);
                if (tid < kItemI / kGroup) {
                    const int64_t left = pjb.n - (pbase + (int64_t)kGroup * tid);
                    pack_group64_inplace(lds + tid * kStride64, b, (uint64_t)d.imin, b == 64 ? ~0ull : ((1ull << b) - 1),
                                         left >= kGroup ? kGroup : (left > 0 ? (int)left : 0));
                }
                lds_barrier();
                const int64_t tile_byte0 = pbase / 8 * b;
                int64_t tile_bytes = (int64_t)kItemI / 8 * b;
                if (tile_byte0 + tile_bytes > d.nbytes) tile_bytes = d.nbytes - tile_byte0;
                store_tile_groups<kThreadsR>(lds, b, (uint8_t*)pjb.dst, tile_byte0, tile_bytes, opaque_tid());
            }
            uint32_t q[kGroup];
            if (ok && tid < kItemI / kGroup) {
                const uint64_t imin = (uint64_t)d.imin;
                const uint32_t* grp = lds + tid * kStride64;
#pragma unroll
                for (int k = 0; k < kGroup / 2; ++k) {
                    const u32x4 r = *reinterpret_cast<const u32x4*>(grp + 4 * k);
                    q[2 * k] = (uint32_t)((((uint64_t)r.y << 32) | r.x) - imin);      // group.go:247; below 2^32 by the choice of the width
                    q[2 * k + 1] = (uint32_t)((((uint64_t)r.w << 32) | r.z) - imin);
                }
                if (pbase + kItemI > pjb.n) {
                    const int64_t e0 = pbase + (int64_t)kGroup * tid;
#pragma unroll
                    for (int i = 0; i < kGroup; ++i) if (e0 + i >= pjb.n) q[i] = 0;
                }
            }
            lds_barrier();                                         // every packing lane holds its group: the stage can be overwritten
            if (ok) {
                const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane(d.bits);
                if (tid < kItemI / kGroup) pack_group32_switch(q, b, lds + tid * pack_stride(b));
                lds_barrier();
                const int64_t tile_byte0 = pbase / 8 * b;
                int64_t tile_bytes = (int64_t)kItemI / 8 * b;
                if (tile_byte0 + tile_bytes > d.nbytes) tile_bytes = d.nbytes - tile_byte0;
                store_tile<kThreadsR>(lds, (uint8_t*)pjb.dst, tile_byte0, tile_bytes, opaque_tid(), (b & 7u) == 0 ? b >> 2 : 0u);
            }
            lds_barrier();
        }
        // ---- park item i, start item i+1
        if (have_cur) {
            if (srem >= kItemI) {
#pragma unroll
                for (int k = 0; k < kV; ++k)
                    *reinterpret_cast<u32x4*>(lds + stage64_idx(2 * (k * kThreadsR + opaque_tid()))) = e[k];
            } else {
#pragma unroll 1
                for (int i = tid; i < kItemI; i += kThreadsR) {
                    const uint64_t v = i < (int)srem ? (uint64_t)ldg<int64_t>(sx + i) : 0ull;
                    *reinterpret_cast<u32x2*>(lds + stage64_idx(i)) = u32x2{(uint32_t)v, (uint32_t)(v >> 32)};
                }
            }
            if (have_next) issue();
        }
        have_prev = have_cur;
        pj = sj; pbase = sbase; pit = sit;
        it = nit;
        have_cur = have_next;
    }
}

// ------------------------------------------------------------------------------------------------
// pack pass, int64 blocks (intGroup.writeData, and raw bit.BufferedArray in explicit mode).
// The width is only known on the device, so the kernel carries both the 32-bit and the 64-bit path.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads, 4) void k_pack_i64(const DevJob* __restrict__ jobs, int njobs,
                                                          int64_t total_tiles,
                                                          const uint64_t* __restrict__ stats,
                                                          Meta* __restrict__ meta) {
    // One 36 KB LDS block for both widths (4 workgroups per CU): a stage of 8,192 32-bit values, or -- for
    // widths above 32 -- a quarter tile of 64-bit values plus a separate output region.
    __shared__ __attribute__((aligned(16))) uint32_t lds[kLds32 / 4];
    constexpr int kV = kTile / (2 * kThreads);       // 16-byte loads per thread per tile (16)
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
        uint64_t st[5] = {0, 0, 0, 0, 0};
        if (jb.mode == kModeStats) load_stats<false>(stats, j, st);
        const Desc d = make_desc_i64(jb, st);
        if (t0 == jb.tile0 && tid == 0) {
            Meta m = {d.imin, (int64_t)d.bits, d.nbytes, (int64_t)d.status};
            meta[j] = m;
        }
        const int ok = __builtin_amdgcn_readfirstlane(d.status == 0 && d.bits != 0);
        if (ok) {
            const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane(d.bits);
            const uint64_t imin = (uint64_t)d.imin;
            const uint64_t mask = b == 64 ? ~0ull : ((1ull << b) - 1);   // bit.go:104-107
            if (b <= 32) {
                // the range is walked backwards, so only the first tile handled can be the block's
                // ragged last one and every prefetched tile is full
                int64_t t = tend - 1;
                i64x2 v[kV];
                bool cur_full = (t - jb.tile0) * kTile + kTile <= jb.n;
                if (cur_full) {
#pragma unroll
                    for (int k = 0; k < kV; ++k) v[k] = ldg<i64x2>(x + (t - jb.tile0) * kTile + 2 * (k * kThreads + tid));
                }
                for (; t >= t0; --t) {
                    const int64_t base = (t - jb.tile0) * kTile;
                    if (cur_full) {
#pragma unroll
                        for (int k = 0; k < kV; ++k) {
                            u32x2 r;
                            r.x = (uint32_t)(((uint64_t)v[k].x - imin) & mask);      // group.go:247
                            r.y = (uint32_t)(((uint64_t)v[k].y - imin) & mask);
                            *reinterpret_cast<u32x2*>(lds + stage32_idx(2 * (k * kThreads + tid))) = r;
                        }
                    } else {
                        for (int i = tid; i < kTile; i += kThreads) {
                            uint64_t r = 0;
                            if (base + i < jb.n) r = ((uint64_t)ldg<int64_t>(x + base + i) - imin) & mask;
                            lds[stage32_idx(i)] = (uint32_t)r;
                        }
                    }
                    if (t - 1 >= t0) {               // next tile's loads stay in flight across the barriers
#pragma unroll
                        for (int k = 0; k < kV; ++k) v[k] = ldg<i64x2>(x + base - kTile + 2 * (k * kThreads + tid));
                    }
                    cur_full = true;
                    lds_barrier();
                    uint32_t g[kGroup];
#pragma unroll
                    for (int k = 0; k < kGroup / 4; ++k) {
                        const u32x4 r = *reinterpret_cast<const u32x4*>(lds + tid * kStride32 + 4 * k);
                        g[4 * k] = r.x; g[4 * k + 1] = r.y; g[4 * k + 2] = r.z; g[4 * k + 3] = r.w;
                    }
                    lds_barrier();
                    pack_group32(g, b, lds + tid * pack_stride(b));
                    lds_barrier();
                    const int64_t tile_byte0 = base / 8 * b;
                    int64_t tile_bytes = (int64_t)kTile / 8 * b;
                    if (tile_byte0 + tile_bytes > d.nbytes) tile_bytes = d.nbytes - tile_byte0;
                    store_tile(lds, (uint8_t*)jb.dst, tile_byte0, tile_bytes, threadIdx.x, (b & 7u) == 0 ? b >> 2 : 0u);
                    lds_barrier();
                }
            } else {
                // widths above 32 (rare): a quarter tile of 64-bit values at a time; lanes 0..63 pack one
                // 32-element group each out of the stage into a separate output region of the same LDS
                constexpr int kQ = kTile / 4;                       // 2,048 elements = 64 groups
                constexpr int kOut = (kQ / kGroup) * kStride64;     // output region starts after the stage (dwords)
                static_assert((kOut + (kQ / kGroup) * 64) * 4 <= kLds32, "wide stage + output fit");
                for (int64_t t = tend - 1; t >= t0; --t) {
                    const int64_t base = (t - jb.tile0) * kTile;
                    for (int h = 0; h < 4; ++h) {
                        const int64_t hbase = base + (int64_t)h * kQ;
                        if (hbase >= jb.n) break;
                        for (int i = 2 * tid; i < kQ; i += 2 * kThreads) {
                            u64x2 r;
                            r.x = hbase + i < jb.n ? (((uint64_t)ldg<int64_t>(x + hbase + i) - imin) & mask) : 0;
                            r.y = hbase + i + 1 < jb.n ? (((uint64_t)ldg<int64_t>(x + hbase + i + 1) - imin) & mask) : 0;
                            *reinterpret_cast<u64x2*>(lds + stage64_idx(i)) = r;
                        }
                        lds_barrier();
                        if (tid < kQ / kGroup) pack_group64_lds(lds + tid * kStride64, b, lds + kOut + tid * b);
                        lds_barrier();
                        const int64_t tile_byte0 = hbase / 8 * b;
                        int64_t tile_bytes = (int64_t)kQ / 8 * b;
                        if (tile_byte0 + tile_bytes > d.nbytes) tile_bytes = d.nbytes - tile_byte0;
                        store_tile(lds + kOut, (uint8_t*)jb.dst, tile_byte0, tile_bytes);
                        lds_barrier();
                    }
                }
            }
        }
        t0 = tend;
        ++j;
    }
}

// ------------------------------------------------------------------------------------------------
// decode pass (tile = kTileDD elements).  The stream of a block may start at any byte (file offsets are byte granular,
// block_index.go:16-23): words are read from the enclosing aligned 64-bit words and funnel-shifted.
// ------------------------------------------------------------------------------------------------
struct StreamView {          // aligned view of a block's packed stream
    const uint64_t* a;       // enclosing aligned words
    int64_t last;            // index of the last aligned word that holds stream bytes
    uint32_t sh;             // bit offset of the stream inside a[0]
};

__device__ __forceinline__ StreamView stream_view(const DevJob& jb, uint32_t b) {
    const uintptr_t p = (uintptr_t)jb.src;
    StreamView v;
    v.a = reinterpret_cast<const uint64_t*>(p & ~(uintptr_t)7);
    v.sh = (uint32_t)(p & 7) * 8;
    const int64_t nbytes = array_bytes(b, jb.n);
    v.last = (int64_t)(((p + nbytes - 1) & ~(uintptr_t)7) - (p & ~(uintptr_t)7)) >> 3;
    return v;
}

// The aligned words that hold one tile's bits (kTileD elements = kTileD/64*b words, one look-ahead word, one more for a stream
// that starts inside a word) -> registers, exactly as they lie in memory: the stream's bit offset `sh` is added where the
// elements are extracted, not shifted out here.  Nothing is computed on the loaded words and none is merged with a constant, so
// all the loads of the tile are in flight together and across the extraction of the tile before (a word past the stream's end is
// read as the last valid one: it only feeds elements beyond n, which are never stored).
template <int NW>
__device__ __forceinline__ void load_tile_words(uint64_t (&w)[NW], const StreamView& sv, int64_t base, uint32_t b) {
    const int64_t w0 = base / 64 * b;
    const int nw = kTileD / 64 * (int)b + 2;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int i = k * kThreads + (int)threadIdx.x;
        if (i < nw) {
            const int64_t kk = w0 + i;
            w[k] = ldg<uint64_t>(sv.a + (kk < sv.last ? kk : sv.last));
        }
    }
}
template <int NW>
__device__ __forceinline__ void stage_tile_words(uint64_t* __restrict__ lw, const uint64_t (&w)[NW], uint32_t b) {
    const int nw = kTileD / 64 * (int)b + 2;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const int i = k * kThreads + (int)threadIdx.x;
        if (i < nw) lw[i] = w[k];
    }
}

// element i of the staged tile; `sh` = bit offset of the stream inside the first staged word (StreamView::sh)
__device__ __forceinline__ uint32_t extract32(const uint32_t* __restrict__ l32, uint32_t i, uint32_t b,
                                              uint32_t mask, uint32_t sh0) {
    const uint32_t bitpos = i * b + sh0;
    const uint32_t di = bitpos >> 5, sh = bitpos & 31;
    const uint64_t win = (uint64_t)l32[di] | ((uint64_t)l32[di + 1] << 32);
    return (uint32_t)(win >> sh) & mask;
}
__device__ __forceinline__ uint64_t extract64(const uint64_t* __restrict__ l64, uint32_t i, uint32_t b,
                                              uint64_t mask, uint32_t sh0) {
    const uint32_t bitpos = i * b + sh0;
    const uint32_t wi = bitpos >> 6, sh = bitpos & 63;
    uint64_t v = l64[wi] >> sh;
    if (sh != 0) v |= l64[wi + 1] << (64 - sh);
    return v & mask;
}

// floatGroup.readData, group.go:299-310.  NARROW: bits <= 32 and min, min+2^bits, pixels fit int32.
// The next tile's stream words are loaded into registers before the current tile is extracted, and
// the LDS word buffer is double buffered: one barrier per tile, loads in flight across it.
// XF: the launch has jobs whose values go through minh's reader transform, float32(math.Pow(10, float64(x)))
// (go/minh/minh.go:315-319; DevJob.xform & kXfLog10), applied to the decoded value before it is stored -- the column is
// written once, not written, read back and written again by a pass of its own.  An instantiation of its own (2.5 KB of
// tables in LDS, float64 arithmetic): the plain kernel is not touched.
// (the XF instantiation needs 124 registers, four waves per SIMD; bounded to five, six or eight waves it spills and takes 7.8, 10.5,
// 14.1 ms where this takes 5.3 for the 24 float columns of tools/bench_c3.py --minh)
template <bool NARROW, bool XF>
__global__ __launch_bounds__(kThreads) void k_unpack_f32(const DevJob* __restrict__ jobs, int njobs,
                                                         int64_t total_tiles) {
    constexpr int kLW = kTileD / 64 * (NARROW ? 32 : 64) + 2;          // words per LDS buffer
    constexpr int kNW = (kLW + kThreads - 1) / kThreads;              // words per thread
    __shared__ __attribute__((aligned(16))) uint64_t lw2[2][kLW];
    __shared__ typename std::conditional<XF, FastTables, double>::type ftab_s;
    FastTables* const ftab_p = XF ? reinterpret_cast<FastTables*>(&ftab_s) : nullptr;
    if (XF) fast_tables_load(ftab_p);
    const int tid = threadIdx.x;
    int64_t t0 = total_tiles * blockIdx.x / gridDim.x;
    const int64_t t1 = total_tiles * (blockIdx.x + 1) / gridDim.x;
    if (t0 >= t1) return;
    int j = find_job_guess<kTileD>(jobs, njobs, t0);
    int buf = 0;

    while (t0 < t1) {
        const DevJob jb = jobs[j];
        const int64_t jend = jb.tile0 + tiles_of_d(jb.n);
        const int64_t tend = t1 < jend ? t1 : jend;
        float* __restrict__ out = (float*)jb.dst;
        const uint32_t b = (uint32_t)jb.bits;
        const float low = jb.low, dx = jb.dxq;
        const bool per = jb.periodic == 1;
        const bool unlog = XF && (jb.xform & kXfLog10);
        const uint64_t mask64 = b == 0 ? 0 : (b == 64 ? ~0ull : ((1ull << b) - 1));
        const StreamView sv = stream_view(jb, b);
        uint64_t w[kNW];
        if (b != 0) load_tile_words<kNW>(w, sv, (t0 - jb.tile0) * kTileD, b);

        for (int64_t t = t0; t < tend; ++t) {
            const int64_t base = (t - jb.tile0) * kTileD;
            const uint64_t* lw = lw2[buf];
            if (b != 0) {
                stage_tile_words<kNW>(lw2[buf], w, b);
                if (t + 1 < tend) load_tile_words<kNW>(w, sv, base + kTileD, b);
                lds_barrier();
            }
            const uint64_t key = dither_key(jb.seed, jb.block, (uint64_t)base >> 32);
            const uint32_t ka = (uint32_t)key, kb = (uint32_t)(key >> 32);
#pragma unroll
            for (int k = 0; k < kTileD / (4 * kThreads); ++k) {
                const uint32_t i0 = 4 * (k * kThreads + tid);
                f32x4 r;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const uint32_t i = i0 + c;
                    double qd;
                    if (NARROW) {
                        const uint32_t v = b == 0 ? 0 : extract32(reinterpret_cast<const uint32_t*>(lw), i, b, (uint32_t)mask64, sv.sh);
                        int32_t q = (int32_t)jb.min + (int32_t)v;            // group.go:262
                        if (per) {                                          // bound(buf, 0, pixels)
                            if (q < 0) q += (int32_t)jb.pixels;
                            else if (q >= (int32_t)jb.pixels) q -= (int32_t)jb.pixels;
                        }
                        qd = (double)q;
                    } else {
                        const uint64_t v = b == 0 ? 0 : extract64(lw, i, b, mask64, sv.sh);
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
                if (unlog) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) r[c] = go_pow10_f32(r[c], ftab_p);
                }
                if (base + i0 + 3 < jb.n) {
                    stg<f32x4>(out + base + i0, r);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) if (base + i0 + c < jb.n) stg<float>(out + base + i0 + c, r[c]);
                }
            }
            buf ^= 1;
        }
        t0 = tend;
        ++j;
    }
}

// intGroup.readData, group.go:257-263; with min == 0 it is (*bit.Array).Slice, bit.go:29-82
__global__ __launch_bounds__(kThreads) void k_unpack_i64(const DevJob* __restrict__ jobs, int njobs,
                                                         int64_t total_tiles) {
    constexpr int kLW = kTileD / 64 * 64 + 2;
    constexpr int kNW = (kLW + kThreads - 1) / kThreads;
    __shared__ __attribute__((aligned(16))) uint64_t lw2[2][kLW];
    const int tid = threadIdx.x;
    int64_t t0 = total_tiles * blockIdx.x / gridDim.x;
    const int64_t t1 = total_tiles * (blockIdx.x + 1) / gridDim.x;
    if (t0 >= t1) return;
    int j = find_job_guess<kTileD>(jobs, njobs, t0);
    int buf = 0;

    while (t0 < t1) {
        const DevJob jb = jobs[j];
        const int64_t jend = jb.tile0 + tiles_of_d(jb.n);
        const int64_t tend = t1 < jend ? t1 : jend;
        int64_t* __restrict__ out = (int64_t*)jb.dst;
        const uint32_t b = (uint32_t)jb.bits;
        const uint64_t mask64 = b == 0 ? 0 : (b == 64 ? ~0ull : ((1ull << b) - 1));
        const uint64_t mn = (uint64_t)jb.min;
        const StreamView sv = stream_view(jb, b);
        uint64_t w[kNW];
        if (b != 0) load_tile_words<kNW>(w, sv, (t0 - jb.tile0) * kTileD, b);

        for (int64_t t = t0; t < tend; ++t) {
            const int64_t base = (t - jb.tile0) * kTileD;
            const uint64_t* lw = lw2[buf];
            if (b != 0) {
                stage_tile_words<kNW>(lw2[buf], w, b);
                if (t + 1 < tend) load_tile_words<kNW>(w, sv, base + kTileD, b);
                lds_barrier();
            }
#pragma unroll
            for (int k = 0; k < kTileD / (2 * kThreads); ++k) {
                const uint32_t i0 = 2 * (k * kThreads + tid);
                const uint64_t v0 = b == 0 ? 0 : extract64(lw, i0, b, mask64, sv.sh);
                const uint64_t v1 = b == 0 ? 0 : extract64(lw, i0 + 1, b, mask64, sv.sh);
                i64x2 r;
                r.x = (int64_t)(mn + v0);
                r.y = (int64_t)(mn + v1);
                if (base + i0 + 1 < jb.n) stg<i64x2>(out + base + i0, r);
                else if (base + i0 < jb.n) stg<int64_t>(out + base + i0, r.x);
            }
            buf ^= 1;
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
// batch set-up in ONE small launch: copy the job records (and the fused launch's group tables) out of
// host-pinned memory and zero the batch's statistics / synchronisation words.  This replaces a
// host-to-device copy plus several memsets, each of which is a separate engine command with its own
// cross-queue dependency (tens of microseconds per batch).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_prepare(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst,
                                                      int64_t copy_words, uint32_t* __restrict__ zero0, int64_t zero0_words,
                                                      uint32_t* __restrict__ zero1, int64_t zero1_words,
                                                      const uint32_t* __restrict__ src2, uint32_t* __restrict__ dst2,
                                                      int64_t copy2_words) {
    const int64_t i0 = (int64_t)blockIdx.x * kThreads + threadIdx.x, step = (int64_t)gridDim.x * kThreads;
    for (int64_t i = i0; i < copy_words; i += step) dst[i] = src[i];
    for (int64_t i = i0; i < copy2_words; i += step) dst2[i] = src2[i];
    for (int64_t i = i0; i < zero0_words; i += step) zero0[i] = 0;
    for (int64_t i = i0; i < zero1_words; i += step) zero1[i] = 0;
}

#include "mnw_subcell.hip"
#include "mnw_minh.hip"
#include "mnw_text.hip"
#include "mnw_cut.hip"

// ------------------------------------------------------------------------------------------------
// launch wrappers (called from mnw_api.cpp)
// ------------------------------------------------------------------------------------------------
void launch_prepare(hipStream_t s, const void* src, void* dst, int64_t copy_bytes, void* zero0, int64_t zero0_bytes,
                    void* zero1, int64_t zero1_bytes, const void* src2, void* dst2, int64_t copy2_bytes) {
    const int64_t words = (copy_bytes + copy2_bytes + zero0_bytes + zero1_bytes) / 4;
    int grid = (int)((words + kThreads * 4 - 1) / (kThreads * 4));
    if (grid < 1) grid = 1;
    if (grid > 256) grid = 256;
    hipLaunchKernelGGL(k_prepare, dim3(grid), dim3(kThreads), 0, s, (const uint32_t*)src, (uint32_t*)dst, copy_bytes / 4,
                       (uint32_t*)zero0, zero0_bytes / 4, (uint32_t*)zero1, zero1_bytes / 4,
                       (const uint32_t*)src2, (uint32_t*)dst2, copy2_bytes / 4);
}
void launch_reduce_f32(hipStream_t s, int grid, const DevJob* jobs, int njobs, int64_t tiles, uint64_t* stats, void* sync, Meta* meta) {
    hipLaunchKernelGGL(k_reduce_f32, dim3(grid), dim3(kThreads), 0, s, jobs, njobs, tiles, stats, (JobSync*)sync, meta);
}
void launch_reduce_i64(hipStream_t s, int grid, const DevJob* jobs, int njobs, int64_t tiles, uint64_t* stats) {
    hipLaunchKernelGGL(k_reduce_i64, dim3(grid), dim3(kThreads), 0, s, jobs, njobs, tiles, stats);
}
void launch_pack_f32(hipStream_t s, int grid, const DevJob* jobs, int njobs, int64_t tiles, void* sync) {
    hipLaunchKernelGGL(k_pack_f32, dim3(grid), dim3(kThreads), 0, s, jobs, njobs, tiles, (JobSync*)sync);
}
void launch_encode_f32(hipStream_t s, int grid, const DevJob* jobs, int njobs, const int64_t* gtile, const int* gjob, int ngroups,
                       uint64_t* stats, void* sync, uint32_t* abort_flag, Meta* meta) {
    hipLaunchKernelGGL(k_encode_f32, dim3(grid), dim3(kThreads), 0, s, jobs, njobs, gtile, gjob, ngroups, stats, (JobSync*)sync, abort_flag, meta);
}
size_t job_sync_bytes() { return sizeof(JobSync); }
void launch_encode_reg_f32(hipStream_t s, int grid, const DevJob* jobs, int njobs, int64_t items, const int* item_job, void* recs,
                           void* grecs, uint32_t gen, uint32_t* abort_flag, uint32_t* abort_host, Meta* meta, int weighted,
                           int64_t test_drop_item, int any_xform, int big, void* drecs = nullptr, int n_sweepers = 0, int flat_lim = kMaxSweep) {
    // grid: the item workgroups AND the sweepers (the last n_sweepers, big launches only; flat_lim: blocks of more items are "large")
    if (!big) n_sweepers = 0;
    const int w = (weighted > 0 && weighted <= kRoundTables && (grid - n_sweepers) % 4 == 0) ? weighted : 0;     // 1 + the round table
#define MNW_LAUNCH_REG(XF, BIG) hipLaunchKernelGGL((k_encode_reg_f32<XF, BIG>), dim3(grid), dim3(kThreadsR), 0, s, jobs, njobs, items, item_job, \
                                                   (ItemRec*)recs, (ItemRec*)grecs, gen, abort_flag, abort_host, meta, w, test_drop_item, \
                                                   (ItemRec*)drecs, n_sweepers, flat_lim)
    if (any_xform) { if (big) MNW_LAUNCH_REG(true, true); else MNW_LAUNCH_REG(true, false); }
    else { if (big) MNW_LAUNCH_REG(false, true); else MNW_LAUNCH_REG(false, false); }
#undef MNW_LAUNCH_REG
}
void launch_encode_reg_i64(hipStream_t s, int grid, const DevJob* jobs, int njobs, int64_t items, const int* item_job, void* recs,
                           uint32_t gen, uint32_t* abort_flag, uint32_t* abort_host, Meta* meta, int weighted, int64_t drop_item,
                           void* drecs, int n_sweepers, int flat_lim) {
    // grid: the item workgroups AND the sweepers (the last n_sweepers; 0 when no block has more than flat_lim items)
    hipLaunchKernelGGL(k_encode_reg_i64, dim3(grid), dim3(kThreadsR), 0, s, jobs, njobs, items, item_job, (ItemRec*)recs, gen,
                       abort_flag, abort_host, meta, (weighted > 0 && weighted <= kRoundTables && n_sweepers == 0 && grid % 4 == 0) ? weighted : 0, drop_item,
                       (ItemRec*)drecs, n_sweepers, flat_lim);
}
int encode_reg_i64_sweepers() { return 8; }
int encode_reg_f32_sweepers() { return 8; }
int64_t reg_item_elems_i64() { return kItemI; }
int encode_reg_i64_blocks_per_cu() {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_encode_reg_i64, kThreadsR, 0) != hipSuccess) return 0;
    return n;
}
size_t item_rec_bytes() { return sizeof(ItemRec); }
int reg_max_items_per_block() { return kMaxSweep; }
int reg_max_items_per_big_block() { return kMaxBig; }
int reg_group_items() { return kGroupItems; }
int64_t reg_item_elems() { return kItem; }
int encode_reg_f32_blocks_per_cu() {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_encode_reg_f32<false, true>, kThreadsR, 0) != hipSuccess) return 0;
    return n;
}
int encode_reg_f32_xf_blocks_per_cu() {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_encode_reg_f32<true, true>, kThreadsR, 0) != hipSuccess) return 0;
    return n;
}
int encode_f32_blocks_per_cu() {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_encode_f32, kThreads, 0) != hipSuccess) return 0;
    return n;
}
void launch_pack_i64(hipStream_t s, int grid, const DevJob* jobs, int njobs, int64_t tiles, const uint64_t* stats, Meta* meta) {
    hipLaunchKernelGGL(k_pack_i64, dim3(grid), dim3(kThreads), 0, s, jobs, njobs, tiles, stats, meta);
}
void launch_unpack_f32(hipStream_t s, int grid, bool narrow, bool xf, const DevJob* jobs, int njobs, int64_t tiles) {
    if (narrow) {
        if (xf) hipLaunchKernelGGL((k_unpack_f32<true, true>), dim3(grid), dim3(kThreads), 0, s, jobs, njobs, tiles);
        else hipLaunchKernelGGL((k_unpack_f32<true, false>), dim3(grid), dim3(kThreads), 0, s, jobs, njobs, tiles);
    } else {
        if (xf) hipLaunchKernelGGL((k_unpack_f32<false, true>), dim3(grid), dim3(kThreads), 0, s, jobs, njobs, tiles);
        else hipLaunchKernelGGL((k_unpack_f32<false, false>), dim3(grid), dim3(kThreads), 0, s, jobs, njobs, tiles);
    }
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
