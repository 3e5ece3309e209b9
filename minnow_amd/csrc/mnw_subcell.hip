// minp's sub-cell layout on the GPU (go/minp/minp.go:246-300): the particles of one file are an
// array-of-structures grid vec[nFile^3][3]; a file stores, per component k and per sub-cell sc, the nSub^3
// values of that sub-cube as one FloatGroup block.  These kernels are the getSubCell / setSubCell loops
// and the bounds() scan; the blocks themselves go through the ordinary float-group kernels.
//
//   k_subcell_gather   vec[i][k] -> out[k][sc][j]      reads 12 B + writes 12 B per particle
//   k_subcell_scatter  in[k][sc][j] -> vec[i][k]       the same in reverse, with the reader's periodic
//                                                      fix-up (minp.go:188-196) folded in
//   k_bounds3          min / max per component         reads 12 B per particle
//
// i = (jx + nSub*sx) + (jy + nSub*sy)*nFile + (jz + nSub*sz)*nFile^2,  j = jx + jy*nSub + jz*nSub^2,
// sc = sx + sy*S + sz*S^2 (minp.go:247-262).  A thread owns one output index g = sc*nSub^3 + j, or four
// consecutive ones when nSub and nFile are multiples of four (the four particles are then 48 contiguous,
// 16-byte aligned bytes of vec, and the three outputs are 16-byte vectors).  All HBM bound; no LDS.
// (This file is included by mnw_kernels.hip, inside namespace mnw, after the float-group helpers it uses.)

#define MNW_GLOBAL_SC __attribute__((address_space(1)))
typedef float sc_f32x4 __attribute__((ext_vector_type(4)));

struct SubcellGeom {
    uint32_t n_file, sub_cells, n_sub;
    uint32_t n_sub2, n_sub3, n_file2;
    int64_t n;                // n_file^3
};

// particle index of output index g (all 32-bit: n < 2^31 is checked on the host)
__device__ __forceinline__ uint32_t particle_of(const SubcellGeom& gm, uint32_t g) {
    const uint32_t sc = g / gm.n_sub3, j = g - sc * gm.n_sub3;
    const uint32_t jz = j / gm.n_sub2, r = j - jz * gm.n_sub2;
    const uint32_t jy = r / gm.n_sub, jx = r - jy * gm.n_sub;
    const uint32_t s2 = gm.sub_cells * gm.sub_cells;
    const uint32_t sz = sc / s2, q = sc - sz * s2;
    const uint32_t sy = q / gm.sub_cells, sx = q - sy * gm.sub_cells;
    return (jx + gm.n_sub * sx) + (jy + gm.n_sub * sy) * gm.n_file + (jz + gm.n_sub * sz) * gm.n_file2;
}

__global__ __launch_bounds__(256) void k_subcell_gather(const float* __restrict__ vec, float* __restrict__ out, SubcellGeom gm, int vec4) {
    const int64_t n = gm.n;
    const int64_t step = (int64_t)gridDim.x * 256;
    if (vec4) {
        for (int64_t g4 = (int64_t)blockIdx.x * 256 + threadIdx.x; 4 * g4 < n; g4 += step) {
            const uint32_t g = (uint32_t)(4 * g4);
            const uint32_t i = particle_of(gm, g);                 // i % 4 == 0, the four particles are consecutive
            const MNW_GLOBAL_SC sc_f32x4* p = (const MNW_GLOBAL_SC sc_f32x4*)(vec + 3 * (int64_t)i);
            const sc_f32x4 a = p[0], b = p[1], c = p[2];           // x0 y0 z0 x1 | y1 z1 x2 y2 | z2 x3 y3 z3
            const sc_f32x4 vx = {a.x, a.w, b.z, c.y}, vy = {a.y, b.x, b.w, c.z}, vz = {a.z, b.y, c.x, c.w};
            *(MNW_GLOBAL_SC sc_f32x4*)(out + g) = vx;
            *(MNW_GLOBAL_SC sc_f32x4*)(out + n + g) = vy;
            *(MNW_GLOBAL_SC sc_f32x4*)(out + 2 * n + g) = vz;
        }
    } else {
        for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < n; g += step) {
            const uint32_t i = particle_of(gm, (uint32_t)g);
            const MNW_GLOBAL_SC float* p = (const MNW_GLOBAL_SC float*)(vec + 3 * (int64_t)i);
            const float x = p[0], y = p[1], z = p[2];
            ((MNW_GLOBAL_SC float*)out)[g] = x;
            ((MNW_GLOBAL_SC float*)out)[n + g] = y;
            ((MNW_GLOBAL_SC float*)out)[2 * n + g] = z;
        }
    }
}

// minp.go:188-196: a decoded coordinate of a periodic box is wrapped once into [0, L)
__device__ __forceinline__ float wrap_once(float x, float L, int periodic) {
    if (periodic) {
        if (x < 0.0f) x = x + L;
        else if (x >= L) x = x - L;
    }
    return x;
}

__global__ __launch_bounds__(256) void k_subcell_scatter(const float* __restrict__ in, float* __restrict__ vec, SubcellGeom gm, int vec4,
                                                         int periodic, float L) {
    const int64_t n = gm.n;
    const int64_t step = (int64_t)gridDim.x * 256;
    if (vec4) {
        for (int64_t g4 = (int64_t)blockIdx.x * 256 + threadIdx.x; 4 * g4 < n; g4 += step) {
            const uint32_t g = (uint32_t)(4 * g4);
            const uint32_t i = particle_of(gm, g);
            sc_f32x4 vx = *(const MNW_GLOBAL_SC sc_f32x4*)(in + g);
            sc_f32x4 vy = *(const MNW_GLOBAL_SC sc_f32x4*)(in + n + g);
            sc_f32x4 vz = *(const MNW_GLOBAL_SC sc_f32x4*)(in + 2 * n + g);
#pragma unroll
            for (int c = 0; c < 4; ++c) { vx[c] = wrap_once(vx[c], L, periodic); vy[c] = wrap_once(vy[c], L, periodic); vz[c] = wrap_once(vz[c], L, periodic); }
            MNW_GLOBAL_SC sc_f32x4* p = (MNW_GLOBAL_SC sc_f32x4*)(vec + 3 * (int64_t)i);
            const sc_f32x4 a = {vx.x, vy.x, vz.x, vx.y}, b = {vy.y, vz.y, vx.z, vy.z}, c2 = {vz.z, vx.w, vy.w, vz.w};
            p[0] = a; p[1] = b; p[2] = c2;
        }
    } else {
        for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < n; g += step) {
            const uint32_t i = particle_of(gm, (uint32_t)g);
            MNW_GLOBAL_SC float* p = (MNW_GLOBAL_SC float*)(vec + 3 * (int64_t)i);
            p[0] = wrap_once(((const MNW_GLOBAL_SC float*)in)[g], L, periodic);
            p[1] = wrap_once(((const MNW_GLOBAL_SC float*)in)[n + g], L, periodic);
            p[2] = wrap_once(((const MNW_GLOBAL_SC float*)in)[2 * n + g], L, periodic);
        }
    }
}

// bounds() of minp.go:290-300 over vec[n][3]: six order-preserving keys combined with atomic max
// (keys[2k] = key(max_k), keys[2k+1] = ~key(min_k); zero-filled memory is the identity).  The scan's
// comparisons `<` and `>` never select a NaN, and neither do v_min/v_max; a NaN in vec[0] (which the
// reference would keep as both bounds) is not reproduced.
__device__ __forceinline__ uint32_t sc_fkey(float f) {
    const uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float sc_wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float sc_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

__global__ __launch_bounds__(256) void k_bounds3(const float* __restrict__ vec, int64_t n, uint32_t* __restrict__ keys) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    // a thread reads 4 particles = 48 contiguous bytes per step when possible
    const int64_t n4 = n / 4;
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < n4; q += step) {
        const MNW_GLOBAL_SC sc_f32x4* p = (const MNW_GLOBAL_SC sc_f32x4*)(vec + 12 * q);
        const sc_f32x4 a = p[0], b = p[1], c = p[2];
        const float xs[4] = {a.x, a.w, b.z, c.y}, ys[4] = {a.y, b.x, b.w, c.z}, zs[4] = {a.z, b.y, c.x, c.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mn[0] = fminf(mn[0], xs[e]); mx[0] = fmaxf(mx[0], xs[e]);
            mn[1] = fminf(mn[1], ys[e]); mx[1] = fmaxf(mx[1], ys[e]);
            mn[2] = fminf(mn[2], zs[e]); mx[2] = fmaxf(mx[2], zs[e]);
        }
    }
    if (blockIdx.x == 0 && (int64_t)threadIdx.x < n - 4 * n4) {
        const float* p = vec + 3 * (4 * n4 + threadIdx.x);
#pragma unroll
        for (int k = 0; k < 3; ++k) { mn[k] = fminf(mn[k], p[k]); mx[k] = fmaxf(mx[k], p[k]); }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float a = sc_wave_max(mx[k]), b = sc_wave_min(mn[k]);
        if ((threadIdx.x & 63) == 0) {
            if (a != -INFINITY) atomicMax(&keys[2 * k], sc_fkey(a));
            if (b != INFINITY) atomicMax(&keys[2 * k + 1], ~sc_fkey(b));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// minp.Writer.Vectors without the staging pass: the float-group reduce and pack read the AoS grid directly.
//
// A vtile = 2,048 consecutive elements j of one sub-cell, all three components: 24 KB of the grid, read as
// 512 groups of four particles (48 contiguous bytes, three 16-byte loads; two groups per lane).  Jobs are
// the 3*S^3 blocks in file order (job = k*S^3 + sc).
//   k_reduce_vec3  folds every vtile into the statistics of its three blocks (stat lines, as k_reduce_f32)
//   k_pack_vec3    derives the three descriptors of a sub-cell from the stat lines itself (the kernel
//                  boundary is the barrier; the workgroup that packs a block's first vtile writes its
//                  metadata), quantizes at load time, parks the three components as three LDS planes in
//                  the pack core's staging layout; wave w packs component w and writes its words out
//                  itself (wave-local: no barrier between pack and store).
// HBM traffic: 12 B/particle twice + the streams, against 12 B read + 12 B written + 12 B read + streams
// of the staging path (k_subcell_gather + the block encoder).
// Only for n_sub, n_file multiples of four and ring sizes the 32-bit pack path takes; anything else, and any
// block the descriptor sends to the exact path, goes through the staging path.
// ------------------------------------------------------------------------------------------------
constexpr int kVTile = 2048;
constexpr int kVPlane = kVTile / kGroup * kStride32;          // dwords of one component plane (2,304)

struct Vec3Args {
    const float* vec;
    SubcellGeom gm;
    uint32_t s3;              // sub_cells^3
    uint32_t tiles_per_sc;    // vtiles per sub-cell
    // Division by a launch constant d as a multiply-high: m = floor(2^32 / d) + 1 gives x / d == umulhi(x, m) whenever x d < 2^32.
    // m_row: d = n_sub / 4 (x < n_sub^3 / 4), m_nsub: d = n_sub (x < n_sub^2); both 0 if n_sub >= 512 (then: ordinary division).
    uint32_t m_row, m_nsub;
};

// first particle of the group of four that starts at element j (a multiple of 4) of sub-cell sc -- particle_of for the vtile kernels:
// the sub-cell's origin is wave-uniform (scalar arithmetic), the lane's part costs two multiply-highs instead of two divisions
__device__ __forceinline__ uint32_t vec3_particle(const Vec3Args& a, uint32_t sc, uint32_t j) {
    const SubcellGeom& gm = a.gm;
    if (a.m_row == 0) return particle_of(gm, sc * gm.n_sub3 + j);
    const uint32_t s2 = gm.sub_cells * gm.sub_cells;
    const uint32_t sz = sc / s2, q = sc - sz * s2, sy = q / gm.sub_cells, sx = q - sy * gm.sub_cells;
    const uint32_t base0 = gm.n_sub * (sx + sy * gm.n_file + sz * gm.n_file2);
    const uint32_t j4 = j >> 2, rowlen = gm.n_sub >> 2;
    const uint32_t row = __umulhi(j4, a.m_row), jx4 = j4 - row * rowlen;          // row = jy + n_sub jz
    const uint32_t jz = __umulhi(row, a.m_nsub), jy = row - jz * gm.n_sub;
    return base0 + 4 * jx4 + jy * gm.n_file + jz * gm.n_file2;
}

// One group of four particles of a vtile as loaded: 48 contiguous bytes x0 y0 z0 x1 | y1 z1 x2 y2 | z2 x3 y3 z3.  The
// vectors stay as the loads delivered them -- component k of particle c is word 3 c + k, a compile-time position -- and are
// left UNASSIGNED past the end of the sub-cell (`valid` false; every user checks it): building de-interleaved vectors, or merging
// with zeros, makes the compiler wait for the loads where they are issued, and nothing is in flight across the work that follows.
struct Vec3Group {
    sc_f32x4 u, v, w;
    bool valid;
    __device__ __forceinline__ float get(int k, int c) const { const int i = 3 * c + k; return i < 4 ? u[i] : i < 8 ? v[i - 4] : w[i - 8]; }
    __device__ __forceinline__ void set(int k, int c, float f) { const int i = 3 * c + k; if (i < 4) u[i] = f; else if (i < 8) v[i - 4] = f; else w[i - 8] = f; }
    __device__ __forceinline__ sc_f32x4 comp(int k) const { return sc_f32x4{get(k, 0), get(k, 1), get(k, 2), get(k, 3)}; }
};
__device__ __forceinline__ Vec3Group vec3_load(const Vec3Args& a, uint32_t sc, uint32_t j) {
    Vec3Group g;
    g.valid = j < a.gm.n_sub3;
    if (g.valid) {
        const uint32_t i = vec3_particle(a, sc, j);
        const MNW_GLOBAL_SC sc_f32x4* p = (const MNW_GLOBAL_SC sc_f32x4*)(a.vec + 3 * (int64_t)i);
        g.u = p[0]; g.v = p[1]; g.w = p[2];
    }
    return g;
}

__global__ __launch_bounds__(256, 4) void k_reduce_vec3(Vec3Args a, const DevJob* __restrict__ jobs, uint64_t* __restrict__ stats) {
    __shared__ float red[12][4];
    __shared__ int redbad[3][4];
    const int tid = threadIdx.x;
    const int64_t total = (int64_t)a.s3 * a.tiles_per_sc;
    int64_t v0 = total * blockIdx.x / gridDim.x;
    const int64_t v1 = total * (blockIdx.x + 1) / gridDim.x;
    while (v0 < v1) {
        const uint32_t sc = (uint32_t)(v0 / a.tiles_per_sc);
        int64_t vend = (int64_t)(sc + 1) * a.tiles_per_sc;
        if (vend > v1) vend = v1;
        float xh[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) xh[k] = jobs[k * a.s3 + sc].xh;
        // four statistics per component; three-input min / max over pairs of particles, one v_cmp_class for NaN / Inf
        float mn[3], mx[3], mxlo[3], mnhi[3];
        bool nonfinite = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) { mn[k] = INFINITY; mx[k] = -INFINITY; mxlo[k] = -INFINITY; mnhi[k] = INFINITY; }
        auto fold = [&](const Vec3Group& g) {
            if (!g.valid) return;
#pragma unroll
            for (int c = 0; c < 4; c += 2) {
                const float v0[3] = {g.get(0, c), g.get(1, c), g.get(2, c)}, v1[3] = {g.get(0, c + 1), g.get(1, c + 1), g.get(2, c + 1)};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    nonfinite |= __builtin_amdgcn_classf(v0[k], 0x207) | __builtin_amdgcn_classf(v1[k], 0x207);
                    mn[k] = min3_raw(mn[k], v0[k], v1[k]);
                    mx[k] = max3_raw(mx[k], v0[k], v1[k]);
                    const bool l0 = v0[k] < xh[k], l1 = v1[k] < xh[k];
                    mxlo[k] = max3_raw(mxlo[k], l0 ? v0[k] : -INFINITY, l1 ? v1[k] : -INFINITY);
                    mnhi[k] = min3_raw(mnhi[k], l0 ? INFINITY : v0[k], l1 ? INFINITY : v1[k]);
                }
            }
        };
        int64_t v = v0;
        for (; v + 1 < vend; v += 2) {                       // two vtiles (twelve 16-byte loads per lane) in flight
            const uint32_t t = (uint32_t)(v - (int64_t)sc * a.tiles_per_sc);
            Vec3Group g[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) g[r] = vec3_load(a, sc, (t + (r >> 1)) * kVTile + 4 * (tid + 256 * (r & 1)));
#pragma unroll
            for (int r = 0; r < 4; ++r) fold(g[r]);
        }
        for (; v < vend; ++v) {
            const uint32_t t = (uint32_t)(v - (int64_t)sc * a.tiles_per_sc);
            Vec3Group g[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) g[r] = vec3_load(a, sc, t * kVTile + 4 * (tid + 256 * r));
#pragma unroll
            for (int r = 0; r < 2; ++r) fold(g[r]);
        }
        // (a NaN or an infinity anywhere in the run flags all three blocks: they then take the exact path, which
        // is always right, only slower)
        FStat st[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) st[k] = FStat{mn[k], mx[k], mxlo[k], mnhi[k], nonfinite};
        // fold the run into the three blocks' stat lines
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float mn = wave_min_dpp(st[k].mn), mx = wave_max_dpp(st[k].mx), mxlo = wave_max_dpp(st[k].mxlo), mnhi = wave_min_dpp(st[k].mnhi);
            const int bad = __any(st[k].bad);
            if (lane_id() == 0) {
                red[4 * k][wave_id()] = mn; red[4 * k + 1][wave_id()] = mx; red[4 * k + 2][wave_id()] = mxlo; red[4 * k + 3][wave_id()] = mnhi;
                redbad[k][wave_id()] = bad;
            }
        }
        __syncthreads();
        if (tid < 15) {
            const int k = tid / 5, w5 = tid % 5;
            float aa = red[4 * k][0], bb = red[4 * k + 1][0], cc = red[4 * k + 2][0], dd = red[4 * k + 3][0];
            int bd = redbad[k][0];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                aa = fminf(aa, red[4 * k][w]); bb = fmaxf(bb, red[4 * k + 1][w]);
                cc = fmaxf(cc, red[4 * k + 2][w]); dd = fminf(dd, red[4 * k + 3][w]);
                bd |= redbad[k][w];
            }
            uint64_t val;
            if (w5 == 0) val = bd ? 0 : fkey(bb);
            else if (w5 == 1) val = bd ? 0 : (uint32_t)~fkey(aa);
            else if (w5 == 2) val = (bd || cc == -INFINITY) ? 0 : fkey(cc);
            else if (w5 == 3) val = (bd || dd == INFINITY) ? 0 : (uint32_t)~fkey(dd);
            else val = bd ? 1 : 0;
            const int64_t j = (int64_t)k * a.s3 + sc;
            uint64_t* rec = stats + (j * kStatSlots + (blockIdx.x % kStatSlots)) * kStatWords;
            if (val != 0) atomicMax((unsigned long long*)(rec + w5), (unsigned long long)val);
        }
        v0 = vend;
    }
}

__global__ __launch_bounds__(256, 4) void k_pack_vec3(Vec3Args a, const DevJob* __restrict__ jobs, const uint64_t* __restrict__ stats,
                                                      Meta* __restrict__ meta) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[3 * kVPlane];
    __shared__ uint64_t gran_s[3];
    const int tid = threadIdx.x;
    const int64_t total = (int64_t)a.s3 * a.tiles_per_sc;
    int64_t v0 = total * blockIdx.x / gridDim.x;
    const int64_t v1 = total * (blockIdx.x + 1) / gridDim.x;
    while (v0 < v1) {
        const uint32_t sc = (uint32_t)(v0 / a.tiles_per_sc);
        int64_t vend = (int64_t)(sc + 1) * a.tiles_per_sc;
        if (vend > v1) vend = v1;
        // ---- the three descriptors of this sub-cell (wave k derives block k's) ------------------------
        __syncthreads();
        if (wave_id() < 3) {
            const int64_t j = (int64_t)wave_id() * a.s3 + sc;
            const DevJob jb = jobs[j];
            uint64_t st[5];
            load_stats<false>(stats, (int)j, st);
            const Desc d = make_desc_f32(jb, st);
            if (lane_id() == 0) {
                gran_s[wave_id()] = gran_encode(d);
                if (v0 == (int64_t)sc * a.tiles_per_sc) {          // this workgroup packs the sub-cell's first vtile
                    Meta m = {d.imin, (int64_t)d.bits, d.nbytes, (int64_t)d.status};
                    meta[j] = m;
                }
            }
        }
        __syncthreads();
        float low[3];
        double rdx[3];
        float dxq[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { const DevJob& jb = jobs[k * a.s3 + sc]; low[k] = jb.low; rdx[k] = jb.rdx; dxq[k] = jb.dxq; }
        // the packing wave's own block
        const int kw = wave_id() < 3 ? wave_id() : 0;
        const DevJob jw = jobs[(int64_t)kw * a.s3 + sc];
        const Desc dw = gran_decode(gran_s[kw], jw.n);
        const int okw = __builtin_amdgcn_readfirstlane(wave_id() < 3 && dw.status == 0 && dw.bits != 0);
        const uint32_t bw = (uint32_t)__builtin_amdgcn_readfirstlane(dw.bits);
        const uint32_t iminw = (uint32_t)(int32_t)dw.imin;
        const uint32_t wrapw = (jw.periodic == 1 && dw.pmin == dw.imin) ? (uint32_t)jw.pixels : 0u;

        // the next vtile's loads are issued as soon as the current one is parked: they fly during the pack
        Vec3Group gl[2];
        {
            const uint32_t t = (uint32_t)(v0 - (int64_t)sc * a.tiles_per_sc);
#pragma unroll
            for (int r = 0; r < 2; ++r) gl[r] = vec3_load(a, sc, t * kVTile + 4 * (tid + 256 * r));
        }
        for (int64_t v = v0; v < vend; ++v) {
            const uint32_t t = (uint32_t)(v - (int64_t)sc * a.tiles_per_sc);
            // ---- quantize, park ---------------------------------------------------------------------
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t e = 4 * (tid + 256 * r);                // element of the vtile
                const Vec3Group g = gl[r];
                u32x4 q[3];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float xv[3] = {g.get(0, c), g.get(1, c), g.get(2, c)};
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        float rq = quant_mul(xv[k], low[k], rdx[k]);
                        if (__builtin_expect(__builtin_amdgcn_classf(rq, 0x90), 0)) rq = __fdiv_rn(__fsub_rn(xv[k], low[k]), dxq[k]);
                        q[k][c] = g.valid ? (uint32_t)cvt_floor_i32(rq) : 0u;
                    }
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) *reinterpret_cast<u32x4*>(lds + k * kVPlane + stage32_idx((int)e)) = q[k];
            }
            if (v + 1 < vend) {
#pragma unroll
                for (int r = 0; r < 2; ++r) gl[r] = vec3_load(a, sc, (t + 1) * kVTile + 4 * (tid + 256 * r));
            }
            lds_barrier();
            // ---- wave w packs component w and writes its words out (wave-local) ------------------------
            if (okw) {
                uint32_t* plane = lds + kw * kVPlane;
                const int l = lane_id();
                uint32_t qq[kGroup];
#pragma unroll
                for (int k8 = 0; k8 < kGroup / 4; ++k8) {
                    const u32x4 rr = *reinterpret_cast<const u32x4*>(plane + l * kStride32 + 4 * k8);
                    qq[4 * k8] = rr.x; qq[4 * k8 + 1] = rr.y; qq[4 * k8 + 2] = rr.z; qq[4 * k8 + 3] = rr.w;
                }
                const int64_t e0 = (int64_t)t * kVTile + kGroup * l;
                if (wrapw == 0) {
#pragma unroll
                    for (int i = 0; i < kGroup; ++i) qq[i] -= iminw;
                } else {
#pragma unroll
                    for (int i = 0; i < kGroup; ++i) { const uint32_t r2 = qq[i] - iminw; qq[i] = min(r2, r2 + wrapw); }
                }
                if (e0 + kGroup > jw.n) {
#pragma unroll
                    for (int i = 0; i < kGroup; ++i) if (e0 + i >= jw.n) qq[i] = 0;
                }
                // all 64 lanes have read their groups before any lane overwrites the plane with words
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                pack_group32_switch(qq, bw, plane + l * bw);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int64_t byte0 = (int64_t)t * (kVTile / 8) * bw;
                int64_t nbytes_t = (int64_t)(kVTile / 8) * bw;
                if (byte0 + nbytes_t > dw.nbytes) nbytes_t = dw.nbytes - byte0;
                uint8_t* dst = (uint8_t*)jw.dst;
                const int full = (int)(nbytes_t >> 4);
                for (int c = l; c < full; c += 64)
                    stg<u32x4>(dst + byte0 + 16 * (int64_t)c, *reinterpret_cast<const u32x4*>(plane + 4 * c));
                const int rem = (int)(nbytes_t & 15);
                if (l < rem) {
                    const uint32_t w = plane[4 * full + (l >> 2)];
                    stg<uint8_t>(dst + byte0 + 16 * (int64_t)full + l, (uint8_t)(w >> (8 * (l & 3))));
                }
            }
            lds_barrier();
        }
        v0 = vend;
    }
}

// ------------------------------------------------------------------------------------------------
// minp.Writer.Vectors with the grid read ONCE (k_encode_reg_vec3): the resident pipeline and tagged-record exchange of
// k_encode_reg_f32 (its header has the protocol) over vtiles, one stage deeper.  An item = one vtile = 2,048 particles of
// one sub-cell, all three components (24 KB of the grid, 24 values per lane).
// The three blocks of a sub-cell have 489 items each at C4's geometry and every workgroup needs all three descriptors every
// trip, so the statistics are exchanged in TWO levels:
//   item records   one per (component, vtile), published in the trip in which the vtile arrives;
//   group records  the fold of kVGroup = 32 consecutive item records of a block, published by EVERY member of the group
//                  (no leader: whichever member gets there first makes it available) in R copies -- copy d is read only by
//                  the workgroups whose vtile is in group d (R = groups per block <= 32);
// a trip reads 3 x (32 + R) records = 4.6 KB for C4, against 3 x 489 = 47 KB for a flat sweep of the item records.
// A record that is read must have been published a whole trip before: under this kernel's load a memory round trip takes most
// of a trip (the queues hold ~32 MB), so a read that has to be repeated costs a trip (measured: the group fold placed in
// the same trip as the publication of its item records, 886 us for C4 against 497 us for the pipeline without any exchange).
// Hence THREE items per workgroup: the NEW one (registers; its loads were issued a trip ago), the MIDDLE one (registers,
// statistics published last trip) and the OLD one (LDS, three planes of pixel indices).  Per trip:
//   all waves    statistics of NEW -> publish its three item records
//   waves 0..2   (= components x, y, z; wave 3 has no part in this) one pair of record loads at the start of the trip, both
//                kinds a trip old by then: lanes 0..31 the R group records of OLD's block, lanes 32..63 the item records of
//                MIDDLE's group.  Upper half -> fold -> publish MIDDLE's group record (the group duty); lower half ->
//                descriptor of OLD's block -> re-base, pack the component's plane in place, store
//   all waves    MIDDLE -> quantize -> LDS; NEW becomes MIDDLE; issue the loads of the next vtile
// No deadlock (unweighted rounds, blocks of at most G vtiles): the duty needs item records only and waits for nothing else; a
// workgroup reaches the trip of a vtile once the blocks of its earlier vtiles are complete, which are strictly earlier blocks --
// induction over the blocks in order.  Waits are bounded (poll_abort): on expiry the launch gives up and the host reruns the
// call through k_reduce_vec3 + k_pack_vec3.
// HBM traffic = algorithmic: 12 B/particle read + the streams (+ the records, 13 % of that at C4).
// ------------------------------------------------------------------------------------------------
constexpr int kVGroup = 32;                 // vtiles per record group

// all the records of a launch behind one buffer descriptor: lane-specific indices
__device__ __forceinline__ void rec_prefetch_any(const ItemRec* base, uint32_t total_recs, int64_t idx, u32x4& a, u32x4& b) {
    const uint64_t p = (uint64_t)base;
    const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(p >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)p);
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)pu, 0, (int)(total_recs * (uint32_t)sizeof(ItemRec)), 0x00020000);
    const int off = (int)((uint32_t)idx * (uint32_t)sizeof(ItemRec));
    a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);
    b = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, 16);
}

struct Vec3RegArgs {
    Vec3Args a;
    ItemRec* recs;            // [3 s3][tiles_per_sc]
    ItemRec* grecs;           // [3 s3][R reader groups][R]
    uint32_t gen;
    uint32_t R;               // groups per block
    uint32_t* abort_flag;
    uint32_t* abort_host;
    int64_t drop_item;        // test hook: the records of that vtile are never published
    uint32_t m_row, m_nsub;   // floor(2^32 / d) + 1 for d = n_sub / 4 and d = n_sub: x / d == umulhi(x, m) whenever x d < 2^32
    uint32_t m_s, m_s2;       // the same for d = sub_cells and sub_cells^2
};

// vec3_load for this kernel: the sub-cell's origin is wave-uniform (scalar arithmetic), the lane's part of the index needs two
// divisions by launch constants, done as multiply-high (exact here: n_sub <= 128, see the launch wrapper).  j4 = element / 4.
__device__ __forceinline__ Vec3Group vec3_load_m(const Vec3RegArgs& ra, uint32_t base0, uint32_t j4) {
    const SubcellGeom& gm = ra.a.gm;
    Vec3Group g;
    g.valid = 4 * j4 < gm.n_sub3;
    if (g.valid) {
        const uint32_t rowlen = gm.n_sub >> 2;
        const uint32_t row = __umulhi(j4, ra.m_row), jx4 = j4 - row * rowlen;          // row = jy + n_sub jz
        const uint32_t jz = __umulhi(row, ra.m_nsub), jy = row - jz * gm.n_sub;
        const uint32_t i = base0 + 4 * jx4 + jy * gm.n_file + jz * gm.n_file2;
        const MNW_GLOBAL_SC sc_f32x4* p = (const MNW_GLOBAL_SC sc_f32x4*)(ra.a.vec + 3 * (int64_t)i);
        g.u = p[0]; g.v = p[1]; g.w = p[2];
    }
    return g;
}

// Twelve per-lane values -> their twelve wave-wide maxima, in lanes 0..3 (three each: lane q holds the values 6 (q & 1) + 3 (q >> 1) + {0, 1, 2}).
// Two halving steps inside each quad (a lane keeps half of its values and takes the partner's half of those), then the three
// survivors are folded over the quads of a row (row rotations keep the lane's position in its quad), then over the rows:
// 42 vector instructions against 12 x 16 for one full-wave reduction per value.
__device__ __forceinline__ void wave_max12(const float (&m)[12], float (&out)[3]) {
    auto dpp = [](float x, auto ctrl) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true)); };
    const int l = lane_id();
    const bool b0 = l & 1, b1 = l & 2;
    float r6[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float keep = b0 ? m[i + 6] : m[i], send = b0 ? m[i] : m[i + 6];
        r6[i] = fmaxf(keep, dpp(send, std::integral_constant<int, 0xB1>{}));          // quad_perm [1,0,3,2]
    }
    float r3[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float keep = b1 ? r6[i + 3] : r6[i], send = b1 ? r6[i] : r6[i + 3];
        r3[i] = fmaxf(keep, dpp(send, std::integral_constant<int, 0x4E>{}));          // quad_perm [2,3,0,1]
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float v = r3[i];
        v = fmaxf(v, dpp(v, std::integral_constant<int, 0x124>{}));                   // row_ror:4
        v = fmaxf(v, dpp(v, std::integral_constant<int, 0x128>{}));                   // row_ror:8
        v = fmaxf(v, __shfl_xor(v, 16));
        v = fmaxf(v, __shfl_xor(v, 32));
        out[i] = v;
    }
}

// what a block's descriptor needs of its job, the same for every sub-cell of a component (LDS copy, made once)
struct Vec3JobPar {
    float low, dxq;
    int periodic, pad;
    int64_t pixels, n, cap;
};

// Lanes 0..11 of one wave: the three descriptors of a sub-cell at once (lane = 4 component + word; desc_from_sweep_f32 across
// three quads).  v = the lane's swept statistic, bad = a member of the lane's block was not finite.  The granule of component c
// comes back in lanes 4c..4c+3; lane 4c writes the block's metadata if asked to.
__device__ __forceinline__ uint64_t desc3_from_sweep_f32(const Vec3JobPar& jp, int64_t j, uint32_t v, bool bad, Meta* __restrict__ meta, bool write_meta) {
    Desc d = {0, 0, 0, 0, 0};
    const int l = lane_id(), r = l & 3, q0 = l & ~3;
    const uint32_t key = (r & 1) ? ~v : v;
    int64_t ql = 0;
    const bool okl = quant_i64(funkey(key), jp.low, jp.dxq, &ql);
    const bool has_lo = __shfl(v, q0 + 2) != 0, has_hi = __shfl(v, q0 + 3) != 0;
    const uint32_t okmask = (uint32_t)(__ballot(okl) >> q0) & 15u;
    const bool ok = (okmask & 3) == 3 && (!has_lo || (okmask & 4)) && (!has_hi || (okmask & 8));
    const int64_t mx0 = __shfl(ql, q0), mn0 = __shfl(ql, q0 + 1), mxlo = __shfl(ql, q0 + 2), mnhi = __shfl(ql, q0 + 3);
    uint64_t maxdelta = 0;
    if (bad || !ok) {
        d.status = (int)kNeedsExact;
    } else if (jp.periodic == 1) {
        if (mn0 < 0 || mx0 >= jp.pixels || jp.pixels > 0x7fffffffll) d.status = (int)kNeedsExact;
        else {
            const int64_t h = jp.pixels / 2;
            if (mx0 - mn0 + 1 <= h) { d.pmin = mn0; d.imin = mn0; maxdelta = (uint64_t)(mx0 - mn0); }
            else if (has_lo && has_hi && (mxlo + jp.pixels) - mnhi + 1 <= h) {
                d.pmin = mnhi; d.imin = mnhi; maxdelta = (uint64_t)(mxlo + jp.pixels - mnhi);
            } else { d.pmin = 0; d.imin = mn0; maxdelta = (uint64_t)(mx0 - mn0); }
        }
    } else {
        if (mn0 < -0x80000000ll || mx0 > 0x7fffffffll) d.status = (int)kNeedsExact;
        else { d.imin = mn0; maxdelta = (uint64_t)(mx0 - mn0); }
    }
    if (d.status == 0) {
        d.bits = precision_needed(maxdelta);
        d.nbytes = d.bits == 0 ? 0 : array_bytes(d.bits, jp.n);
        if (d.nbytes > jp.cap) d.status = -5;
    }
    if (write_meta && r == 0) {
        Meta m = {d.imin, (int64_t)d.bits, d.nbytes, (int64_t)d.status};
        meta[j] = m;
    }
    return gran_encode(d);
}

#ifndef MNW_V3_DIAG
#define MNW_V3_DIAG 0          // diagnostic builds only (wrong output): 1 = never wait for a group record, 2 = no group duty, 4 = duty without waiting
#endif

__global__ __launch_bounds__(256, 3) void k_encode_reg_vec3(Vec3RegArgs ra, const DevJob* __restrict__ jobs, Meta* __restrict__ meta) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[3 * kVPlane];
    __shared__ float red[12][4];                         // NEW's statistics per wave, all as maxima (minima negated)
    __shared__ int redbad[4];
    __shared__ uint32_t swp[3][5];                       // OLD's block per component: the four swept statistics, flags
    __shared__ uint64_t gran_s[3];
    __shared__ Vec3JobPar jpar[3];
    __shared__ int abort_s;
    const Vec3Args& a = ra.a;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(wave_id()), l = lane_id();    // (uniform: the roles below are scalar branches)
    const uint32_t T = a.tiles_per_sc, R = ra.R, s3 = a.s3, gen = ra.gen;
    const uint32_t total = s3 * T;                       // (below 2^26: the host's record budget)
    const uint32_t n_item_recs = 3u * s3 * T, n_group_recs = 3u * s3 * R * R;
    // a round moves every workgroup on by G items = dsc sub-cells and dt vtiles
    const uint32_t G = gridDim.x, dsc = G / T, dt = G - dsc * T;
    if (tid < 3) {
        const DevJob& jb = jobs[tid * s3];
        Vec3JobPar p = {jb.low, jb.dxq, jb.periodic, 0, jb.pixels, jb.n, jb.cap};
        jpar[tid] = p;
    }
    if (tid == 0) abort_s = 0;
    // the quantization constants of the three components (the same for every sub-cell: mnw_vectors_encode fills them so)
    float qlow[3], qdx[3], qxh[3];
    double qrdx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { const DevJob& jb = jobs[k * s3]; qlow[k] = jb.low; qdx[k] = jb.dxq; qrdx[k] = jb.rdx; qxh[k] = jb.xh; }

    Vec3Group gn[2], gm[2];                             // the NEW item (in flight / just arrived) and the MIDDLE one
    auto issue = [&](uint32_t sc, uint32_t t) {
        // (scalar: the item is wave-uniform; sc < 2^15, so the divisions by sub_cells are multiply-highs as well)
        const uint32_t S = a.gm.sub_cells;
        const uint32_t sz = __umulhi(sc, ra.m_s2), q = sc - sz * S * S, sy = __umulhi(q, ra.m_s), sx = q - sy * S;
        const uint32_t base0 = a.gm.n_sub * (sx + sy * a.gm.n_file + sz * a.gm.n_file2);
#pragma unroll
        for (int r = 0; r < 2; ++r) gn[r] = vec3_load_m(ra, base0, t * (kVTile / 4) + tid + 256 * r);
    };
    // MIDDLE -> pixel indices, in place (its floats are not needed once its statistics are out): arithmetic only
    auto quantize_mid = [&]() {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            Vec3Group& g = gm[r];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float xv[3] = {g.get(0, c), g.get(1, c), g.get(2, c)};
                uint32_t qv[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float rq = quant_mul(xv[k], qlow[k], qrdx[k]);
                    if (__builtin_expect(__builtin_amdgcn_classf(rq, 0x90), 0)) rq = __fdiv_rn(__fsub_rn(xv[k], qlow[k]), qdx[k]);
                    qv[k] = g.valid ? (uint32_t)cvt_floor_i32(rq) : 0u;
                }
                g.set(0, c, __uint_as_float(qv[0])); g.set(1, c, __uint_as_float(qv[1])); g.set(2, c, __uint_as_float(qv[2]));
            }
        }
    };
    // The record loads of a trip (waves 0..2, component wv), issued at the end of the trip before, ahead of the loads of the next
    // item, so that they are back when those are.  Lanes 0..31: the group records of (osc_, ot_)'s block, the copy row of that vtile's
    // own group -- published during the trip that is ending; lanes 32..63: the item records of (msc_, mt_)'s group -- published at
    // its start.  Both arrays sit behind one buffer descriptor (the group records follow the item records).
    u32x4 ga, gb;
    bool mine = false;
    auto issue_records = [&](bool old_, uint32_t osc_, uint32_t ot_, bool mid_, uint32_t msc_, uint32_t mt_) {
        mine = false;
        if (wv < 3) {
            // (32-bit, scalar but for the lane number, which is taken through opaque_tid: nothing lane-specific is kept across the trip)
            const int lo = opaque_tid() & 63;
            const uint32_t mfirst_ = mt_ / kVGroup * kVGroup;
            const int mcnt_ = (int)(T - mfirst_ < (uint32_t)kVGroup ? T - mfirst_ : (uint32_t)kVGroup);
            const uint32_t gbase = n_item_recs + ((uint32_t)(wv * s3 + osc_) * R + ot_ / kVGroup) * R;
            const uint32_t ibase = (uint32_t)(wv * s3 + msc_) * T + mfirst_ - 32u;
            const bool lower = lo < 32;
            mine = lower ? (old_ && lo < (int)R) : (mid_ && lo - 32 < mcnt_ && !(MNW_V3_DIAG & 2));
            if (mine) rec_prefetch_any(ra.recs, n_item_recs + n_group_recs, (int64_t)((lower ? gbase : ibase) + (uint32_t)lo), ga, gb);
        }
    };
    // NEW, MIDDLE, OLD as (sub-cell, vtile)
    uint32_t sc = blockIdx.x / T, t = blockIdx.x - sc * T, msc = 0, mt = 0, osc = 0, ot = 0;
    bool have_new = blockIdx.x < total, have_mid = false, have_old = false;
    if (have_new) issue(sc, t);
    MNW_STAMP_DECL
    MNW_STAMP_START
    lds_barrier();

    while (have_new || have_mid || have_old) {
        uint32_t nsc = sc + dsc, nt = t + dt;
        if (nt >= T) { nt -= T; ++nsc; }
        const bool have_next = have_new && nsc < s3;

        // ---- 1. (the record loads of this trip were issued at the end of the previous one, ahead of NEW's loads: see 5)
        // NEW is complete from here on, on every path: the compiler waits once, here, not wherever NEW is touched next
#pragma unroll
        for (int r = 0; r < 2; ++r) asm volatile("" :: "v"(gn[r].u), "v"(gn[r].v), "v"(gn[r].w));
        const uint32_t mgrp = mt / kVGroup, mfirst = mgrp * kVGroup;
        const int mcnt = (int)(T - mfirst < (uint32_t)kVGroup ? T - mfirst : (uint32_t)kVGroup);

        // ---- 2. statistics of NEW, per component (as k_reduce_vec3), all twelve as maxima
        if (have_new) {
            float m12[12];
            bool nonfinite = false;
#pragma unroll
            for (int k = 0; k < 12; ++k) m12[k] = -INFINITY;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const Vec3Group& g = gn[r];
                if (g.valid) {
#pragma unroll
                    for (int c = 0; c < 4; c += 2) {
                        const float v0[3] = {g.get(0, c), g.get(1, c), g.get(2, c)}, v1[3] = {g.get(0, c + 1), g.get(1, c + 1), g.get(2, c + 1)};
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            nonfinite |= __builtin_amdgcn_classf(v0[k], 0x207) | __builtin_amdgcn_classf(v1[k], 0x207);
                            m12[4 * k] = max3_raw(m12[4 * k], -v0[k], -v1[k]);                      // -min
                            m12[4 * k + 1] = max3_raw(m12[4 * k + 1], v0[k], v1[k]);                 // max
                            const bool l0 = v0[k] < qxh[k], l1 = v1[k] < qxh[k];
                            m12[4 * k + 2] = max3_raw(m12[4 * k + 2], l0 ? v0[k] : -INFINITY, l1 ? v1[k] : -INFINITY);     // max below the split
                            m12[4 * k + 3] = max3_raw(m12[4 * k + 3], l0 ? -INFINITY : -v0[k], l1 ? -INFINITY : -v1[k]);   // -min at / above it
                        }
                    }
                }
            }
            float o3[3];
            wave_max12(m12, o3);
            if (l < 4) {
                const int base = 6 * (l & 1) + 3 * (l >> 1);
#pragma unroll
                for (int i = 0; i < 3; ++i) red[base + i][wv] = o3[i];
            }
            const int bad = __any(nonfinite);
            if (l == 0) redbad[wv] = bad;
        }
        MNW_STAMP(0)      // wait for NEW's loads (the records arrive before them), lane statistics, wave reductions
        // waves 0..2: fold both halves (wave k: block k's group records, and component k's group duty)
        uint32_t gv[4] = {0, 0, 0, 0}, dv[4] = {0, 0, 0, 0};
        uint32_t gfl = 0, dfl = 0;                         // bit 0: a record is missing, bit 1: a member was not finite (lower / upper half)
        if (wv < 3) {
            uint32_t missing = 0, bad = 0;
            uint32_t vl[4] = {0, 0, 0, 0};
            if (mine) {
                const uint32_t tg[4] = {ga.y, ga.w, gb.y, gb.w};
                vl[0] = ga.x; vl[1] = ga.z; vl[2] = gb.x; vl[3] = gb.z;
#pragma unroll
                for (int k = 0; k < 4; ++k) { missing |= (tg[k] & ~kRecBad) != gen; bad |= tg[k] & kRecBad; }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) half_wave_max_dpp(vl[k], gv[k], dv[k]);
            const uint64_t mm = __ballot(missing != 0), bm = __ballot(bad != 0);
            gfl = ((uint32_t)mm ? 1u : 0u) | ((uint32_t)bm ? 2u : 0u);
            dfl = ((uint32_t)(mm >> 32) ? 1u : 0u) | ((uint32_t)(bm >> 32) ? 2u : 0u);
            if (have_old) {
                if (__builtin_expect((gfl & 1u) && !(MNW_V3_DIAG & 1), 0)) {
                    // a group record was not there yet (its group more than a trip behind): poll the row
                    uint32_t spins = 0;
                    const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
                    const ItemRec* row = ra.grecs + ((int64_t)(wv * s3 + osc) * R + ot / kVGroup) * R;
                    for (;;) {
                        uint64_t w[4] = {0, 0, 0, 0};
                        uint32_t ms = 0, bd = 0;
                        if (l < (int)R) {
                            load_rec_sc1(row + l, w);
#pragma unroll
                            for (int k = 0; k < 4; ++k) { const uint32_t tag = (uint32_t)(w[k] >> 32); ms |= (tag & ~kRecBad) != gen; bd |= tag & kRecBad; }
                        }
                        if (!__any(ms)) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) gv[k] = wave_max_dpp(l < (int)R ? (uint32_t)w[k] : 0u);
                            gfl = __any(bd) ? 2u : 0u;
                            break;
                        }
                        if (poll_abort(spins, t_start, ra.abort_flag, ra.abort_host)) { if (l == 0) abort_s = 1; break; }
                    }
                }
                if (l < 5) swp[wv][l] = l == 0 ? gv[0] : l == 1 ? gv[1] : l == 2 ? gv[2] : l == 3 ? gv[3] : gfl;
            }
        }
        lds_barrier();
        // ---- 3. publish NEW's three records (lanes 0..11: component tid / 4, word tid % 4)
        if (have_new && tid < 12 && sc * T + t != (uint32_t)ra.drop_item) {
            const int k = tid >> 2, w4 = tid & 3;
            // word 0: max, 1: min, 2: max below the split, 3: min at / above it (red rows 4k + {1, 0, 2, 3})
            const int row = 4 * k + (w4 == 0 ? 1 : w4 == 1 ? 0 : w4);
            float v = fmaxf(fmaxf(red[row][0], red[row][1]), fmaxf(red[row][2], red[row][3]));
            const int bd = redbad[0] | redbad[1] | redbad[2] | redbad[3];
            uint32_t val;
            if (w4 == 0) val = bd ? 0 : fkey(v);
            else if (w4 == 1) val = bd ? 0 : (uint32_t)~fkey(-v);
            else if (w4 == 2) val = (bd || v == -INFINITY) ? 0 : fkey(v);
            else val = (bd || v == -INFINITY) ? 0 : (uint32_t)~fkey(-v);
            const uint64_t w64 = (uint64_t)val | ((uint64_t)(gen | (bd ? kRecBad : 0u)) << 32);
            __hip_atomic_store(&ra.recs[(int64_t)(k * s3 + sc) * T + t].g[w4], w64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        MNW_STAMP(1)      // record fold (waits for the record loads) + barrier + publish
        uint32_t qq[kGroup];
        if (wv == 3) {
            // ---- 4a. wave 3: the three descriptors of OLD's sub-cell, lanes 0..11
            if (have_old && l < 12) {
                const int c = l >> 2;
                const uint64_t g = desc3_from_sweep_f32(jpar[c], (int64_t)c * s3 + osc, swp[c][l & 3], (swp[c][4] & 2u) != 0, meta, ot == 0);
                if ((l & 3) == 0) gran_s[c] = g;
            }
        } else {
            if (have_mid) quantize_mid();                 // (wave 3 does its share after the descriptors, while the others pack)
            if (have_mid && !(MNW_V3_DIAG & 2)) {
                // ---- 4b. waves 0..2: the group duty of MIDDLE for this component: publish the fold of its group's item records
                const int member = (int)(mt - mfirst);
                if (__builtin_expect((dfl & 1u) && !(MNW_V3_DIAG & 4), 0)) {
                    // an item record of the group was not there yet (a neighbour more than a trip behind): poll
                    uint32_t spins = 0;
                    const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
                    const ItemRec* r0 = ra.recs + (int64_t)(wv * s3 + msc) * T + mfirst;
                    for (;;) {
                        uint64_t w[4] = {0, 0, 0, 0};
                        uint32_t ms = 0, bd = 0;
                        if (l < mcnt) {
                            load_rec_sc1(r0 + l, w);
#pragma unroll
                            for (int k = 0; k < 4; ++k) { const uint32_t tag = (uint32_t)(w[k] >> 32); ms |= (tag & ~kRecBad) != gen; bd |= tag & kRecBad; }
                        }
                        if (!__any(ms)) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) dv[k] = wave_max_dpp(l < mcnt ? (uint32_t)w[k] : 0u);
                            dfl = __any(bd) ? 2u : 0u;
                            break;
                        }
                        if (poll_abort(spins, t_start, ra.abort_flag, ra.abort_host)) { dfl = 1u; if (l == 0) abort_s = 1; break; }
                    }
                }
                if (!(dfl & 1u) || (MNW_V3_DIAG & 4))
                    publish_group_rec(ra.grecs + (int64_t)(wv * s3 + msc) * R * R, (int)R, (int)mgrp, member, mcnt,
                                      dv[0], dv[1], dv[2], dv[3], (dfl & 2u) != 0, gen);
            }
            if (have_old) {
                // the lane's 32 pixel indices of OLD's plane, while wave 3 works out the descriptors
                const uint32_t* plane = lds + wv * kVPlane;
#pragma unroll
                for (int k8 = 0; k8 < kGroup / 4; ++k8) {
                    const u32x4 rr = *reinterpret_cast<const u32x4*>(plane + l * kStride32 + 4 * k8);
                    qq[4 * k8] = rr.x; qq[4 * k8 + 1] = rr.y; qq[4 * k8 + 2] = rr.z; qq[4 * k8 + 3] = rr.w;
                }
            }
        }
        MNW_STAMP(2)      // group duty, LDS read | descriptors
        lds_barrier();                                   // descriptors visible; every lane holds its group: the planes can be overwritten
        if (abort_s) break;
        if (wv == 3 && have_mid) quantize_mid();
        if (wv < 3 && have_old) {
            // ---- 4c. re-base, pack, store
            const Vec3JobPar& jp = jpar[wv];
            const Desc dw = gran_decode(gran_s[wv], jp.n);
            if (__builtin_amdgcn_readfirstlane(dw.status == 0 && dw.bits != 0)) {
                const uint32_t bw = (uint32_t)__builtin_amdgcn_readfirstlane(dw.bits);
                const uint32_t iminw = (uint32_t)(int32_t)dw.imin;
                const uint32_t wrapw = (jp.periodic == 1 && dw.pmin == dw.imin) ? (uint32_t)jp.pixels : 0u;
                uint32_t* plane = lds + wv * kVPlane;
                const int64_t e0 = (int64_t)ot * kVTile + kGroup * l;
                if (wrapw == 0) {
#pragma unroll
                    for (int i = 0; i < kGroup; ++i) qq[i] -= iminw;
                } else {
#pragma unroll
                    for (int i = 0; i < kGroup; ++i) { const uint32_t r2 = qq[i] - iminw; qq[i] = min(r2, r2 + wrapw); }
                }
                if (e0 + kGroup > jp.n) {
#pragma unroll
                    for (int i = 0; i < kGroup; ++i) if (e0 + i >= jp.n) qq[i] = 0;
                }
                pack_group32_switch(qq, bw, plane + l * bw);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                MNW_STAMP(3)      // re-base, pack
                const int64_t byte0 = (int64_t)ot * (kVTile / 8) * bw;
                int64_t nbytes_t = (int64_t)(kVTile / 8) * bw;
                if (byte0 + nbytes_t > dw.nbytes) nbytes_t = dw.nbytes - byte0;
                uint8_t* dst = (uint8_t*)jobs[(int64_t)wv * s3 + osc].dst;
                const int full = (int)(nbytes_t >> 4);
                for (int c = l; c < full; c += 64)
                    stg<u32x4>(dst + byte0 + 16 * (int64_t)c, *reinterpret_cast<const u32x4*>(plane + 4 * c));
                const int rem = (int)(nbytes_t & 15);
                if (l < rem) {
                    const uint32_t w = plane[4 * full + (l >> 2)];
                    stg<uint8_t>(dst + byte0 + 16 * (int64_t)full + l, (uint8_t)(w >> (8 * (l & 3))));
                }
            }
        }
        lds_barrier();                                   // planes free
        MNW_STAMP(4)      // stores issued + barrier
        // ---- 5. next trip's record loads; MIDDLE's pixel indices -> LDS; NEW becomes MIDDLE; start the next item
        issue_records(have_mid, msc, mt, have_new, sc, t);
        if (have_mid) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t e = 4 * (tid + 256 * r);                // element of the vtile
                const Vec3Group& g = gm[r];
#pragma unroll
                for (int k = 0; k < 3; ++k) *reinterpret_cast<sc_f32x4*>(lds + k * kVPlane + stage32_idx((int)e)) = g.comp(k);
            }
        }
        if (have_new) {
            // (the copy is pinned here, ahead of the loads: left to the compiler it becomes a copy at the loop's back edge, the
            // loads then go to temporaries and are WAITED FOR there to be moved into NEW's registers)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                gm[r] = gn[r];
                asm volatile("" : "+v"(gm[r].u), "+v"(gm[r].v), "+v"(gm[r].w));
            }
            if (have_next) issue(nsc, nt);
        }
        MNW_STAMP(5)      // quantize + park MIDDLE, next loads issued
        have_old = have_mid; osc = msc; ot = mt;
        have_mid = have_new; msc = sc; mt = t;
        sc = nsc; t = nt;
        have_new = have_next;
    }
    MNW_STAMP_FLUSH
}

// ------------------------------------------------------------------------------------------------
// minp.Reader.Vectors without the staging pass: the three streams of a sub-cell are unpacked vtile by vtile
// and written straight into the AoS grid (dequantize, group.go:299-310; periodic fix-up, minp.go:188-196;
// setSubCell).  32-bit streams only (bits <= 32, ring and minimum in int32: the NARROW class of
// k_unpack_f32); anything else goes through the staging path.  HBM traffic: the streams + 12 B/particle
// written, against streams + 12 B written + 12 B read + 12 B written of the staging path.
//
// Round 3: written against the instruction costs measured on gfx950 (tools/microbench/valu_cost.hip: add / sub / mul / fma
// of float32, and / or / xor, add / sub of integers and right shifts issue every 2 cycles per SIMD; compares, selects, min / max,
// left shifts, bit-field and funnel operations, conversions, integer multiplies and all float64 arithmetic every 4).  What a
// sub-cell's block makes unnecessary is decided once per (sub-cell, component) in scalar code, not per element:
//   * bound(buf, 0, pixels) (group.go:303): nothing when min + 2^bits - 1 < pixels, one unsigned minimum when min >= 0;
//   * the reader's fix-up (minp.go:188-196): nothing when the block's largest / smallest possible value is inside [0, L) --
//     every step of the dequantization is monotone in q + u, so the extremes of x are x of the extreme (q, u);
//   * element c of a lane's group of four sits at bit (group start) + c b: one add, the dword address, one ds_read2_b32,
//     one v_alignbit_b32 (which takes the shift from the low five bits itself) and the mask.
// float32(float64(q) + u) is one float64 fma (u = z 2^-32 is an exact product, so the fused form rounds once, like the
// addition) between three conversions.  The stream words are double buffered in LDS (one barrier per vtile), sized by the
// widest block of the call: 26 KB per workgroup for C4's 17 bits, six workgroups per CU.
// ------------------------------------------------------------------------------------------------
struct V3Dec {                  // one component's block of the current sub-cell; wave-uniform
    uint32_t b, mask;
    uint32_t mn, wrap;          // minimum (>= 0 on this path); pixels if bound() can fire, else 0 (then min(q, q - 0) = q)
    uint32_t ka, kb, sh;        // dither key halves; bit offset of the stream inside its first staged word
    float low, dx;
    bool fix_lo, fix_hi;        // the periodic fix-up can fire below 0 / at or above L
};

__device__ __forceinline__ V3Dec v3dec_make(const DevJob& jb, int periodic, float L) {
    V3Dec d;
    d.b = (uint32_t)jb.bits;
    d.mask = d.b == 0 ? 0u : (d.b == 32 ? ~0u : ((1u << d.b) - 1u));
    d.mn = (uint32_t)jb.min;
    d.low = jb.low; d.dx = jb.dxq;
    const uint64_t key = dither_key(jb.seed, jb.block, 0);       // element indices stay below 2^32
    d.ka = (uint32_t)key; d.kb = (uint32_t)(key >> 32);
    d.sh = (uint32_t)((uintptr_t)jb.src & 7) * 8;
    const int64_t top = jb.min + (int64_t)d.mask;                // largest value the stream can hold (host: min >= 0, top fits int32)
    const bool wraps = top >= jb.pixels;
    d.wrap = wraps ? (uint32_t)jb.pixels : 0u;
    // extremes of the dequantized value over everything the block can hold: q + u is in [qmin, qmax + 1 - 2^-32] and
    // x = fl(fl(dx * fl32(q + u)) + low) is monotone non-decreasing in it for dx >= 0
    d.fix_lo = d.fix_hi = periodic != 0;
    if (periodic && d.dx >= 0.0f) {
        const int64_t qmax = wraps ? jb.pixels - 1 : top, qmin = wraps ? 0 : jb.min;
        const float xmax = __fadd_rn(__fmul_rn(d.dx, (float)__fma_rn((double)0xffffffffu, 0x1p-32, (double)qmax)), d.low);
        const float xmin = __fadd_rn(__fmul_rn(d.dx, (float)(double)qmin), d.low);
        d.fix_hi = !(xmax < L);
        d.fix_lo = !(xmin >= 0.0f);
    }
    return d;
}

// element at bit `pos` of the staged words
__device__ __forceinline__ uint32_t v3_extract(const uint32_t* __restrict__ l32, uint32_t pos, uint32_t mask) {
    const uint32_t* p = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(l32) + ((pos >> 3) & ~3u));
    return __builtin_amdgcn_alignbit(p[1], p[0], pos) & mask;
}

// four consecutive elements of one component, starting at bit pos0 of the staged words and at element j of the block -> floats
__device__ __forceinline__ sc_f32x4 v3_dequant4(const V3Dec& d, const uint32_t* __restrict__ l32, uint32_t pos0, uint32_t j, float L) {
    sc_f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        // group.go:262, then bound(buf, 0, pixels) with q >= 0: one wrap from above, as an unsigned minimum
        uint32_t q = d.mn + v3_extract(l32, pos0 + (uint32_t)c * d.b, d.mask);
        q = min(q, q - d.wrap);
#ifdef MNW_DIAG_NO_DITHER       // diagnostic builds only: price the dither and the float64 detour
        const float f = (float)(int32_t)q;
#else
        const uint32_t z = dither_bits(j + (uint32_t)c, d.ka, d.kb);
        const float f = (float)__fma_rn((double)z, 0x1p-32, (double)(int32_t)q); // float32(float64(q) + u), u = z 2^-32 exactly
#endif
        o[c] = __fadd_rn(__fmul_rn(d.dx, f), d.low);                             // group.go:308: float32 multiply, then add
    }
    // minp.go:188-196, only for blocks in which it can fire (selects, not branches: the conditions differ from lane to lane)
    if (d.fix_hi) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float y = __fsub_rn(o[c], L); o[c] = o[c] >= L ? y : o[c]; }
    }
    if (d.fix_lo) {
        // (after the upper test: a value that was >= L is now in [0, L) and is not touched again; one below 0 was not touched above)
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float y = __fadd_rn(o[c], L); o[c] = o[c] < 0.0f ? y : o[c]; }
    }
    return o;
}

// Grid stores: a lane's four particles are 48 contiguous bytes, but three 16-byte stores at a 48-byte lane stride make every
// store instruction touch 24 cache lines a third each.  A wave's 256 particles of one pass -- 3 KB of the grid, contiguous but
// for the ends of sub-cell rows -- go through a wave-local LDS image (no barrier: a wave's LDS operations complete in order)
// and leave as three instructions of 64 consecutive 16-byte chunks each; the grid offset of every group of four travels
// through LDS beside it.
// vmcnt: a wave's vector-memory operations complete in issue order and the counter counts stores too.  The words of vtile
// v + 1 are requested BEFORE the grid stores of vtile v are issued, so waiting for them must not wait for those stores
// (measured, round 3: with `s_waitcnt vmcnt(0)` in front of the staging -- what the compiler emits as soon as a load or a store
// sits in divergent or conditional control flow, because it can then no longer count -- the launch took 573 us with the stores,
// 370 us without them, and not a microsecond less when the dither was removed).  Hence every load and store of the loop is an
// UNCONDITIONAL raw-buffer operation in straight-line code: lanes with nothing to do are given an offset outside the buffer
// (a load then returns zero and a store is dropped, neither makes a request), and the compiler's own count is exact.
constexpr int kV3ImgDw = 64 * 12 + 64;          // dwords per wave: the image and the 64 grid offsets
constexpr uint32_t kV3NoStore = 0xfffff800u;    // a grid offset beyond every buffer's range (they end below 0xfffff001), also with 32 added

__device__ __forceinline__ __amdgpu_buffer_rsrc_t v3_rsrc(const void* base, uint32_t bytes) {
    const uint64_t p = (uint64_t)base;
    const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(p >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)p);
    return __builtin_amdgcn_make_buffer_rsrc((void*)pu, 0, (int)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

// The aligned 64-bit words that hold vtile t of a stream -> registers (lane i of the workgroup takes words i, i + 256, ...);
// words past the stream's end read as zero (they only feed elements beyond n, which are never stored).
template <int NW>
__device__ __forceinline__ void v3_load_words(u32x2 (&w)[NW], const StreamView& sv, uint32_t t, uint32_t b, int tid) {
    const int64_t w0 = (int64_t)t * (kVTile / 64) * b;
    int64_t left = sv.last + 1 - w0;                             // words of the stream from w0 on
    if (left < 0 || b == 0) left = 0;
    if (left > NW * 256) left = NW * 256;
    const __amdgpu_buffer_rsrc_t r = v3_rsrc(sv.a + w0, (uint32_t)left * 8u);
#pragma unroll
    for (int k = 0; k < NW; ++k) w[k] = __builtin_amdgcn_raw_buffer_load_b64(r, (k * 256 + tid) * 8, 0, 0);
}

template <int NW, bool DB>
__global__ __launch_bounds__(256) void k_unpack_vec3(Vec3Args a, float* __restrict__ vec_out, const DevJob* __restrict__ jobs,
                                                     int periodic, float L, int tiles_per_wg) {
    static_assert(kTileD == kVTile, "the stream helpers are written for 2,048-element tiles");
    constexpr int kLW = NW * 256;                                       // 64-bit words of one component's buffer
    extern __shared__ __attribute__((aligned(16))) uint64_t v3_lds[];    // [1 or 2 buffers][3 components][kLW], then the images
    const SubcellGeom& gm = a.gm;
    int64_t v0, v1;
    if (tiles_per_wg > 0) {
        // Workgroups in dispatch order walk the sub-cells of one x-row together (sx fastest): the five row segments that make up
        // one row of the grid are then written at about the same time.  A sub-cell at a time leaves 1,200 of every 6,000 bytes
        // of the grid's rows to each of five sub-cells processed one after the other, and rows written with such gaps move at
        // 2.9 TB/s where contiguous ones move at 6.9 (tools/microbench/rw_mix.hip).
        const uint32_t S = gm.sub_cells, tq_n = (a.tiles_per_sc + tiles_per_wg - 1) / tiles_per_wg;
        const uint32_t w = blockIdx.x, sx = w % S, q = w / S, tq = q % tq_n, syz = q / tq_n;
        const uint32_t t0 = tq * tiles_per_wg, t1 = min(t0 + (uint32_t)tiles_per_wg, a.tiles_per_sc);
        v0 = (int64_t)(sx + S * syz) * a.tiles_per_sc + t0;
        v1 = v0 + (t1 - t0);
    } else {
        const int64_t total = (int64_t)a.s3 * a.tiles_per_sc;
        v0 = total * blockIdx.x / gridDim.x;
        v1 = total * (blockIdx.x + 1) / gridDim.x;
    }
    int buf = 0;
    uint32_t* const img = reinterpret_cast<uint32_t*>(v3_lds + (size_t)(DB ? 2 : 1) * 3 * kLW) + wave_id() * kV3ImgDw;
    while (v0 < v1) {
        const uint32_t sc = (uint32_t)(v0 / a.tiles_per_sc);
        int64_t vend = (int64_t)(sc + 1) * a.tiles_per_sc;
        if (vend > v1) vend = v1;
        V3Dec d[3];
        StreamView sv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const DevJob& jb = jobs[(int64_t)k * a.s3 + sc];
            d[k] = v3dec_make(jb, periodic, L);
            sv[k] = stream_view(jb, d[k].b);
        }
        // the sub-cell's part of the grid behind one buffer descriptor: offsets are relative to its first particle
        const uint32_t s2 = gm.sub_cells * gm.sub_cells;
        const uint32_t sz = sc / s2, sq = sc - sz * s2, sy = sq / gm.sub_cells, sx = sq - sy * gm.sub_cells;
        const uint32_t base0 = gm.n_sub * (sx + sy * gm.n_file + sz * gm.n_file2);
        const int64_t gleft = (gm.n - (int64_t)base0) * 12;             // bytes of the grid from there on (host: the sub-cell's span < 2^32)
        const __amdgpu_buffer_rsrc_t grid_r = v3_rsrc(vec_out + 3 * (int64_t)base0, gleft > 0xfffff000ll ? 0xfffff000u : (uint32_t)gleft);
        // the words of the next vtile are loaded into registers while the current one is extracted
        u32x2 w[3][NW];
        {
            const uint32_t t = (uint32_t)(v0 - (int64_t)sc * a.tiles_per_sc);
#pragma unroll
            for (int k = 0; k < 3; ++k) v3_load_words<NW>(w[k], sv[k], t, d[k].b, (int)threadIdx.x);
            // Six stores that store nothing (out of range: no request), so that the loop is entered with the same sequence of
            // outstanding operations -- the words' loads, then six stores -- as its back edge carries: the compiler merges the two
            // states at the loop header, and with different ones it can only wait for everything (vmcnt(0), i.e. for the grid
            // stores of the vtile before).
            const u32x4 nothing = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 6; ++i) __builtin_amdgcn_raw_buffer_store_b128(nothing, grid_r, (int)kV3NoStore, 0, 0);
        }
        for (int64_t v = v0; v < vend; ++v) {
            const uint32_t t = (uint32_t)(v - (int64_t)sc * a.tiles_per_sc);
            uint64_t* lw = v3_lds + (size_t)buf * 3 * kLW;
            if (!DB && v != v0) lds_barrier();                          // every wave is done with the words of the vtile before
#pragma unroll
            for (int k = 0; k < 3; ++k) {
#pragma unroll
                for (int i = 0; i < NW; ++i) *reinterpret_cast<u32x2*>(lw + k * kLW + i * 256 + threadIdx.x) = w[k][i];
                v3_load_words<NW>(w[k], sv[k], t + 1, d[k].b, (int)threadIdx.x);    // (past the sub-cell's end: zeros, no request)
            }
            lds_barrier();
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int lane = opaque_tid() & 63;
                const uint32_t e0 = 4 * (64 * wave_id() + 256 * r);    // the wave's first element of this pass
                const uint32_t e = e0 + 4 * lane;
                const uint32_t j = t * kVTile + e;
                // (lanes past the end of the sub-cell -- its last vtile -- compute on whatever the buffers hold and store nothing)
                sc_f32x4 o[3];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    o[k] = v3_dequant4(d[k], reinterpret_cast<const uint32_t*>(lw + k * kLW), e * d[k].b + d[k].sh, j, L);
                const sc_f32x4 q0 = {o[0].x, o[1].x, o[2].x, o[0].y}, q1 = {o[1].y, o[2].y, o[0].z, o[1].z},
                               q2 = {o[2].z, o[0].w, o[1].w, o[2].w};
                sc_f32x4* ip = reinterpret_cast<sc_f32x4*>(img + 12 * lane);
                ip[0] = q0; ip[1] = q1; ip[2] = q2;
                const uint32_t rel = vec3_particle(a, sc, j) - base0;                   // (garbage, and unused, for j >= n_sub^3)
                img[64 * 12 + lane] = j < gm.n_sub3 ? rel * 12u : kV3NoStore;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const uint32_t c = 64 * i + lane;                   // chunk of the wave's image
                    const uint32_t g = __umulhi(c, 0x55555556u), part = c - 3 * g;      // c / 3 for c < 2^31
                    const u32x4 ch = *reinterpret_cast<const u32x4*>(img + 4 * c);
                    const uint32_t off = img[64 * 12 + g];
#ifdef MNW_DIAG_V3_NOSTORE      // diagnostic builds only: price the grid stores
                    __builtin_amdgcn_raw_buffer_store_b128(ch, grid_r, (int)(ch.x == 0x7fc00001u ? off + 16 * part : kV3NoStore), 0, 0);
#else
                    __builtin_amdgcn_raw_buffer_store_b128(ch, grid_r, (int)(off + 16 * part), 0, 0);       // (kV3NoStore + 16 part: out of range)
#endif
                }
                // (the next pass overwrites the image: its writes follow these reads in the wave's LDS queue)
            }
            if (DB) buf ^= 1;
        }
        v0 = vend;
    }
}

static SubcellGeom make_geom(int64_t n_file, int64_t sub_cells) {
    SubcellGeom g;
    g.n_file = (uint32_t)n_file; g.sub_cells = (uint32_t)sub_cells; g.n_sub = (uint32_t)(n_file / sub_cells);
    g.n_sub2 = g.n_sub * g.n_sub; g.n_sub3 = g.n_sub2 * g.n_sub; g.n_file2 = g.n_file * g.n_file;
    g.n = n_file * n_file * n_file;
    return g;
}
static int subcell_grid(int64_t work, int cus) {
    int64_t g = (work + 255) / 256;
    const int64_t cap = (int64_t)cus * 16;
    if (g > cap) g = cap;
    return g < 1 ? 1 : (int)g;
}

void launch_subcell_gather(hipStream_t s, int cus, const float* vec, float* out, int64_t n_file, int64_t sub_cells) {
    const SubcellGeom g = make_geom(n_file, sub_cells);
    const int vec4 = (g.n_sub % 4 == 0) && (g.n_file % 4 == 0) && (((uintptr_t)vec | (uintptr_t)out) % 16 == 0);
    hipLaunchKernelGGL(k_subcell_gather, dim3(subcell_grid(vec4 ? g.n / 4 : g.n, cus)), dim3(256), 0, s, vec, out, g, vec4);
}
void launch_subcell_scatter(hipStream_t s, int cus, const float* in, float* vec, int64_t n_file, int64_t sub_cells, int periodic, float L) {
    const SubcellGeom g = make_geom(n_file, sub_cells);
    const int vec4 = (g.n_sub % 4 == 0) && (g.n_file % 4 == 0) && (((uintptr_t)vec | (uintptr_t)in) % 16 == 0);
    hipLaunchKernelGGL(k_subcell_scatter, dim3(subcell_grid(vec4 ? g.n / 4 : g.n, cus)), dim3(256), 0, s, in, vec, g, vec4, periodic, L);
}
void launch_bounds3(hipStream_t s, int cus, const float* vec, int64_t n, uint32_t* keys) {
    hipLaunchKernelGGL(k_bounds3, dim3(subcell_grid(n / 4 + 1, cus)), dim3(256), 0, s, vec, n, keys);
}

static Vec3Args make_vec3_args(const float* vec, int64_t n_file, int64_t sub_cells) {
    Vec3Args a;
    a.vec = vec; a.gm = make_geom(n_file, sub_cells);
    a.s3 = (uint32_t)(sub_cells * sub_cells * sub_cells);
    a.tiles_per_sc = (a.gm.n_sub3 + kVTile - 1) / kVTile;
    const bool magic = a.gm.n_sub % 4 == 0 && a.gm.n_sub >= 8 && a.gm.n_sub < 512;      // x d < 2^32 for both divisions (and d >= 2)
    a.m_row = magic ? (uint32_t)((1ull << 32) / (a.gm.n_sub / 4)) + 1u : 0u;
    a.m_nsub = magic ? (uint32_t)((1ull << 32) / a.gm.n_sub) + 1u : 0u;
    return a;
}
void launch_reduce_vec3(hipStream_t s, int grid, const float* vec, int64_t n_file, int64_t sub_cells, const DevJob* jobs, uint64_t* stats) {
    const Vec3Args a = make_vec3_args(vec, n_file, sub_cells);
    hipLaunchKernelGGL(k_reduce_vec3, dim3(grid), dim3(256), 0, s, a, jobs, stats);
}
void launch_pack_vec3(hipStream_t s, int grid, const float* vec, int64_t n_file, int64_t sub_cells, const DevJob* jobs, const uint64_t* stats,
                      Meta* meta) {
    const Vec3Args a = make_vec3_args(vec, n_file, sub_cells);
    hipLaunchKernelGGL(k_pack_vec3, dim3(grid), dim3(256), 0, s, a, jobs, stats, meta);
}
// single-read vectors encode.  recs: 3 s3 T item records, then 3 s3 R R group records (R = ceil(T / kVGroup)).
int64_t vec3_reg_recs(int64_t n_file, int64_t sub_cells) {
    const SubcellGeom g = make_geom(n_file, sub_cells);
    const int64_t s3 = sub_cells * sub_cells * sub_cells, T = (g.n_sub3 + kVTile - 1) / kVTile, R = (T + kVGroup - 1) / kVGroup;
    return 3 * s3 * (T + R * R);
}
int64_t vec3_reg_max_tiles_per_block() { return 32 * kVGroup; }           // R <= 32: one group record per lane of a half wave
// geometries k_encode_reg_vec3 takes: its multiply-high divisions need 8 <= n_sub <= 128 and sub_cells <= 32, its record
// offsets fewer than 2^26 records
bool vec3_reg_geometry_ok(int64_t n_file, int64_t sub_cells) {
    if (sub_cells < 1 || sub_cells > 32 || n_file % sub_cells) return false;
    const int64_t n_sub = n_file / sub_cells;
    if (n_sub < 8 || n_sub > 128 || n_sub % 4 || n_file % 4) return false;
    return vec3_reg_recs(n_file, sub_cells) < (1ll << 26);
}
int encode_reg_vec3_blocks_per_cu() {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_encode_reg_vec3, 256, 0) != hipSuccess) return 0;
    return n;
}
void launch_encode_reg_vec3(hipStream_t s, int grid, const float* vec, int64_t n_file, int64_t sub_cells, const DevJob* jobs, void* recs,
                            uint32_t gen, uint32_t* abort_flag, uint32_t* abort_host, Meta* meta, int64_t drop_item) {
    Vec3RegArgs ra;
    ra.a = make_vec3_args(vec, n_file, sub_cells);
    ra.R = (ra.a.tiles_per_sc + kVGroup - 1) / kVGroup;
    ra.recs = (ItemRec*)recs;
    ra.grecs = ra.recs + (size_t)3 * ra.a.s3 * ra.a.tiles_per_sc;
    ra.gen = gen; ra.abort_flag = abort_flag; ra.abort_host = abort_host; ra.drop_item = drop_item;
    // multiply-high division: exact for x d < 2^32; here x < n_sub^3 / 4 with d = n_sub / 4, and x < n_sub^2 with d = n_sub
    // (the host only sends n_sub <= 128 here: vec3_reg_max_tiles_per_block)
    ra.m_row = (uint32_t)((1ull << 32) / (ra.a.gm.n_sub / 4)) + 1u;
    ra.m_nsub = (uint32_t)((1ull << 32) / ra.a.gm.n_sub) + 1u;
    ra.m_s = sub_cells == 1 ? 0u : (uint32_t)((1ull << 32) / sub_cells) + 1u;              // (one sub-cell: every quotient is 0)
    ra.m_s2 = sub_cells == 1 ? 0u : (uint32_t)((1ull << 32) / (sub_cells * sub_cells)) + 1u;
    hipLaunchKernelGGL(k_encode_reg_vec3, dim3(grid), dim3(256), 0, s, ra, jobs, meta);
}
int64_t vec3_tiles(int64_t n_file, int64_t sub_cells) {
    const int64_t n_sub = n_file / sub_cells, n3 = n_sub * n_sub * n_sub;
    return sub_cells * sub_cells * sub_cells * ((n3 + kVTile - 1) / kVTile);
}
// max_bits: the widest block of the call (sizes the LDS word buffers and the prefetch registers); double_buffer: two word
// buffers and one barrier per vtile instead of one and two
// tiles_per_wg > 0: x-interleaved order, the grid is sub_cells^3 * ceil(T / tiles_per_wg) workgroups (`grid` is ignored)
void launch_unpack_vec3(hipStream_t s, int grid, float* vec_out, int64_t n_file, int64_t sub_cells, const DevJob* jobs, int periodic, float L,
                        int max_bits, bool double_buffer, int tiles_per_wg) {
    const Vec3Args a = make_vec3_args(vec_out, n_file, sub_cells);
    if (tiles_per_wg > 0) grid = (int)(a.s3 * ((a.tiles_per_sc + tiles_per_wg - 1) / tiles_per_wg));
    if (max_bits < 1) max_bits = 1;
    const int words = kVTile / 64 * max_bits + 2;                 // 64-bit words of one component's vtile
    const int nw = words <= 2 * 256 ? 2 : words <= 3 * 256 ? 3 : 5;
    const size_t lds = (size_t)(double_buffer ? 2 : 1) * 3 * nw * 256 * sizeof(uint64_t) + (size_t)4 * kV3ImgDw * sizeof(uint32_t);
#define MNW_LAUNCH_UV3(NWV) do { if (double_buffer) hipLaunchKernelGGL((k_unpack_vec3<NWV, true>), dim3(grid), dim3(256), lds, s, a, vec_out, jobs, periodic, L, tiles_per_wg); \
                                 else hipLaunchKernelGGL((k_unpack_vec3<NWV, false>), dim3(grid), dim3(256), lds, s, a, vec_out, jobs, periodic, L, tiles_per_wg); } while (0)
    if (nw == 2) MNW_LAUNCH_UV3(2);
    else if (nw == 3) MNW_LAUNCH_UV3(3);
    else MNW_LAUNCH_UV3(5);
#undef MNW_LAUNCH_UV3
}
// the grid offsets of k_unpack_vec3 are 32-bit and relative to a sub-cell's first particle
bool vec3_unpack_geometry_ok(int64_t n_file, int64_t sub_cells) {
    return (n_file / sub_cells) * n_file * n_file * 12 < 0xfff00000ll;
}
