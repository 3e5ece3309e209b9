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

// Round 3: the grid is read as consecutive 16-byte chunks, one per lane (64 lanes x 16 B contiguous per instruction), not as 48 bytes
// per lane at a 48-byte stride (every instruction then touched 24 cache lines for 1 KB of data: 1.2 TB/s).  Float f of the grid is
// component f mod 3; chunk c starts at float 4 c, i.e. at component c mod 3.  A workgroup takes 768 chunks per step (a multiple of
// three), lane t the chunks t, t + 256, t + 512 of them: the j-th of these starts at component (t + j) mod 3, so position p of it is
// component (t + j + p) mod 3 and an accumulator indexed by (j + p) mod 3 -- fixed at compile time -- holds component
// (index + t) mod 3 of the lane: three selects at the very end.
__global__ __launch_bounds__(256) void k_bounds3(const float* __restrict__ vec, int64_t n, uint32_t* __restrict__ keys) {
    float amn[3] = {INFINITY, INFINITY, INFINITY}, amx[3] = {-INFINITY, -INFINITY, -INFINITY};
    const int64_t nchunks = (3 * n) / 4;                 // whole 16-byte chunks of the grid
    const int t = threadIdx.x;
    for (int64_t s0 = blockIdx.x; s0 * 768 < nchunks; s0 += gridDim.x) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int64_t c = s0 * 768 + t + 256 * j;
            if (c < nchunks) {
                const sc_f32x4 v = *((const MNW_GLOBAL_SC sc_f32x4*)vec + c);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    amn[(j + p) % 3] = fminf(amn[(j + p) % 3], v[p]);
                    amx[(j + p) % 3] = fmaxf(amx[(j + p) % 3], v[p]);
                }
            }
        }
    }
    float mn[3], mx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int idx = (k + 3 - t % 3) % 3;               // the accumulator that holds component k in this lane
        mn[k] = idx == 0 ? amn[0] : idx == 1 ? amn[1] : amn[2];
        mx[k] = idx == 0 ? amx[0] : idx == 1 ? amx[1] : amx[2];
    }
    if (blockIdx.x == 0 && (int64_t)t < 3 * n - 4 * nchunks) {        // the floats behind the last whole chunk (at most three)
        const int64_t f = 4 * nchunks + t;
        const float x = vec[f];
        const int k = (int)(f % 3);
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) if (kk == k) { mn[kk] = fminf(mn[kk], x); mx[kk] = fmaxf(mx[kk], x); }
    }
    // one set of six atomics per WORKGROUP, and only where it would change the key: atomics on one address are served one after
    // the other (13 ns each, measured on the item counter of k_encode_reg_vec3's dynamic hand-out) -- six per wave of 4,096
    // workgroups were 1.2 ms of this kernel's 1.28
    __shared__ float wred[6][4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float a = sc_wave_max(mx[k]), b = sc_wave_min(mn[k]);
        if ((threadIdx.x & 63) == 0) { wred[2 * k][threadIdx.x >> 6] = a; wred[2 * k + 1][threadIdx.x >> 6] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int q = threadIdx.x;
        float v = wred[q][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) v = (q & 1) ? fminf(v, wred[q][w]) : fmaxf(v, wred[q][w]);
        if (v != ((q & 1) ? INFINITY : -INFINITY)) {
            const uint32_t key = (q & 1) ? ~sc_fkey(v) : sc_fkey(v);
            if (key > __hip_atomic_load(&keys[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&keys[q], key);
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
// minp.Writer.Vectors with the grid read ONCE: k_encode_reg_vec3 (round 3 form).
//
// An item = one vtile = 2,048 particles of one sub-cell, all three components (24 KB of the grid, 24 values per lane); items
// are dealt in rounds over the G resident workgroups (item i to workgroup i mod G).  A sub-cell's three blocks have T items
// each (489 at C4's geometry), and a block's width is known only when all of them have been seen.  Round 2's kernel let every
// workgroup fold the records of its block's items (in two levels, 3 x (32 + R) records per trip and workgroup): bit-exact and
// no faster than two passes over the grid, because at ~40 vector instructions per element it was bound by instruction issue.
// This form is written against the measured instruction costs (tools/microbench/valu_cost.hip) and does per element only
// what cannot be done per item, per sub-cell or per launch:
//   * Quantize on arrival, statistics on the pixel indices (integers): min / max of q and of q - pixels/2 (mod 2^32) with
//     three-input min / max, one subtraction, and one float addition that carries NaN / Inf -- 7 instructions per element
//     where the float statistics took 12.  The float64-multiply quotient (k_pack_f32's header) needs no per-element guard for
//     the denormal range: a quotient down there quantizes to 0 or -1, so a wave whose minimum is <= 0 reloads its floats and
//     redoes that component with the IEEE division (rare: only vtiles that touch the bottom of the range).
//   * ONE workgroup per sub-cell derives its three descriptors, and it is none of those that hold items: the last n_sweepers
//     workgroups of the grid are sweepers (below).  Everybody else reads 16 bytes per component and trip instead of folding records.
//   * DEPTH items per workgroup (a record that is read must be at least a trip old; with one sweep per sub-cell, two): F, floats
//     in flight for the next trip; Q, the pixel indices of the item quantized in this trip (registers; the quantization moves F
//     to Q, so nothing is copied); P (DEPTH >= 4) the item of the trip before; and the OLD item, parked in LDS as three planes in
//     the pack core's staging layout.
// Per trip:   1. waves 0..2 pack OLD's component w from its plane with the descriptor requested a trip ago (polling for it if it
//                is not there yet) and store the words (wave-local)
//             barrier -- 2. P (or Q) -> planes  3. F (arrived) -> quantize -> Q, lane statistics, wave reduction -> LDS
//                        4. request the descriptor of the sub-cell now in the planes, then the next item's floats -> F
//             barrier -- 5. wave 3 publishes the item's three records
// vmcnt (see k_unpack_vec3): every wave issues the SAME vector-memory instructions every trip -- eight 16-byte stores and one
// byte store of packed words, one descriptor load, six loads of floats, one record store -- as unconditional raw-buffer
// operations whose idle lanes (or whole idle waves) carry an out-of-range offset, so that the compiler's counted waits let the
// stores of a trip stay in flight under the loads that were issued before them.
// No deadlock (blocks of at most G vtiles, all workgroups resident): a workgroup publishes item i before it needs anything
// that depends on item i; a sweeper needs item records only and waits for nothing else; an item workgroup reaches the trip
// that publishes an item of sub-cell s once the descriptors of its earlier items are out, which belong to sub-cells <= s whose
// records were all published at least a trip earlier -- induction over the sub-cells in order (a sweeper takes its sub-cells in
// increasing order too).  Waits are bounded (poll_abort): on expiry the launch gives up and the host reruns the call through
// k_reduce_vec3 + k_pack_vec3.   HBM traffic = algorithmic: 12 B/particle read + the streams (+ 96 B of records per item).
// ------------------------------------------------------------------------------------------------
// Round sizes of k_encode_reg_vec3's weighted hand-out, in THIRDS of the grid (three workgroups per CU: a CU's three come from the
// three thirds of the grid in dispatch order, and the hardware favours the oldest -- measured: the first eighth of the workgroups
// found its descriptors late 65 % of the time, the last 19 %, i.e. the late ones set the pace).  Table k gives the thirds
// 8:7:6, 7:6:5, 9:8:7, 10:9:8 items per cycle; no two partial rounds adjacent (k_encode_reg_f32's rule).
constexpr int kV3Tables = 4;
__constant__ int kV3RoundCount[kV3Tables] = {8, 7, 9, 10};
__constant__ int kV3RoundTable[kV3Tables][10] = {
    {3, 3, 1, 3, 3, 3, 2, 3},
    {3, 3, 1, 3, 3, 2, 3},
    {3, 3, 3, 1, 3, 3, 3, 2, 3},
    {3, 3, 3, 1, 3, 3, 3, 2, 3, 3},
};
struct Vec3RegArgs {
    Vec3Args a;
    ItemRec* recs;            // [3 s3][T] item records, then [3 s3] 16-byte descriptor records
    uint32_t gen;
    uint32_t* abort_flag;
    uint32_t* abort_host;
    int64_t drop_item;        // test hook: the records of that vtile are never published
    uint32_t n_sweepers;      // workgroups at the end of the grid that derive the sub-cells' descriptors
    int weighted;             // 0: every round is a whole grid; k > 0: round sizes from kRoundTable[k - 1] (k_encode_reg_f32: the quarters of the grid run at different speeds)
    uint32_t m_T;             // floor(2^32 / T) + 1: item / T == umulhi(item, m_T) for item T < 2^32
};

constexpr uint32_t kV3Oob = 0xfffff800u;        // an offset beyond every buffer of this kernel, also with a few hundred bytes added

// Twelve per-lane values -> their twelve wave-wide unsigned maxima, in lanes 0..3 (three each: lane q holds the values
// 6 (q & 1) + 3 (q >> 1) + {0, 1, 2}).  Two halving steps inside each quad (a lane keeps half of its values and takes the
// partner's half of those), then the three survivors are folded over the quads of a row and over the rows.
__device__ __forceinline__ void wave_max12_u32(const uint32_t (&m)[12], uint32_t (&out)[3]) {
    auto dpp = [](uint32_t x, auto ctrl) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, decltype(ctrl)::value, 0xf, 0xf, true); };
    const int l = lane_id();
    const bool b0 = l & 1, b1 = l & 2;
    uint32_t r6[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const uint32_t keep = b0 ? m[i + 6] : m[i], send = b0 ? m[i] : m[i + 6];
        r6[i] = max(keep, dpp(send, std::integral_constant<int, 0xB1>{}));            // quad_perm [1,0,3,2]
    }
    uint32_t r3[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const uint32_t keep = b1 ? r6[i + 3] : r6[i], send = b1 ? r6[i] : r6[i + 3];
        r3[i] = max(keep, dpp(send, std::integral_constant<int, 0x4E>{}));            // quad_perm [2,3,0,1]
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        uint32_t v = r3[i];
        v = max(v, dpp(v, std::integral_constant<int, 0x124>{}));                     // row_ror:4
        v = max(v, dpp(v, std::integral_constant<int, 0x128>{}));                     // row_ror:8
        v = max(v, (uint32_t)__shfl_xor((int)v, 16));
        v = max(v, (uint32_t)__shfl_xor((int)v, 32));
        out[i] = v;
    }
}

// The statistics of a set of pixel indices, all four as unsigned maxima ("max form": what the item records hold and what
// every fold is): 0: max q, 1: min q, 2: max of u = q - h mod 2^32, 3: min of u (h = pixels / 2: for q in [0, pixels) the
// indices below h are exactly the u >= 2^32 - h, so max u belongs to the largest q below h and min u to the least q at or
// above it -- the two extra statistics of periodicMin's second arc, make_desc_f32).
__device__ __forceinline__ uint32_t v3_enc_max(int32_t mx) { return (uint32_t)mx ^ 0x80000000u; }
__device__ __forceinline__ uint32_t v3_enc_min(int32_t mn) { return ~((uint32_t)mn ^ 0x80000000u); }
__device__ __forceinline__ int32_t v3_dec_max(uint32_t w) { return (int32_t)(w ^ 0x80000000u); }
__device__ __forceinline__ int32_t v3_dec_min(uint32_t w) { return (int32_t)(~w ^ 0x80000000u); }

// what a block's descriptor needs of its job, the same for every sub-cell of a component (LDS copy, made once)
struct Vec3JobPar {
    float low, dxq;
    int64_t pixels, n, cap;
};

// a block's descriptor from its swept statistics (max form) -- the decisions of make_desc_f32 on integers
// (wraps: the block is the ring's second arc -- values below the arc's start exist and bound() lifts them by `pixels`; in the other
// two cases every value is at or above the block minimum and the re-basing needs no wrap)
__device__ __forceinline__ Desc v3_desc(const Vec3JobPar& jp, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, bool bad, bool& wraps) {
    Desc d = {0, 0, 0, 0, 0};
    wraps = false;
    const int64_t mx0 = v3_dec_max(w0), mn0 = v3_dec_min(w1);
    const uint32_t umx = w2, umn = ~w3;
    // (a quotient beyond int32 saturates the conversion: INT32_MAX >= pixels and INT32_MIN < 0 send the block to the exact path too)
    if (bad || mn0 < 0 || mx0 >= jp.pixels || jp.pixels > 0x7fffffffll) { d.status = (int)kNeedsExact; return d; }
    const int64_t h = jp.pixels / 2;
    uint64_t maxdelta;
    const bool has_lo = (int32_t)umx < 0, has_hi = (int32_t)umn >= 0;
    const int64_t mxlo = (int64_t)(uint32_t)(umx + (uint32_t)h), mnhi = (int64_t)(uint32_t)(umn + (uint32_t)h);
    if (mx0 - mn0 + 1 <= h) { d.pmin = mn0; d.imin = mn0; maxdelta = (uint64_t)(mx0 - mn0); }
    else if (has_lo && has_hi && (mxlo + jp.pixels) - mnhi + 1 <= h) { d.pmin = mnhi; d.imin = mnhi; maxdelta = (uint64_t)(mxlo + jp.pixels - mnhi); wraps = true; }
    else { d.pmin = 0; d.imin = mn0; maxdelta = (uint64_t)(mx0 - mn0); }
    d.bits = precision_needed(maxdelta);
    d.nbytes = d.bits == 0 ? 0 : array_bytes(d.bits, jp.n);
    if (d.nbytes > jp.cap) d.status = -5;
    return d;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t v3_rsrc(const void* base, uint32_t bytes) {
    const uint64_t p = (uint64_t)base;
    const uint64_t pu = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(p >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)p);
    return __builtin_amdgcn_make_buffer_rsrc((void*)pu, 0, (int)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}

// DEPTH: items between a vtile's publication and its pack.  3: F, Q, planes (128 registers, four workgroups per CU).  4: one more
// item in registers (P, quantized two trips ago).  5: another one waiting in lane-private LDS slots (W) -- at three workgroups per
// CU, which the registers of 4 already cost, half the LDS is idle.  A descriptor is needed DEPTH - 2 whole trips after its
// sub-cell's last record; it arrives two memory hops and a sweep after that (10-17 us measured, against trips of 5.5-7.5 us).
template <int DEPTH>
__global__ __launch_bounds__(256, DEPTH == 3 ? 4 : 3) void k_encode_reg_vec3(Vec3RegArgs ra, const DevJob* __restrict__ jobs, Meta* __restrict__ meta) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[3 * kVPlane];
    __shared__ __attribute__((aligned(16))) uint32_t wait_s[DEPTH == 5 ? 6 : 1][256][4];   // DEPTH 5: the item W, lane-private
    __shared__ uint32_t red[12][4];                      // the item's statistics per wave, max form
    __shared__ int redbad[4];
    __shared__ Vec3JobPar jpar[3];
    __shared__ int abort_s;
    const Vec3Args& a = ra.a;
    const SubcellGeom& gm = a.gm;
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(wave_id()), l = lane_id();    // (uniform: the roles below are scalar branches)
    const uint32_t T = a.tiles_per_sc, s3 = a.s3, gen = ra.gen;
    const uint32_t total = s3 * T;                       // (below 2^26: the host's record budget)
    const uint32_t n_item_recs = 3u * s3 * T;
    // a round moves every workgroup on by G items = dsc sub-cells and dt vtiles
    const uint32_t G = gridDim.x - ra.n_sweepers;          // (the last n_sweepers workgroups hold no items)
    if (tid < 3) {
        const DevJob& jb = jobs[tid * s3];
        Vec3JobPar p = {jb.low, jb.dxq, jb.pixels, jb.n, jb.cap};
        jpar[tid] = p;
    }
    if (tid == 0) abort_s = 0;
    // the quantization constants of the three components (the same for every sub-cell: mnw_vectors_encode fills them so)
    float qlow[3], qdx[3];
    double qrdx[3];
    uint32_t qh[3], qpx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { const DevJob& jb = jobs[k * s3]; qlow[k] = jb.low; qdx[k] = jb.dxq; qrdx[k] = jb.rdx; qh[k] = (uint32_t)(jb.pixels / 2); qpx[k] = (uint32_t)jb.pixels; }
    // everything the launch exchanges behind one buffer descriptor: item records, then the descriptor records
    const __amdgpu_buffer_rsrc_t rec_r = v3_rsrc(ra.recs, (n_item_recs * 2u + 3u * s3) * 16u);
    const uint32_t desc_off0 = n_item_recs * 32u;

    // ---- sweepers: the last ra.n_sweepers workgroups of the grid hold no items.  Sweeper j derives the descriptors of the sub-cells
    // j, j + n_sweepers, ...: it waits for the sub-cell's last record, folds the 3 T item records (all four waves, 512 records per
    // batch of loads; again until none is missing), and lanes 0..2 of wave 3 decide and publish the three descriptor records and the
    // blocks' metadata.  Nobody's
    // pipeline stands still for this: when the workgroup that holds a sub-cell's last vtile did it, its sweep -- two dependent
    // round trips and a wait for the slowest of 489 workgroups -- made it the straggler of the next rounds, whose own records
    // the next sub-cells' owners then waited for: every round paid for a sweep (measured: 80 % of the descriptor requests came
    // back empty, 4.7 us of every 11 us trip were spent polling).
    if (blockIdx.x >= G) {
        // (the sweepers are the youngest waves of their CUs, and the hardware favours a CU's oldest: without a raised priority a
        // sweep waits behind the item workgroups' instructions)
        __builtin_amdgcn_s_setprio(3);
        for (uint32_t sc_ = blockIdx.x - G; sc_ < s3; sc_ += ra.n_sweepers) {
            uint32_t spins = 0;
            const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
            // a hint that the sub-cell is (nearly) complete: the last word of its last vtile's records
            for (;;) {
                const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rec_r, (int)(((2u * s3 + sc_) * T + (T - 1)) * 32u + 24u), 0, 16);
                if (((w.y | kRecBad) ^ (gen | kRecBad)) == 0) break;
                if (poll_abort(spins, t_start, ra.abort_flag, ra.abort_host)) { if (l == 0) abort_s = 1; break; }
            }
            if (wv == 0) { MNW_V3_TRACE(3 * total + 2 * sc_) }
            lds_barrier();
            if (abort_s) return;
            for (;;) {
                uint32_t acc[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) acc[i] = 0;
                uint32_t missing = 0, anybad = 0;
                for (uint32_t t0 = 0; t0 < T; t0 += 512) {
                    u32x4 ra_[2][3], rb_[2][3];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t tt = t0 + 256u * h + (uint32_t)opaque_tid();
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const int off = (int)(tt < T ? (((uint32_t)k * s3 + sc_) * T + tt) * 32u : kV3Oob);
                            ra_[h][k] = __builtin_amdgcn_raw_buffer_load_b128(rec_r, off, 0, 16);
                            rb_[h][k] = __builtin_amdgcn_raw_buffer_load_b128(rec_r, off + 16, 0, 16);
                        }
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t tt = t0 + 256u * h + (uint32_t)opaque_tid();
                        if (tt < T) {
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                const uint32_t tg = ra_[h][k].y | ra_[h][k].w | rb_[h][k].y | rb_[h][k].w, ta = ra_[h][k].y & ra_[h][k].w & rb_[h][k].y & rb_[h][k].w;
                                missing |= ((tg | kRecBad) ^ (gen | kRecBad)) | ((ta | kRecBad) ^ (gen | kRecBad));
                                anybad |= (tg >> 31) << k;
                                acc[4 * k] = max(acc[4 * k], ra_[h][k].x); acc[4 * k + 1] = max(acc[4 * k + 1], ra_[h][k].z);
                                acc[4 * k + 2] = max(acc[4 * k + 2], rb_[h][k].x); acc[4 * k + 3] = max(acc[4 * k + 3], rb_[h][k].z);
                            }
                        }
                    }
                }
                uint32_t o3[3];
                wave_max12_u32(acc, o3);
                const uint32_t wbad = (__any(anybad & 1u) ? 1u : 0u) | (__any(anybad & 2u) ? 2u : 0u) | (__any(anybad & 4u) ? 4u : 0u) | (__any(missing != 0) ? 8u : 0u);
                if (l < 4) {
                    const int base = 6 * (l & 1) + 3 * (l >> 1);
#pragma unroll
                    for (int i = 0; i < 3; ++i) red[base + i][wv] = o3[i];
                }
                if (l == 0) redbad[wv] = (int)wbad;
                lds_barrier();
                const uint32_t flags = (uint32_t)(redbad[0] | redbad[1] | redbad[2] | redbad[3]);
                if ((flags & 8u) == 0) {
                    if (tid >= 192 && tid < 195) {                 // lanes 0..2 of wave 3: component tid - 192
                        const int k = tid - 192;
                        uint32_t w4[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) w4[j] = max(max(red[4 * k + j][0], red[4 * k + j][1]), max(red[4 * k + j][2], red[4 * k + j][3]));
                        bool wraps;
                        const Desc d = v3_desc(jpar[k], w4[0], w4[1], w4[2], w4[3], ((flags >> k) & 1u) != 0, wraps);
                        const int64_t j = (int64_t)k * s3 + sc_;
                        Meta m = {d.imin, (int64_t)d.bits, d.nbytes, (int64_t)d.status};
                        meta[j] = m;
                        const uint32_t code = d.status == 0 ? 0u : d.status == (int)kNeedsExact ? 1u : d.status == -5 ? 2u : 3u;
                        const uint32_t info = (uint32_t)d.bits | (wraps ? 0x80u : 0u) | (code << 8);
                        const u32x4 rec = {(uint32_t)(int32_t)d.imin, gen, info, gen};
                        __builtin_amdgcn_raw_buffer_store_b128(rec, rec_r, (int)(desc_off0 + (uint32_t)j * 16u), 0, 16);
                    }
                    if (wv == 3) { MNW_V3_TRACE(3 * total + 2 * sc_ + 1) }
                    lds_barrier();                               // (red / redbad are free again)
                    break;
                }
                if (wv == 0 && poll_abort(spins, t_start, ra.abort_flag, ra.abort_host) && l == 0) abort_s = 1;
                lds_barrier();
                if (abort_s) return;
            }
        }
        return;
    }

    // origin (particle index) of a sub-cell: scalar
    auto origin = [&](uint32_t sc_) {
        const uint32_t S = gm.sub_cells, s2 = S * S;
        const uint32_t sz = sc_ / s2, q_ = sc_ - sz * s2, sy = q_ / S, sx = q_ - sy * S;
        return gm.n_sub * (sx + sy * gm.n_file + sz * gm.n_file2);
    };
    // grid offset (bytes, relative to the sub-cell's origin) of the group of four particles that starts at element j; out of range
    // past the end of the sub-cell
    auto group_off = [&](uint32_t j) -> uint32_t {
        const uint32_t j4 = j >> 2, rowlen = gm.n_sub >> 2;
        const uint32_t row = __umulhi(j4, a.m_row), jx4 = j4 - row * rowlen;              // row = jy + n_sub jz
        const uint32_t jz = __umulhi(row, a.m_nsub), jy = row - jz * gm.n_sub;
        const uint32_t rel = 4 * jx4 + jy * gm.n_file + jz * gm.n_file2;
        return j < gm.n_sub3 ? rel * 12u : kV3Oob;
    };
    // the floats of vtile (sc_, t_): lane = two groups of four particles, three 16-byte loads each (zeros, and no request, past the end)
    sc_f32x4 F[6];
    auto issue_floats = [&](bool have, uint32_t sc_, uint32_t t_) {
        const uint32_t b0 = origin(sc_);
        const int64_t left = (gm.n - (int64_t)b0) * 12;
        const __amdgpu_buffer_rsrc_t r = v3_rsrc(a.vec + 3 * (int64_t)b0, !have ? 0u : left > 0xfffff000ll ? 0xfffff000u : (uint32_t)left);
        const int ot = opaque_tid();
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const uint32_t off = group_off(t_ * kVTile + 4 * (uint32_t)(ot + 256 * g));
#pragma unroll
            for (int i = 0; i < 3; ++i)
                F[3 * g + i] = __builtin_bit_cast(sc_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(off + 16 * i), 0, 0));
        }
    };
    // the descriptor record of component wv (waves 0..2) of sub-cell sc_: {imin, tag, info, tag}
    u32x4 dsc_rec;
    auto issue_desc = [&](bool have, uint32_t sc_) {
        const uint32_t k = wv < 3 ? wv : 0;
        dsc_rec = __builtin_amdgcn_raw_buffer_load_b128(rec_r, (int)(have ? desc_off0 + (k * s3 + sc_) * 16u : kV3Oob), 0, 16);     // sc1
    };

    // items as (sub-cell, vtile): N = the one whose floats are in flight, Q = quantized in the last trip, O = parked in LDS
    // Item hand-out as in k_encode_reg_f32: in round k the workgroups [0, size_k) take the items base_k + w; with `weighted` the
    // round sizes cycle through a table in quarters of the grid (no two partial rounds adjacent: two consecutive items of a
    // workgroup stay at least a whole grid, hence at least T items, apart)
    const int weighted = ra.weighted;
    const uint32_t unit = weighted ? G / 3 : G;
    int round = 0;
    uint32_t round_base = 0;
    auto next_item = [&]() -> uint32_t {
        for (;;) {
            round_base += (uint32_t)(weighted ? kV3RoundTable[weighted - 1][round] : 1) * unit;
            round = round + 1 == (weighted ? kV3RoundCount[weighted - 1] : 1) ? 0 : round + 1;
            if (round_base >= total) return total;
            if (blockIdx.x < (uint32_t)(weighted ? kV3RoundTable[weighted - 1][round] : 1) * unit) return round_base + blockIdx.x;
        }
    };
    uint32_t nit = blockIdx.x;                          // round 0 is a full one
    uint32_t nsc = T == 1 ? nit : __umulhi(nit, ra.m_T), nt = nit - nsc * T, qsc = 0, qt = 0, psc = 0, pt = 0, wsc = 0, wt = 0, osc = 0, ot = 0;
    bool have_n = nit < total, have_q = false, have_p = false, have_w = false, have_o = false;
    uint32_t P[3][8];                                    // DEPTH >= 4: the item quantized two trips ago
    uint32_t Q[3][8];                                    // pixel indices of the Q item: component k, the lane's 8 particles
    issue_desc(false, 0);
    issue_floats(have_n, nsc, nt);
    MNW_STAMP_DECL
    MNW_V3_COUNT_DECL
    MNW_STAMP_START
    lds_barrier();

    while (have_n || have_q || have_p || have_w || have_o) {
        // ---- 1. pack the parked item (waves 0..2; wave 3 issues the same, idle, memory instructions) ---------------------------
        {
            // (all waves run the same vector-memory instructions: eight chunk stores and one tail store; wave 3's are out of range)
            // (lane-specific addresses are built from an opaque lane number inside the trip: hoisted out of the loop they are kept,
            // and spilled, across everything else)
            const int lq = opaque_tid() & 63;
            uint32_t bw = 0, full = 0, rem = 0, lpad = 0;
            int64_t byte0 = 0;
            uint32_t* plane = lds + (wv < 3 ? wv : 0) * kVPlane;
            __amdgpu_buffer_rsrc_t dst_r = v3_rsrc(nullptr, 0);
            if (wv < 3 && have_o) {
                // the descriptor, requested a trip ago; poll if the sub-cell's owner has not got there yet
                const Vec3JobPar& jp = jpar[wv];
                uint32_t imin = dsc_rec.x, info = dsc_rec.z;
                if (wv == 0) { MNW_V3_TRACE(total + osc * T + ot) }
                MNW_V3_COUNT(0);
                if (__builtin_expect(((dsc_rec.y ^ gen) | (dsc_rec.w ^ gen)) != 0, 0)) {
                    uint32_t spins = 0;
                    const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
                    MNW_V3_COUNT(1);
                    for (;;) {
                        MNW_V3_COUNT(2);
                        // (a workgroup that waits here is AHEAD of the ones whose records the descriptor is waiting for, some of them
                        // on this CU: it must not take issue slots and memory requests from them -- sleep between looks)
                        __builtin_amdgcn_s_sleep(64);
                        const u32x4 d2 = __builtin_amdgcn_raw_buffer_load_b128(rec_r, (int)(desc_off0 + ((uint32_t)wv * s3 + osc) * 16u), 0, 16);
                        if (((d2.y ^ gen) | (d2.w ^ gen)) == 0) { imin = d2.x; info = d2.z; break; }
                        if (poll_abort(spins, t_start, ra.abort_flag, ra.abort_host)) { if (l == 0) abort_s = 1; info = 0x100u; break; }
                    }
                }
                // info: bits 0..6 width, bit 7 the block is the ring's second arc (re-basing wraps), bits 8.. status code (0 ok)
                imin = (uint32_t)__builtin_amdgcn_readfirstlane((int)imin); info = (uint32_t)__builtin_amdgcn_readfirstlane((int)info);
                if (wv == 0) { MNW_V3_TRACE(2 * total + osc * T + ot) }
                MNW_STAMP(0)      // the descriptor is there (wait for its load, or poll)
                bw = (info >> 8) == 0 ? (info & 0x7fu) : 0u;
                if (bw != 0) {
                    const uint32_t wrapw = (info & 0x80u) ? (uint32_t)jp.pixels : 0u;
                    uint32_t qq[kGroup];
#pragma unroll
                    for (int k8 = 0; k8 < kGroup / 4; ++k8) {
                        const u32x4 rr = *reinterpret_cast<const u32x4*>(plane + lq * kStride32 + 4 * k8);
                        qq[4 * k8] = rr.x; qq[4 * k8 + 1] = rr.y; qq[4 * k8 + 2] = rr.z; qq[4 * k8 + 3] = rr.w;
                    }
                    const int64_t e0 = (int64_t)ot * kVTile + kGroup * lq;
                    if (wrapw == 0) {
#pragma unroll
                        for (int i = 0; i < kGroup; ++i) qq[i] -= imin;
                    } else {
#pragma unroll
                        for (int i = 0; i < kGroup; ++i) { const uint32_t r2 = qq[i] - imin; qq[i] = min(r2, r2 + wrapw); }
                    }
                    if (e0 + kGroup > jp.n) {
#pragma unroll
                        for (int i = 0; i < kGroup; ++i) if (e0 + i >= jp.n) qq[i] = 0;
                    }
                    // (every lane holds its group: the plane can be overwritten.)  A lane's bw words start bw dwords after its
                    // neighbour's; for bw = 8, 16, 24, 32 the 16-byte writes of eight neighbouring lanes then fall on two or four
                    // bank groups instead of eight (LDS bank conflicts: 56 % of this kernel's LDS cycles by the counters, positions
                    // being 16 and 17 bits wide): those widths get one empty 16-byte unit per lane (stride bw + 4: an odd number of
                    // units), which the read-out below skips
                    lpad = (bw & 7u) == 0 ? 4u : 0u;
                    pack_group32_switch(qq, bw, plane + lq * (bw + lpad));
                    byte0 = (int64_t)ot * (kVTile / 8) * bw;
                    const int64_t nbytes = array_bytes(bw, jp.n);
                    int64_t nbytes_t = (int64_t)(kVTile / 8) * bw;
                    if (byte0 + nbytes_t > nbytes) nbytes_t = nbytes - byte0;
                    full = (uint32_t)(nbytes_t >> 4); rem = (uint32_t)(nbytes_t & 15);
                    const DevJob& jb = jobs[(int64_t)wv * s3 + osc];
                    dst_r = v3_rsrc((const uint8_t*)jb.dst + byte0, (uint32_t)nbytes_t);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // 16-byte unit c of the stream sits at unit c + c / (bw / 4) of the plane when the lanes' outputs are padded
            const uint32_t m_units = lpad ? 0xffffffffu / (bw >> 2) + 1u : 0u;          // (scalar; c / d == umulhi(c, m) for c < 2^16)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t c = (uint32_t)lq + 64u * i;
                const uint32_t pc = c + __umulhi(c, m_units);
                const u32x4 ch = *reinterpret_cast<const u32x4*>(plane + 4 * (pc < (uint32_t)(kVPlane / 4) ? pc : 0u));
                __builtin_amdgcn_raw_buffer_store_b128(ch, dst_r, (int)(c < full ? 16u * c : kV3Oob), 0, 0);
            }
            {
                const uint32_t tu = full + ((uint32_t)lq >> 4);                            // the unit that holds the lane's tail byte
                const uint32_t pw = 4 * (tu + __umulhi(tu, m_units)) + (((uint32_t)lq >> 2) & 3u);
                const uint32_t wd = plane[pw < (uint32_t)kVPlane ? pw : 0u];
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(wd >> (8 * (lq & 3))), dst_r, (int)((uint32_t)lq < rem ? 16u * full + (uint32_t)lq : kV3Oob), 0, 0);
            }
        }
        MNW_STAMP(1)          // plane read, re-base, pack, stores issued
        lds_barrier();                                   // every lane holds what it needed of the planes
        MNW_STAMP(2)          // barrier
        if (abort_s) break;
        // ---- 2. the oldest waiting item -> the planes; the others move up (Q -> P -> W) -----------------------------------------------
        {
            const int ot_ = opaque_tid();
            const bool have_x = DEPTH == 5 ? have_w : DEPTH == 4 ? have_p : have_q;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const uint32_t e = 4 * (uint32_t)(ot_ + 256 * g);                          // element of the vtile
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    u32x4 v;
                    if (DEPTH == 5) {
                        u32x4* slot = reinterpret_cast<u32x4*>(&wait_s[3 * g + k][ot_][0]);
                        v = *slot;
                        const u32x4 pv = {P[k][4 * g], P[k][4 * g + 1], P[k][4 * g + 2], P[k][4 * g + 3]};
                        *slot = pv;
                    } else if (DEPTH == 4) v = u32x4{P[k][4 * g], P[k][4 * g + 1], P[k][4 * g + 2], P[k][4 * g + 3]};
                    else v = u32x4{Q[k][4 * g], Q[k][4 * g + 1], Q[k][4 * g + 2], Q[k][4 * g + 3]};
                    if (have_x) *reinterpret_cast<u32x4*>(lds + k * kVPlane + stage32_idx((int)e)) = v;
                }
            }
        }
        if (DEPTH >= 4) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) P[k][i] = Q[k][i];
        }
        MNW_STAMP(3)          // park
        // ---- 3. F -> Q: quantize, statistics ---------------------------------------------------------------------------------------
        const uint32_t nit2 = have_n ? next_item() : total;
        const uint32_t nsc2 = T == 1 ? nit2 : __umulhi(nit2, ra.m_T), nt2 = nit2 - nsc2 * T;
        const bool have_next = nit2 < total;
        if (have_n) {
            // component k of particle c of group g is word 3 c + k of the group's three vectors
            auto fget = [&](int g, int k, int c) -> float { const int i = 3 * c + k; return F[3 * g + (i >> 2)][i & 3]; };
            const bool partial = (nt + 1) * (uint32_t)kVTile > gm.n_sub3;                 // the sub-cell's last, short vtile (uniform)
            uint32_t m12[12];
            float nansum = 0.0f;
            bool redo[3] = {false, false, false};
            // (two instantiations: the whole-vtile one has no per-lane validity in it -- a select per element costs as much as the
            // statistics -- the other is the sub-cell's last vtile)
            auto quantize = [&](auto whole) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    int32_t mn = 0x7fffffff, mx = (int32_t)0x80000000;
                    uint32_t umn = 0xffffffffu, umx = 0u;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const bool valid = decltype(whole)::value || nt * (uint32_t)kVTile + 4 * (uint32_t)(opaque_tid() + 256 * g) < gm.n_sub3;
                        int32_t q4[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float rq = quant_mul(fget(g, k, c), qlow[k], qrdx[k]);
                            nansum = __fadd_rn(nansum, valid ? rq : 0.0f);
                            q4[c] = cvt_floor_i32(rq);
                            Q[k][4 * g + c] = valid ? (uint32_t)q4[c] : 0u;
                        }
                        if (valid) {
                            const uint32_t u0 = (uint32_t)q4[0] - qh[k], u1 = (uint32_t)q4[1] - qh[k], u2 = (uint32_t)q4[2] - qh[k], u3 = (uint32_t)q4[3] - qh[k];
                            mn = min(min(mn, min(q4[0], q4[1])), min(q4[2], q4[3]));
                            mx = max(max(mx, max(q4[0], q4[1])), max(q4[2], q4[3]));
                            umn = min(min(umn, min(u0, u1)), min(u2, u3));
                            umx = max(max(umx, max(u0, u1)), max(u2, u3));
                        }
                    }
                    m12[4 * k] = v3_enc_max(mx); m12[4 * k + 1] = v3_enc_min(mn); m12[4 * k + 2] = umx; m12[4 * k + 3] = ~umn;
                }
            };
            if (!partial) quantize(std::true_type{}); else quantize(std::false_type{});
            uint32_t o3[3];
            wave_max12_u32(m12, o3);
            // Two rare cases send a wave back to its floats (reloaded; IEEE division):
            //  * its least pixel index is <= 0: it may hold a quotient in the denormal range, where the float64 multiply can round
            //    differently from the division;
            //  * its largest pixel index is exactly `pixels`: a value that quantizes to the top of the ring (the data's maximum under
            //    limits taken from the data, go/minp/minp.go:87-96: high = Nextafter32(max) can give (max - low) / dx == pixels).  The
            //    reference's sequential periodicMin treats it as ring position 0 (periodicDistance wraps it), and bound() moves it
            //    there whenever the arc's start is 0; recorded as 0 -- in the statistics and in the parked item -- it gives the
            //    reference's (min, bits) and bytes in both cases: start a > 0 (narrow arc [a, pixels]: delta = 0 + pixels - a, the
            //    wrapped-arc formulas) and start 0 (bound() makes it 0).  Anything beyond `pixels` stays out of the domain.
            {
                // wave_max12_u32: lane 0 holds the values 0..2, lane 1 the values 6..8, lane 2 the values 3..5, lane 3 the values 9..11
                const int32_t mn0 = v3_dec_min((uint32_t)__shfl((int)o3[1], 0)), mn1 = v3_dec_min((uint32_t)__shfl((int)o3[2], 2)), mn2 = v3_dec_min((uint32_t)__shfl((int)o3[0], 3));
                const int32_t mx0 = v3_dec_max((uint32_t)__shfl((int)o3[0], 0)), mx1 = v3_dec_max((uint32_t)__shfl((int)o3[1], 2)), mx2 = v3_dec_max((uint32_t)__shfl((int)o3[2], 1));
                redo[0] = mn0 <= 0 || (uint32_t)mx0 == qpx[0]; redo[1] = mn1 <= 0 || (uint32_t)mx1 == qpx[1]; redo[2] = mn2 <= 0 || (uint32_t)mx2 == qpx[2];
            }
            if (__builtin_expect(redo[0] || redo[1] || redo[2], 0)) {
                // (the floats are gone -- they became Q --: fetch them again; everything here is waited for before it is left)
                const uint32_t b0 = origin(nsc);
                const float* gp = a.vec + 3 * (int64_t)b0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (!redo[k]) continue;
                    int32_t mn = 0x7fffffff, mx = (int32_t)0x80000000;
                    uint32_t umn = 0xffffffffu, umx = 0u;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const uint32_t off = group_off(nt * kVTile + 4 * (uint32_t)(opaque_tid() + 256 * g));
                        if (off != kV3Oob) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const float x = *(const MNW_GLOBAL_SC float*)(gp + (off >> 2) + 3 * c + k);
                                const float rq = __fdiv_rn(__fsub_rn(x, qlow[k]), qdx[k]);
                                int32_t q = cvt_floor_i32(rq);
                                if ((uint32_t)q == qpx[k]) q = 0;                         // the top of the ring is its position 0
                                Q[k][4 * g + c] = (uint32_t)q;
                                const uint32_t u = (uint32_t)q - qh[k];
                                mn = min(mn, q); mx = max(mx, q); umn = min(umn, u); umx = max(umx, u);
                            }
                        }
                    }
                    m12[4 * k] = v3_enc_max(mx); m12[4 * k + 1] = v3_enc_min(mn); m12[4 * k + 2] = umx; m12[4 * k + 3] = ~umn;
                }
                wave_max12_u32(m12, o3);
            }
            if (l < 4) {
                const int base = 6 * (l & 1) + 3 * (l >> 1);
#pragma unroll
                for (int i = 0; i < 3; ++i) red[base + i][wv] = o3[i];
            }
            const int bad = __any(__builtin_amdgcn_classf(nansum, 0x203));               // a NaN or an infinity among the quotients
            if (l == 0) redbad[wv] = bad;
        }
        MNW_STAMP(4)          // wait for the floats, quantize, statistics, wave reduction
        lds_barrier();                                   // the planes and the statistics are in LDS
        // ---- 5. wave 3 publishes the item's three records (lanes 0..11: component l / 4, word l % 4); a NaN flags all three blocks
        {
            uint32_t off = kV3Oob;
            u32x2 w64 = {0, 0};
            const int lp = opaque_tid() & 63;
            if (wv == 3 && have_n && lp < 12 && (int64_t)nit != ra.drop_item) {
                const int k = lp >> 2, w4 = lp & 3;
                const uint32_t v = max(max(red[lp][0], red[lp][1]), max(red[lp][2], red[lp][3]));
                const int bd = redbad[0] | redbad[1] | redbad[2] | redbad[3];
                w64.x = v; w64.y = gen | (bd ? kRecBad : 0u);
                off = (((uint32_t)k * s3 + nsc) * T + nt) * 32u + 8u * (uint32_t)w4;
            }
            __builtin_amdgcn_raw_buffer_store_b64(w64, rec_r, (int)off, 0, 16);           // sc1
            if (wv == 3 && have_n) { MNW_V3_TRACE(nit) }
        }
        // ---- 6. requests of the next trip: the descriptor of the sub-cell now parked (a sweeper publishes it once the sub-cell's last
        // record is out; a request that comes back empty costs polled round trips at the start of the next trip, so it is made as
        // late as possible), then the next item's floats
        issue_desc(DEPTH == 5 ? have_w : DEPTH == 4 ? have_p : have_q, DEPTH == 5 ? wsc : DEPTH == 4 ? psc : qsc);
        issue_floats(have_next, nsc2, nt2);
        MNW_STAMP(5)          // barrier, publish (and the sweep duty, once per sub-cell), requests
        if (DEPTH == 5) { have_o = have_w; osc = wsc; ot = wt; have_w = have_p; wsc = psc; wt = pt; have_p = have_q; psc = qsc; pt = qt; }
        else if (DEPTH == 4) { have_o = have_p; osc = psc; ot = pt; have_p = have_q; psc = qsc; pt = qt; }
        else { have_o = have_q; osc = qsc; ot = qt; }
        have_q = have_n; qsc = nsc; qt = nt;
        nit = nit2; nsc = nsc2; nt = nt2;
        have_n = have_next;
    }
    MNW_STAMP_FLUSH
    MNW_V3_COUNT_FLUSH
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
        const uint32_t z = dither_bits(j + (uint32_t)c, d.ka, d.kb);
        const float f = (float)__fma_rn((double)z, 0x1p-32, (double)(int32_t)q); // float32(float64(q) + u), u = z 2^-32 exactly
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


// The aligned 64-bit words that hold vtile t of a stream -> registers (lane i of the workgroup takes words i, i + 256, ...);
// words past the stream's end read as zero (they only feed elements beyond n, which are never stored).
template <int NW>
__device__ __forceinline__ void v3_load_words(u32x2 (&w)[NW], const StreamView& sv, uint32_t t, uint32_t b, int tid) {
    const int64_t w0 = (int64_t)t * (kVTile / 64) * b;
    int64_t left = sv.last + 1 - w0;                             // words of the stream from w0 on
    if (left < 0 || b == 0) left = 0;
    const int64_t nw = kVTile / 64 * (int64_t)b + 2;            // words that hold the vtile's bits (one look-ahead word, one for a stream that starts inside a word)
    if (left > nw) left = nw;                                    // (lanes past them are out of range: no request -- not the next vtile's words)
    const __amdgpu_buffer_rsrc_t r = v3_rsrc(sv.a + w0, (uint32_t)left * 8u);
#pragma unroll
    for (int k = 0; k < NW; ++k) w[k] = __builtin_amdgcn_raw_buffer_load_b64(r, (k * 256 + tid) * 8, 0, 0);
}

template <int NW, bool DB>
__global__ __launch_bounds__(256) void k_unpack_vec3(Vec3Args a, float* __restrict__ vec_out, const DevJob* __restrict__ jobs,
                                                     int periodic, float L, int tiles_per_wg, int xcd_order) {
    static_assert(kTileD == kVTile, "the stream helpers are written for 2,048-element tiles");
    constexpr int kLW = NW * 256;                                       // 64-bit words of one component's buffer
    extern __shared__ __attribute__((aligned(16))) uint64_t v3_lds[];    // [1 or 2 buffers][3 components][kLW], then the images
    const SubcellGeom& gm = a.gm;
    int64_t v0, v1;
    if (tiles_per_wg > 0) {
        // The five row segments that make up one row of the grid are written at about the same time AND through the same L2: a
        // sub-cell's rows are 1,200 of every 6,000 bytes of the grid's rows, their ends fall inside 64-byte granules that the
        // neighbouring sub-cell's rows complete, and a granule that leaves an L2 half written costs the memory side a
        // read-modify-write (tools/microbench/rows_order.hip, nothing but the stores of this kernel: 459 us sub-cell by sub-cell,
        // 451 us with the x-neighbours adjacent in dispatch order -- workgroup w runs on XCD w mod 8, so they were in five
        // different L2s --, 333-349 us with the x-neighbours on one XCD, 264 us for whole rows).  xcd_order: workgroup w is the
        // (w / 8)-th of XCD w mod 8; consecutive workgroups of an XCD take the sub-cells of one x-row (sx fastest).
        const uint32_t S = gm.sub_cells, tq_n = (a.tiles_per_sc + tiles_per_wg - 1) / tiles_per_wg;
        uint32_t sx, q;
        if (xcd_order) {
            const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
            sx = slot % S; q = (slot / S) * 8 + xcd;
            if (q >= tq_n * (a.s3 / S)) return;                        // (the grid is rounded up to whole groups of 8 S workgroups)
        } else { sx = blockIdx.x % S; q = blockIdx.x / S; }
        const uint32_t tq = q % tq_n, syz = q / tq_n;
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
                    __builtin_amdgcn_raw_buffer_store_b128(ch, grid_r, (int)(off + 16 * part), 0, 0);       // (kV3NoStore + 16 part: out of range)
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
    // (768 chunks of 16 bytes per workgroup and step; at most 16 workgroups per CU)
    hipLaunchKernelGGL(k_bounds3, dim3(subcell_grid((3 * n / 4 + 767) / 768 * 256 + 1, cus)), dim3(256), 0, s, vec, n, keys);
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
// single-read vectors encode.  recs (in units of ItemRec): 3 s3 T item records, then 3 s3 descriptor records of 16 bytes
int64_t vec3_reg_recs(int64_t n_file, int64_t sub_cells) {
    const SubcellGeom g = make_geom(n_file, sub_cells);
    const int64_t s3 = sub_cells * sub_cells * sub_cells, T = (g.n_sub3 + kVTile - 1) / kVTile;
    return 3 * s3 * T + (3 * s3 + 1) / 2;
}
static int g_v3_sweepers = 8;
int vec3_reg_sweepers() { return g_v3_sweepers; }                          // workgroups of the grid that hold no items
void vec3_reg_set_sweepers(int n) { g_v3_sweepers = n < 1 ? 1 : n > 64 ? 64 : n; }   // (tuning knob: option vec3_sweepers)
int64_t vec3_reg_max_tiles_per_block() { return 1ll << 20; }               // (the sweep takes any T; what bounds T is the resident grid)
// geometries k_encode_reg_vec3 takes: its multiply-high divisions need 8 <= n_sub < 512, n_sub a multiple of four, its grid
// offsets a sub-cell's span of the grid below 2^32 bytes, its record offsets fewer than 2^26 records
bool vec3_reg_geometry_ok(int64_t n_file, int64_t sub_cells) {
    if (sub_cells < 1 || n_file % sub_cells) return false;
    const int64_t n_sub = n_file / sub_cells;
    if (n_sub < 8 || n_sub >= 512 || n_sub % 4 || n_file % 4) return false;
    if (n_sub * n_file * n_file * 12 >= 0xfff00000ll) return false;
    return vec3_reg_recs(n_file, sub_cells) < (1ll << 26);
}
int encode_reg_vec3_blocks_per_cu(int depth) {
    if (depth == 3) { int m = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&m, k_encode_reg_vec3<3>, 256, 0) != hipSuccess) return 0; return m; }
    if (depth == 4) { int m = 0; if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&m, k_encode_reg_vec3<4>, 256, 0) != hipSuccess) return 0; return m; }
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_encode_reg_vec3<5>, 256, 0) != hipSuccess) return 0;
    return n;
}
void launch_encode_reg_vec3(hipStream_t s, int grid, const float* vec, int64_t n_file, int64_t sub_cells, const DevJob* jobs, void* recs,
                            uint32_t gen, uint32_t* abort_flag, uint32_t* abort_host, Meta* meta, int64_t drop_item, int weighted, int depth) {
    Vec3RegArgs ra;
    ra.a = make_vec3_args(vec, n_file, sub_cells);
    ra.recs = (ItemRec*)recs;
    ra.gen = gen; ra.abort_flag = abort_flag; ra.abort_host = abort_host; ra.drop_item = drop_item;
    ra.n_sweepers = (uint32_t)vec3_reg_sweepers();
    ra.weighted = ((grid - (int)ra.n_sweepers) % 3 == 0 && depth > 3 && weighted >= 0 && weighted <= kV3Tables) ? weighted : 0;     // (thirds: three workgroups per CU)
    ra.m_T = ra.a.tiles_per_sc > 1 ? (uint32_t)((1ull << 32) / ra.a.tiles_per_sc) + 1u : 0u;
    if (depth >= 5) hipLaunchKernelGGL(k_encode_reg_vec3<5>, dim3(grid), dim3(256), 0, s, ra, jobs, meta);
    else if (depth == 4) hipLaunchKernelGGL(k_encode_reg_vec3<4>, dim3(grid), dim3(256), 0, s, ra, jobs, meta);
    else hipLaunchKernelGGL(k_encode_reg_vec3<3>, dim3(grid), dim3(256), 0, s, ra, jobs, meta);
}
int64_t vec3_tiles(int64_t n_file, int64_t sub_cells) {
    const int64_t n_sub = n_file / sub_cells, n3 = n_sub * n_sub * n_sub;
    return sub_cells * sub_cells * sub_cells * ((n3 + kVTile - 1) / kVTile);
}
// max_bits: the widest block of the call (sizes the LDS word buffers and the prefetch registers); double_buffer: two word
// buffers and one barrier per vtile instead of one and two
// tiles_per_wg > 0: x-interleaved order, the grid is sub_cells^3 * ceil(T / tiles_per_wg) workgroups (`grid` is ignored)
void launch_unpack_vec3(hipStream_t s, int grid, float* vec_out, int64_t n_file, int64_t sub_cells, const DevJob* jobs, int periodic, float L,
                        int max_bits, bool double_buffer, int tiles_per_wg, int xcd_order) {
    const Vec3Args a = make_vec3_args(vec_out, n_file, sub_cells);
    if (tiles_per_wg > 0) {
        const int64_t q_n = (int64_t)(a.s3 / a.gm.sub_cells) * ((a.tiles_per_sc + tiles_per_wg - 1) / tiles_per_wg);     // pieces per sub-cell of an x-row
        grid = (int)(xcd_order ? (q_n + 7) / 8 * 8 * a.gm.sub_cells : q_n * a.gm.sub_cells);
    }
    if (max_bits < 1) max_bits = 1;
    const int words = kVTile / 64 * max_bits + 2;                 // 64-bit words of one component's vtile
    const int nw = words <= 2 * 256 ? 2 : words <= 3 * 256 ? 3 : 5;
    const size_t lds = (size_t)(double_buffer ? 2 : 1) * 3 * nw * 256 * sizeof(uint64_t) + (size_t)4 * kV3ImgDw * sizeof(uint32_t);
#define MNW_LAUNCH_UV3(NWV) do { if (double_buffer) hipLaunchKernelGGL((k_unpack_vec3<NWV, true>), dim3(grid), dim3(256), lds, s, a, vec_out, jobs, periodic, L, tiles_per_wg, xcd_order); \
                                 else hipLaunchKernelGGL((k_unpack_vec3<NWV, false>), dim3(grid), dim3(256), lds, s, a, vec_out, jobs, periodic, L, tiles_per_wg, xcd_order); } while (0)
    if (nw == 2) MNW_LAUNCH_UV3(2);
    else if (nw == 3) MNW_LAUNCH_UV3(3);
    else MNW_LAUNCH_UV3(5);
#undef MNW_LAUNCH_UV3
}
// the grid offsets of k_unpack_vec3 are 32-bit and relative to a sub-cell's first particle
bool vec3_unpack_geometry_ok(int64_t n_file, int64_t sub_cells) {
    return (n_file / sub_cells) * n_file * n_file * 12 < 0xfff00000ll;
}
