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
};

// the lane's two groups of four particles of vtile (sc, t): de-interleaved values, or `valid` false past the end
struct Vec3Group {
    sc_f32x4 x, y, z;
    bool valid;
};
__device__ __forceinline__ Vec3Group vec3_load(const Vec3Args& a, uint32_t sc, uint32_t j) {
    Vec3Group g;
    g.valid = j < a.gm.n_sub3;
    if (g.valid) {
        const uint32_t i = particle_of(a.gm, sc * a.gm.n_sub3 + j);
        const MNW_GLOBAL_SC sc_f32x4* p = (const MNW_GLOBAL_SC sc_f32x4*)(a.vec + 3 * (int64_t)i);
        const sc_f32x4 u = p[0], v = p[1], w = p[2];
        g.x = sc_f32x4{u.x, u.w, v.z, w.y}; g.y = sc_f32x4{u.y, v.x, v.w, w.z}; g.z = sc_f32x4{u.z, v.y, w.x, w.w};
    } else {
        g.x = g.y = g.z = sc_f32x4{0, 0, 0, 0};
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
                const float v0[3] = {g.x[c], g.y[c], g.z[c]}, v1[3] = {g.x[c + 1], g.y[c + 1], g.z[c + 1]};
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
                    const float xv[3] = {g.x[c], g.y[c], g.z[c]};
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
// minp.Reader.Vectors without the staging pass: the three streams of a sub-cell are unpacked vtile by vtile
// and written straight into the AoS grid (dequantize, group.go:299-310; periodic fix-up, minp.go:188-196;
// setSubCell).  32-bit streams only (bits <= 32, ring and minimum in int32: the NARROW class of
// k_unpack_f32); anything else goes through the staging path.  HBM traffic: the streams + 12 B/particle
// written, against streams + 12 B written + 12 B read + 12 B written of the staging path.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_unpack_vec3(Vec3Args a, float* __restrict__ vec_out, const DevJob* __restrict__ jobs,
                                                     int periodic, float L) {
    constexpr int kLW = kTileD / 64 * 32 + 2;                     // 64-bit words of one component's vtile
    constexpr int kNW = (kLW + 255) / 256;
    static_assert(kTileD == kVTile, "the stream helpers are written for 2,048-element tiles");
    __shared__ __attribute__((aligned(16))) uint64_t lw[3][kLW];
    const int tid = threadIdx.x;
    const int64_t total = (int64_t)a.s3 * a.tiles_per_sc;
    int64_t v0 = total * blockIdx.x / gridDim.x;
    const int64_t v1 = total * (blockIdx.x + 1) / gridDim.x;
    while (v0 < v1) {
        const uint32_t sc = (uint32_t)(v0 / a.tiles_per_sc);
        int64_t vend = (int64_t)(sc + 1) * a.tiles_per_sc;
        if (vend > v1) vend = v1;
        uint32_t b[3], mask[3], ka[3], kb[3];
        int32_t mn[3], pix[3];
        float low[3], dx[3];
        StreamView sv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const DevJob jb = jobs[(int64_t)k * a.s3 + sc];
            b[k] = (uint32_t)jb.bits;
            mask[k] = b[k] == 0 ? 0u : (b[k] == 32 ? ~0u : ((1u << b[k]) - 1u));
            mn[k] = (int32_t)jb.min; pix[k] = (int32_t)jb.pixels;
            low[k] = jb.low; dx[k] = jb.dxq;
            sv[k] = stream_view(jb, b[k]);
            const uint64_t key = dither_key(jb.seed, jb.block, 0);   // element indices stay below 2^32
            ka[k] = (uint32_t)key; kb[k] = (uint32_t)(key >> 32);
        }
        // the words of the next vtile are loaded into registers while the current one is extracted
        uint64_t w[3][kNW];
        {
            const uint32_t t = (uint32_t)(v0 - (int64_t)sc * a.tiles_per_sc);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (b[k] != 0) load_tile_words<kNW>(w[k], sv[k], (int64_t)t * kVTile, b[k]);
        }
        for (int64_t v = v0; v < vend; ++v) {
            const uint32_t t = (uint32_t)(v - (int64_t)sc * a.tiles_per_sc);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (b[k] != 0) {
                    stage_tile_words<kNW>(lw[k], w[k], b[k]);
                    if (v + 1 < vend) load_tile_words<kNW>(w[k], sv[k], (int64_t)(t + 1) * kVTile, b[k]);
                }
            }
            lds_barrier();
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint32_t e = 4 * (tid + 256 * r);
                const uint32_t j = t * kVTile + e;
                if (j < a.gm.n_sub3) {
                    sc_f32x4 o[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const uint32_t val = b[k] == 0 ? 0u : extract32(reinterpret_cast<const uint32_t*>(lw[k]), e + c, b[k], mask[k]);
                            int32_t q = mn[k] + (int32_t)val;                 // group.go:262
                            if (q < 0) q += pix[k];                           // bound(buf, 0, pixels): FloatGroups are periodic
                            else if (q >= pix[k]) q -= pix[k];
                            const uint32_t z = dither_bits(j + c, ka[k], kb[k]);
                            const double u = (double)z * (1.0 / 4294967296.0);
                            const float x = __fadd_rn(__fmul_rn(dx[k], (float)__dadd_rn((double)q, u)), low[k]);   // group.go:308
                            o[k][c] = wrap_once(x, L, periodic);
                        }
                    }
                    const uint32_t i = particle_of(a.gm, sc * a.gm.n_sub3 + j);
                    MNW_GLOBAL_SC sc_f32x4* p = (MNW_GLOBAL_SC sc_f32x4*)(vec_out + 3 * (int64_t)i);
                    const sc_f32x4 q0 = {o[0].x, o[1].x, o[2].x, o[0].y}, q1 = {o[1].y, o[2].y, o[0].z, o[1].z},
                                   q2 = {o[2].z, o[0].w, o[1].w, o[2].w};
                    p[0] = q0; p[1] = q1; p[2] = q2;
                }
            }
            lds_barrier();
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

void launch_reduce_vec3(hipStream_t s, int grid, const float* vec, int64_t n_file, int64_t sub_cells, const DevJob* jobs, uint64_t* stats) {
    Vec3Args a;
    a.vec = vec; a.gm = make_geom(n_file, sub_cells);
    a.s3 = (uint32_t)(sub_cells * sub_cells * sub_cells);
    a.tiles_per_sc = (a.gm.n_sub3 + kVTile - 1) / kVTile;
    hipLaunchKernelGGL(k_reduce_vec3, dim3(grid), dim3(256), 0, s, a, jobs, stats);
}
void launch_pack_vec3(hipStream_t s, int grid, const float* vec, int64_t n_file, int64_t sub_cells, const DevJob* jobs, const uint64_t* stats,
                      Meta* meta) {
    Vec3Args a;
    a.vec = vec; a.gm = make_geom(n_file, sub_cells);
    a.s3 = (uint32_t)(sub_cells * sub_cells * sub_cells);
    a.tiles_per_sc = (a.gm.n_sub3 + kVTile - 1) / kVTile;
    hipLaunchKernelGGL(k_pack_vec3, dim3(grid), dim3(256), 0, s, a, jobs, stats, meta);
}
int64_t vec3_tiles(int64_t n_file, int64_t sub_cells) {
    const int64_t n_sub = n_file / sub_cells, n3 = n_sub * n_sub * n_sub;
    return sub_cells * sub_cells * sub_cells * ((n3 + kVTile - 1) / kVTile);
}
void launch_unpack_vec3(hipStream_t s, int grid, float* vec_out, int64_t n_file, int64_t sub_cells, const DevJob* jobs, int periodic, float L) {
    Vec3Args a;
    a.vec = vec_out; a.gm = make_geom(n_file, sub_cells);
    a.s3 = (uint32_t)(sub_cells * sub_cells * sub_cells);
    a.tiles_per_sc = (a.gm.n_sub3 + kVTile - 1) / kVTile;
    hipLaunchKernelGGL(k_unpack_vec3, dim3(grid), dim3(256), 0, s, a, vec_out, jobs, periodic, L);
}
