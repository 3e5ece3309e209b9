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
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mnw {

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

}  // namespace mnw
