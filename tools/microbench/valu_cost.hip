// Micro-benchmark (not product code): issue cost of the vector instructions the codec kernels are made of, on gfx950.
// One kernel per instruction: every wave runs kIter x 32 independent copies (8 destination registers in rotation) between two
// s_memtime reads; run with 1, 2, 4 and 8 waves per SIMD (workgroups of 256 = one wave per SIMD each).  Reported: cycles per
// wave-instruction seen by one wave (its own issue cost when alone) and SIMD cycles per wave-instruction at 8 waves (throughput).
// Also prints the in-kernel clock (s_memtime against the 100 MHz s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int kIter = 2048;

#define REP8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define REP32(I) REP8(I) REP8(I) REP8(I) REP8(I)

// DT = type of the eight destinations, ST = type of the two sources
#define DEF_K(NAME, TEXT, DT, ST, AINIT, BINIT)                                                                                 \
    __global__ __launch_bounds__(256) void k_##NAME(unsigned long long* out, unsigned* sink) {                                   \
        DT d0 = (DT)threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;      \
        ST a = AINIT, b = BINIT;                                                                                                 \
        asm volatile("" : "+v"(a), "+v"(b));                                                                                     \
        const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();                                                          \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                              \
        for (int it = 0; it < kIter; ++it) {                                                                                     \
            asm volatile(TEXT(0) TEXT(1) TEXT(2) TEXT(3) TEXT(4) TEXT(5) TEXT(6) TEXT(7)                                         \
                         TEXT(0) TEXT(1) TEXT(2) TEXT(3) TEXT(4) TEXT(5) TEXT(6) TEXT(7)                                         \
                         TEXT(0) TEXT(1) TEXT(2) TEXT(3) TEXT(4) TEXT(5) TEXT(6) TEXT(7)                                         \
                         TEXT(0) TEXT(1) TEXT(2) TEXT(3) TEXT(4) TEXT(5) TEXT(6) TEXT(7)                                         \
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)                        \
                         : "v"(a), "v"(b) : "vcc", "s10", "s11");                                                                              \
        }                                                                                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                              \
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();                                                          \
        if ((threadIdx.x & 63) == 0) { out[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0; out[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0; } \
        if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 == (DT)12345) sink[0] = 1;                                                    \
    }
#define U32A (threadIdx.x * 2654435761u + 12345u)
#define U32B ((threadIdx.x ^ 0x5bd1e995u) | 1u)
#define F64A (1.0 + threadIdx.x * 1e-3)
#define F64B (0.999 + threadIdx.x * 1e-6)
#define DEF_K32(NAME, TEXT) DEF_K(NAME, TEXT, unsigned, unsigned, U32A, U32B)
#define DEF_K64(NAME, TEXT) DEF_K(NAME, TEXT, double, double, F64A, F64B)
#define DEF_K64_32(NAME, TEXT) DEF_K(NAME, TEXT, double, unsigned, U32A, U32B)
#define DEF_K32_64(NAME, TEXT) DEF_K(NAME, TEXT, unsigned, double, F64A, F64B)

// mixed: 64-bit destination from 32-bit sources, or 32-bit destination (low half of the pair) from 64-bit sources
#define T_add_f32(i) "v_add_f32 %" #i ", %8, %9\n"
#define T_fma_f32(i) "v_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define T_mul_lo_u32(i) "v_mul_lo_u32 %" #i ", %8, %9\n"
#define T_mul_hi_u32(i) "v_mul_hi_u32 %" #i ", %8, %9\n"
#define T_mul_u32_u24(i) "v_mul_u32_u24 %" #i ", %8, %9\n"
#define T_mad_u32_u24(i) "v_mad_u32_u24 %" #i ", %8, %9, %" #i "\n"
#define T_xor(i) "v_xor_b32 %" #i ", %8, %9\n"
#define T_lshl_or(i) "v_lshl_or_b32 %" #i ", %8, 5, %9\n"
#define T_alignbit(i) "v_alignbit_b32 %" #i ", %8, %9, %" #i "\n"
#define T_bfe(i) "v_bfe_u32 %" #i ", %8, 5, 17\n"
#define T_perm(i) "v_perm_b32 %" #i ", %8, %9, %" #i "\n"
#define T_cvt_f32_u32(i) "v_cvt_f32_u32 %" #i ", %8\n"
#define T_cvt_f32_i32(i) "v_cvt_f32_i32 %" #i ", %8\n"
#define T_cvt_flr(i) "v_cvt_flr_i32_f32 %" #i ", %8\n"
#define T_min3_f32(i) "v_min3_f32 %" #i ", %8, %9, %" #i "\n"
#define T_max3_i32(i) "v_max3_i32 %" #i ", %8, %9, %" #i "\n"
#define T_min_u32(i) "v_min_u32 %" #i ", %8, %9\n"
#define T_cndmask(i) "v_cndmask_b32 %" #i ", %8, %9, vcc\n"
#define T_cmp_cnd(i) "v_cmp_lt_f32 vcc, %8, %" #i "\nv_cndmask_b32 %" #i ", %8, %9, vcc\n"
#define T_cmp_class(i) "v_cmp_class_f32 vcc, %" #i ", %9\n"
#define T_ldexp_f32(i) "v_ldexp_f32 %" #i ", %8, %9\n"
#define T_rcp_f32(i) "v_rcp_f32 %" #i ", %8\n"
#define T_ffbh(i) "v_ffbh_u32 %" #i ", %8\n"
#define T_add_u32(i) "v_add_u32 %" #i ", %8, %9\n"
#define T_add3_u32(i) "v_add3_u32 %" #i ", %8, %9, %" #i "\n"
#define T_xad(i) "v_xad_u32 %" #i ", %8, %9, %" #i "\n"
#define T_and_or(i) "v_and_or_b32 %" #i ", %8, %9, %" #i "\n"
#define T_mov_dpp(i) "v_mov_b32_dpp %" #i ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define T_sub_f32(i) "v_sub_f32 %" #i ", %8, %9\n"
#define T_mul_f32(i) "v_mul_f32 %" #i ", %8, %9\n"

#define T_and(i) "v_and_b32 %" #i ", %8, %9\n"
#define T_or(i) "v_or_b32 %" #i ", %8, %9\n"
#define T_lshlrev(i) "v_lshlrev_b32 %" #i ", 5, %8\n"
#define T_lshrrev(i) "v_lshrrev_b32 %" #i ", %9, %8\n"
#define T_ashrrev(i) "v_ashrrev_i32 %" #i ", 7, %8\n"
#define T_max_f32(i) "v_max_f32 %" #i ", %8, %9\n"
#define T_min_f32(i) "v_min_f32 %" #i ", %8, %9\n"
#define T_sub_u32(i) "v_sub_u32 %" #i ", %8, %9\n"
#define T_max_u32(i) "v_max_u32 %" #i ", %8, %9\n"
#define T_max_i32(i) "v_max_i32 %" #i ", %8, %9\n"
#define T_mov(i) "v_mov_b32 %" #i ", %8\n"
#define T_cmp_f32(i) "v_cmp_lt_f32 vcc, %8, %" #i "\n"
#define T_cmp_u32(i) "v_cmp_lt_u32 vcc, %8, %" #i "\n"
#define T_cnd_s(i) "v_cndmask_b32 %" #i ", %8, %9, s[10:11]\n"
#define T_med3_f32(i) "v_med3_f32 %" #i ", %8, %9, %" #i "\n"
#define T_bfi(i) "v_bfi_b32 %" #i ", %8, %9, %" #i "\n"
#define T_lshl_add(i) "v_lshl_add_u32 %" #i ", %8, 3, %9\n"
#define T_add_lshl(i) "v_add_lshl_u32 %" #i ", %8, %9, 3\n"
#define T_or3(i) "v_or3_b32 %" #i ", %8, %9, %" #i "\n"
#define T_fmac(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define T_floor_f32(i) "v_floor_f32 %" #i ", %8\n"
#define T_rndne_f32(i) "v_rndne_f32 %" #i ", %8\n"
#define T_mad_i32_i24(i) "v_mad_i32_i24 %" #i ", %8, %9, %" #i "\n"
#define T_mul_f32_sdwa(i) "v_mul_f32 %" #i ", 2.0, %8\n"
#define T_max3_f32(i) "v_max3_f32 %" #i ", %8, %9, %" #i "\n"
#define T_sub_co(i) "v_sub_co_u32 %" #i ", vcc, %8, %9\n"
#define T_cvt_u32_f32(i) "v_cvt_u32_f32 %" #i ", %8\n"
#define T_readlane(i) "v_readfirstlane_b32 s10, %8\n"
#define T_add_f32_abs(i) "v_add_f32 %" #i ", |%8|, -%9\n"
#define T_lshlrev_v(i) "v_lshlrev_b32 %" #i ", %9, %8\n"

#define T_add_f64(i) "v_add_f64 %" #i ", %8, %9\n"
#define T_mul_f64(i) "v_mul_f64 %" #i ", %8, %9\n"
#define T_fma_f64(i) "v_fma_f64 %" #i ", %8, %9, %" #i "\n"
#define T_lshr_b64(i) "v_lshrrev_b64 %" #i ", 7, %8\n"
#define T_pk_add_f32(i) "v_pk_add_f32 %" #i ", %8, %9\n"
#define T_pk_mul_f32(i) "v_pk_mul_f32 %" #i ", %8, %9\n"
#define T_pk_fma_f32(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define T_cvt_f64_u32(i) "v_cvt_f64_u32 %" #i ", %8\n"
#define T_cvt_f64_i32(i) "v_cvt_f64_i32 %" #i ", %8\n"
#define T_cvt_f64_f32(i) "v_cvt_f64_f32 %" #i ", %8\n"
#define T_cvt_f32_f64(i) "v_cvt_f32_f64 %" #i ", %8\n"
#define T_cvt_i32_f64(i) "v_cvt_i32_f64 %" #i ", %8\n"
#define T_mad_u64_u32(i) "v_mad_u64_u32 %" #i ", vcc, %8, %9, %" #i "\n"
#define T_floor_f64(i) "v_floor_f64 %" #i ", %8\n"
#define T_rcp_f64(i) "v_rcp_f64 %" #i ", %8\n"

DEF_K32(add_f32, T_add_f32) DEF_K32(fma_f32, T_fma_f32) DEF_K32(mul_lo_u32, T_mul_lo_u32) DEF_K32(mul_hi_u32, T_mul_hi_u32)
DEF_K32(mul_u32_u24, T_mul_u32_u24) DEF_K32(mad_u32_u24, T_mad_u32_u24) DEF_K32(xor_b32, T_xor) DEF_K32(lshl_or, T_lshl_or)
DEF_K32(alignbit, T_alignbit) DEF_K32(bfe, T_bfe) DEF_K32(perm, T_perm) DEF_K32(cvt_f32_u32, T_cvt_f32_u32) DEF_K32(cvt_f32_i32, T_cvt_f32_i32)
DEF_K32(cvt_flr, T_cvt_flr) DEF_K32(min3_f32, T_min3_f32) DEF_K32(max3_i32, T_max3_i32) DEF_K32(min_u32, T_min_u32) DEF_K32(cndmask, T_cndmask)
DEF_K32(cmp_cnd, T_cmp_cnd) DEF_K32(cmp_class, T_cmp_class) DEF_K32(ldexp_f32, T_ldexp_f32) DEF_K32(rcp_f32, T_rcp_f32) DEF_K32(ffbh, T_ffbh)
DEF_K32(add_u32, T_add_u32) DEF_K32(add3_u32, T_add3_u32) DEF_K32(xad, T_xad) DEF_K32(and_or, T_and_or) DEF_K32(mov_dpp, T_mov_dpp)
DEF_K32(sub_f32, T_sub_f32) DEF_K32(mul_f32, T_mul_f32)
DEF_K32(and, T_and) DEF_K32(or, T_or) DEF_K32(lshlrev, T_lshlrev) DEF_K32(lshrrev, T_lshrrev) DEF_K32(ashrrev, T_ashrrev) DEF_K32(max_f32, T_max_f32) DEF_K32(min_f32, T_min_f32) DEF_K32(sub_u32, T_sub_u32) DEF_K32(max_u32, T_max_u32) DEF_K32(max_i32, T_max_i32) DEF_K32(mov, T_mov) DEF_K32(cmp_f32, T_cmp_f32) DEF_K32(cmp_u32, T_cmp_u32) DEF_K32(cnd_s, T_cnd_s) DEF_K32(med3_f32, T_med3_f32) DEF_K32(bfi, T_bfi) DEF_K32(lshl_add, T_lshl_add) DEF_K32(add_lshl, T_add_lshl) DEF_K32(or3, T_or3) DEF_K32(fmac, T_fmac) DEF_K32(floor_f32, T_floor_f32) DEF_K32(rndne_f32, T_rndne_f32) DEF_K32(mad_i32_i24, T_mad_i32_i24) DEF_K32(mul_f32_sdwa, T_mul_f32_sdwa) DEF_K32(max3_f32, T_max3_f32) DEF_K32(sub_co, T_sub_co) DEF_K32(cvt_u32_f32, T_cvt_u32_f32) DEF_K32(add_f32_abs, T_add_f32_abs) DEF_K32(lshlrev_v, T_lshlrev_v)
DEF_K64(add_f64, T_add_f64) DEF_K64(mul_f64, T_mul_f64) DEF_K64(fma_f64, T_fma_f64) DEF_K64(lshr_b64, T_lshr_b64)
DEF_K64(pk_add_f32, T_pk_add_f32) DEF_K64(pk_mul_f32, T_pk_mul_f32) DEF_K64(pk_fma_f32, T_pk_fma_f32)
DEF_K64_32(cvt_f64_u32, T_cvt_f64_u32) DEF_K64_32(cvt_f64_i32, T_cvt_f64_i32) DEF_K64_32(cvt_f64_f32, T_cvt_f64_f32) DEF_K32_64(cvt_f32_f64, T_cvt_f32_f64)
DEF_K32_64(cvt_i32_f64, T_cvt_i32_f64) DEF_K64_32(mad_u64_u32, T_mad_u64_u32) DEF_K64(floor_f64, T_floor_f64)
DEF_K64(rcp_f64, T_rcp_f64)

typedef void (*kern_t)(unsigned long long*, unsigned*);
struct Entry { const char* name; kern_t k; int per_text; };
#define E(NAME) {#NAME, k_##NAME, 1}

int main(int argc, char** argv) {
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, pci bus %d\n", prop.gcnArchName, cus, prop.pciBusID);
    unsigned long long* out; unsigned* sink;
    const int max_wg = cus * 8;
    CHK(hipMalloc(&out, max_wg * 4 * 2 * sizeof(unsigned long long))); CHK(hipMalloc(&sink, 4)); CHK(hipMemset(sink, 0, 4));
    std::vector<Entry> es = {E(add_f32), E(fma_f32), E(sub_f32), E(mul_f32), E(mul_lo_u32), E(mul_hi_u32), E(mul_u32_u24), E(mad_u32_u24), E(xor_b32), E(lshl_or), E(alignbit), E(bfe), E(perm),
                             E(cvt_f32_u32), E(cvt_f32_i32), E(cvt_flr), E(min3_f32), E(max3_i32), E(min_u32), E(cndmask), {"cmp+cndmask (pair)", k_cmp_cnd, 2}, E(cmp_class), E(ldexp_f32), E(rcp_f32),
                             E(ffbh), E(add_u32), E(add3_u32), E(xad), E(and_or), E(mov_dpp),
                             E(and), E(or), E(lshlrev), E(lshrrev), E(ashrrev), E(max_f32), E(min_f32), E(sub_u32), E(max_u32), E(max_i32), E(mov), E(cmp_f32), E(cmp_u32), E(cnd_s), E(med3_f32), E(bfi), E(lshl_add), E(add_lshl), E(or3), E(fmac), E(floor_f32), E(rndne_f32), E(mad_i32_i24), E(mul_f32_sdwa), E(max3_f32), E(sub_co), E(cvt_u32_f32), E(add_f32_abs), E(lshlrev_v),
                             E(add_f64), E(mul_f64), E(fma_f64), E(lshr_b64), E(pk_add_f32), E(pk_mul_f32), E(pk_fma_f32), E(cvt_f64_u32), E(cvt_f64_i32), E(cvt_f64_f32),
                             E(cvt_f32_f64), E(cvt_i32_f64), E(mad_u64_u32), E(floor_f64), E(rcp_f64)};
    printf("%-22s %10s %10s %10s %10s   (cycles per wave-instruction as one wave sees it; last column: SIMD cycles per wave-instruction by the WALL CLOCK of the 8-wave launch)\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");
    std::vector<unsigned long long> h(max_wg * 8);
    hipEvent_t ev0, ev1; CHK(hipEventCreate(&ev0)); CHK(hipEventCreate(&ev1));
    double clk_sum = 0; int clk_n = 0;
    for (auto& e : es) {
        printf("%-22s", e.name);
        double last = 0;
        for (int wps : {1, 2, 4, 8}) {
            const int grid = cus * wps;
            hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, out, sink);        // warm-up
            CHK(hipEventRecord(ev0));
            hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, out, sink);
            CHK(hipEventRecord(ev1));
            CHK(hipDeviceSynchronize());
            float ms = 0; CHK(hipEventElapsedTime(&ms, ev0, ev1));
            CHK(hipMemcpy(h.data(), out, grid * 4 * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            std::vector<double> cyc, clk;
            for (int i = 0; i < grid * 4; ++i) { cyc.push_back((double)h[2 * i]); if (h[2 * i + 1] > 0) clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1); }
            std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
            const double med = cyc[cyc.size() / 2] / (kIter * 32.0 * e.per_text);
            printf(" %10.2f", med);
            // wall clock: all wave-instructions of the launch over all SIMDs, in cycles of the measured in-kernel clock
            const double ghz = clk.empty() ? 2.3 : clk[clk.size() / 2];
            last = ms * 1e-3 * ghz * 1e9 / ((double)grid * 4 * kIter * 32.0 * e.per_text / (cus * 4.0));
            if (wps == 8 && !clk.empty()) { clk_sum += clk[clk.size() / 2]; ++clk_n; }
        }
        printf("   -> %.2f\n", last);
    }
    printf("in-kernel clock (median over waves, mean over the 8-wave runs): %.3f GHz\n", clk_sum / (clk_n ? clk_n : 1));
    return 0;
}
