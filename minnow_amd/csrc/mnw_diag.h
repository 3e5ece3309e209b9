// mnw_diag.h -- everything the DIAGNOSTIC builds of the kernels need (tools/microbench/*_stamps.hip define MNW_STAMPS before they
// include mnw_kernels.hip): per-phase cycle stamps, counters and the device globals they write to.  In the library build
// (minnow_amd/build.py defines none of the MNW_STAMPS* macros) every macro below expands to nothing and no global exists: the
// shipped kernels carry the one-word markers only.
#pragma once

#ifdef MNW_STAMPS
// per-phase cycle sums of wave 0 of every workgroup, written to a buffer no other code reads
__device__ unsigned long long* g_stamp_abs = nullptr;
__device__ unsigned long long* g_stamp_buf = nullptr;
#define MNW_STAMP_DECL unsigned long long st_t = 0, st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long st_rt0 = __builtin_amdgcn_s_memrealtime(), st_c0 = __builtin_amdgcn_s_memtime();
#define MNW_STAMP_START { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#ifdef MNW_STAMPS_LIGHT     // start and end of every workgroup only: the kernel as shipped but for two clock reads
#define MNW_STAMP(i)
#else
#define MNW_STAMP(i) { unsigned long long t2_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2_) :: "memory"); __builtin_amdgcn_sched_barrier(0); st_acc[i] += t2_ - st_t; st_t = t2_; }
#endif
#define MNW_STAMP_FLUSH if (g_stamp_buf && threadIdx.x == 0) { const unsigned long long e_ = __builtin_amdgcn_s_memrealtime(); st_acc[7] = e_ - st_rt0; st_acc[6] = __builtin_amdgcn_s_memtime() - st_c0; for (int i_ = 0; i_ < 8; ++i_) g_stamp_buf[blockIdx.x * 8 + i_] += st_acc[i_]; if (g_stamp_abs) { g_stamp_abs[blockIdx.x * 2] = st_rt0; g_stamp_abs[blockIdx.x * 2 + 1] = e_; } }
// k_encode_reg_f32<., BIG>: calls, group polls, block polls, cycles: all / group phase / block phase
__device__ unsigned long long g_big_diag[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define MNW_BIG_DIAG(i, v) if ((threadIdx.x & 63) == 0) atomicAdd(&g_big_diag[i], (unsigned long long)(v))
// k_encode_reg_f32: sweeps that found a record missing
__device__ unsigned long long g_slow_sweeps = 0;
#define MNW_DIAG_SLOW_SWEEP if ((threadIdx.x & 63) == 0) atomicAdd(&g_slow_sweeps, 1ull);
// k_encode_reg_vec3: how often a descriptor is late, per workgroup (wave 0)
__device__ unsigned long long* g_v3_diag = nullptr;
#define MNW_V3_COUNT_DECL unsigned long long v3c_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define MNW_V3_COUNT(i) { v3c_[i] += 1; }
#define MNW_V3_COUNT_FLUSH if (g_v3_diag && threadIdx.x == 0) { for (int i_ = 0; i_ < 8; ++i_) g_v3_diag[blockIdx.x * 8 + i_] += v3c_[i_]; }
// k_encode_reg_vec3 time line (100 MHz clock): [0 .. items): when an item's records were published; [items .. 2 items): when its
// descriptor was first looked for; [2 items .. 3 items): when it was there; then per sub-cell: sentinel seen, descriptors published
__device__ unsigned long long* g_v3_trace = nullptr;
#define MNW_V3_TRACE(idx) if (g_v3_trace && (threadIdx.x & 63) == 0) g_v3_trace[idx] = __builtin_amdgcn_s_memrealtime();
#else
#define MNW_V3_TRACE(idx)
#define MNW_STAMP_DECL
#define MNW_STAMP_START
#define MNW_STAMP(i)
#define MNW_STAMP_FLUSH
#define MNW_BIG_DIAG(i, v)
#define MNW_DIAG_SLOW_SWEEP
#define MNW_V3_COUNT_DECL
#define MNW_V3_COUNT(i)
#define MNW_V3_COUNT_FLUSH
#endif
