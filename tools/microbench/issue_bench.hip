// issue_bench.hip -- SIMD issue cost of the wave64 instructions the compositing kernels are made of, on MI355X.
// Every kernel runs ONE instruction form on 8 independent registers per lane (no dependent chains), 256-thread
// blocks, 4 waves per SIMD resident (the compositing kernels' occupancy), long enough that launch overhead vanishes.
// Prints real-time nanoseconds per wave-instruction per SIMD and the same in cycles at the nominal 2.4 GHz.
// build: hipcc --offload-arch=gfx950 -O3 issue_bench.hip -o issue_bench ; run: gpurun -- ./tools/microbench/issue_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 3000
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL(NAME, ASMSTR)                                                                          \
  __global__ void __launch_bounds__(256, 4) k_##NAME(float* out, float a, float b, int iters) {      \
    float x[8];                                                                                       \
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;                                      \
    unsigned lds_addr = (threadIdx.x * 16) & 0x3ff0;                                                  \
    (void)lds_addr;                                                                                   \
    for (int it = 0; it < iters; ++it) {                                                              \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASMSTR : "+v"(x[i]) : "v"(a), "v"(b)); \
    }                                                                                                 \
    float s = 0;                                                                                      \
    for (int i = 0; i < 8; ++i) s += x[i];                                                            \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                                          \
  }

KERNEL(fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL(fmac, "v_fmac_f32 %0, %1, %2")
KERNEL(fmac_dpp_bcast, "v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf")
KERNEL(fma_abs, "v_fma_f32 %0, |%0|, |%1|, %2")
KERNEL(mul, "v_mul_f32 %0, %0, %1")
KERNEL(add, "v_add_f32 %0, %0, %1")
KERNEL(sub_dpp, "v_subrev_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf")
KERNEL(min_f32, "v_min_f32 %0, %0, %1")
KERNEL(mov, "v_mov_b32 %0, %1")
KERNEL(mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
KERNEL(lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
KERNEL(ffbh, "v_ffbh_u32 %0, %0")
KERNEL(bfe, "v_bfe_u32 %0, %0, 0, %1")
KERNEL(min_u32, "v_min_u32 %0, %0, %1")
KERNEL(xor_b32, "v_xor_b32 %0, %0, %1")
KERNEL(lshlrev, "v_lshlrev_b32 %0, 2, %0")
KERNEL(add_u32, "v_add_u32 %0, %0, %1")
KERNEL(sub_u32, "v_sub_u32 %0, %0, %1")
KERNEL(and_b32, "v_and_b32 %0, %0, %1")
KERNEL(or_b32, "v_or_b32 %0, %0, %1")
KERNEL(and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL(lshl_or, "v_lshl_or_b32 %0, %0, 3, %1")
KERNEL(max_f32, "v_max_f32 %0, %0, %1")
KERNEL(med3_f32, "v_med3_f32 %0, %0, %1, %2")
KERNEL(mul_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL(mul_e64_neg, "v_mul_f32_e64 %0, %0, -%1")
KERNEL(sub_f32, "v_sub_f32 %0, %0, %1")
KERNEL(fma_sgpr_lit, "v_fmamk_f32 %0, %0, 0x40490fdb, %1")
KERNEL(add_dpp_quad, "v_add_f32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(add_dpp_rowshr, "v_add_f32_dpp %0, %1, %0 row_shr:4 row_mask:0xf bank_mask:0xf")
KERNEL(bcnt, "v_bcnt_u32_b32 %0, %0, %1")
KERNEL(exp, "v_exp_f32 %0, %0")
KERNEL(rcp, "v_rcp_f32 %0, %0")
KERNEL(cmp_vcc, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL(cndmask_vcc, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL(cmp_cndmask, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
KERNEL(mov_dpp_quad, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(permlane32_swap, "v_permlane32_swap_b32 %0, %1")
KERNEL(readlane_like_bpermute, "ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)")
KERNEL(swizzle, "ds_swizzle_b32 %0, %0 offset:0x401f\n s_waitcnt lgkmcnt(0)")

// packed-f32 forms (VOP3P): operands are 64-bit register pairs, 2 FMAs per lane and instruction
#define PK_KERNEL(NAME, ASMSTR)                                                                       \
  __global__ void __launch_bounds__(256, 4) k_##NAME(float* out, float a, float b, int iters) {      \
    v2f x[8];                                                                                         \
    for (int i = 0; i < 8; ++i) { x[i].x = threadIdx.x * 0.001f + i; x[i].y = x[i].x + 0.5f; }        \
    const v2f a2 = {a, b}, b2 = {b, a};                                                               \
    for (int it = 0; it < iters; ++it) {                                                              \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASMSTR : "+v"(x[i]) : "v"(a2), "v"(b2)); \
    }                                                                                                 \
    float s = 0;                                                                                      \
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;                                                 \
    out[blockIdx.x * 256 + threadIdx.x] = s;                                                          \
  }
PK_KERNEL(pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
PK_KERNEL(pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
PK_KERNEL(pk_add_f32, "v_pk_add_f32 %0, %0, %1")
PK_KERNEL(pk_fma_opsel, "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]")
KERNEL(permlane16_swap, "v_permlane16_swap_b32 %0, %1")

// LDS forms: 8 independent accesses per iteration, then one wait
#define LDS_KERNEL(NAME, BODY)                                                                         \
  __global__ void __launch_bounds__(256, 4) k_##NAME(float* out, float a, float b, int iters) {       \
    __shared__ float4 sm[2048];                                                                        \
    for (int i = threadIdx.x; i < 2048; i += 256) sm[i] = make_float4(a, b, a, b);                     \
    __syncthreads();                                                                                   \
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                                   \
    unsigned addr_seq = (wave * 512 + lane) * 16;            /* consecutive 16-byte slots */          \
    unsigned addr_rnd = (wave * 512 + ((lane * 37 + 11) & 15) * 3) * 16; /* 16 distinct 48-byte records */ \
    unsigned addr_row = (wave * 512) * 16 + lane * 8;        /* 8-byte pixel-major */                 \
    (void)addr_seq; (void)addr_rnd; (void)addr_row;                                                    \
    v4f v = {a, b, a, b}, acc = v; v2f v2 = {a, b};                                                       \
    for (int it = 0; it < iters; ++it) { BODY }                                                        \
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w + v.x + v2.x;                         \
  }

#define RD128(A, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(v) : "v"(A));
LDS_KERNEL(ds_read_b128_seq, RD128(addr_seq, 0) RD128(addr_seq, 1024) RD128(addr_seq, 2048) RD128(addr_seq, 3072)
           RD128(addr_seq, 4096) RD128(addr_seq, 5120) RD128(addr_seq, 6144) RD128(addr_seq, 7168)
           asm volatile("s_waitcnt lgkmcnt(0)"); acc.x += v.x;)
LDS_KERNEL(ds_read_b128_rec16, RD128(addr_rnd, 0) RD128(addr_rnd, 16) RD128(addr_rnd, 32) RD128(addr_rnd, 768)
           RD128(addr_rnd, 784) RD128(addr_rnd, 800) RD128(addr_rnd, 1536) RD128(addr_rnd, 1552)
           asm volatile("s_waitcnt lgkmcnt(0)"); acc.x += v.x;)
#define RD64(A, OFF) asm volatile("ds_read_b64 %0, %1 offset:" #OFF : "=v"(v2) : "v"(A));
LDS_KERNEL(ds_read_b64_row, RD64(addr_row, 0) RD64(addr_row, 520) RD64(addr_row, 1040) RD64(addr_row, 1560)
           RD64(addr_row, 2080) RD64(addr_row, 2600) RD64(addr_row, 3120) RD64(addr_row, 3640)
           asm volatile("s_waitcnt lgkmcnt(0)"); acc.x += v.x;)
#define RD32(A, OFF) asm volatile("ds_read_b32 %0, %1 offset:" #OFF : "=v"(v.x) : "v"(A));
LDS_KERNEL(ds_read_b32_row, RD32(addr_row, 0) RD32(addr_row, 260) RD32(addr_row, 520) RD32(addr_row, 780)
           RD32(addr_row, 1040) RD32(addr_row, 1300) RD32(addr_row, 1560) RD32(addr_row, 1820)
           asm volatile("s_waitcnt lgkmcnt(0)"); acc.x += v.x;)
#define WR64(A, OFF) asm volatile("ds_write_b64 %0, %1 offset:" #OFF :: "v"(A), "v"(v2));
LDS_KERNEL(ds_write_b64_row, WR64(addr_row, 0) WR64(addr_row, 520) WR64(addr_row, 1040) WR64(addr_row, 1560)
           WR64(addr_row, 2080) WR64(addr_row, 2600) WR64(addr_row, 3120) WR64(addr_row, 3640)
           asm volatile("s_waitcnt lgkmcnt(0)");)
#define WR32(A, OFF) asm volatile("ds_write_b32 %0, %1 offset:" #OFF :: "v"(A), "v"(v.x));
LDS_KERNEL(ds_write_b32_row, WR32(addr_row, 0) WR32(addr_row, 260) WR32(addr_row, 520) WR32(addr_row, 780)
           WR32(addr_row, 1040) WR32(addr_row, 1300) WR32(addr_row, 1560) WR32(addr_row, 1820)
           asm volatile("s_waitcnt lgkmcnt(0)");)
#define WR128(A, OFF) asm volatile("ds_write_b128 %0, %1 offset:" #OFF :: "v"(A), "v"(v));
LDS_KERNEL(ds_write_b128_seq, WR128(addr_seq, 0) WR128(addr_seq, 1024) WR128(addr_seq, 2048) WR128(addr_seq, 3072)
           WR128(addr_seq, 4096) WR128(addr_seq, 5120) WR128(addr_seq, 6144) WR128(addr_seq, 7168)
           asm volatile("s_waitcnt lgkmcnt(0)");)

#define WR2_64(A, O0, O1) asm volatile("ds_write2_b64 %0, %1, %1 offset0:" #O0 " offset1:" #O1 :: "v"(A), "v"(v2));
LDS_KERNEL(ds_write2_b64_row, WR2_64(addr_row, 0, 65) WR2_64(addr_row, 130, 195) WR2_64(addr_row, 4, 69) WR2_64(addr_row, 134, 199)
           WR2_64(addr_row, 8, 73) WR2_64(addr_row, 138, 203) WR2_64(addr_row, 12, 77) WR2_64(addr_row, 142, 207)
           asm volatile("s_waitcnt lgkmcnt(0)");)
#define WR64S(A, OFF) asm volatile("ds_write_b64 %0, %1 offset:" #OFF :: "v"(A), "v"(v2));
LDS_KERNEL(ds_write_b64_seq, WR64S(addr_row, 0) WR64S(addr_row, 512) WR64S(addr_row, 1024) WR64S(addr_row, 1536)
           WR64S(addr_row, 2048) WR64S(addr_row, 2560) WR64S(addr_row, 3072) WR64S(addr_row, 3584)
           asm volatile("s_waitcnt lgkmcnt(0)");)
typedef double v2d __attribute__((ext_vector_type(2)));
#define RD2_64(A, O0, O1) asm volatile("ds_read2_b64 %0, %1 offset0:" #O0 " offset1:" #O1 : "=v"(v) : "v"(A));
LDS_KERNEL(ds_read2_b64_row, RD2_64(addr_row, 0, 1) RD2_64(addr_row, 65, 66) RD2_64(addr_row, 130, 131) RD2_64(addr_row, 195, 196)
           RD2_64(addr_row, 4, 5) RD2_64(addr_row, 69, 70) RD2_64(addr_row, 134, 135) RD2_64(addr_row, 199, 200)
           asm volatile("s_waitcnt lgkmcnt(0)"); acc.x += v.x;)

typedef void (*kern_t)(float*, float, float, int);
struct Entry { const char* name; kern_t fn; int per_iter; };

int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
#define E(NAME, N) {#NAME, k_##NAME, N}
  Entry tab[] = {E(fma, 8), E(fmac, 8), E(fmac_dpp_bcast, 8), E(fma_abs, 8), E(mul, 8), E(add, 8), E(sub_dpp, 8),
                 E(min_f32, 8), E(mov, 8), E(mad_u24, 8), E(lshl_add, 8), E(ffbh, 8), E(bfe, 8), E(min_u32, 8),
                 E(xor_b32, 8), E(lshlrev, 8), E(add_u32, 8), E(sub_u32, 8), E(and_b32, 8), E(or_b32, 8), E(and_or, 8), E(lshl_or, 8), E(max_f32, 8), E(med3_f32, 8), E(mul_u24, 8), E(mul_e64_neg, 8), E(sub_f32, 8), E(fma_sgpr_lit, 8), E(add_dpp_quad, 8), E(add_dpp_rowshr, 8), E(bcnt, 8), E(exp, 8), E(rcp, 8), E(cmp_vcc, 8), E(cndmask_vcc, 8), E(cmp_cndmask, 16),
                 E(mov_dpp_quad, 8), E(permlane32_swap, 8), E(permlane16_swap, 8), E(pk_fma_f32, 8), E(pk_mul_f32, 8), E(pk_add_f32, 8), E(pk_fma_opsel, 8), E(readlane_like_bpermute, 8), E(swizzle, 8),
                 E(ds_read_b128_seq, 8), E(ds_read_b128_rec16, 8), E(ds_read_b64_row, 8), E(ds_read_b32_row, 8),
                 E(ds_write_b64_row, 8), E(ds_write2_b64_row, 8), E(ds_write_b64_seq, 8), E(ds_read2_b64_row, 8), E(ds_write_b32_row, 8), E(ds_write_b128_seq, 8)};
  const int nblk = 4096;  // 16 blocks per CU: every SIMD always has 4 resident waves
  for (auto& t : tab) {
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(t.fn, dim3(nblk), dim3(256), 0, 0, out, 1.0001f, 0.5f, ITERS);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double wave_instr_per_simd = (double)nblk * 4 * ITERS * t.per_iter / 1024.0;  // 1024 SIMDs
    const double ns = best * 1e6 / wave_instr_per_simd;
    printf("%-26s %8.3f ms  %6.3f ns per wave-instruction per SIMD = %5.2f cycles at 2.4 GHz\n", t.name, best, ns, ns * 2.4);
  }
  return 0;
}
