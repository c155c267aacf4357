// Micro-benchmark: issue rate of a few VALU instruction kinds on gfx950, N waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ __launch_bounds__ (256) void k (uint32_t *out, int iters)
{
  uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) { REP16 (asm volatile ("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));) }
    if (KIND == 1) { REP16 (asm volatile ("v_add_u32 %0, %0, %0\n v_add_u32 %1, %1, %1\n v_add_u32 %2, %2, %2\n v_add_u32 %3, %3, %3\n v_add_u32 %4, %4, %4\n v_add_u32 %5, %5, %5\n v_add_u32 %6, %6, %6\n v_add_u32 %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (KIND == 2) { REP16 (asm volatile ("v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %4\n v_and_b32 %4, %4, %5\n v_and_b32 %5, %5, %6\n v_and_b32 %6, %6, %7\n v_and_b32 %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (KIND == 3) { REP16 (asm volatile ("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %4\n v_mul_u32_u24 %4, %4, %5\n v_mul_u32_u24 %5, %5, %6\n v_mul_u32_u24 %6, %6, %7\n v_mul_u32_u24 %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (KIND == 4) { REP16 (asm volatile ("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_u32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n v_cmp_lt_u32 vcc, %1, %2\n v_cndmask_b32 %3, %3, %4, vcc\n v_cmp_lt_u32 vcc, %5, %6\n v_cndmask_b32 %7, %7, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");) }
    if (KIND == 5) { REP16 (asm volatile ("v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %1, %1, %2\n v_pk_max_i16 %2, %2, %3\n v_pk_max_i16 %3, %3, %4\n v_pk_max_i16 %4, %4, %5\n v_pk_max_i16 %5, %5, %6\n v_pk_max_i16 %6, %6, %7\n v_pk_max_i16 %7, %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (KIND == 6) { REP16 (asm volatile ("v_alignbit_b32 %0, %0, %1, 7\n v_alignbit_b32 %1, %1, %2, 7\n v_alignbit_b32 %2, %2, %3, 7\n v_alignbit_b32 %3, %3, %4, 7\n v_alignbit_b32 %4, %4, %5, 7\n v_alignbit_b32 %5, %5, %6, 7\n v_alignbit_b32 %6, %6, %7, 7\n v_alignbit_b32 %7, %7, %0, 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (KIND == 7) { REP16 (asm volatile ("v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_min_u32_sdwa %1, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n v_mul_u32_u24_sdwa %2, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_min_u32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n v_mul_u32_u24_sdwa %4, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_min_u32_sdwa %5, %5, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n v_mul_u32_u24_sdwa %6, %6, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n v_min_u32_sdwa %7, %7, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (KIND == 8) { REP16 (asm volatile ("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (KIND == 9) { REP16 (asm volatile ("v_dot2_u32_u16 %0, %0, %1, %2\n v_dot2_u32_u16 %1, %1, %2, %3\n v_dot2_u32_u16 %2, %2, %3, %4\n v_dot2_u32_u16 %3, %3, %4, %5\n v_dot2_u32_u16 %4, %4, %5, %6\n v_dot2_u32_u16 %5, %5, %6, %7\n v_dot2_u32_u16 %6, %6, %7, %0\n v_dot2_u32_u16 %7, %7, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (KIND == 10) { REP16 (asm volatile ("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %4\n v_mad_u32_u24 %3, %3, %4, %5\n v_mad_u32_u24 %4, %4, %5, %6\n v_mad_u32_u24 %5, %5, %6, %7\n v_mad_u32_u24 %6, %6, %7, %0\n v_mad_u32_u24 %7, %7, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (KIND == 11) { REP16 (asm volatile ("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3" : "+v"(*(double *)&f0), "+v"(*(double *)&f2), "+v"(*(double *)&f4), "+v"(*(double *)&f6));) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + uint32_t (f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
}

template <int KIND> void run (const char *name, int per_iter, uint32_t *out)
{
  for (int wps = 1; wps <= 8; wps *= 2) {        // waves per SIMD
    const int blocks = 256 * wps;                 // 256 CUs, 4 waves per block = 1 per SIMD
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate (&e0); hipEventCreate (&e1);
    hipLaunchKernelGGL (k<KIND>, dim3 (blocks), dim3 (256), 0, 0, out, 10);
    hipEventRecord (e0);
    hipLaunchKernelGGL (k<KIND>, dim3 (blocks), dim3 (256), 0, 0, out, iters);
    hipEventRecord (e1); hipEventSynchronize (e1);
    float ms; hipEventElapsedTime (&ms, e0, e1);
    const double insts_per_simd = double (iters) * per_iter * wps;
    printf ("%-22s waves/SIMD %d: %.3f ms, %.2f cycles@2.4GHz per wave-instruction per SIMD\n", name, wps, ms, ms * 1e-3 * 2.4e9 / insts_per_simd);
  }
}

int main ()
{
  uint32_t *out; hipMalloc (&out, 256 * 8 * 256 * 4);
  run<0> ("v_fma_f32", 128, out);
  run<1> ("v_add_u32", 128, out);
  run<2> ("v_and_b32", 128, out);
  run<3> ("v_mul_u32_u24", 128, out);
  run<4> ("v_cmp+v_cndmask", 128, out);
  run<5> ("v_pk_max_i16", 128, out);
  run<6> ("v_alignbit_b32", 128, out);
  run<7> ("sdwa mul/min", 128, out);
  run<8> ("v_mov_b32_dpp", 128, out);
  run<9> ("v_dot2_u32_u16", 128, out);
  run<10> ("v_mad_u32_u24", 128, out);
  run<11> ("v_pk_fma_f32", 64, out);
  return 0;
}
