// Micro-benchmark: dependent v_mul_f32 -> v_add_f32 chain under different exec masks, 1..4 waves per workgroup (gfx950).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__ (256) void k (float *out, long long *clk, int iters, float B, unsigned long long mask)
{
  const int lane = threadIdx.x & 63;
  float z = out[threadIdx.x];
  const long long c0 = clock64 ();
  if ((mask >> lane) & 1) {
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int u = 0; u < 64; u++) z = __fadd_rn (1e-3f, __fmul_rn (z, B));
    }
  }
  const long long c1 = clock64 ();
  out[threadIdx.x] = z;
  if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}
int main ()
{
  float *out; long long *clk;
  hipMalloc (&out, 1024 * 4); hipMemset (out, 0, 1024 * 4); hipMalloc (&clk, 8 * 1024);
  const unsigned long long masks[] = { ~0ull, 1ull, 0xFFFFull, 0xFFFFFFFFull, 0x1111111111111111ull };
  for (int threads : { 64, 128, 256 }) for (int blocks : { 1, 8, 256, 1024 }) for (unsigned long long m : masks) {
    const int iters = 3000;
    hipLaunchKernelGGL (k, dim3 (blocks), dim3 (threads), 0, 0, out, clk, 10, 0.999f, m);
    hipLaunchKernelGGL (k, dim3 (blocks), dim3 (threads), 0, 0, out, clk, iters, 0.999f, m);
    long long h; hipMemcpy (&h, clk, 8, hipMemcpyDeviceToHost);
    printf ("%4d blocks x %3d threads, exec %016llx (%2d lanes): %.1f ticks per mul+add pair\n", blocks, threads, m, __builtin_popcountll (m), double (h) / (double (iters) * 64));
  }
  return 0;
}
