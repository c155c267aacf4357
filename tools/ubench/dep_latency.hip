// Micro-benchmark: latency of a dependent v_mul_f32 -> v_add_f32 pair on gfx950 and the shader clock while it runs.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off dep_latency.hip -o dep_latency ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k (float *out, long long *clk, int iters, float B, float t)
{
  float z = out[threadIdx.x];
  const long long c0 = clock64 (), w0 = wall_clock64 ();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 64; u++) z = __fadd_rn (t, __fmul_rn (z, B));
  }
  const long long c1 = clock64 (), w1 = wall_clock64 ();
  out[threadIdx.x] = z;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
int main ()
{
  float *out; long long *clk, h[2];
  hipMalloc (&out, 4096 * 4); hipMemset (out, 0, 4096 * 4); hipMalloc (&clk, 16);
  for (int blocks : { 1, 1024, 4096 }) for (int threads : { 4, 64, 256 }) {
    const int iters = 20000;
    hipLaunchKernelGGL (k, dim3 (blocks), dim3 (threads), 0, 0, out, clk, 100, 0.999f, 1e-3f);
    hipEvent_t a, b; hipEventCreate (&a); hipEventCreate (&b);
    hipEventRecord (a);
    hipLaunchKernelGGL (k, dim3 (blocks), dim3 (threads), 0, 0, out, clk, iters, 0.999f, 1e-3f);
    hipEventRecord (b); hipEventSynchronize (b);
    float ms; hipEventElapsedTime (&ms, a, b);
    hipMemcpy (h, clk, 16, hipMemcpyDeviceToHost);
    const double steps = double (iters) * 64;
    printf ("blocks %4d threads %3d: %.3f ms, %.2f ns per mul+add pair, clock64 %.1f ticks/pair, wall_clock64 %.2f ticks/pair -> clock64 runs at %.0f MHz if wall is 100 MHz\n",
            blocks, threads, ms, ms * 1e6 / steps, h[0] / steps, h[1] / steps, 100.0 * h[0] / h[1]);
  }
  return 0;
}
