// Micro-benchmark: a dependent v_mul_f32 -> v_add_f32 chain run by ONE lane of a full wave, with and without
// LDS traffic per four steps.  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__ (256) void k (float *out, long long *clk, int iters, float B)
{
  __shared__ __attribute__ ((aligned (16))) float t[4][1024], zo[4][1024];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = lane; i < 1024; i += 64) t[wv][i] = 1e-3f * float (i & 7);
  __syncthreads ();
  float z = out[threadIdx.x];
  const long long c0 = clock64 ();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {                       // all lanes, registers only
#pragma unroll
      for (int u = 0; u < 64; u++) z = __fadd_rn (1e-3f, __fmul_rn (z, B));
    } else if (lane == 0) {                // one lane
      if (MODE == 1) {
#pragma unroll
        for (int u = 0; u < 64; u++) z = __fadd_rn (1e-3f, __fmul_rn (z, B));
      } else {                             // + LDS: 16-byte read ahead, 16-byte write behind every four steps
        const float *tt = t[wv] + (it & 15) * 64; float *oo = zo[wv] + (it & 15) * 64;
        float4 nx = *reinterpret_cast<const float4 *> (tt);
        for (int i = 0; i < 64; i += 4) {
          const float4 x = nx;
          if (i + 8 <= 64) nx = *reinterpret_cast<const float4 *> (tt + i + 4);
          float4 o;
          z = __fadd_rn (x.x, __fmul_rn (z, B)); o.x = z; z = __fadd_rn (x.y, __fmul_rn (z, B)); o.y = z;
          z = __fadd_rn (x.z, __fmul_rn (z, B)); o.z = z; z = __fadd_rn (x.w, __fmul_rn (z, B)); o.w = z;
          *reinterpret_cast<float4 *> (oo + i) = o;
        }
      }
    }
  }
  const long long c1 = clock64 ();
  out[threadIdx.x] = z + zo[wv][lane];
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = c1 - c0;
}
template <int MODE> void run (const char *name, float *out, long long *clk)
{
  const int iters = 5000;
  for (int blocks : { 1, 256 }) {
    hipLaunchKernelGGL (k<MODE>, dim3 (blocks), dim3 (256), 0, 0, out, clk, 10, 0.999f);
    hipLaunchKernelGGL (k<MODE>, dim3 (blocks), dim3 (256), 0, 0, out, clk, iters, 0.999f);
    long long h; hipMemcpy (&h, clk, 8, hipMemcpyDeviceToHost);
    printf ("%-28s blocks %3d x 256 threads: %.1f ticks per step\n", name, blocks, double (h) / (double (iters) * 64));
  }
}
int main ()
{
  float *out; long long *clk;
  hipMalloc (&out, 1024 * 4); hipMemset (out, 0, 1024 * 4); hipMalloc (&clk, 16);
  run<0> ("all lanes, registers", out, clk);
  run<1> ("one lane, registers", out, clk);
  run<2> ("one lane, LDS in and out", out, clk);
  return 0;
}
