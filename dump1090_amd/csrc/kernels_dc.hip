// gfx950 kernels for the float converters with the DC block ON over a whole recording
// (ref: src/convert.c:112-245: convert_uc8_generic / convert_sc16_generic / convert_sc16q11_generic).
//
// The one-pole filter   z1 = fI * DC_A + z1 * DC_B   (one rounding per product and per sum, never fused)
// is a chain over every sample of the stream, and DC_B = exp(-2 pi / rate) is so close to 1 that a
// wrong start value is forgotten only after ~35 / DC_A samples.  What makes it parallel all the same:
//
//   g(z) = RN (t + RN (z * B))   is monotone non-decreasing in z  (RN = round to nearest, B > 0).
//
// A chunk of the stream therefore starts its chain `warm` samples early from a bracket lo <= z <= hi that
// is known to hold the true state (|z1| never exceeds the largest |f|), steps lo and hi through the very
// float operations of the reference, and once lo == hi the value IS the reference's z1 -- whatever the
// start was.  Chunks are independent; one wave runs one chunk: lanes 0..3 carry I-lo, I-hi, Q-lo, Q-hi,
// all 64 lanes fetch/convert samples ahead of the chain and form the magnitudes behind it.
// A chunk whose bracket has not closed at its first sample is reported (`unmerged`) and re-run by the
// host from the exact end state of the chunk before it.
//
// "ref:" citations are relative to the reference's root.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"
#include "device_common.h"
#include "kernels.h"

namespace modes {

constexpr int kDcBatch = 256;          // samples per step of the wave: four per lane

// raw IQ of four consecutive samples
template <int CONV> struct DcRaw { uint32_t w[CONV == kConvUc8Generic ? 2 : 4]; };

template <int CONV>
__device__ __forceinline__ void dc_load4 (const uint8_t *iq, long long s, long long s_end, DcRaw<CONV> &r)
{
  constexpr int NW = CONV == kConvUc8Generic ? 2 : 4;
  if (s + 4 <= s_end) {
    if (CONV == kConvUc8Generic) { const uint2 v = *reinterpret_cast<const uint2 *> (iq + 2 * s); r.w[0] = v.x; r.w[1] = v.y; }
    else { const uint4 v = *reinterpret_cast<const uint4 *> (iq + 4 * s); r.w[0] = v.x; r.w[1] = v.y; r.w[2] = v.z; r.w[3] = v.w; }
    return;
  }
#pragma unroll
  for (int k = 0; k < NW; k++) r.w[k] = 0;
  for (int k = 0; k < 4 && s + k < s_end; k++) {       // the ragged end of the stream
    if (CONV == kConvUc8Generic) {
      const uint32_t v = reinterpret_cast<const uint16_t *> (iq)[s + k];
      r.w[k >> 1] |= v << (16 * (k & 1));
    } else r.w[k] = reinterpret_cast<const uint32_t *> (iq)[s + k];
  }
}

template <int CONV>
__device__ __forceinline__ void dc_to_float (const DcRaw<CONV> &r, const float *uc8f, float *fI, float *fQ)
{
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (CONV == kConvUc8Generic) {                                   // ref: src/convert.c:131-132
      const uint32_t iqp = r.w[k >> 1] >> (16 * (k & 1));
      fI[k] = uc8f[iqp & 0xFF]; fQ[k] = uc8f[(iqp >> 8) & 0xFF];
    } else {                                                         // ref: src/convert.c:176-177, 221-222
      const float sc = CONV == kConvSc16Generic ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
      fI[k] = float (int16_t (r.w[k] & 0xFFFF)) * sc; fQ[k] = float (int16_t (r.w[k] >> 16)) * sc;
    }
  }
}

template <int CONV>
__global__ __launch_bounds__ (64)
void dc_chain_kernel (const DcParams p)
{
  __shared__ __attribute__ ((aligned (16))) float s_t[2][kDcBatch];   // f * DC_A per channel
  __shared__ __attribute__ ((aligned (16))) float s_z[4][kDcBatch];   // z1 after every sample, per chain lane
  __shared__ float s_uc8[256];
  const int lane = threadIdx.x;
  if (CONV == kConvUc8Generic) {
    for (int b = lane; b < 256; b += 64) s_uc8[b] = float ((double (b) - 127.5) / 127.5);   // ref: src/convert.c:131
    __builtin_amdgcn_s_waitcnt (0xc07f);
    __builtin_amdgcn_wave_barrier ();
  }
  const DcChunk d = p.desc[blockIdx.x];
  const long long c0 = (long long) d.chunk * (long long) p.chunk_len;
  const long long c1 = min (c0 + (long long) p.chunk_len, (long long) p.n);
  float z = lane == 0 ? d.zIlo : lane == 1 ? d.zIhi : lane == 2 ? d.zQlo : d.zQhi;
  const float A = p.A, B = p.B;
  const int ch = (lane >> 1) & 1;

  DcRaw<CONV> raw;
  dc_load4<CONV> (p.iq, d.start + 4 * lane, c1, raw);
  for (long long b = d.start; b < c1; b += kDcBatch) {
    float fI[4], fQ[4];
    dc_to_float<CONV> (raw, s_uc8, fI, fQ);
    if (b + kDcBatch < c1) dc_load4<CONV> (p.iq, b + kDcBatch + 4 * lane, c1, raw);     // next batch, in flight behind the chain
    {
      float4 tI, tQ;
      tI.x = __fmul_rn (fI[0], A); tI.y = __fmul_rn (fI[1], A); tI.z = __fmul_rn (fI[2], A); tI.w = __fmul_rn (fI[3], A);
      tQ.x = __fmul_rn (fQ[0], A); tQ.y = __fmul_rn (fQ[1], A); tQ.z = __fmul_rn (fQ[2], A); tQ.w = __fmul_rn (fQ[3], A);
      *reinterpret_cast<float4 *> (&s_t[0][4 * lane]) = tI;
      *reinterpret_cast<float4 *> (&s_t[1][4 * lane]) = tQ;
    }
    __builtin_amdgcn_s_waitcnt (0xc07f);
    __builtin_amdgcn_wave_barrier ();
    const bool region = b >= c0;
    if (b == c0) {                      // first sample of the chunk: has the bracket closed?
      const float other = __shfl_xor (z, 1, 64);
      const bool open = lane < 4 && !(z == other);
      const u64 m = __ballot (open);
      if (lane < 4) p.z_start[4 * size_t (d.chunk) + lane] = z;
      if (lane == 0) p.unmerged[d.chunk] = m ? 1u : 0u;
    }
    const int cnt = int (min ((long long) kDcBatch, c1 - b));
    if (lane < 4) {
      const float *t = s_t[ch];
      float *zo = s_z[lane];
      if (cnt == kDcBatch) {
        // 32 samples at a time: their eight LDS reads are in flight while the 32 before them are stepped
        constexpr int G = 8;                        // float4 per group
        float4 cur[G], nxt[G];
#pragma unroll
        for (int k = 0; k < G; k++) cur[k] = *reinterpret_cast<const float4 *> (t + 4 * k);
#pragma unroll
        for (int g = 0; g < kDcBatch / (4 * G); g++) {
          if (g + 1 < kDcBatch / (4 * G)) {
#pragma unroll
            for (int k = 0; k < G; k++) nxt[k] = *reinterpret_cast<const float4 *> (t + 4 * G * (g + 1) + 4 * k);
          }
#pragma unroll
          for (int k = 0; k < G; k++) {
            float4 o;
            z = __fadd_rn (cur[k].x, __fmul_rn (z, B)); o.x = z;       // ref: src/convert.c:137-140 (one rounding per operation)
            z = __fadd_rn (cur[k].y, __fmul_rn (z, B)); o.y = z;
            z = __fadd_rn (cur[k].z, __fmul_rn (z, B)); o.z = z;
            z = __fadd_rn (cur[k].w, __fmul_rn (z, B)); o.w = z;
            *reinterpret_cast<float4 *> (zo + 4 * G * g + 4 * k) = o;
          }
#pragma unroll
          for (int k = 0; k < G; k++) cur[k] = nxt[k];
        }
      } else {
        for (int i = 0; i < cnt; i++) { z = __fadd_rn (t[i], __fmul_rn (z, B)); zo[i] = z; }
      }
    }
    __builtin_amdgcn_s_waitcnt (0xc07f);
    __builtin_amdgcn_wave_barrier ();
    if (region) {
      const float4 zI = *reinterpret_cast<const float4 *> (&s_z[0][4 * lane]);
      const float4 zQ = *reinterpret_cast<const float4 *> (&s_z[2][4 * lane]);
      const float zi[4] = { zI.x, zI.y, zI.z, zI.w }, zq[4] = { zQ.x, zQ.y, zQ.z, zQ.w };
      uint32_t m[4];
#pragma unroll
      for (int k = 0; k < 4; k++)                                  // ref: src/convert.c:139-147
        m[k] = mag_from_float (__fsub_rn (fI[k], zi[k]), __fsub_rn (fQ[k], zq[k]), nullptr);
      uint16_t *dst = p.mag + b + 4 * lane;
      if (4 * lane + 4 <= cnt) {
        uint32_t *d32 = reinterpret_cast<uint32_t *> (dst);        // p.mag is 4-byte aligned, b and 4 * lane are even
        d32[0] = m[0] | (m[1] << 16); d32[1] = m[2] | (m[3] << 16);
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) if (4 * lane + k < cnt) dst[k] = uint16_t (m[k]);
      }
      __builtin_amdgcn_s_waitcnt (0xc07f);
      __builtin_amdgcn_wave_barrier ();                            // s_z is rewritten by the next batch's chain
    }
  }
  const float zq_end = __shfl (z, 2, 64);
  if (lane == 0) { p.z_end[2 * size_t (d.chunk)] = z; p.z_end[2 * size_t (d.chunk) + 1] = zq_end; }
}

hipError_t launch_dc_chain (int conv, const DcParams &p, uint32_t ndesc, hipStream_t stream)
{
  if (ndesc == 0) return hipSuccess;
  const dim3 grid (ndesc), block (64);
  switch (conv) {
    case kConvUc8Generic:     hipLaunchKernelGGL (dc_chain_kernel<kConvUc8Generic>, grid, block, 0, stream, p); break;
    case kConvSc16Generic:    hipLaunchKernelGGL (dc_chain_kernel<kConvSc16Generic>, grid, block, 0, stream, p); break;
    case kConvSc16Q11Generic: hipLaunchKernelGGL (dc_chain_kernel<kConvSc16Q11Generic>, grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError ();
}

}  // namespace modes
