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
// start was.  Chunks are independent; one workgroup of four waves runs one chunk, wave w one of the four chains
// I-lo, I-hi, Q-lo, Q-hi (the hi waves idle once their bracket has closed); all 256 threads fetch/convert the next
// 4096 samples and form the magnitudes behind the chains.
// A chunk whose bracket has not closed at its first sample is reported (`unmerged`) and re-run by the
// host from the exact end state of the chunk before it.
//
// A chain is a latency problem: two dependent operations per sample.  A wave therefore steps 1024 samples at once,
// sixteen per lane, from guessed lane starts and keeps what it can prove (see dc_round below); where the guesses
// keep failing (z1 next to zero, or sitting on a boundary of the product's rounding) it steps plainly for a while.
//
// "ref:" citations are relative to the reference's root.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"
#include "device_common.h"
#include "kernels.h"

namespace modes {


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


// inclusive float prefix sum over the 64 lanes (row_shr 1,2,4,8, row_bcast:15, row_bcast:31; zero fill)
__device__ __forceinline__ float wave_incl_scan_f (float v)
{
#define DCADD(CTRL, MASK, BOUND) v = __fadd_rn (v, __uint_as_float (dpp_mov<CTRL, MASK, BOUND> (__float_as_uint (v))))
  DCADD (0x111, 0xF, true); DCADD (0x112, 0xF, true); DCADD (0x114, 0xF, true); DCADD (0x118, 0xF, true);
  DCADD (0x142, 0xA, false); DCADD (0x143, 0xC, false);
#undef DCADD
  return v;
}

// LDS index of sample i of a batch.  (Padding 16 bytes behind every 64 floats, to spread the lanes' 64-byte-stride
// groups over the banks, was measured and bought nothing: the rounds are bound by dependent latency, not by LDS.)
__device__ __forceinline__ int dcp (int i) { return i; }
constexpr int kDcBatchLds = kDcBatch;

constexpr int kDcPer = 16;                      // consecutive samples per lane in one guess
constexpr int kDcRound = 64 * kDcPer;           // samples per guess

__device__ __forceinline__ float bcast_lane (float x, int l)
{
  return __uint_as_float (uint32_t (__builtin_amdgcn_readlane (int (__float_as_uint (x)), l)));
}

// ---- one round of a chain: kDcRound samples, kDcPer consecutive ones per lane ----------------------------------
// Every lane steps its own samples with the REAL operations (ref: src/convert.c:137-140), so what happens inside a
// lane -- ties of the rounding included -- is the reference's arithmetic; only the value a lane STARTS from is
// guessed.  While z1 stays inside one binade and RN (z * B) keeps its distance to z, moving the start by a whole
// number of ulps moves every later value by the same amount, except that a tie of RN (t + z * B) falls the other
// way when the start moves by an odd number.  So each lane first steps its samples from z0 and from z0 + 1 ulp:
// that yields its increment and the parity of its increment for an even and for an odd start offset, i.e. a map
// (offset parity in) -> (offset parity out, increment).  The maps compose associatively; a scan over the lanes gives
// every lane its start offset, the lane steps again from there, and the round is accepted up to the first lane
// whose start differs, bit for bit, from the value the lane below ended on (exact by induction from z0).
struct DcMap { uint32_t n; float d0, d1; };        // n bit p: the parity flips for an input parity p; d_p: increment

__device__ __forceinline__ DcMap dc_compose (const DcMap y, const DcMap x)      // y first, then x
{
  DcMap o;
  const bool f0 = y.n & 1u, f1 = y.n & 2u;         // parity after y for inputs 0, 1 is 0 ^ f0, 1 ^ f1
  const uint32_t x0 = f0 ? (x.n >> 1) & 1u : x.n & 1u;          // x's flip at parity (0 ^ f0)
  const uint32_t x1 = f1 ? x.n & 1u : (x.n >> 1) & 1u;          // x's flip at parity (1 ^ f1)
  o.n = ((y.n & 1u) ^ x0) | ((((y.n >> 1) & 1u) ^ x1) << 1);
  o.d0 = __fadd_rn (y.d0, f0 ? x.d1 : x.d0);
  o.d1 = __fadd_rn (y.d1, f1 ? x.d0 : x.d1);
  return o;
}

template <int CTRL, int MASK, bool BOUND>
__device__ __forceinline__ DcMap dc_map_dpp (const DcMap x)
{
  DcMap y;
  y.n = dpp_mov<CTRL, MASK, BOUND> (x.n);
  y.d0 = __uint_as_float (dpp_mov<CTRL, MASK, BOUND> (__float_as_uint (x.d0)));
  y.d1 = __uint_as_float (dpp_mov<CTRL, MASK, BOUND> (__float_as_uint (x.d1)));
  return y;
}

// inclusive scan of the lanes' maps (all-zero = the identity, which is what DPP fills in)
__device__ __forceinline__ DcMap dc_map_scan (DcMap x)
{
  x = dc_compose (dc_map_dpp<0x111, 0xF, true> (x), x);
  x = dc_compose (dc_map_dpp<0x112, 0xF, true> (x), x);
  x = dc_compose (dc_map_dpp<0x114, 0xF, true> (x), x);
  x = dc_compose (dc_map_dpp<0x118, 0xF, true> (x), x);
  x = dc_compose (dc_map_dpp<0x142, 0xA, false> (x), x);
  x = dc_compose (dc_map_dpp<0x143, 0xC, false> (x), x);
  return x;
}

// tk[q] = t of sample kDcPer * lane + q.  On return zt[q] = z1 behind that sample as stepped from the lane's start;
// the result is the number of leading lanes whose values are proven (64 = the whole round).
__device__ __forceinline__ int dc_round (const float *tk, float z0, float B, int lane, float *zt)
{
  const uint32_t zb = __float_as_uint (z0);
  const float z1u = __uint_as_float (zb + 1u);                   // one ulp further from zero: the odd start
  float za = z0, zo = z1u;
#pragma unroll
  for (int q = 0; q < kDcPer; q++) {
    za = __fadd_rn (tk[q], __fmul_rn (za, B));
    zo = __fadd_rn (tk[q], __fmul_rn (zo, B));
  }
  DcMap mp;
  mp.d0 = __fsub_rn (za, z0);
  mp.d1 = __fsub_rn (zo, z1u);
  mp.n = ((__float_as_uint (za) ^ zb) & 1u) | (((__float_as_uint (zo) ^ zb ^ 1u) & 1u) << 1);
  const DcMap in = dc_map_scan (mp);
  DcMap ex = dc_map_dpp<0x138, 0xF, true> (in);                  // wave_shr:1: the lanes below
  if (lane == 0) { ex.n = 0; ex.d0 = 0.0f; ex.d1 = 0.0f; }
  // the chain enters the round at offset 0 (even): start of this lane = z0 + increment of the lanes below
  const float start = __fadd_rn (z0, ex.d0);
  float z = start;
#pragma unroll
  for (int q = 0; q < kDcPer; q++) { z = __fadd_rn (tk[q], __fmul_rn (z, B)); zt[q] = z; }
  float below = __uint_as_float (dpp_mov<0x138, 0xF, true> (__float_as_uint (z)));
  if (lane == 0) below = z0;
  const u64 bad = __ballot (__float_as_uint (below) != __float_as_uint (start));
  return bad ? __builtin_ctzll (bad) : 64;
}

// One chain over t[0 .. cnt): returns z1 behind the last sample, zo[i] = z1 behind sample i.  z0 is uniform.
// t and zo are 16-byte aligned.
__device__ __forceinline__ float dc_chain_batch (const float *t, float *zo, bool store, int cnt, float z0, float B, int lane, uint32_t &n_guess, uint32_t &n_plain, int &backoff)
{
  int base = 0, plain = 0;
  while (base < cnt) {
    const int rem = cnt - base;
    if (plain || rem < kDcRound || (base & 3)) {  // plain steps (ref: src/convert.c:137-140), then guess again
      int m = plain ? 64 * plain : rem < kDcRound ? rem : 4 - (base & 3);
      if (m > rem) m = rem;
      // every lane steps the same chain: a dependent VALU operation of a wave with one active lane takes four times
      // as long as one with all lanes active (tools/ubench/dep_latency2.hip: 35 against 8.6 cycles per step)
      float z = z0;
      const bool st = store && lane == 0;
      int i = 0;
      for (; i < m && ((base + i) & 3); i++) { z = __fadd_rn (t[dcp (base + i)], __fmul_rn (z, B)); if (st) zo[dcp (base + i)] = z; }
      float4 nx = i + 4 <= m ? *reinterpret_cast<const float4 *> (t + dcp (base + i)) : make_float4 (0.0f, 0.0f, 0.0f, 0.0f);
      for (; i + 4 <= m; i += 4) {
        const float4 x = nx;
        if (i + 8 <= m) nx = *reinterpret_cast<const float4 *> (t + dcp (base + i + 4));  // in flight behind the four steps
        float4 o;
        z = __fadd_rn (x.x, __fmul_rn (z, B)); o.x = z;
        z = __fadd_rn (x.y, __fmul_rn (z, B)); o.y = z;
        z = __fadd_rn (x.z, __fmul_rn (z, B)); o.z = z;
        z = __fadd_rn (x.w, __fmul_rn (z, B)); o.w = z;
        if (st) *reinterpret_cast<float4 *> (zo + dcp (base + i)) = o;
      }
      for (; i < m; i++) { z = __fadd_rn (t[dcp (base + i)], __fmul_rn (z, B)); if (st) zo[dcp (base + i)] = z; }
      z0 = z;
      base += m; n_plain += uint32_t ((m + 63) / 64); plain = 0;
      continue;
    }
    n_guess++;
    float tk[kDcPer], zt[kDcPer];
    {
      const float4 *src = reinterpret_cast<const float4 *> (t + dcp (base + kDcPer * lane));
#pragma unroll
      for (int g = 0; g < kDcPer / 4; g++) { const float4 x = src[g]; tk[4 * g] = x.x; tk[4 * g + 1] = x.y; tk[4 * g + 2] = x.z; tk[4 * g + 3] = x.w; }
    }
    const int good = dc_round (tk, z0, B, lane, zt);             // >= 1: lane 0 starts from z0 itself
    if (store && lane < good) {
      float4 *dst = reinterpret_cast<float4 *> (zo + dcp (base + kDcPer * lane));
#pragma unroll
      for (int g = 0; g < kDcPer / 4; g++) { float4 o; o.x = zt[4 * g]; o.y = zt[4 * g + 1]; o.z = zt[4 * g + 2]; o.w = zt[4 * g + 3]; dst[g] = o; }
    }
    z0 = bcast_lane (zt[kDcPer - 1], good - 1);
    base += kDcPer * good;
    // a round that got nowhere: step plainly for a while, twice as long every time it happens again (z1 sitting on a
    // boundary of the product's rounding, or next to zero, fails round after round)
    if (good < 16) { backoff = backoff ? (backoff < 32 ? 2 * backoff : 32) : 1; plain = backoff; }
    else if (good >= 32) backoff = 0;
  }
  return z0;
}

template <int CONV>
__global__ __launch_bounds__ (256)
void dc_chain_kernel (const DcParams p)
{
  __shared__ __attribute__ ((aligned (16))) float s_t[2][kDcBatchLds];   // f * DC_A per channel
  __shared__ __attribute__ ((aligned (16))) float s_z[2][kDcBatchLds];   // z1 behind every sample, lo chains (the hi chains only serve to close the brackets)
  __shared__ float s_uc8[256];
  __shared__ float s_state[4];                                        // I-lo, I-hi, Q-lo, Q-hi behind the last batch
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const DcChunk d = p.desc[blockIdx.x];
  if (CONV == kConvUc8Generic) s_uc8[tid] = float ((double (tid) - 127.5) / 127.5);       // ref: src/convert.c:131
  if (tid == 0) { s_state[0] = d.zIlo; s_state[1] = d.zIhi; s_state[2] = d.zQlo; s_state[3] = d.zQhi; }
  __syncthreads ();
  const long long c0 = (long long) d.chunk * (long long) p.chunk_len;
  const long long c1 = min (c0 + (long long) p.chunk_len, (long long) p.n);
  float z = wv == 0 ? d.zIlo : wv == 1 ? d.zIhi : wv == 2 ? d.zQlo : d.zQhi;
  const float A = p.A, B = p.B;
  uint32_t n_guess = 0, n_plain = 0;
  int backoff = 0;
  long long tm_chain = 0, tm_bar = 0, tm_total = clock64 ();

  constexpr int NG = kDcBatch / 1024;                                // groups of 1024 samples: four per thread each
  DcRaw<CONV> raw[NG];
#pragma unroll
  for (int g = 0; g < NG; g++) dc_load4<CONV> (p.iq, d.start + 1024 * g + 4 * tid, c1, raw[g]);
  for (long long b = d.start; b < c1; b += kDcBatch) {
    float fI[NG][4], fQ[NG][4];
#pragma unroll
    for (int g = 0; g < NG; g++) {
      dc_to_float<CONV> (raw[g], s_uc8, fI[g], fQ[g]);
      if (b + kDcBatch < c1) dc_load4<CONV> (p.iq, b + kDcBatch + 1024 * g + 4 * tid, c1, raw[g]);   // next batch, in flight behind the chains
      float4 tI, tQ;
      tI.x = __fmul_rn (fI[g][0], A); tI.y = __fmul_rn (fI[g][1], A); tI.z = __fmul_rn (fI[g][2], A); tI.w = __fmul_rn (fI[g][3], A);
      tQ.x = __fmul_rn (fQ[g][0], A); tQ.y = __fmul_rn (fQ[g][1], A); tQ.z = __fmul_rn (fQ[g][2], A); tQ.w = __fmul_rn (fQ[g][3], A);
      *reinterpret_cast<float4 *> (&s_t[0][dcp (1024 * g + 4 * tid)]) = tI;
      *reinterpret_cast<float4 *> (&s_t[1][dcp (1024 * g + 4 * tid)]) = tQ;
    }
    // the states the chains hold in front of this batch
    const float sIlo = s_state[0], sIhi = s_state[1], sQlo = s_state[2], sQhi = s_state[3];
    long long tb = clock64 ();
    __syncthreads ();
    tm_bar += clock64 () - tb;
    const bool region = b >= c0;
    if (b == c0 && tid == 0) {            // first sample of the chunk: have the brackets closed?
      float *zs = p.z_start + 4 * size_t (d.chunk);
      zs[0] = sIlo; zs[1] = sIhi; zs[2] = sQlo; zs[3] = sQhi;
      p.unmerged[d.chunk] = (sIlo == sIhi && sQlo == sQhi) ? 0u : 1u;
    }
    const int cnt = int (min ((long long) kDcBatch, c1 - b));
    const bool closed = (wv & 1) && (wv == 1 ? sIlo == sIhi : sQlo == sQhi);     // a hi chain that has met its lo chain
    if (!closed) {
      const long long tc = clock64 ();
      z = dc_chain_batch (s_t[wv >> 1], s_z[wv >> 1], !(wv & 1), cnt, z, B, lane, n_guess, n_plain, backoff);
      if (lane == 0) {
        s_state[wv] = z;
        if (!(wv & 1) && (wv == 0 ? sIlo == sIhi : sQlo == sQhi)) s_state[wv + 1] = z;   // the closed hi chain follows
      }
      tm_chain += clock64 () - tc;
    }
    tb = clock64 ();
    __syncthreads ();
    tm_bar += clock64 () - tb;
    if (closed) z = s_state[wv];
    if (region) {
#pragma unroll
      for (int g = 0; g < NG; g++) {
        const int o = 1024 * g + 4 * tid;
        if (o >= cnt) break;
        const float4 zI = *reinterpret_cast<const float4 *> (&s_z[0][dcp (o)]);
        const float4 zQ = *reinterpret_cast<const float4 *> (&s_z[1][dcp (o)]);
        const float zi[4] = { zI.x, zI.y, zI.z, zI.w }, zq[4] = { zQ.x, zQ.y, zQ.z, zQ.w };
        uint32_t m[4];
#pragma unroll
        for (int k = 0; k < 4; k++)                                // ref: src/convert.c:139-147
          m[k] = mag_from_float (__fsub_rn (fI[g][k], zi[k]), __fsub_rn (fQ[g][k], zq[k]), nullptr);
        uint16_t *dst = p.mag + b + o;
        if (o + 4 <= cnt) {
          uint32_t *d32 = reinterpret_cast<uint32_t *> (dst);      // p.mag is 4-byte aligned, b and o are even
          d32[0] = m[0] | (m[1] << 16); d32[1] = m[2] | (m[3] << 16);
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++) if (o + k < cnt) dst[k] = uint16_t (m[k]);
        }
      }
    }
  }
  __syncthreads ();
  if (tid == 0) { p.z_end[2 * size_t (d.chunk)] = s_state[0]; p.z_end[2 * size_t (d.chunk) + 1] = s_state[2]; }
  if (lane == 0 && p.ticks) { unsigned long long *o = p.ticks + 16 * size_t (d.chunk) + 5 + 2 * wv; o[0] = n_guess; o[1] = n_plain; }
  if (lane == 0 && p.rounds) { atomicAdd (&p.rounds[0], (unsigned long long) n_guess); atomicAdd (&p.rounds[1], (unsigned long long) n_plain);
    if (wv == 0) { atomicAdd (&p.rounds[2], (unsigned long long) tm_chain); atomicAdd (&p.rounds[3], (unsigned long long) tm_bar); atomicAdd (&p.rounds[4], (unsigned long long) (clock64 () - tm_total)); atomicMax (&p.rounds[5], (unsigned long long) (clock64 () - tm_total)); if (p.ticks) { unsigned long long *o = p.ticks + 16 * size_t (d.chunk); o[0] = (unsigned long long) (clock64 () - tm_total); o[1] = n_guess; o[2] = n_plain; o[3] = (unsigned long long) tm_chain; o[4] = (unsigned long long) tm_bar; } } }
}

hipError_t launch_dc_chain (int conv, const DcParams &p, uint32_t ndesc, hipStream_t stream)
{
  if (ndesc == 0) return hipSuccess;
  const dim3 grid (ndesc), block (256);
  switch (conv) {
    case kConvUc8Generic:     hipLaunchKernelGGL (dc_chain_kernel<kConvUc8Generic>, grid, block, 0, stream, p); break;
    case kConvSc16Generic:    hipLaunchKernelGGL (dc_chain_kernel<kConvSc16Generic>, grid, block, 0, stream, p); break;
    case kConvSc16Q11Generic: hipLaunchKernelGGL (dc_chain_kernel<kConvSc16Q11Generic>, grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError ();
}

}  // namespace modes
