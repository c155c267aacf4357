// gfx950 kernels for the float converters with the DC block ON over a whole recording
// (ref: src/convert.c:112-245: convert_uc8_generic / convert_sc16_generic / convert_sc16q11_generic).
//
// The one-pole filter   z1 = fI * DC_A + z1 * DC_B   (one rounding per product and per sum, never fused)
// is a chain over every sample of the stream, and DC_B = exp(-2 pi / rate) is so close to 1 that a
// wrong start value is forgotten only after ~35 / DC_A samples.  What makes it parallel all the same:
//
//   g(z) = RN (t + RN (z * B))   is monotone non-decreasing in z  (RN = round to nearest, B > 0).
//
// A chunk of the stream starts its chain `warm` samples early, where the filter is in real arithmetic (dc_linear_kernel),
// and steps it through the very float operations of the reference.  By the chunk's first sample the start is forgotten --
// nearly always; the host accepts a chunk only when it enters with the very state the chunk before ended on, so every
// accepted chunk is the reference's chain by induction from the stream's first sample, and one that is not accepted is
// stepped again from the right state until it has met the first pass (dc_chain_kernel, prepare_dc in engine.cpp).  A
// shard's first chunk behind a lead-in has no chunk before it: it steps both ends of a bracket lo <= z <= hi, which the
// monotone g keeps in order, and is right once lo == hi.  Chunks are independent; one workgroup of eight waves runs one:
// two chain waves (I, Q), and helpers that stage the next 4096 samples and form the magnitudes behind the chains.
//
// A chain is two dependent operations per sample.  A wave therefore steps 1024 samples at once, sixteen per lane,
// from guessed lane starts that it refines until it can prove them (see dc_round below).
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

// FULL: the caller knows that all four samples lie in front of s_end
template <int CONV, bool FULL = false>
__device__ __forceinline__ void dc_load4 (const uint8_t *iq, long long s, long long s_end, DcRaw<CONV> &r)
{
  constexpr int NW = CONV == kConvUc8Generic ? 2 : 4;
  if (FULL || s + 4 <= s_end) {
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

#ifndef DC_PER
#define DC_PER 16
#endif
#ifndef DC_ITERS
#define DC_ITERS 16
#endif
constexpr int kDcPer = DC_PER;                  // consecutive samples per lane in one round
constexpr int kDcRound = 64 * kDcPer;           // samples per round
constexpr int kDcIters = DC_ITERS;              // refinements of the lanes' start values per round

__device__ __forceinline__ float bcast_lane (float x, int l)
{
  return __uint_as_float (uint32_t (__builtin_amdgcn_readlane (int (__float_as_uint (x)), l)));
}

// ---- one round of a chain: up to kDcRound samples, kDcPer consecutive ones per lane ----------------------------
// Every lane steps its own samples with the REAL operations (ref: src/convert.c:137-140), so what happens inside a
// lane -- ties of the rounding included -- is the reference's arithmetic; only the value a lane STARTS from is
// guessed.  The guesses are refined: a lane's increment (end - start) depends on its start only through the few
// places where RN (z * B) or a tie falls differently, so  start[l] = z0 + sum of the increments of the lanes below
// converges in two to three passes even where z1 sits on a boundary of the product's rounding -- which is where the
// filter settles whenever the input has a DC component: RN (z * B) = z - k ulp with k = round (z * (1 - B) / ulp), the
// mean of RN (t) falls between two integers, and z1 dithers across the point where k changes.
// A pass is accepted up to the first lane whose start differs, bit for bit, from the value the lane below ended on:
// exact by induction from z0, whatever the guesses were.
// tk[q] = t of sample kDcPer * lane + q; nl = lanes that hold samples (lanes beyond hold anything), nlmask = their mask.
// On return zt[q] = z1 behind that sample as stepped from the lane's start; the result is the number of leading lanes
// whose values are proven (>= 1).
// The round is a string of dependent operations and is paid for in latency, so: the sum of the first guess is a tree,
// the lane below is fetched with the proven z0 as lane 0's old value (no select), the verdict is one compare.
__device__ __forceinline__ int dc_round (const float *tk, float z0, float B, int lane, int nl, u64 nlmask, float *zt, uint32_t &n_pass)
{
  // First guess: every sample's step taken from z0 itself, d = RN (t + RN (z0 * B)) - z0.  While z1 stays in z0's binade
  // and on z0's side of the next point where RN (z * B) leaks one ulp more or less (such points lie 1 / DC_A ulps apart),
  // RN (z * B) = z - k ulp with z0's k, every z1 is a grid point, and a step adds the same whole number of ulps to any of
  // them: the prefix sums of d (exact: small multiples of the ulp) ARE the lanes' starts, and the first pass proves all
  // of them.  Where the round crosses such a point the lanes behind the crossing miss and are refined as below.
#ifndef DC_LINEAR_GUESS
  float start;
  {
    const float y0 = __fmul_rn (z0, B);
    float part[kDcPer];
#pragma unroll
    for (int q = 0; q < kDcPer; q++) part[q] = __fsub_rn (__fadd_rn (tk[q], y0), z0);
#pragma unroll
    for (int w = kDcPer / 2; w >= 1; w >>= 1)
#pragma unroll
      for (int q = 0; q < w; q++) part[q] += part[q + w];
    const float inc = part[0];
    const float incl = wave_incl_scan_f (inc);
    start = z0 + (incl - inc);
  }
#else
  // (the filter in real arithmetic, z' = z + t - z (1 - B), with z ~ z0 over the round: what ran before)
  float start;
  {
    float part[kDcPer];
#pragma unroll
    for (int q = 0; q < kDcPer; q++) part[q] = tk[q];
#pragma unroll
    for (int w = kDcPer / 2; w >= 1; w >>= 1)
#pragma unroll
      for (int q = 0; q < w; q++) part[q] += part[q + w];
    // what z1 loses per step: z - RN (z * B) at z0, an integer number of ulps
    const float leak = z0 - z0 * B;
    const float inc = part[0] - float (kDcPer) * leak;
    const float incl = wave_incl_scan_f (inc);
    start = z0 + (incl - inc);
  }
#endif
  int good = 1;
  // Lanes below `good` are proven and keep their starts; the others take theirs from the end of the last proven lane
  // plus the increments the lanes between them made in this pass -- sums that start at the proven front, so they stay
  // small and exact where the chain sits still, and the pass after a miss starts from this pass's knowledge instead of
  // the first guess a new round would make.
#pragma unroll 1
  for (int it = 0; it < kDcIters; it++) {
    float z = start;
#ifdef DC_COUNT_PASSES
    n_pass++;
#endif
#pragma unroll
    for (int q = 0; q < kDcPer; q++) { z = __fadd_rn (tk[q], __fmul_rn (z, B)); zt[q] = z; }
    // wave_shr:1 = the lane below; lane 0 has none and keeps the `old` operand: z0 itself
    const uint32_t below = uint32_t (__builtin_amdgcn_update_dpp (int (__float_as_uint (z0)), int (__float_as_uint (z)), 0x138, 0xF, 0xF, false));
    const u64 bad = __ballot (below != __float_as_uint (start)) & nlmask;
    good = bad ? __builtin_ctzll (bad) : nl;      // (a proven lane's start never changes below, so it is never the first miss)
    if (good == nl || it == kDcIters - 1) break;
    const float zg = bcast_lane (z, good - 1);    // the proven front: z1 in front of lane `good`
    const float inc = lane >= good ? __fsub_rn (z, start) : 0.0f;
    const float incl = wave_incl_scan_f (inc);
    if (lane >= good) start = __fadd_rn (zg, __fsub_rn (incl, inc));
  }
  return good;
}

// One chain over t[0 .. cnt): returns z1 behind the last sample, zo[i] = z1 behind sample i.  z0 is uniform.
// t and zo are 16-byte aligned and hold kDcBatch entries (entries at and beyond cnt are never used in a result).
// Two register sets alternate: while one round is stepped the t values of the round behind it are on their way from LDS
// (on the bet that the round is accepted whole, as nearly all are; after a partial round the set is fetched again).
__device__ __forceinline__ float dc_chain_batch (const float *t, float *zo, bool store, int cnt, float z0, float B, int lane, uint32_t &n_guess, uint32_t &n_plain)
{
  int base = 0;
  // (lanes beyond the round's last sample fetch whatever the buffer's last entries hold: nothing of theirs reaches a
  // result -- the prefix sums run upwards, the verdict is masked, the broadcasts read proven lanes)
  auto load = [&] (float *tk, int at) {
    const float4 *src = reinterpret_cast<const float4 *> (t + dcp (min (at + kDcPer * lane, kDcBatch - kDcPer)));
#pragma unroll
    for (int g = 0; g < kDcPer / 4; g++) { const float4 x = src[g]; tk[4 * g] = x.x; tk[4 * g + 1] = x.y; tk[4 * g + 2] = x.z; tk[4 * g + 3] = x.w; }
  };
  auto lanes_at = [&] (int at) { const int rem = cnt - at; return rem >= kDcRound ? 64 : rem / kDcPer; };
  // one round on `cur` (fetched here unless the round before has done so), the next one's values into `nxt` meanwhile;
  // false: the batch is through
  auto step = [&] (float *cur, float *nxt, bool &have) -> bool {
    if (base >= cnt) return false;
    const int rem = cnt - base;
    if (rem < kDcPer) {                            // ragged end of the stream: plain steps (ref: src/convert.c:137-140)
      float z = z0;
      const bool st = store && lane == 0;
      for (int i = 0; i < rem; i++) { z = __fadd_rn (t[dcp (base + i)], __fmul_rn (z, B)); if (st) zo[dcp (base + i)] = z; }
      z0 = z;
      base += rem; n_plain++;
      return false;
    }
    n_guess++;
    const int nl = lanes_at (base);                // lanes that hold samples of this round
    if (!have) load (cur, base);
    const int nb = base + kDcPer * nl;
    const bool ahead = cnt - nb >= kDcPer;
    if (ahead) load (nxt, nb);
    float zt[kDcPer];
    const int good = dc_round (cur, z0, B, lane, nl, nl == 64 ? ~0ull : (1ull << nl) - 1, zt, n_plain);      // >= 1: lane 0 starts from z0 itself
    if (store && lane < good) {
      float4 *dst = reinterpret_cast<float4 *> (zo + dcp (base + kDcPer * lane));
#pragma unroll
      for (int g = 0; g < kDcPer / 4; g++) { float4 o; o.x = zt[4 * g]; o.y = zt[4 * g + 1]; o.z = zt[4 * g + 2]; o.w = zt[4 * g + 3]; dst[g] = o; }
    }
    z0 = bcast_lane (zt[kDcPer - 1], good - 1);
    base += kDcPer * good;
    have = ahead && good == nl;
    return true;
  };
  float tkA[kDcPer], tkB[kDcPer];
  bool have = false;
  for (;;) {
    if (!step (tkA, tkB, have)) break;
    if (!step (tkB, tkA, have)) break;
  }
  return z0;
}

// Magnitudes of four consecutive samples per lane -- samples [b + o, b + o + 4) of a batch whose z1 values lie in zI / zQ
// (ref: src/convert.c:139-147); `raw` holds their IQ.
// FULL: the batch holds kDcBatch samples (no bounds to look at)
template <int CONV, bool FULL>
__device__ __forceinline__ void dc_magnitudes_go (const DcParams &p, const float *uc8f, const float *zI, const float *zQ,
                                                  const DcRaw<CONV> &raw, long long b, int o, int cnt)
{
  if (!FULL && o >= cnt) return;
  float fI[4], fQ[4];
  dc_to_float<CONV> (raw, uc8f, fI, fQ);
  const float4 a = *reinterpret_cast<const float4 *> (zI + o);
  const float4 c = *reinterpret_cast<const float4 *> (zQ + o);
  const float zi[4] = { a.x, a.y, a.z, a.w }, zq[4] = { c.x, c.y, c.z, c.w };
  uint32_t m[4];
  float sq[4];
#pragma unroll
  for (int k = 0; k < 4; k++)
    m[k] = mag_from_float (__fsub_rn (fI[k], zi[k]), __fsub_rn (fQ[k], zq[k]), &sq[k]);
  if (p.mag_sq) {                      // b and o are multiples of 4: whole float4s (the slack behind [n] takes the ragged end)
    float4 q4; q4.x = sq[0]; q4.y = sq[1]; q4.z = sq[2]; q4.w = sq[3];
    *reinterpret_cast<float4 *> (p.mag_sq + b + o) = q4;
  }
  uint16_t *dst = p.mag + b + o;
  if (FULL || o + 4 <= cnt) {
    uint32_t *d32 = reinterpret_cast<uint32_t *> (dst);          // p.mag is 4-byte aligned, b and o are even
    d32[0] = m[0] | (m[1] << 16); d32[1] = m[2] | (m[3] << 16);
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) if (o + k < cnt) dst[k] = uint16_t (m[k]);
  }
}

// One workgroup of eight waves per chunk.  A chain is a string of dependent operations: what a chunk costs is the
// number of its rounds times the latency of one, and nothing else may stand in the way of the wave that steps it.
//   waves 0 / 2   the chains of I and Q (the `lo` values of the descriptor) -- they do nothing but step, batch after
//                 batch, and leave z1 over the t values they have consumed (same LDS slot, same lane);
//   waves 1 / 3   the same samples from the `hi` values where the descriptor holds a bracket (a shard's first chunk behind
//                 a lead-in: nothing earlier can vouch for its state); helpers otherwise;
//   helpers       (waves 4 .. 7, and 1 / 3 when there is no bracket) take every fourth / sixth go of 256 samples of every
//                 batch: the magnitudes of the batch BEFORE the one being stepped (from the z1 the chains left), then
//                 t = f * DC_A of the batch AFTER it into the very slots just read; the raw samples of the batch behind
//                 that are in flight meanwhile.
// Two batch buffers alternate; one barrier a batch.
// Exactness does not rest on a bracket: the host accepts a chunk when the state it was in at its first sample IS, bit for
// bit, the state the chunk before ended on (induction from the stream's first sample).  To repair a chunk that was not,
// the kernel writes the chains' state every kDcCkpt samples (`ckpt`); a descriptor with stop == -1 steps the chunk again
// from the exact state and stops at the first checkpoint where it has met the trajectory of the first pass -- from there
// on both are the same chain.
template <int CONV>
__global__ __launch_bounds__ (512) __attribute__ ((amdgpu_waves_per_eu (4, 4)))
void dc_chain_kernel (const DcParams p)
{
  __shared__ __attribute__ ((aligned (16))) float s_buf[2][2][kDcBatchLds];   // [batch parity][channel]: f * DC_A, then z1 in place
  __shared__ float s_uc8[256];
  __shared__ float s_state[2][4];                                     // [batch parity] I-lo, I-hi, Q-lo, Q-hi in front of the batch
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane (tid >> 6);
  const DcChunk d = p.desc[blockIdx.x];
  const bool bracket = __float_as_uint (d.zIlo) != __float_as_uint (d.zIhi) || __float_as_uint (d.zQlo) != __float_as_uint (d.zQhi);
  const bool helper = wv >= 4 || (!bracket && (wv & 1));
  const int nh = bracket ? 4 : 6;                                     // helpers
  const int hidx = wv >= 4 ? wv - 4 : 4 + (wv >> 1);                  // this helper takes goes hidx, hidx + nh, ... of every batch
  constexpr int kGoes = kDcBatch / 256, NG = (kGoes + 3) / 4;         // goes of 256 samples (four per lane) in a batch; most goes a helper takes
  if (CONV == kConvUc8Generic && tid < 256) s_uc8[tid] = float ((double (tid) - 127.5) / 127.5);       // ref: src/convert.c:131
  if (tid == 0) { s_state[0][0] = d.zIlo; s_state[0][1] = d.zIhi; s_state[0][2] = d.zQlo; s_state[0][3] = d.zQhi; }
  __syncthreads ();
  const long long c0 = (long long) d.chunk * (long long) p.chunk_len;
  const long long c1 = d.stop > 0 ? d.stop : min (c0 + (long long) p.chunk_len, (long long) p.n);
  const bool cmp = d.stop == -1;
  bool noted = false;
  float z = wv == 0 ? d.zIlo : wv == 1 ? d.zIhi : wv == 2 ? d.zQlo : d.zQhi;
  const float A = p.A, B = p.B;
  uint32_t n_guess = 0, n_plain = 0;
  // (clock reads only while a profile is asked for -- MODES_DC_PROFILE sets `ticks`: each costs the chain wave a scalar memory round trip)
  const bool prof = p.ticks != nullptr;
  auto now = [&] () -> long long { return prof ? clock64 () : 0ll; };
  long long tm_chain = 0, tm_bar = 0, tm_bar_region = 0, tm_total = now ();

  DcRaw<CONV> raw[NG];                                                // IQ of the batch to stage next
  // (whole batches -- all but a stream's last -- take the paths without bounds: a fifth fewer operations for the helpers,
  // whose issue rate is what the chunk's own samples cost)
  auto fetch = [&] (long long b, DcRaw<CONV> *r) {
    if (b + kDcBatch <= c1) {
#pragma unroll
      for (int k = 0; k < NG; k++) { const int go = hidx + nh * k; if (go < kGoes) dc_load4<CONV, true> (p.iq, b + 256 * go + 4 * lane, c1, r[k]); }
    } else {
#pragma unroll
      for (int k = 0; k < NG; k++) { const int go = hidx + nh * k; if (go < kGoes) dc_load4<CONV, false> (p.iq, b + 256 * go + 4 * lane, c1, r[k]); }
    }
  };
  auto stage = [&] (int par) {
#pragma unroll
    for (int k = 0; k < NG; k++) {
      const int go = hidx + nh * k;
      if (go >= kGoes) break;
      float fI[4], fQ[4];
      dc_to_float<CONV> (raw[k], s_uc8, fI, fQ);
      float4 tI, tQ;
      tI.x = __fmul_rn (fI[0], A); tI.y = __fmul_rn (fI[1], A); tI.z = __fmul_rn (fI[2], A); tI.w = __fmul_rn (fI[3], A);
      tQ.x = __fmul_rn (fQ[0], A); tQ.y = __fmul_rn (fQ[1], A); tQ.z = __fmul_rn (fQ[2], A); tQ.w = __fmul_rn (fQ[3], A);
      *reinterpret_cast<float4 *> (&s_buf[par][0][dcp (256 * go + 4 * lane)]) = tI;
      *reinterpret_cast<float4 *> (&s_buf[par][1][dcp (256 * go + 4 * lane)]) = tQ;
    }
  };
  auto magnitudes = [&] (int par, long long b, int cnt) {             // of batch b, whose z1 lie in s_buf[par]; its IQ is read again (L2 holds it)
    DcRaw<CONV> rawm[NG];
    fetch (b, rawm);
    if (cnt == kDcBatch) {
#pragma unroll
      for (int k = 0; k < NG; k++) {
        const int go = hidx + nh * k;
        if (go >= kGoes) break;
        dc_magnitudes_go<CONV, true> (p, s_uc8, s_buf[par][0], s_buf[par][1], rawm[k], b, 256 * go + 4 * lane, cnt);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NG; k++) {
        const int go = hidx + nh * k;
        if (go >= kGoes) break;
        dc_magnitudes_go<CONV, false> (p, s_uc8, s_buf[par][0], s_buf[par][1], rawm[k], b, 256 * go + 4 * lane, cnt);
      }
    }
  };
  if (helper && d.start < c1) {
    fetch (d.start, raw);
    stage (0);
    if (d.start + kDcBatch < c1) fetch (d.start + kDcBatch, raw);
  }
  __syncthreads ();
  int pb = 0;                                                        // parity of the batch being stepped
  int ck_wait = 0, ck_idx = 0;                                       // batches until the next checkpoint (the first is the chunk's first sample), its index
  bool prev_ok = false;                                              // the batch before holds z1 of both channels and wants magnitudes
  long long prev_b = 0;
  int prev_cnt = 0;
  for (long long b = d.start; b < c1; b += kDcBatch, pb ^= 1) {
    // the states the chains hold in front of this batch
    const float sIlo = s_state[pb][0], sIhi = s_state[pb][1], sQlo = s_state[pb][2], sQhi = s_state[pb][3];
    const bool region = b >= c0;
    const bool both_closed = sIlo == sIhi && sQlo == sQhi;
    if (b == c0 && tid == 0) {            // first sample of the chunk: the state it is entered with (have the brackets closed?)
      float *zs = p.z_start + 4 * size_t (d.slot);
      zs[0] = sIlo; zs[1] = sIhi; zs[2] = sQlo; zs[3] = sQhi;
      p.unmerged[d.slot] = both_closed ? 0u : 1u;
    }
    if (region && ck_wait-- == 0) {
      ck_wait = kDcCkpt / kDcBatch - 1;
      float *ck = p.ckpt + 2 * (size_t (d.chunk) * p.nck + size_t (ck_idx++));
      if (cmp) {                          // has this chain met the first pass's?  (uniform: every thread reads the same two values)
        const uint32_t eI = __builtin_amdgcn_readfirstlane (__float_as_uint (ck[0])), eQ = __builtin_amdgcn_readfirstlane (__float_as_uint (ck[1]));
        if (eI == __float_as_uint (sIlo) && eQ == __float_as_uint (sQlo)) {
          noted = true;
          if (tid == 0) { p.closed_at[d.slot] = b; p.z_close[2 * size_t (d.slot)] = sIlo; p.z_close[2 * size_t (d.slot) + 1] = sQlo; }
          break;
        }
      } else if (tid == 0) { ck[0] = sIlo; ck[1] = sQlo; }
    }
    if (!cmp && region && !noted && both_closed) {                  // from here on the chunk's values are those of ONE chain
      noted = true;
      if (tid == 0) { p.closed_at[d.slot] = b; p.z_close[2 * size_t (d.slot)] = sIlo; p.z_close[2 * size_t (d.slot) + 1] = sQlo; }
    }
    const int cnt = int (min ((long long) kDcBatch, c1 - b));
    const bool keep_z = region && both_closed;                      // the lo chains leave z1 in the batch buffer
    if (!helper) {
      const bool closed = (wv & 1) && (wv == 1 ? sIlo == sIhi : sQlo == sQhi);     // a hi chain that has met its lo chain
      if (!closed) {
        const long long tc = now ();
        float *t = s_buf[pb][wv >> 1];
        z = dc_chain_batch (t, t, !(wv & 1) && keep_z, cnt, z, B, lane, n_guess, n_plain);
        if (lane == 0) {
          s_state[pb ^ 1][wv] = z;
          if (!(wv & 1) && (wv == 0 ? sIlo == sIhi : sQlo == sQhi)) s_state[pb ^ 1][wv + 1] = z;   // the closed hi chain follows
        }
        tm_chain += now () - tc;
      }
    } else {
      const long long tc = now ();
      if (prev_ok) magnitudes (pb ^ 1, prev_b, prev_cnt);
      if (b + kDcBatch < c1) {
        stage (pb ^ 1);
        if (b + 2 * kDcBatch < c1) fetch (b + 2 * kDcBatch, raw);
      }
      tm_chain += now () - tc;
    }
    const long long tb = now ();
    __syncthreads ();
    tm_bar += now () - tb;
    if (region) tm_bar_region += now () - tb;
    if (!helper && (wv & 1)) z = s_state[pb ^ 1][wv];              // (a closed hi chain follows its lo chain)
    prev_ok = keep_z; prev_b = b; prev_cnt = cnt;
  }
  if (prev_ok && helper) magnitudes (pb ^ 1, prev_b, prev_cnt);      // the last batch stepped
  if (tid == 0) {
    p.z_end[2 * size_t (d.slot)] = s_state[pb][0]; p.z_end[2 * size_t (d.slot) + 1] = s_state[pb][2];
    if (!noted) p.closed_at[d.slot] = -1;
  }
  if (lane == 0 && wv == 4 && p.ticks) { unsigned long long *o = p.ticks + 16 * size_t (d.chunk) + 13; o[0] = (unsigned long long) tm_chain; o[1] = (unsigned long long) tm_bar; o[2] = (unsigned long long) tm_bar_region; }
  if (lane == 0 && wv == 0 && p.ticks) p.ticks[16 * size_t (d.chunk) + 12] = (unsigned long long) tm_bar_region;
  if (lane == 0 && !helper && p.ticks) { unsigned long long *o = p.ticks + 16 * size_t (d.chunk) + 5 + 2 * wv; o[0] = n_guess; o[1] = n_plain; }
  if (lane == 0 && !helper && p.rounds) { atomicAdd (&p.rounds[0], (unsigned long long) n_guess); atomicAdd (&p.rounds[1], (unsigned long long) n_plain);
    if (wv == 0) { atomicAdd (&p.rounds[2], (unsigned long long) tm_chain); atomicAdd (&p.rounds[3], (unsigned long long) tm_bar); atomicAdd (&p.rounds[4], (unsigned long long) (now () - tm_total)); atomicMax (&p.rounds[5], (unsigned long long) (now () - tm_total)); if (p.ticks) { unsigned long long *o = p.ticks + 16 * size_t (d.chunk); o[0] = (unsigned long long) (now () - tm_total); o[1] = n_guess; o[2] = n_plain; o[3] = (unsigned long long) tm_chain; o[4] = (unsigned long long) tm_bar; } } }
}

// ---- where the filter would be in real arithmetic ---------------------------------------------------------------
// z1 behind a segment = z1 in front of it * B^len + sum over its samples of f * A * B^(len - 1 - i): the sums of all
// segments are independent (double precision here), the host chains them.  The float filter follows that value to
// within a few per cent (it settles where RN (z * B) changes its step, see dc_round), which is what the chunks'
// starting brackets are built from.  One workgroup per segment of kDcSeg samples.
template <int CONV>
__global__ __launch_bounds__ (256)
void dc_linear_kernel (const uint8_t *iq, long long n, double A, const double *wtab /* [1025]: B^(1023 - j), B^1024 */, double *out /* [segments][2] */)
{
  __shared__ float s_uc8[256];
  __shared__ double s_part[2][4];
  const int tid = threadIdx.x;
  if (CONV == kConvUc8Generic) s_uc8[tid] = float ((double (tid) - 127.5) / 127.5);
  __syncthreads ();
  const long long s0 = (long long) blockIdx.x * kDcSeg;
  const long long s1 = min (s0 + (long long) kDcSeg, n);
  // sample 1024 g + 4 tid + k of the segment carries the weight B^(kDcSeg - 1 - that): Horner over g with B^1024
  // (the weights come from a table the host made once: five pow () per thread were a third of this kernel)
  double w[4];
#pragma unroll
  for (int k = 0; k < 4; k++) w[k] = wtab[4 * tid + k];
  const double B1024 = wtab[1024];
  double aI = 0.0, aQ = 0.0;
  for (int g = 0; g < kDcSeg / 1024; g++) {
    DcRaw<CONV> raw;
    dc_load4<CONV> (iq, s0 + 1024 * g + 4 * tid, s1, raw);       // zeros beyond the end of the stream
    float fI[4], fQ[4];
    dc_to_float<CONV> (raw, s_uc8, fI, fQ);
    double sI = 0.0, sQ = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool in = s0 + 1024 * g + 4 * tid + k < s1;
      sI += in ? double (fI[k]) * w[k] : 0.0; sQ += in ? double (fQ[k]) * w[k] : 0.0;
    }
    aI = aI * B1024 + sI; aQ = aQ * B1024 + sQ;
  }
  for (int d = 32; d >= 1; d >>= 1) { aI += __shfl_xor (aI, d, 64); aQ += __shfl_xor (aQ, d, 64); }
  if ((tid & 63) == 0) { s_part[0][tid >> 6] = aI; s_part[1][tid >> 6] = aQ; }
  __syncthreads ();
  if (tid < 2) out[2 * size_t (blockIdx.x) + tid] = A * (s_part[tid][0] + s_part[tid][1] + s_part[tid][2] + s_part[tid][3]);
}

// ---- mean_power of every block from the stored squared magnitudes ----------------------------------------------
// `power += mag_sq` (ref: src/convert.c:146,153-154) is a float running sum over the block's samples, in order: the
// result depends on that order.  One wave per block; like the chains above, every lane adds its own sixteen values from
// a guessed start, a lane is proven when its start IS, bit for bit, what the lane below ended on, and the others take
// the proven front plus the increments in between -- at least one more lane is proven with every pass.
__global__ __launch_bounds__ (256)
void dc_block_power_kernel (const float *sq, uint64_t n, uint32_t block_samples, uint32_t nblocks, double *mean_power)
{
  const uint32_t blk = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (blk >= nblocks) return;
  const uint64_t n0 = uint64_t (blk) * block_samples, n1 = min (n0 + uint64_t (block_samples), n);
  float p0 = 0.0f;
  uint64_t base = n0;
  while (base < n1) {
    const uint64_t rem = n1 - base;
    if (rem < 16) {                                    // ragged end: plain adds
      for (uint64_t i = 0; i < rem; i++) p0 = __fadd_rn (p0, sq[base + i]);
      break;
    }
    const int nl = rem >= 1024 ? 64 : int (rem / 16);
    float x[16];
    if (lane < nl) {
      const float4 *src = reinterpret_cast<const float4 *> (sq + base + 16 * lane);        // base stays a multiple of 16 floats
#pragma unroll
      for (int g = 0; g < 4; g++) { const float4 v = src[g]; x[4 * g] = v.x; x[4 * g + 1] = v.y; x[4 * g + 2] = v.z; x[4 * g + 3] = v.w; }
    } else {
#pragma unroll
      for (int q = 0; q < 16; q++) x[q] = 0.0f;
    }
    float tot = 0.0f;
#pragma unroll
    for (int q = 0; q < 16; q++) tot += x[q];
    float start = p0 + (wave_incl_scan_f (tot) - tot);
    float pe = 0.0f;
    for (;;) {
      float v = start;
#pragma unroll
      for (int q = 0; q < 16; q++) v = __fadd_rn (v, x[q]);
      float below = __uint_as_float (dpp_mov<0x138, 0xF, true> (__float_as_uint (v)));
      if (lane == 0) below = p0;
      const u64 bad = __ballot (__float_as_uint (below) != __float_as_uint (start) && lane < nl);
      const int good = bad ? __builtin_ctzll (bad) : nl;
      if (good == nl) { pe = bcast_lane (v, nl - 1); break; }
      const float pg = bcast_lane (v, good - 1);
      const float inc = lane >= good ? __fsub_rn (v, start) : 0.0f;
      const float incl = wave_incl_scan_f (inc);
      if (lane >= good) start = __fadd_rn (pg, __fsub_rn (incl, inc));
    }
    p0 = pe;
    base += 16ull * nl;
  }
  if (lane == 0) mean_power[blk] = double (p0);
}

hipError_t launch_dc_block_power (const float *sq, uint64_t n, uint32_t block_samples, uint32_t nblocks, double *mean_power, hipStream_t stream)
{
  if (nblocks == 0) return hipSuccess;
  hipLaunchKernelGGL (dc_block_power_kernel, dim3 ((nblocks + 3) / 4), dim3 (256), 0, stream, sq, n, block_samples, nblocks, mean_power);
  return hipGetLastError ();
}

hipError_t launch_dc_linear (int conv, const uint8_t *iq, uint64_t n, double A, const double *wtab, double *out, hipStream_t stream)
{
  if (n == 0) return hipSuccess;
  const dim3 grid ((unsigned) ((n + kDcSeg - 1) / kDcSeg)), block (256);
  switch (conv) {
    case kConvUc8Generic:     hipLaunchKernelGGL (dc_linear_kernel<kConvUc8Generic>, grid, block, 0, stream, iq, (long long) n, A, wtab, out); break;
    case kConvSc16Generic:    hipLaunchKernelGGL (dc_linear_kernel<kConvSc16Generic>, grid, block, 0, stream, iq, (long long) n, A, wtab, out); break;
    case kConvSc16Q11Generic: hipLaunchKernelGGL (dc_linear_kernel<kConvSc16Q11Generic>, grid, block, 0, stream, iq, (long long) n, A, wtab, out); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError ();
}

hipError_t launch_dc_chain (int conv, const DcParams &p, uint32_t ndesc, hipStream_t stream)
{
  if (ndesc == 0) return hipSuccess;
  const dim3 grid (ndesc), block (512);
  switch (conv) {
    case kConvUc8Generic:     hipLaunchKernelGGL (dc_chain_kernel<kConvUc8Generic>, grid, block, 0, stream, p); break;
    case kConvSc16Generic:    hipLaunchKernelGGL (dc_chain_kernel<kConvSc16Generic>, grid, block, 0, stream, p); break;
    case kConvSc16Q11Generic: hipLaunchKernelGGL (dc_chain_kernel<kConvSc16Q11Generic>, grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError ();
}

}  // namespace modes
