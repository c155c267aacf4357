// gfx950 kernels of libmodes_hip.so.
//
// scan2400: IQ bytes in HBM -> (magnitudes only ever live in LDS) -> preamble detect ->
//           phase pick -> bit slicing -> CRC syndrome -> error-table diagnose -> records.
//
// One wavefront (64 lanes) owns one strip of kStrip scan positions.  No __syncthreads():
// every LDS buffer is private to a wave, all hand-offs are wave-synchronous.
//
// "ref:" citations are relative to the reference's root.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"
#include "device_common.h"
#include "kernels.h"

#ifndef SCANF_WAVES
#define SCANF_WAVES 5
#endif
namespace modes {



// A wave scans PAIR consecutive strips and feeds their candidates through ONE queue, so that the 64-lane batches of
// stages C and D run fuller.  Measured (profiles/r02_pmc_scan_stages.json, tools/ab.sh): PAIR = 2 cuts the VALU
// instructions per strip by 18 % (uc8 byte table) / 10 % (sc16), but its 8 KB of magnitudes per wave leave room for 4
// waves per SIMD instead of 6, and the kernel is as sensitive to that (4 against 6 waves with single strips: +20 %
// time) as to the instruction count: the byte-table instance comes out 3.6 % faster, the sc16 one 6 % slower.  So the
// byte-table converters run with PAIR = 2, the others with single strips.  kConvUc8LutR8 -- the byte-table instance whose
// tile holds ranks, one byte per sample (device_common.h) -- has the pair AND the 6 waves: a further 8 % faster.
template <int CONV> struct ScanPair { static constexpr int n = (CONV == kConvUc8Lut || CONV == kConvUc8LutPower || CONV == kConvUc8LutR8 || CONV == kConvUc8LutRaw) ? 2 : 1; };
// Waves per SIMD the register allocation aims for.  With the wave number read as a scalar (below) the integer sc16
// instance and the rank tile fit 72 registers, and their 21 / 22.6 KB of LDS per workgroup leave room for seven
// workgroups per CU: 3 % faster than six.  The float converters spill at six already and stay there.
template <int CONV, bool PE> struct ScanWaves {
  static constexpr int n = (CONV == kConvSc16Generic || CONV == kConvSc16Q11Generic) ? SCANF_WAVES :
                           (!PE && (CONV == kConvUc8Lut || CONV == kConvUc8LutPower)) ? 4 :       // two strips of magnitudes per wave: 39 KB of LDS per workgroup
                           (!PE && (CONV == kConvUc8LutR8 || CONV == kConvUc8LutRaw || CONV == kConvSc16Int || CONV == kConvMagU16)) ? 7 : 6;
};
constexpr int kWavesPerBlock = 4;
constexpr int kQ1Seg = 256;
constexpr uint32_t kSynSkipped = 0x80000000u;      // stage D1 did not slice the candidate: the reference never looks at it
template <bool PE, int PAIR> struct WaveLds {
  uint16_t q1[kQ1Seg];                              // off | pattern << 12, kQ1Seg candidates at a time
  uint32_t q2[PE ? 5 * 64 : 64];                    // off | (t-4) << 12 | df << 16, one stage-C batch at a time
  uint32_t vp[PAIR * kStrip / 32];                  // valid-preamble bit per position of the span
};


__device__ __forceinline__ uint32_t alignbit (uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit (hi, lo, sh); }

// One data bit of the 2.4 MS/s slicers, weight folded in, on the differences e[j] = d[j] - d[j+1]:
//   5a-3b-2c = 5e0+2e1, 4a-b-3c = 4e0+3e1, 3a+b-4c = 3e0+4e1, 2a+3b-5c = 2e0+5e1, a+5b-5c-d = e0+6e1+e2
// (ref: src/demod-2400.c:32-55); F = fifths of a sample after d[0].
template <int F>
__device__ __forceinline__ int wslice (const int *e)
{
  constexpr int i = F / 5, r = F % 5, w = corr_weight<r> ();
  int v = __mul24 (e[i], w * (5 - r)) + __mul24 (e[i + 1], w * (2 + r));
  if (r == 4) v += __mul24 (e[i + 2], w);
  return v;
}
// correlate_check for start phase T (ref: src/demod-2400.c:86-129): returns the score, shifts the five sliced bits into F5
template <int T>
__device__ __forceinline__ int phase_score (const int *e, uint32_t &F5)
{
  const int a = wslice<T> (e), b = wslice<T + 12> (e), c = wslice<T + 24> (e), d = wslice<T + 36> (e), g = wslice<T + 48> (e);
  const int na = -a, nb = -b, nc = -c, nd = -d, ng = -g;
  F5 = alignbit (F5, uint32_t (na), 31); F5 = alignbit (F5, uint32_t (nb), 31); F5 = alignbit (F5, uint32_t (nc), 31);
  F5 = alignbit (F5, uint32_t (nd), 31); F5 = alignbit (F5, uint32_t (ng), 31);
  return max (a, na) + max (b, nb) + max (c, nc) + max (d, nd) + max (g, ng);
}

// one data bit that starts f fifths of a sample after m[0]: > 0 means 1 (ref: src/demod-2400.c:32-55, 339-412)
template <typename Ref>
__device__ __forceinline__ int slice_at (const Ref &m, int f)
{
  const int i = (f * 0xCCCD) >> 18, ph = f - 5 * i;
  const int m0 = m.mag (i), m1 = m.mag (i + 1), m2 = m.mag (i + 2), m3 = m.mag (i + 3);
  const int e0 = m0 - m1, e1 = m1 - m2, e2 = m2 - m3;
  return __mul24 (e0, 5 - ph) + __mul24 (e1, 2 + ph) + __mul24 (e2, ph >> 2);
}

// XOR over the 64 lanes, result valid in lane 63
__device__ __forceinline__ uint32_t wave_xor_to63 (uint32_t v)
{
  v = row_xor (v);
  v ^= dpp_mov<0x142, 0xA, false> (v);
  v ^= dpp_mov<0x143, 0xC, false> (v);
  return v;
}

// ------------------------------------------------------------------------------------------
// the scan kernel
// ------------------------------------------------------------------------------------------
template <int CONV, bool PE>
__global__ __launch_bounds__ (64 * kWavesPerBlock) __attribute__ ((amdgpu_waves_per_eu (ScanWaves<CONV, PE>::n, ScanWaves<CONV, PE>::n)))
void scan2400_kernel (const ScanParams p)
{
  constexpr bool kRaw = CONV == kConvUc8LutRaw;     // the tile holds the input bytes (LDS-DMA), ranks and magnitudes are looked up
  constexpr bool kR8 = CONV == kConvUc8LutR8 || kRaw;   // the tile holds one byte per sample (ranks, or input bytes), not magnitudes
  constexpr bool kLut = CONV == kConvUc8Lut || CONV == kConvUc8LutPower || kR8;
  constexpr int kPair = PE ? 1 : ScanPair<CONV>::n;
  constexpr int kSpan = kPair * kStrip;                           // positions one wave scans
  constexpr int kGroupSpan = kWavesPerBlock * kSpan;              // positions one workgroup scans
  constexpr int kStageHalf = kGroupSpan / 2;                      // the tile is staged in two goes (registers)
  // raw tile: s_tile[x] = input byte of sample a0 + x, a0 = (sample of V[G0]) - kRawShift2400, in whole 16-byte chunks
  constexpr int kRawBytes = (kRawShift2400 + kGroupSpan + kLook2400 + 15) / 16 * 16;
  constexpr int kTileN = kRaw ? kRawBytes : kStageHalf + 8 * TileChunks<kStageHalf + kLook2400>::n;
  constexpr int kLead = kRaw ? 0 : kR8 ? kTileLead8 : kTileLead;
  static_assert (kSpan <= 4096, "queue entries keep a position of the span in 12 bits");
  static_assert (!(kR8 && PE), "the rank tile is built for the plain scan only");
  using TileT = typename std::conditional<kR8, uint8_t, uint16_t>::type;
  __shared__ WaveLds<PE, kPair> lds[kWavesPerBlock];
  __shared__ __attribute__ ((aligned (16))) TileT s_tile[kLead + kTileN];   // s_tile[kLead + x] = V[G0 + x], shared by the four waves
  __shared__ uint16_t s_lut[kLut ? 256 : 1];        // as-written uc8 table; rank tile: rank -> magnitude
  __shared__ uint8_t s_rank[kR8 ? 256 : 1];         // rank tile: byte -> rank
  const int lane = lane_id ();
  const int wv = __builtin_amdgcn_readfirstlane (int (threadIdx.x >> 6));    // wave-uniform: what derives from it is scalar arithmetic
  WaveLds<PE, kPair> &w = lds[wv];
  // The tables are fetched first and stored last: the raw tile's DMA is issued between the two, so the wait for the
  // tables does not stand in front of it (its staging reads them only on its slow path, behind a barrier of its own).
  uint16_t lut_mine = 0;
  uint8_t rank_mine = 0;
  if (kLut) {
    lut_mine = (kR8 && !kRaw) ? p.rankmag[threadIdx.x] : p.lut256[threadIdx.x];
    if (kR8) rank_mine = p.rank256[threadIdx.x];
    if (!kRaw) {
      s_lut[threadIdx.x] = lut_mine;
      if (kR8) s_rank[threadIdx.x] = rank_mine;
      __syncthreads ();
    }
  }
  const uint32_t strip = p.strip_begin + (blockIdx.x * kWavesPerBlock + wv) * kPair;   // first strip of this wave's span
  const uint32_t strip_end = p.strip_end ? p.strip_end : p.nstrips;
#pragma unroll
  for (int h = 0; h < kPair; h++) w.vp[64 * h + lane] = 0;

  // stage D maps lane -> message bit (lane ^ 7) (+ 64 in the second round): a ballot then holds the message bytes as they
  // lie in memory.  Single-bit syndromes of those bits for 112-bit and 56-bit CRCs (ref: src/crc.c:35,56-61)
  const int bl = lane ^ 7;

  // ---- stage A: magnitudes V[G0 .. G0 + 4 spans + kLook2400) into LDS, staged by the whole workgroup in two goes ----
  const long long G0 = (long long) p.strip_begin * kStrip + (long long) blockIdx.x * kGroupSpan;
  const long long s0 = G0 - (long long) (CONV == kConvMagU16 ? 0u : p.ov);   // sample index of V[G0] (may be < 0)
  if constexpr (kRaw) {
    const long long a0 = s0 - kRawShift2400;        // G0 and the launch's first strip are multiples of 16, ov = kOv2400
    const bool interior = a0 >= 0 && a0 + kRawBytes <= (long long) p.nsamples;     // workgroup-uniform
    if (!interior) { s_lut[threadIdx.x] = lut_mine; s_rank[threadIdx.x] = rank_mine; __syncthreads (); }    // the slow path reads s_lut
    if (p.debug_stop != 9)                            // (9: the stage probe's run without loads, on whatever tile the LDS holds)
      stage_tile_raw<kRawBytes, kWavesPerBlock> (s_tile, s_lut, p, a0, wv, lane, int (threadIdx.x));
    if (interior) { s_lut[threadIdx.x] = lut_mine; s_rank[threadIdx.x] = rank_mine; }
    asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");       // the DMA writes land before the barrier lets anyone read
  } else if constexpr (kR8) {
    stage_tile_rank8<kStageHalf, 64 * kWavesPerBlock> (s_tile + kLead, s_rank, s_lut, p, s0, int (threadIdx.x));
    stage_tile_rank8<kStageHalf + kLook2400, 64 * kWavesPerBlock> (s_tile + kLead + kStageHalf, s_rank, s_lut, p, s0 + kStageHalf, int (threadIdx.x));
  } else {
    stage_tile_pos<CONV, kStageHalf, 64 * kWavesPerBlock> (s_tile + kLead, s_lut, p, s0, int (threadIdx.x));
    stage_tile_pos<CONV, kStageHalf + kLook2400, 64 * kWavesPerBlock> (s_tile + kLead + kStageHalf, s_lut, p, s0 + kStageHalf, int (threadIdx.x));
  }
  __syncthreads ();
  if (strip >= strip_end) return;
  const long long P0 = (long long) strip * kStrip;
  const TileRef<kR8> V { s_tile + (kRaw ? kRawShift2400 : kLead) + wv * kSpan, s_lut };
  // order key of one sample: the magnitude, its rank, or (raw tile) the rank of the byte
  auto ordv = [&] (const TileRef<kR8> &T, int i) -> uint32_t { if constexpr (kRaw) return s_rank[T.p[i]]; else return T.ord (i); };

  if (p.debug_stop == 1) { if (lane == 0) p.counts[strip] = V.ord (lane); return; }

  // ---- stage B, once per strip of the span: lane l owns positions [32 l, 32 l + 32) of the strip: LT/GT bit planes ----
  uint32_t any[kPair], c0[kPair], c1[kPair], c2[kPair];
#pragma unroll
  for (int h = 0; h < kPair; h++) {
    const TileRef<kR8> Vh = V.at (h * kStrip);
    uint32_t LT = 0, GT = 0;
    if constexpr (kR8) {
      uint32_t wd[9];
      if constexpr (kRaw) {
        // the lane's 33 input bytes sit kRawShift2400 bytes into three aligned 16-byte reads; their ranks are gathered
        // from the 256-byte table and packed four to a word
        const uint4 *vw = reinterpret_cast<const uint4 *> (s_tile + wv * kSpan + h * kStrip + 32 * lane);
        const uint4 y0 = vw[0], y1 = vw[1], y2 = vw[2];
        const uint32_t W[12] = { y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w, y2.x, y2.y, y2.z, y2.w };
        constexpr int kW = kRawShift2400 / 4, kB = kRawShift2400 % 4;
#pragma unroll
        for (int k = 0; k < 9; k++) {
          const uint32_t b4 = kB ? __builtin_amdgcn_alignbyte (W[k + kW + 1], W[k + kW], kB) : W[k + kW];
          if (k < 8) {
            const uint32_t r0 = s_rank[b4 & 0xFF], r1 = s_rank[(b4 >> 8) & 0xFF], r2 = s_rank[(b4 >> 16) & 0xFF], r3 = s_rank[b4 >> 24];
            const uint32_t lo = __builtin_amdgcn_perm (r1, r0, 0x0c0c0400u), hi = __builtin_amdgcn_perm (r3, r2, 0x0c0c0400u);
            wd[k] = __builtin_amdgcn_perm (hi, lo, 0x05040100u);
          } else wd[k] = s_rank[b4 & 0xFF];
        }
      } else {
        const uint4 *vw = reinterpret_cast<const uint4 *> (Vh.p + 32 * lane);
        const uint4 x0 = vw[0], x1 = vw[1];
        wd[0] = x0.x; wd[1] = x0.y; wd[2] = x0.z; wd[3] = x0.w; wd[4] = x1.x; wd[5] = x1.y; wd[6] = x1.z; wd[7] = x1.w;
        wd[8] = Vh.ord (32 * lane + 32);
      }
      // Four neighbour pairs are compared per word: with H = 0x80 and L = 0x7F in every byte, (x | H) - (y & L) never
      // borrows between bytes and keeps bit 7 where the low seven bits of x are >= those of y; the top bits decide
      // otherwise.  v_dot4 with (1, 2, 4, 8) / (16, 32, 64, 128) then packs the four verdicts of a word next to those
      // of its neighbour: 3.75 operations per position instead of 5 with one subtraction per pair.
      const uint32_t H = 0x80808080u, L = 0x7F7F7F7Fu;
      uint32_t l7[4], g7[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        uint32_t al = 0, ag = 0;
#pragma unroll
        for (int h = 1; h >= 0; h--) {
          const uint32_t x = wd[2 * i + h], y = __builtin_amdgcn_alignbyte (wd[2 * i + h + 1], x, 1);   // the four samples one further on
          const uint32_t t1 = (x | H) - (y & L), t2 = (y | H) - (x & L);
          const uint32_t ltw = ((~x & y) | (~(x ^ y) & ~t1)) & H;                                         // bit 7 of a byte: x < y
          const uint32_t gtw = ((~y & x) | (~(x ^ y) & ~t2)) & H;                                         //                  x > y
          const uint32_t K = h ? 0x80402010u : 0x08040201u;
          al = __builtin_amdgcn_udot4 (ltw, K, al, false);
          ag = __builtin_amdgcn_udot4 (gtw, K, ag, false);
        }
        l7[i] = al; g7[i] = ag;                                                                           // eight verdicts, times 128
      }
      LT = (l7[0] >> 7) | (l7[1] << 1) | (l7[2] << 9) | (l7[3] << 17);
      GT = (g7[0] >> 7) | (g7[1] << 1) | (g7[2] << 9) | (g7[3] << 17);
    } else {
      const uint4 *vw = reinterpret_cast<const uint4 *> (Vh.p + 32 * lane);
      const uint4 x0 = vw[0], x1 = vw[1], x2 = vw[2], x3 = vw[3];
      const uint32_t wd[17] = { x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w, x3.x, x3.y, x3.z, x3.w,
                                Vh.ord (32 * lane + 32) };
#pragma unroll
      for (int k = 31; k >= 0; k--) {
        const int a = int ((wd[k >> 1] >> (16 * (k & 1))) & 0xFFFFu);
        const int b = int ((wd[(k + 1) >> 1] >> (16 * ((k + 1) & 1))) & 0xFFFFu);
        const int d = a - b;
        LT = alignbit (LT, uint32_t (d), 31);        // shifts in (a < b)
        GT = alignbit (GT, uint32_t (-d), 31);       // shifts in (a > b)
      }
    }
    uint32_t LTn = dpp_mov<0x130, 0xF, true> (LT), GTn = dpp_mov<0x130, 0xF, true> (GT);   // wave_shl:1 = lane + 1's word
    {   // the word behind the strip, for lane 63
      const uint32_t a = ordv (Vh, kStrip + (lane & 15)), b = ordv (Vh, kStrip + (lane & 15) + 1);
      const uint32_t l = uint32_t (__ballot (a < b)) & 0xFFFFu, g = uint32_t (__ballot (a > b)) & 0xFFFFu;
      if (lane == 63) { LTn = l; GTn = g; }
    }

    // peaks P_k = (p[k] < p[k+1] > p[k+2]); the five layouts of ref: src/demod-2400.c:225-282 with the
    // quick check (p0 < p1, p12 > p13) folded in.
#define LTk(k) alignbit (LTn, LT, k)
#define GTk(k) alignbit (GTn, GT, k)
    const uint32_t P0m = LT & GTk (1), P2m = LTk (2) & GTk (3), P8m = LTk (8) & GTk (9), P11m = LTk (11) & GTk (12);
    const uint32_t P3m = LTk (3) & GTk (4), P9m = LTk (9) & GTk (10);
    const uint32_t t389 = P0m & P2m & P8m;
    const uint32_t ph3 = t389 & LTk (10) & GTk (12);
    const uint32_t ph4 = t389 & P11m;
    const uint32_t ph5 = P0m & P11m & LTk (2) & GTk (4) & LTk (8) & GTk (10);
    const uint32_t u39b = P3m & P9m & P11m;
    const uint32_t ph6 = P0m & u39b;
    const uint32_t ph7 = LT & GTk (2) & u39b;
#undef LTk
#undef GTk
    uint32_t anyh = ph3 | ph4 | ph5 | ph6 | ph7;
    {   // positions at or beyond the end of the stream are not scanned (ref: src/demod-2400.c:200)
      const long long base = P0 + (long long) h * kStrip + 32ll * lane, N = (long long) p.nsamples;
      if (base >= N) anyh = 0;
      else if (base + 32 > N) anyh &= (~0u) >> (32 - int (N - base));
    }
    // first matching layout wins (else-if chain): code planes of the layout index 0..4
    const uint32_t n3 = ~ph3, n34 = n3 & ~ph4, n345 = n34 & ~ph5;
    const uint32_t e6 = ph6 & n345, e7 = anyh & n345 & ~ph6;
    any[h] = anyh; c0[h] = (ph4 & n3) | e6; c1[h] = (ph5 & n34) | e6; c2[h] = e7;
  }

  if (p.debug_stop == 2) { if ((any[0] ^ any[kPair - 1]) == 0x1234567u) p.counts[strip] = 1; return; }
  // ---- candidates in ascending order (strip by strip, lane by lane), kQ1Seg at a time through queue 1 ----
  int mycnt[kPair], first_slot[kPair];
  int n1 = 0;
#pragma unroll
  for (int h = 0; h < kPair; h++) {
    mycnt[h] = __popc (any[h]);
    const int incl = wave_incl_scan (mycnt[h], lane);
    first_slot[h] = n1 + incl - mycnt[h];
    n1 += __builtin_amdgcn_readlane (incl, 63);             // scalar: the loops below branch on the scalar unit
  }
  bool overflow = false;
  uint32_t valid_preamble[kPair], nrec[kPair];
#pragma unroll
  for (int h = 0; h < kPair; h++) { valid_preamble[h] = 0; nrec[h] = 0; }
  Record *out = p.records + size_t (strip) * p.cap;
  // Skip-ahead known inside the span (ref: src/demod-2400.c:514): behind a message that the reference certainly
  // reaches and certainly accepts (DF11/17/18 with syndrome 0: score and accept rules ask the ICAO cache nothing,
  // ref: src/dump1090.c:3309-3353,2440-2578) it never looks at the next msglen*12/5 positions, so candidates there
  // are dropped before the slicer.  c_end: positions below are certainly skipped; u_end: positions below may be
  // skipped by a message whose acceptance depends on the ICAO cache or on what earlier spans found (a block's first
  // strip starts with nothing pending: each demod_2400 call starts at j = 0; blocks are whole spans).
  int c_end = 0;
  uint32_t r_m0lo = 0, r_m0hi = 0, r_m1lo = 0, r_m1hi = 0, r_syn = 0;   // per-lane results of stage D1 (lane = candidate of the batch)
  int u_end = (p.block_strips ? strip % p.block_strips == 0 : strip == 0) ? 0 : kLongBits * 12 / 5;

  for (int seg = 0; seg < n1 && !overflow; seg += kQ1Seg) {
#pragma unroll
    for (int h = 0; h < kPair; h++) {
      int slot = first_slot[h] - seg;
      uint32_t mm = any[h];
      while (mm) {
        const int b = __builtin_ctz (mm);
        mm &= mm - 1;
        // (the layout index stays in the code planes of the lane that owns the position: stage C1 fetches it from there)
        if (uint32_t (slot) < uint32_t (kQ1Seg)) w.q1[slot] = uint16_t (h * kStrip + 32 * lane + b);
        slot++;
      }
    }
    __builtin_amdgcn_s_waitcnt (0xc07f);
    __builtin_amdgcn_wave_barrier ();
    const int nseg = min (n1 - seg, kQ1Seg);

    // ---- stage C1: amplitude tests, one lane per candidate; survivors (about one in four) are compacted in place ----
    int nsurv = 0;
    for (int base = 0; base < nseg; base += 64) {
      const int qi = base + lane;
      bool amp = false;
      int off = qi < nseg ? int (w.q1[qi]) : 0;
      // layout index 0..4 of the candidate: bit (off & 31) of the three code planes of lane (off >> 5) & 63
      int pat;
      {
        const int src = (off >> 3) & 0xFC;                          // byte address of the owning lane for ds_bpermute
        uint32_t k0 = uint32_t (__builtin_amdgcn_ds_bpermute (src, int (c0[0]))), k1 = uint32_t (__builtin_amdgcn_ds_bpermute (src, int (c1[0]))),
                 k2 = uint32_t (__builtin_amdgcn_ds_bpermute (src, int (c2[0])));
        if (kPair > 1) {
          const uint32_t j0 = uint32_t (__builtin_amdgcn_ds_bpermute (src, int (c0[kPair - 1]))), j1 = uint32_t (__builtin_amdgcn_ds_bpermute (src, int (c1[kPair - 1]))),
                         j2 = uint32_t (__builtin_amdgcn_ds_bpermute (src, int (c2[kPair - 1])));
          const bool hi = off >= kStrip;
          k0 = hi ? j0 : k0; k1 = hi ? j1 : k1; k2 = hi ? j2 : k2;
        }
        const int b = off & 31;
        pat = int (((k0 >> b) & 1u) | (((k1 >> b) & 1u) << 1) | (((k2 >> b) & 1u) << 2));
      }
      if (qi < nseg) {
        const TileRef<kR8> pp = V.at (off);
        int s[19];
#pragma unroll
        for (int k = 0; k < 19; k++) s[k] = pp.mag (k);
        // ref: src/demod-2400.c:233-281; which samples enter high / base_signal / base_noise, by layout index
        const int s39 = s[3] + s[9], s410 = s[4] + s[10];
        const int hsum = s[1] + s[12] + (pat <= 2 ? s39 : 0) + (pat >= 2 ? s410 : 0) + (pat == 4 ? s[2] : 0) + (pat == 0 ? s[11] : 0);
        const int high = hsum >> 2;
        const uint32_t sig = uint32_t ((pat == 4 ? 0 : s[1]) + (pat <= 1 ? s39 : 0) + (pat >= 3 ? s410 : 0) + (pat == 0 ? 0 : s[12]));
        const uint32_t noise = uint32_t (s[6] + s[7] + ((pat == 2 || pat == 4) ? 0 : s[5]) + ((pat == 0 || pat == 2) ? 0 : s[8]));
        const int quiet = max (max (max (s[5], s[6]), max (s[7], s[8])), max (max (max (s[14], s[15]), max (s[16], s[17])), s[18]));
        amp = !(sig * 2 < 3 * noise) && quiet < high;                                     // ref: :292, :297-308
      }
      const u64 am = __ballot (amp);
      if (amp) w.q1[nsurv + __popcll (am & ((1ull << lane) - 1))] = uint16_t (off);       // nsurv + rank <= qi: in place
      nsurv += __popcll (am);
    }
    __builtin_amdgcn_s_waitcnt (0xc07f);
    __builtin_amdgcn_wave_barrier ();

    for (int base = 0; base < nsurv && !overflow; base += 64) {
      // ---- stage C2: phase pick (best_phase), one lane per surviving candidate ----
      const int qi = base + lane;
      bool pass = false;
      int off = 0, nent = 0;
      uint32_t ent[PE ? 5 : 1];
      if (qi < nsurv) {
        off = w.q1[qi];
        const TileRef<kR8> pp = V.at (off);
        pass = true;
        {
          int e[14], thr;
          {
            int d[15];
#pragma unroll
            for (int k = 0; k < 15; k++) d[k] = pp.mag (19 + k);
#pragma unroll
            for (int k = 0; k < 14; k++) e[k] = d[k] - d[k + 1];
            thr = d[0] + d[1] + d[2] + d[3] + d[4] + d[5];       // threshold of best_phase (ref: :137)
          }
          uint32_t F5 = 0;                              // five sliced bits per start phase, phase 4 in bits 24..20
          const int cv0 = phase_score<4> (e, F5), cv1 = phase_score<5> (e, F5), cv2 = phase_score<6> (e, F5),
                    cv3 = phase_score<7> (e, F5), cv4 = phase_score<8> (e, F5);
          if (PE) {                                                                       // ref: :310-314
#pragma unroll
            for (int t = 0; t < 5; t++) {
              const int df = int ((F5 >> (5 * (4 - t))) & 31u);
              if (slice_bytes_for_df (df)) ent[nent++] = uint32_t (off) | (uint32_t (t) << 12) | (uint32_t (df) << 16);
            }
          } else {                                                                        // ref: :134-179, 319-323
            // strict > against the running best: the earliest phase holding the maximum wins
            const int key = max (max (max ((cv0 << 3) | 4, (cv1 << 3) | 3), max ((cv2 << 3) | 2, (cv3 << 3) | 1)), cv4 << 3);
            pass = (key >> 3) > thr;
            const int sh = key & 7;                    // 4 - phase index
            const int df = int ((F5 >> (5 * sh)) & 31u);
            if (pass && slice_bytes_for_df (df)) ent[nent++] = uint32_t (off) | (uint32_t (4 - sh) << 12) | (uint32_t (df) << 16);
          }
        }
      }
      {                                                                                   // ref: :326
        const u64 pm = __ballot (pass), hi = kPair > 1 ? __ballot (pass && off >= kStrip) : 0ull;
        valid_preamble[0] += __popcll (pm & ~hi);
        if (kPair > 1) valid_preamble[kPair - 1] += __popcll (hi);
      }
      if (pass) atomicOr (&w.vp[off >> 5], 1u << (off & 31));
      const int inc = wave_incl_scan (nent, lane);
      const int n2 = __builtin_amdgcn_readlane (inc, 63);   // <= kQ2Cap by construction
      {
        const int slot = inc - nent;
        for (int k = 0; k < nent; k++) w.q2[slot + k] = ent[k];
      }
      __builtin_amdgcn_s_waitcnt (0xc07f);
      __builtin_amdgcn_wave_barrier ();
      if (p.debug_stop == 3) continue;

      for (int b2 = 0; b2 < n2 && !overflow; b2 += 64) {
        const int nq = min (n2 - b2, 64);
        // (fetched here, not at the top of the kernel: three registers less to carry through stages A to C)
        const uint32_t sbL0 = c_single_bit[bl], sbL1 = bl < 48 ? c_single_bit[64 + bl] : 0u, sbS = bl < 56 ? c_single_bit[56 + bl] : 0u;
        // ---- stage D1: 56/112-bit slicing + CRC syndrome, one candidate at a time, one lane per message bit ----
        for (int ci = 0; ci < nq; ci++) {
          const uint32_t qe = uint32_t (__builtin_amdgcn_readfirstlane (int (w.q2[b2 + ci])));
          const int coff = qe & 0xFFF, t = 4 + int ((qe >> 12) & 7), df = int (qe >> 16);
          const bool longmsg = slice_bytes_for_df (df) == kLongBytes;   // 14 bytes sliced, else 7 (ref: src/demod-2400.c:416-435)
          const bool longcrc = msg_bits_for_df (df) == kLongBits;       // bits the score's CRC runs over
          if (!PE && coff < c_end) { if (lane == ci) r_syn = kSynSkipped; continue; }
          const TileRef<kR8> m = V.at (coff + 19);
          // syndrome = XOR of the single-bit syndromes of the set bits (the CRC is linear, ref: src/crc.c:56-61)
          const bool bit0 = slice_at (m, t + 12 * bl) > 0;
          u64 m0 = __ballot (bit0), m1 = 0;
          uint32_t x = bit0 ? (longcrc ? sbL0 : sbS) : 0u;
          if (longmsg) {
            const bool bit1 = bl < 48 && slice_at (m, t + 12 * (64 + bl)) > 0;
            m1 = __ballot (bit1);
            if (longcrc) x ^= bit1 ? sbL1 : 0u;
          } else m0 &= (1ull << 56) - 1;
          const uint32_t syn = uint32_t (__builtin_amdgcn_readlane (int (wave_xor_to63 (x)), 63));
          if (!PE) {
            const int next = coff + (longcrc ? kLongBits : kShortBits) * 12 / 5 + 1;
            if (syn == 0 && (df == 17 || df == 18 || df == 11) && coff >= u_end) c_end = next;
            else u_end = max (u_end, next);
          }
          if (lane == ci) { r_m0lo = uint32_t (m0); r_m0hi = uint32_t (m0 >> 32); r_m1lo = uint32_t (m1); r_m1hi = uint32_t (m1 >> 32); r_syn = syn; }
        }
        // ---- stage D2: classes, error-table diagnosis, records; one lane per candidate ----
        const bool act = lane < nq;
        const uint32_t qe = act ? w.q2[b2 + lane] : 0u;
        const int coff = qe & 0xFFF, t = 4 + int ((qe >> 12) & 7), df = int (qe >> 16);
        const bool longcrc = msg_bits_for_df (df) == kLongBits;
        const uint32_t syn = r_syn;
        const u64 m0 = (u64 (r_m0hi) << 32) | r_m0lo, m1 = (u64 (r_m1hi) << 32) | r_m1lo;
        const u64 crc_bits = longcrc ? (m0 | m1) : (m0 & ((1ull << 56) - 1));
        bool keep = act && crc_bits != 0 && syn != kSynSkipped;                               // all-zero message -> -2 (ref: src/dump1090.c:3295)
        int cls, nerr = 0, b0 = 0, b1 = 0;
        const bool es = df == 17 || df == 18, acq = df == 11;
        const uint32_t key = acq ? (syn & 0xFFFF80u) : syn;
        const ErrEntry *tab = longcrc ? p.long_tab : p.short_tab;
        const int tn = longcrc ? p.long_n : p.short_n;
        bool need = keep && (es || acq) && key != 0;
        if (es || acq) cls = acq ? kClsDF11 : kClsES;
        else           cls = (df == 20 || df == 21) ? kClsAP2 : kClsAP;
        if (need) {                                                     // 64 Kbit filter in front of the table
          const uint32_t h = (key ^ (key >> 12)) & 0xFFFF;
          const uint32_t *bloom = longcrc ? p.bloom_long : p.bloom_short;
          need = tn > 0 && ((bloom[h >> 5] >> (h & 31)) & 1);
          if (!need) keep = false;                                      // uncorrectable (ref: src/dump1090.c:3315,3343)
        }
        for (u64 nm = __ballot (need); nm; nm &= nm - 1) {              // rare: wave-cooperative table search
          const int l = __builtin_ctzll (nm);
          const uint32_t lkey = uint32_t (__builtin_amdgcn_readlane (int (key), l));
          const bool llong = (__ballot (longcrc) >> l) & 1;
          const ErrEntry *ltab = llong ? p.long_tab : p.short_tab;
          const int idx = table_find (ltab, llong ? p.long_n : p.short_n, lkey, lane);
          if (lane == l) {
            if (idx >= 0) {
              const ErrEntry te = ltab[idx];
              if (acq && te.nerr > 1) keep = false;                     // ref: src/dump1090.c:3322
              else { nerr = te.nerr; b0 = te.bit0; b1 = te.bit1; }
            } else keep = false;
          }
        }
        (void) tab;
        // records go to the slots of the strip they lie in; the batch is in ascending order, so first strip's come first
        const bool second = kPair > 1 && coff >= kStrip;
        const u64 kept = __ballot (keep), kept_hi = kPair > 1 ? __ballot (keep && second) : 0ull;
        const u64 mine = second ? kept_hi : kept & ~kept_hi;
        const uint32_t slot = (second ? nrec[kPair - 1] : nrec[0]) + uint32_t (__popcll (mine & ((1ull << lane) - 1)));
        const uint32_t total0 = nrec[0] + uint32_t (__popcll (kept & ~kept_hi)), total1 = nrec[kPair - 1] + uint32_t (__popcll (kept_hi));
        if (total0 > p.cap || total1 > p.cap) { overflow = true; break; }
        if (keep && p.write_records) {
          // Record layout: off,phase,df | cls,nerr,bit0,bit1 | crc | msg[0..13] | flags,pad
          uint4 lo4, hi4;
          lo4.x = uint32_t (coff & (kStrip - 1)) | (uint32_t (t) << 16) | (uint32_t (df) << 24);
          lo4.y = uint32_t (cls) | (uint32_t (uint8_t (nerr)) << 8) | (uint32_t (uint8_t (b0)) << 16) | (uint32_t (uint8_t (b1)) << 24);
          lo4.z = syn;
          lo4.w = r_m0lo;
          hi4.x = r_m0hi; hi4.y = r_m1lo; hi4.z = r_m1hi & 0xFFFFu; hi4.w = 0;
          uint4 *dst = reinterpret_cast<uint4 *> (out + (second ? p.cap : 0u) + slot);
          dst[0] = lo4; dst[1] = hi4;
        }
        nrec[0] = total0; if (kPair > 1) nrec[kPair - 1] = total1;
      }
    }
  }
  if (p.debug_stop == 3) { if (lane == 0) p.counts[strip] = valid_preamble[0] + valid_preamble[kPair - 1]; return; }

#pragma unroll
  for (int h = 0; h < kPair; h++) {
    if (strip + h >= strip_end) break;
    if (p.vp_bitmap) reinterpret_cast<uint32_t *> (p.vp_bitmap)[size_t (strip + h) * (kStrip / 32) + lane] = w.vp[64 * h + lane];
    if (lane == 0) {      // per-strip results only: no contended atomics
      p.counts[strip + h] = overflow ? 0xFFFFFFFFu : nrec[h];
      p.vp_counts[strip + h] = valid_preamble[h];
    }
  }
}

// ------------------------------------------------------------------------------------------
// stand-alone stateless converters (convert_func on device buffers)
// ------------------------------------------------------------------------------------------
template <int CONV>
__global__ __launch_bounds__ (256)
void convert_kernel (const uint8_t *iq, uint16_t *mag, size_t n, const uint16_t *lut256, unsigned long long *power_u64)
{
  __shared__ uint16_t lut[256];
  if (CONV == kConvUc8Lut || CONV == kConvUc8LutPower) {
    lut[threadIdx.x] = lut256[threadIdx.x];
    __syncthreads ();
  }
  u64 power = 0;
  const size_t stride = size_t (gridDim.x) * blockDim.x;
  for (size_t i = size_t (blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint16_t m;
    if (CONV == kConvUc8Lut || CONV == kConvUc8LutPower) {
      m = lut[iq[i]];
      if (CONV == kConvUc8LutPower) power += uint32_t (m) * uint32_t (m);
    } else if (CONV == kConvUc8Generic) {
      m = mag_from_float (c_uc8_float[iq[2 * i]], c_uc8_float[iq[2 * i + 1]], nullptr);
    } else {
      const int16_t *s = reinterpret_cast<const int16_t *> (iq) + 2 * i;
      if (CONV == kConvSc16Int) m = mag_sc16_int (s[0], s[1]);
      else {
        const float sc = CONV == kConvSc16Generic ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
        m = mag_from_float (float (s[0]) * sc, float (s[1]) * sc, nullptr);
      }
    }
    mag[i] = m;
  }
  if (CONV == kConvUc8LutPower) {
    for (int d = 32; d >= 1; d >>= 1) power += __shfl_xor (power, d, 64);
    if ((threadIdx.x & 63) == 0 && power) atomicAdd (power_u64, power);
  }
}

// Sum of squared magnitudes V[pos[i] + 19 .. + len[i]) for accepted messages
// (ref: src/demod-2400.c:486-497).  One wave per message.
template <int CONV>
__global__ __launch_bounds__ (256)
void rssi2400_kernel (const uint8_t *iq, uint64_t nsamples, const uint16_t *halo, uint32_t ov, uint32_t block_samples,
                      const uint16_t *lut256, const uint64_t *pos, const uint32_t *len, uint64_t *sum_out, uint32_t nmsg)
{
  const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= nmsg) return;
  const long long base = (long long) pos[i] + 19;
  const int n = int (len[i]);
  u64 acc = 0;
  for (int k = lane; k < n; k += 64) {
    const uint32_t m = mag_of_V<CONV> (iq, nsamples, halo, ov, block_samples, lut256, base + k);
    acc += u64 (m * m);
  }
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor (acc, d, 64);
  if (lane == 0) sum_out[i] = acc;
}

// ------------------------------------------------------------------------------------------
// exclusive scan of the per-strip record counts (overflowed strips count as 0) + gather
// ------------------------------------------------------------------------------------------
constexpr int kScanChunk = 1024;

__device__ __forceinline__ uint32_t clean_count (uint32_t c) { return c == 0xFFFFFFFFu ? 0u : c; }

__global__ __launch_bounds__ (256)
void chunk_sums_kernel (const uint32_t *counts, const uint32_t *vp_counts, uint32_t n, uint64_t *chunk_sum,
                        uint64_t *chunk_vp, uint64_t *chunk_ovf)
{
  __shared__ uint32_t part[3][4];
  const uint32_t base = blockIdx.x * kScanChunk;
  uint32_t s = 0, v = 0, o = 0;
  for (uint32_t i = threadIdx.x; i < kScanChunk; i += 256)
    if (base + i < n) { const uint32_t c = counts[base + i]; s += clean_count (c); o += c == 0xFFFFFFFFu; v += vp_counts[base + i]; }
  for (int d = 32; d >= 1; d >>= 1) { s += __shfl_xor (int (s), d, 64); v += __shfl_xor (int (v), d, 64); o += __shfl_xor (int (o), d, 64); }
  if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = s; part[1][threadIdx.x >> 6] = v; part[2][threadIdx.x >> 6] = o; }
  __syncthreads ();
  if (threadIdx.x == 0) {
    chunk_sum[blockIdx.x] = u64 (part[0][0]) + part[0][1] + part[0][2] + part[0][3];
    chunk_vp[blockIdx.x]  = u64 (part[1][0]) + part[1][1] + part[1][2] + part[1][3];
    chunk_ovf[blockIdx.x] = u64 (part[2][0]) + part[2][1] + part[2][2] + part[2][3];
  }
}

// one wave: exclusive scan of chunk sums in place; totals go to totals[0..2] = records, valid preambles, overflowed strips.
// (No barriers: the launch often shares the GPU with a scan kernel, where a 16-wave workgroup waits long for a CU.)
__global__ __launch_bounds__ (64)
void scan_chunk_sums_kernel (uint64_t *chunk_sum, const uint64_t *chunk_vp, const uint64_t *chunk_ovf, uint32_t nchunks,
                             uint64_t *totals)
{
  __builtin_amdgcn_s_setprio (3);
  const int lane = threadIdx.x;
  uint64_t carry = 0, vp = 0, ovf = 0;
  for (uint32_t base = 0; base < nchunks; base += 64) {
    const uint32_t i = base + lane;
    const uint64_t v = i < nchunks ? chunk_sum[i] : 0;
    if (i < nchunks) { vp += chunk_vp[i]; ovf += chunk_ovf[i]; }
    uint64_t incl = v;
    for (int d = 1; d < 64; d <<= 1) { const uint64_t o = __shfl_up (incl, d, 64); if (lane >= d) incl += o; }
    if (i < nchunks) chunk_sum[i] = carry + incl - v;
    carry += __shfl (incl, 63, 64);
  }
  for (int d = 32; d >= 1; d >>= 1) { vp += __shfl_xor (vp, d, 64); ovf += __shfl_xor (ovf, d, 64); }
  if (lane == 0) { totals[0] = carry; totals[1] = vp; totals[2] = ovf; }
}

// one wave per strip: offsets[strip] = chunk base + scan inside the chunk; copies the strip's records
__global__ __launch_bounds__ (256)
void gather_kernel (const uint32_t *counts, uint32_t nstrips, const uint64_t *chunk_sum, const Record *records,
                    uint32_t cap, Record *packed, uint64_t *packed_pos, uint64_t *offsets)
{
  const uint32_t strip = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (strip >= nstrips) return;
  const uint32_t chunk = strip / kScanChunk, first = chunk * kScanChunk;
  uint64_t before = 0;
  for (uint32_t i = first + lane; i < strip; i += 64) before += clean_count (counts[i]);
  for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor (before, d, 64);
  const uint64_t off = chunk_sum[chunk] + before;
  const uint32_t cnt = clean_count (counts[strip]);
  if (lane == 0) offsets[strip] = off;
  const uint4 *src = reinterpret_cast<const uint4 *> (records + size_t (strip) * cap);
  uint4 *dst = reinterpret_cast<uint4 *> (packed + off);
  for (uint32_t i = lane; i < cnt * 2; i += 64) dst[i] = src[i];
  for (uint32_t i = lane; i < cnt; i += 64) packed_pos[off + i] = uint64_t (strip) * kStrip + records[size_t (strip) * cap + i].off;
}


// ------------------------------------------------------------------------------------------
// float converters with the DC block on (ref: src/convert.c:112-245).  The one-pole filter
// z1 = f*DC_A + z1*DC_B is evaluated in the reference's order, one rounding per operation:
//   dc_zscan_kernel : one wave; all lanes stage f*DC_A into LDS, lanes 0/1 run the I and Q chains
//   dc_mag_kernel   : wide; magnitude (and mag_sq) from the stored z1 sequences
//   power_seq_kernel: one lane adds mag_sq in sample order (float, like `power += mag_sq`)
// ------------------------------------------------------------------------------------------
template <int CONV>
__device__ __forceinline__ void sample_to_float (const uint8_t *iq, size_t i, float *fI, float *fQ)
{
  if (CONV == kConvUc8Generic) { *fI = c_uc8_float[iq[2 * i]]; *fQ = c_uc8_float[iq[2 * i + 1]]; }
  else {
    const int16_t *s = reinterpret_cast<const int16_t *> (iq) + 2 * i;
    const float sc = CONV == kConvSc16Generic ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
    *fI = float (s[0]) * sc; *fQ = float (s[1]) * sc;
  }
}

constexpr int kZChunk = 2048;

template <int CONV>
__global__ __launch_bounds__ (64)
void dc_zscan_kernel (const uint8_t *iq, size_t n, float A, float B, float zI0, float zQ0, float *zI, float *zQ, float *z_final)
{
  __shared__ float c[2][kZChunk];
  const int lane = threadIdx.x;
  float z = lane == 0 ? zI0 : zQ0;
  for (size_t base = 0; base < n; base += kZChunk) {
    const int cnt = int (n - base < size_t (kZChunk) ? n - base : size_t (kZChunk));
    for (int i = lane; i < cnt; i += 64) {
      float fI, fQ;
      sample_to_float<CONV> (iq, base + i, &fI, &fQ);
      c[0][i] = __fmul_rn (fI, A);
      c[1][i] = __fmul_rn (fQ, A);
    }
    __syncthreads ();
    if (lane < 2) {
      float *cc = c[lane];
      int i = 0;
      for (; i + 4 <= cnt; i += 4) {
        const float4 v = *reinterpret_cast<const float4 *> (cc + i);
        float4 o;
        z = __fadd_rn (v.x, __fmul_rn (z, B)); o.x = z;
        z = __fadd_rn (v.y, __fmul_rn (z, B)); o.y = z;
        z = __fadd_rn (v.z, __fmul_rn (z, B)); o.z = z;
        z = __fadd_rn (v.w, __fmul_rn (z, B)); o.w = z;
        *reinterpret_cast<float4 *> (cc + i) = o;
      }
      for (; i < cnt; i++) { z = __fadd_rn (cc[i], __fmul_rn (z, B)); cc[i] = z; }
    }
    __syncthreads ();
    for (int i = lane; i < cnt; i += 64) { zI[base + i] = c[0][i]; zQ[base + i] = c[1][i]; }
    __syncthreads ();
  }
  if (lane < 2) z_final[lane] = z;
}

template <int CONV>
__global__ __launch_bounds__ (256)
void dc_mag_kernel (const uint8_t *iq, size_t n, const float *zI, const float *zQ, uint16_t *mag, float *mag_sq)
{
  const size_t stride = size_t (gridDim.x) * blockDim.x;
  for (size_t i = size_t (blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    float fI, fQ, sq;
    sample_to_float<CONV> (iq, i, &fI, &fQ);
    fI = __fsub_rn (fI, zI[i]);
    fQ = __fsub_rn (fQ, zQ[i]);
    mag[i] = mag_from_float (fI, fQ, &sq);
    if (mag_sq) mag_sq[i] = sq;
  }
}

// stateless variant of the same converters (DC_A = 0, DC_B = 1: z1 stays 0) that also stores mag_sq
template <int CONV>
__global__ __launch_bounds__ (256)
void nodc_mag_kernel (const uint8_t *iq, size_t n, uint16_t *mag, float *mag_sq)
{
  const size_t stride = size_t (gridDim.x) * blockDim.x;
  for (size_t i = size_t (blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    float fI, fQ, sq;
    sample_to_float<CONV> (iq, i, &fI, &fQ);
    mag[i] = mag_from_float (fI, fQ, &sq);
    if (mag_sq) mag_sq[i] = sq;
  }
}

__global__ __launch_bounds__ (64)
void power_seq_kernel (const float *mag_sq, size_t n, float *out)
{
  if (threadIdx.x != 0) return;
  float p = 0.0f;
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    const float4 a = *reinterpret_cast<const float4 *> (mag_sq + i), b = *reinterpret_cast<const float4 *> (mag_sq + i + 4);
    p = __fadd_rn (p, a.x); p = __fadd_rn (p, a.y); p = __fadd_rn (p, a.z); p = __fadd_rn (p, a.w);
    p = __fadd_rn (p, b.x); p = __fadd_rn (p, b.y); p = __fadd_rn (p, b.z); p = __fadd_rn (p, b.w);
  }
  for (; i < n; i++) p = __fadd_rn (p, mag_sq[i]);
  *out = p;
}


// ------------------------------------------------------------------------------------------
// Ordered accept pass on the device (what demod_2400's loop does with every candidate that
// reached modeS_message_score: ICAO cache tests/updates, decode_mode_S_message's accept rules,
// skip-ahead; ref: src/demod-2400.c:438-518, src/dump1090.c:2372-2578,2676-2687,3279-3365).
//
// One wave per 256 KiB block.  The wave keeps the block's running ICAO cache in LDS, seeded from
// snap_in[block].  Blocks depend on each other only through that cache, so the host iterates:
// accept -> propagate the per-block writes (delta) down the block axis -> accept again, until no
// delta changes; decisions at a position depend only on earlier positions, so the fixed point is
// the sequential result.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t icao_hash (uint32_t a)      // ref: src/dump1090.c:1438-1447
{
  a = ((a >> 16) ^ a) * 0x45D9F3Bu;
  a = ((a >> 16) ^ a) * 0x45D9F3Bu;
  a = (a >> 16) ^ a;
  return a & (kIcaoSlots - 1);
}

// MODES_NON_ICAO_ADDRESS for DF17/18 (ref: src/dump1090.c:1918-1963,1990,2097,2130,2220,2239); m = message words, little endian
__device__ __forceinline__ uint32_t non_icao_flag_dev (const uint8_t *msg, int df)
{
  bool look = false;
  if (df == 18) {
    const int cf = msg[0] & 7;
    if (cf == 1 || cf == 5 || cf == 4 || cf == 7) return kNonIcao;
    if (cf == 3) return (msg[4] & 0x80) ? kNonIcao : 0u;
    look = cf == 2 || cf == 6;
  }
  if (!look) return 0;
  const int type = msg[4] >> 3;
  bool imf = false;
  if (type == 19) imf = msg[5] & 0x80;
  else if (type >= 5 && type <= 8) imf = msg[6] & 0x08;
  else if (type == 0 || (type >= 9 && type <= 18) || (type >= 20 && type <= 22)) imf = msg[4] & 0x01;
  else if (type == 28) imf = (msg[4] & 7) == 1 && (msg[10] & 0x01);
  else if (type == 31) imf = msg[10] & 0x01;
  return imf ? kNonIcao : 0u;
}

struct AcceptLds {
  uint32_t cache[kIcaoSlots];
  uint32_t written[kIcaoSlots / 32];
  uint32_t counter[kBlkStats];
  uint32_t prefix[64];        // records in the block's earlier strips
  uint32_t accn[64];          // accepted messages per strip
};

__device__ __forceinline__ uint32_t rdlane (uint32_t v, int l) { return uint32_t (__builtin_amdgcn_readlane (int (v), l)); }

// One wave per workgroup and 5 KiB of LDS: the pass of one range of blocks runs while the next range is scanned, and a
// workgroup this small finds room on a CU the scan kernel has filled.
template <int MODE>
__global__ __launch_bounds__ (64)
void accept_kernel (const AcceptParams p)
{
  __shared__ AcceptLds lds[1];
  const int lane = threadIdx.x & 63, wv = 0;
  const uint32_t blk = p.blk_begin + blockIdx.x;
  if (blk >= (p.blk_end ? p.blk_end : p.nblocks)) return;
  if (p.dirty && !p.dirty[blk]) return;                      // same entry cache as in the round before: same results
  __builtin_amdgcn_s_setprio (3);                            // the pass is latency bound and often shares the GPU with a scan
  AcceptLds &w = lds[wv];
  const uint32_t first_strip = blk * p.strips_per_block;
  const uint32_t last_strip = min (first_strip + p.strips_per_block, p.nstrips);
  // the block's records: counts of its strips (<= 64 of them)
  const uint32_t nst = last_strip - first_strip;
  const uint32_t my_count = uint32_t (lane) < nst ? clean_count (p.counts[first_strip + lane]) : 0u;
  const uint32_t my_incl = uint32_t (wave_incl_scan (int (my_count), lane));
  const uint32_t total = uint32_t (__builtin_amdgcn_readlane (int (my_incl), 63));
  if (total == 0) {
    // nothing to walk (most blocks of a quiet recording): the block accepts nothing and writes no slot; the 4 KB of its
    // entry cache are not even fetched
    if (uint32_t (lane) < nst && p.acc_counts[first_strip + lane] != 0) p.acc_counts[first_strip + lane] = 0;
    const bool prev = !p.first_round && p.blk_wrote[blk] != 0;
    if (lane == 0 && (p.first_round || prev)) p.blk_wrote[blk] = 0;
    if (lane == 0 && prev) atomicOr (p.changed, 1u);
    if (lane < kBlkStats) p.blk_stats[size_t (blk) * kBlkStats + lane] = 0;
    return;
  }
  const uint32_t *entry = p.first_round ? p.cache_in : p.snap_in + size_t (blk) * kIcaoSlots;     // the cache as the block starts with it
  for (int i = lane; i < int (kIcaoSlots); i += 64) w.cache[i] = entry[i];
  if (lane < int (kIcaoSlots / 32)) w.written[lane] = 0;
  if (lane < kBlkStats) w.counter[lane] = 0;
  __builtin_amdgcn_s_waitcnt (0xc07f);
  __builtin_amdgcn_wave_barrier ();

  const uint64_t blk_pos0 = uint64_t (blk) * p.block_samples;
  const uint32_t blk_len = uint32_t (min (uint64_t (p.block_samples), p.nsamples - blk_pos0));
  uint32_t skip_until = 0;                                   // each demod_2400 call starts at j = 0
  uint32_t st_demod = 0, st_oop = 0, st_good = 0, st_bad = 0, st_fix1 = 0, st_fix2 = 0, st_unk = 0, st_vp = 0, st_modeac = 0;

  // the block's records, 64 at a time across its strips: exclusive prefix of the counts in LDS, record g -> (strip, index)
  w.prefix[lane] = my_incl - my_count;
  w.accn[lane] = 0;
  __builtin_amdgcn_s_waitcnt (0xc07f);
  __builtin_amdgcn_wave_barrier ();
  // A block whose entry cache changed repeats itself all the same unless the change shows where a record looks: every
  // decision and score reads the cache either at a slot the block wrote before (same history, same value) or as the block
  // found it.  Record.flags bit 1 keeps "entry[slot] == address" from the block's last pass; if it holds for every record
  // under the new entry cache, the last pass's results stand.
  if (p.dirty) {
    bool differs = false;
    for (uint32_t base = 0; base < total; base += 64) {
      const uint32_t g = base + lane;
      const bool have = g < total;
      uint32_t sl = 0;
#pragma unroll
      for (uint32_t step = 32; step >= 1; step >>= 1)
        if (sl + step < nst && w.prefix[sl + step] <= g) sl += step;
      const Record *rp = p.records + size_t (first_strip + sl) * p.cap + (have ? g - w.prefix[sl] : 0u);
      Record r;
      if (have) { const uint4 *s4 = reinterpret_cast<const uint4 *> (rp); *reinterpret_cast<uint4 *> (&r) = s4[0]; *(reinterpret_cast<uint4 *> (&r) + 1) = s4[1]; }
      else memset (&r, 0, sizeof r);
      uint32_t aa = (uint32_t (r.msg[1]) << 16) | (uint32_t (r.msg[2]) << 8) | r.msg[3];
      if (r.nerr >= 1 && r.bit0 >= 8 && r.bit0 <= 31) aa ^= 1u << (31 - r.bit0);
      if (r.nerr >= 2 && r.bit1 >= 8 && r.bit1 <= 31) aa ^= 1u << (31 - r.bit1);
      const uint32_t test_addr = (r.cls == kClsAP || r.cls == kClsAP2) ? r.crc : aa;
      const uint32_t ev = w.cache[icao_hash (test_addr)];         // nothing written yet: still the entry cache
      const bool eq = ev != 0 && ev == test_addr;
      if (have && eq != bool ((r.flags >> 1) & 1)) differs = true;
    }
    if (!__ballot (differs)) return;
  }
  {
    for (uint32_t base = 0; base < total; base += 64) {
      const uint32_t g = base + lane;
      const bool have = g < total;
      uint32_t sl = 0;                                                   // largest strip with prefix <= g
#pragma unroll
      for (uint32_t step = 32; step >= 1; step >>= 1)
        if (sl + step < nst && w.prefix[sl + step] <= g) sl += step;
      const uint32_t strip = first_strip + sl;
      Record *rp = p.records + size_t (strip) * p.cap + (have ? g - w.prefix[sl] : 0u);
      Record r;
      if (have) { const uint4 *s4 = reinterpret_cast<const uint4 *> (rp); *reinterpret_cast<uint4 *> (&r) = s4[0]; *(reinterpret_cast<uint4 *> (&r) + 1) = s4[1]; }
      else memset (&r, 0, sizeof r);
      const uint32_t j = sl * kStrip + r.off;
      // --- per-record facts ---
      uint32_t aa = (uint32_t (r.msg[1]) << 16) | (uint32_t (r.msg[2]) << 8) | r.msg[3];
      bool aa_changed = false;
      if (r.nerr >= 1 && r.bit0 >= 8 && r.bit0 <= 31) { aa ^= 1u << (31 - r.bit0); aa_changed = true; }
      if (r.nerr >= 2 && r.bit1 >= 8 && r.bit1 <= 31) { aa ^= 1u << (31 - r.bit1); aa_changed = true; }
      const bool ap = r.cls == kClsAP || r.cls == kClsAP2, acq = r.cls == kClsDF11;
      const uint32_t iid = r.crc & 0x7F, masked = r.crc & 0xFFFF80u;
      const uint32_t test_addr = ap ? r.crc : aa;
      const uint32_t slot = icao_hash (test_addr);
      bool uncond, need_known, adder;
      if (ap)       { uncond = false; need_known = true; adder = false; }
      else if (acq) {
        // demod_2400 scores first: a clean DF11 with IID != 0 needs a known address (ref: src/dump1090.c:3332-3338);
        // demod_2000 goes straight to decode_mode_S_message, which takes any clean DF11 (ref: src/dump1090.c:2471-2490)
        uncond = masked == 0 && (MODE == 2000 || iid == 0); need_known = !uncond; adder = masked == 0 && iid == 0;
      }
      else          { uncond = r.crc == 0 || (r.nerr == 1 && !aa_changed); need_known = !uncond; adder = r.crc == 0; }
      // demod_2000: records exist for every valid preamble; only some reach decode_mode_S_message
      const bool modea = MODE == 2000 && r.cls == kClsModeA;             // emitted as found (ref: src/demod-2000.c:677-701)
      const bool decodes = MODE != 2000 || ((r.phase & kInfoDecode) && !(r.phase & kInfoStaticReject));
      if (modea) { uncond = true; need_known = false; adder = false; }
      uncond = uncond && have && (decodes || modea); need_known = need_known && have && decodes; adder = adder && have && decodes;
      const uint32_t add_val = acq ? aa : (aa | non_icao_flag_dev (r.msg, r.df));
      const uint32_t add_slot = icao_hash (add_val);
      const uint32_t skip_len = modea ? 50u                                             // ref: src/demod-2000.c:698
                              : MODE == 2000 ? 2u * (8u + r.pad[2]) - 1u                // ref: src/demod-2000.c:997
                                             : uint32_t (msg_bits_for_df (r.df)) * 12 / 5;   // ref: src/demod-2400.c:514
      const uint32_t c0 = w.cache[slot], e0 = entry[slot];
      bool known = have && c0 != 0 && c0 == test_addr;
      const bool eq_entry = e0 != 0 && e0 == test_addr;

      // --- which records can change anything: acceptable now, or their slot is written inside this batch ---
      bool touched = false;
      for (u64 am = __ballot (adder); am; am &= am - 1) {
        const int a = __builtin_ctzll (am);
        if (need_known && lane > a && slot == rdlane (add_slot, a)) touched = true;
      }
      u64 todo = __ballot (uncond || (need_known && known) || touched);
      u64 accepted = 0;
      uint32_t my_skip = 0;
      const uint32_t skip_in = skip_until;
      while (todo) {
        const int i = __builtin_ctzll (todo);
        todo &= todo - 1;
        const uint32_t ji = rdlane (j, i);
        if (ji < skip_until) continue;
        const uint32_t ta = rdlane (test_addr, i), sl = rdlane (slot, i);
        const uint32_t cur = w.cache[sl];
        const bool kn = cur != 0 && cur == ta;
        if (lane == i) known = kn;
        const bool acc = ((__ballot (uncond) >> i) & 1) || kn;
        if (!acc) continue;
        accepted |= 1ull << i;
        skip_until = ji + rdlane (skip_len, i) + 1;
        if (lane == i) my_skip = skip_until;
        if ((__ballot (adder) >> i) & 1) {                                            // ref: src/dump1090.c:2676-2687
          const uint32_t as = rdlane (add_slot, i), av = rdlane (add_val, i);
          if (lane == 0) { w.cache[as] = av; atomicOr (&w.written[as >> 5], 1u << (as & 31)); }
          __builtin_amdgcn_s_waitcnt (0xc07f);
          __builtin_amdgcn_wave_barrier ();
        }
      }
      // --- what the reference would have counted for every record of the batch ---
      const u64 before = accepted & ((1ull << lane) - 1);
      uint32_t skip_at = skip_in;
      {
        const int prev = before ? 63 - __builtin_clzll (before) : 0;
        const uint32_t v = uint32_t (__shfl (int (my_skip), prev, 64));
        if (before) skip_at = v;
      }
      const bool reached = have && j >= skip_at;
      int score;
      if (ap)       score = known ? 1000 : (r.cls == kClsAP ? -1 : -2);
      else if (acq) score = iid == 0 ? (known ? 1600 : 750) / (r.nerr + 1) : (known ? 1000 / (r.nerr + 1) : -1);
      else          score = (known ? 1800 : 1400) / (r.nerr + 1);
      const bool is_acc = (accepted >> lane) & 1;
      if (MODE == 2000) {
        score = -1;
        const bool called = reached && !modea && (r.phase & kInfoDecode);          // decode_mode_S_message ran
        st_vp    += __popcll (__ballot (reached && !modea));                         // ref: src/demod-2000.c:751
        st_modeac += __popcll (__ballot (reached && modea));                         // ref: src/demod-2000.c:699
        const uint32_t inc = reached && !modea ? uint32_t (r.phase & kInfoDemodMask) : 0u;   // ref: src/demod-2000.c:848,934
        st_demod += __popcll (__ballot (inc & 1)) + 2 * __popcll (__ballot (inc & 2));
        st_bad   += __popcll (__ballot (called && !is_acc));
        st_unk   += __popcll (__ballot (called && !is_acc && !(r.phase & kInfoStaticReject)));
      } else {
        const bool scored = reached && score >= 0;
        st_oop   += __popcll (__ballot (reached && score > -2));
        st_demod += __popcll (__ballot (scored));
        st_bad   += __popcll (__ballot (scored && !is_acc));
        st_unk   += __popcll (__ballot (scored && !is_acc));      // every reject past the score is "unknown ICAO" (-1)
      }
      const int ebits = (acq ? (masked ? r.nerr : 0) : ap ? 0 : (r.crc ? r.nerr : 0));
      st_good  += __popcll (__ballot (is_acc && ebits == 0 && !modea));
      st_fix1  += __popcll (__ballot (is_acc && ebits == 1));
      st_fix2  += __popcll (__ballot (is_acc && ebits == 2));
      if (is_acc) atomicAdd (&w.accn[sl], 1u);
      // write back: flags bit0 = accepted, bit1 = address found in the entry cache, pad[0..1] = score
      if (have) {
        const uint8_t nf = (is_acc ? 1 : 0) | (eq_entry ? 2 : 0);
        const int16_t sc16 = int16_t (score);
        if (r.flags != nf || *reinterpret_cast<const int16_t *> (r.pad) != sc16) {
          rp->flags = nf;
          *reinterpret_cast<int16_t *> (rp->pad) = sc16;
        }
      }
      // valid preambles hidden inside the skipped-over part of accepted messages: positions (j, my_skip) never examined
      if (p.vp_bitmap && is_acc) {
        const uint32_t lo = j + 1, hi = min (my_skip, blk_len);
        if (lo < hi) {
          const uint64_t g0 = blk_pos0 + lo, g1 = blk_pos0 + hi;
          uint32_t c = 0;
          for (uint64_t wi = g0 >> 6; wi <= ((g1 - 1) >> 6); wi++) {
            u64 bits = p.vp_bitmap[wi];
            if (wi == (g0 >> 6)) bits &= ~0ull << (g0 & 63);
            if (wi == ((g1 - 1) >> 6) && (g1 & 63)) bits &= (1ull << (g1 & 63)) - 1;
            c += __popcll (bits);
          }
          if (c) atomicAdd (&w.counter[kStSkippedVp], c);
        }
      }
    }
  }
  __builtin_amdgcn_s_waitcnt (0xc07f);
  __builtin_amdgcn_wave_barrier ();
  if (uint32_t (lane) < nst) {
    const uint32_t nacc = w.accn[lane];
    if (p.acc_counts[first_strip + lane] != nacc) p.acc_counts[first_strip + lane] = nacc;
  }
  {
    uint32_t nacc = uint32_t (lane) < nst ? w.accn[lane] : 0u;
    for (int d = 32; d >= 1; d >>= 1) nacc += uint32_t (__shfl_xor (int (nacc), d, 64));
    if (lane == 0) w.counter[kStAccepted] = nacc;
  }
  __builtin_amdgcn_s_waitcnt (0xc07f);
  __builtin_amdgcn_wave_barrier ();
  // delta of this block + change detection against the previous round.  A block that wrote no slot only says so
  // (blk_wrote): its delta row is neither written nor read.
  const bool wrote = __ballot (lane < int (kIcaoSlots / 32) && w.written[lane & (kIcaoSlots / 32 - 1)] != 0) != 0;
  const bool prev = !p.first_round && p.blk_wrote[blk] != 0;
  bool diff = false;
  if (wrote) {
    for (int i = lane; i < int (kIcaoSlots); i += 64) {
      const bool wr = (w.written[i >> 5] >> (i & 31)) & 1;
      const uint32_t nv = wr ? (kDeltaWritten | w.cache[i]) : 0u;
      uint32_t *d = p.delta + size_t (blk) * kIcaoSlots + i;
      if (!prev) { *d = nv; diff = diff || nv != 0; }
      else if (*d != nv) { *d = nv; diff = true; }
    }
  } else diff = prev;
  if (lane == 0 && (p.first_round || prev != wrote)) p.blk_wrote[blk] = wrote ? 1u : 0u;
  if (__ballot (diff) && lane == 0) atomicOr (p.changed, 1u);
  if (lane == 0) {
    uint32_t *o = p.blk_stats + size_t (blk) * kBlkStats;
    o[kStDemodulated] = st_demod; o[kStOutOfPhase] = st_oop; o[kStGood] = st_good; o[kStBad] = st_bad;
    o[kStFix1] = st_fix1; o[kStFix2] = st_fix2; o[kStUnknown] = st_unk;
    o[kStSkippedVp] = w.counter[kStSkippedVp]; o[kStAccepted] = w.counter[kStAccepted]; o[kStVp] = st_vp; o[kStModeAC] = st_modeac;
  }
}

// ---- propagate the per-block cache writes down the block axis ----
constexpr int kSnapChunk = 32;

// The three kernels below run behind every accept round.  `changed` is that round's word: when no delta changed (in the
// first round: no block wrote anything) the snapshots and the cache behind the last block are what they were, no block
// is marked dirty, and the kernels leave at once -- a pass over a shard without accepted squitters costs no snapshot traffic.

// last write per (chunk of blocks, slot)
__global__ __launch_bounds__ (256)
void snap_chunk_last_kernel (const uint32_t *delta, const uint32_t *blk_wrote, uint32_t nblocks, uint32_t *chunk_last, uint32_t *dirty,
                             const uint32_t *changed)
{
  const uint32_t s = blockIdx.x * 256 + threadIdx.x, chunk = blockIdx.y;
  if (s >= kIcaoSlots) return;
  uint32_t v = 0;
  const uint32_t b0 = chunk * kSnapChunk, b1 = min (b0 + kSnapChunk, nblocks);
  // snap_fill_kernel, next in the stream, marks the blocks to repeat.  The marks are cleared after a round without a
  // change as well: the engine queues one round beyond the last, and that round must not touch a block.
  if (s < kSnapChunk && b0 + s < b1) dirty[b0 + s] = 0u;
  if (!*changed) return;
  for (uint32_t b = b0; b < b1; b++) {
    if (!blk_wrote[b]) continue;
    const uint32_t d = delta[size_t (b) * kIcaoSlots + s];
    if (d & kDeltaWritten) v = d;
  }
  chunk_last[size_t (chunk) * kIcaoSlots + s] = v;
}

// serial over chunks, one thread per slot: cache at the entry of every chunk; also the cache after the last block
__global__ __launch_bounds__ (256)
void snap_chunk_scan_kernel (const uint32_t *chunk_last, uint32_t nchunks, const uint32_t *cache_in, uint32_t *chunk_entry,
                             uint32_t *cache_out, const uint32_t *changed, int first_round)
{
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s >= kIcaoSlots) return;
  if (!*changed) { if (first_round) cache_out[s] = cache_in[s]; return; }
  uint32_t cur = cache_in[s];
  uint32_t c = 0;
  for (; c + 8 <= nchunks; c += 8) {
    uint32_t d[8];
#pragma unroll
    for (int k = 0; k < 8; k++) d[k] = chunk_last[size_t (c + k) * kIcaoSlots + s];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      chunk_entry[size_t (c + k) * kIcaoSlots + s] = cur;
      if (d[k] & kDeltaWritten) cur = d[k] & ~kDeltaWritten;
    }
  }
  for (; c < nchunks; c++) {
    chunk_entry[size_t (c) * kIcaoSlots + s] = cur;
    const uint32_t d = chunk_last[size_t (c) * kIcaoSlots + s];
    if (d & kDeltaWritten) cur = d & ~kDeltaWritten;
  }
  cache_out[s] = cur;
}

__global__ __launch_bounds__ (256)
void snap_fill_kernel (const uint32_t *delta, const uint32_t *blk_wrote, uint32_t nblocks, const uint32_t *chunk_entry, uint32_t *snap_in,
                       uint32_t *dirty, const uint32_t *changed, const uint32_t *cache_in, int first_round, uint32_t *marked, uint32_t *marked_host)
{
  if (!*changed) return;
  const uint32_t s = blockIdx.x * 256 + threadIdx.x, chunk = blockIdx.y;
  if (s >= kIcaoSlots) return;
  uint32_t cur = chunk_entry[size_t (chunk) * kIcaoSlots + s];
  const uint32_t first = cache_in[s];                              // what every block started from in the first round
  const uint32_t b0 = chunk * kSnapChunk, b1 = min (b0 + kSnapChunk, nblocks);
  bool any = false;
  for (uint32_t b = b0; b < b1; b++) {
    // a block whose entry cache is what it was in the round before would repeat itself: only the others run again
    uint32_t *sn = snap_in + size_t (b) * kIcaoSlots + s;
    if (first_round) { *sn = cur; if (cur != first) { dirty[b] = 1u; any = true; } }
    else if (*sn != cur) { *sn = cur; dirty[b] = 1u; any = true; }
    if (blk_wrote[b]) {
      const uint32_t d = delta[size_t (b) * kIcaoSlots + s];
      if (d & kDeltaWritten) cur = d & ~kDeltaWritten;
    }
  }
  // Deltas can change without any block's entry cache changing (a slot written again with what it held): then no block
  // is marked, the round just run was already the last one, and `marked` stays clear to say so.
  // (marked_host: the same word in pinned host memory, where the host looks when the round's event has fired)
  if (__ballot (any) && (threadIdx.x & 63) == 0) { atomicOr (marked, 1u); if (marked_host) *marked_host = 1u; }
}

// sums of the per-block statistics: out[k] = sum over blocks of blk_stats[block][k]
__global__ __launch_bounds__ (256)
void blk_stats_sum_kernel (const uint32_t *blk_stats, uint32_t nblocks, unsigned long long *out)
{
  unsigned long long acc[kBlkStats];
#pragma unroll
  for (int k = 0; k < kBlkStats; k++) acc[k] = 0;
  for (uint32_t b = blockIdx.x * 256 + threadIdx.x; b < nblocks; b += gridDim.x * 256) {
    const uint4 *q = reinterpret_cast<const uint4 *> (blk_stats + size_t (b) * kBlkStats);
    const uint4 a = q[0], c = q[1], d = q[2];
    acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w; acc[4] += c.x; acc[5] += c.y; acc[6] += c.z; acc[7] += c.w;
    acc[8] += d.x; acc[9] += d.y; acc[10] += d.z; acc[11] += d.w;
  }
#pragma unroll
  for (int k = 0; k < kBlkStats; k++) {
    unsigned long long v = acc[k];
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor (v, d, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd (&out[k], v);
  }
}

hipError_t launch_blk_stats_sum (const uint32_t *blk_stats, uint32_t nblocks, unsigned long long *out, hipStream_t stream)
{
  static_assert (kBlkStats == 12, "three uint4 per block");
  hipError_t e = hipMemsetAsync (out, 0, kBlkStats * sizeof (unsigned long long), stream);
  if (e != hipSuccess || nblocks == 0) return e;
  const uint32_t grid = (nblocks + 255) / 256 < 64 ? (nblocks + 255) / 256 : 64;
  hipLaunchKernelGGL (blk_stats_sum_kernel, dim3 (grid), dim3 (256), 0, stream, blk_stats, nblocks, out);
  return hipGetLastError ();
}

// ---- emit: accepted records -> modes_message, in order, with the signal level ----
struct MsgOut {            // mirrors modes_message (include/modes_hip.h), 56 bytes
  uint64_t timestamp; uint8_t msg[14]; uint8_t msg_bits, msg_type; uint32_t addr, crc; int16_t score; uint8_t error_bits, phase;
  uint32_t block; double sig_level; uint32_t j, reserved;
};
static_assert (sizeof (MsgOut) == 56, "modes_message layout");

// MODES_NON_ICAO_ADDRESS from the message held as two little-endian words (bytes 0..7, 8..13)
__device__ __forceinline__ uint32_t non_icao_flag_words (u64 lo, u64 hi, int df)
{
  uint8_t msg[14];
#pragma unroll
  for (int b = 0; b < 8; b++) msg[b] = uint8_t (lo >> (8 * b));
#pragma unroll
  for (int b = 0; b < 6; b++) msg[8 + b] = uint8_t (hi >> (8 * b));
  return non_icao_flag_dev (msg, df);
}

// offsets[strip] = messages of all earlier strips: chunk base + exclusive scan inside the 1024-strip chunk
__global__ __launch_bounds__ (256)
void strip_offsets_kernel (const uint32_t *acc_counts, uint32_t nstrips, const uint64_t *chunk_sum, uint64_t *offsets)
{
  __shared__ uint32_t wsum[4];
  const uint32_t base = blockIdx.x * kScanChunk + threadIdx.x * 4;
  uint32_t c[4];
#pragma unroll
  for (int k = 0; k < 4; k++) c[k] = base + k < nstrips ? acc_counts[base + k] : 0u;
  const uint32_t mine = c[0] + c[1] + c[2] + c[3];
  const uint32_t incl = uint32_t (wave_incl_scan (int (mine), threadIdx.x & 63));
  if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads ();
  uint32_t before = incl - mine;
  for (uint32_t wv = 0; wv < (threadIdx.x >> 6); wv++) before += wsum[wv];
  uint64_t o = chunk_sum[blockIdx.x] + before;
#pragma unroll
  for (int k = 0; k < 4; k++) { if (base + k < nstrips) offsets[base + k] = o; o += c[k]; }
}

hipError_t launch_strip_offsets (const uint32_t *acc_counts, uint32_t nstrips, const uint64_t *chunk_sum, uint64_t *offsets, hipStream_t stream)
{
  if (nstrips == 0) return hipSuccess;
  static_assert (kScanChunk == 1024, "256 threads x 4 strips");
  hipLaunchKernelGGL (strip_offsets_kernel, dim3 ((nstrips + kScanChunk - 1) / kScanChunk), dim3 (256), 0, stream, acc_counts, nstrips, chunk_sum, offsets);
  return hipGetLastError ();
}

__device__ __forceinline__ uint32_t bcast (uint32_t v, int l) { return uint32_t (__builtin_amdgcn_readlane (int (v), l)); }

constexpr int kEmitStrips = 16;

template <int CONV>
__global__ __launch_bounds__ (256)
void emit2400_kernel (const EmitParams p)
{
  // A wave looks at kEmitStrips strips at once (most hold no accepted message), FOUR at a time: a message's signal
  // level is a sum over at most 268 samples, and what a message costs is the latency of fetching its record and then
  // its samples, so each 16-lane row of the wave takes a strip of its own and walks that strip's accepted records --
  // four chains of dependent loads in flight per wave instead of one.  Everything "uniform" below is uniform per row.
  constexpr bool kEmitLut = CONV == kConvUc8Lut || CONV == kConvUc8LutPower;
  __shared__ uint16_t s_lut[kEmitLut ? 256 : 1];          // the byte table: a gather from LDS instead of one more trip to L2 per message
  if (kEmitLut) { s_lut[threadIdx.x] = p.lut256[threadIdx.x]; __syncthreads (); }
  const uint32_t strip_base = p.strip_begin + (blockIdx.x * 4 + (threadIdx.x >> 6)) * kEmitStrips;
  const int lane = threadIdx.x & 63, row = lane >> 4, sub = lane & 15;
  if (strip_base >= p.strip_end) return;
  const uint32_t cnt = (lane < kEmitStrips && strip_base + lane < p.strip_end) ? p.acc_counts[strip_base + lane] : 0u;
  if (__ballot (cnt != 0) == 0) return;
  for (int si = row; si < kEmitStrips; si += 4) {
  const uint32_t strip = strip_base + uint32_t (si);
  const uint32_t nacc = uint32_t (__shfl (int (cnt), si, 64));
  if (nacc == 0) continue;
  MsgOut *out = reinterpret_cast<MsgOut *> (p.messages) + p.offsets[strip];
  const Record *recs = p.records + size_t (strip) * p.cap;
  const uint32_t nrec = p.counts[strip];
  constexpr int bps = (CONV == kConvUc8Generic || CONV == kConvMagU16) ? 2 : 4;      // bytes per sample where the fast path applies
  // one accepted record -> one message (ref: src/demod-2400.c:463-497, src/demod-2000.c:960-966); `acc` = the sum of
  // squares behind the signal level, `slot` = where the message goes in the strip's part of the list
  auto build = [&] (uint32_t d0, uint32_t d1, uint32_t crc, uint32_t d3, uint32_t d4, uint32_t d5, uint32_t d6, uint32_t d7,
                    u64 acc, uint32_t slot) {
    const uint32_t off = d0 & 0xFFFF, phase = (d0 >> 16) & 0xFF, df = d0 >> 24;
    const uint32_t cls = d1 & 0xFF, nerr = (d1 >> 8) & 0xFF, bit0 = (d1 >> 16) & 0xFF, bit1 = d1 >> 24;
    u64 mlo = (u64 (d4) << 32) | d3, mhi = (u64 (d6 & 0xFFFF) << 32) | d5;           // msg[0..7], msg[8..13]
    const int16_t score = int16_t ((d6 >> 24) | ((d7 & 0xFF) << 8));                   // pad[0..1]
    const uint64_t P = uint64_t (strip) * kStrip + off;
    const int mbits = msg_bits_for_df (int (df));
    const int slen = mbits * 12 / 5;
    // (a block is 256 KiB of input: a power of two samples, so no 64-bit divisions per message)
    const uint64_t blk = P >> p.block_shift, j = P & uint64_t (p.block_samples - 1);
    const uint64_t counter = p.first_sample + blk * p.block_samples;
    const uint64_t block_ts = uint64_t ((double (counter) * 12E6) / double (p.sample_rate));        // ref: src/infile.c:143
    uint64_t ts = p.mode == 2000 ? block_ts + j * 6                                                  // ref: src/demod-2000.c:960
                                 : block_ts + j * 5 + phase;                                         // ref: src/demod-2400.c:467
    const bool ap = cls == kClsAP || cls == kClsAP2, acq = cls == kClsDF11;
    uint32_t msg_bits, msg_type, addr, ocrc = 0, eb = 0, oph = 0;
    int16_t oscore;
    double sig;
    if (cls == kClsModeA) {                     // ref: src/demod-2000.c:318,677-701, src/dump1090.c:2722-2766
      const uint32_t code = ((d3 & 0xFF) << 8) | ((d3 >> 8) & 0xFF);
      ts = block_ts + (j + 1) * 6;
      msg_bits = 16; msg_type = 32;
      addr = (code & 0x0000FF7Fu) | kNonIcao;
      oscore = -1;
      const int lv = int (crc);
      sig = double (int (uint32_t (lv) * uint32_t (lv))) / (65535.0 * 65535.0);   // int product as the reference forms it
    } else {
      if (acq && (crc & 0xFFFF80u)) eb = nerr;
      if (!ap && !acq && crc) eb = nerr;
      if (eb >= 1) { const uint32_t b = bit0 >> 3; const u64 f = u64 (0x80u >> (bit0 & 7)); if (b < 8) mlo ^= f << (8 * b); else mhi ^= f << (8 * (b - 8)); }
      if (eb >= 2) { const uint32_t b = bit1 >> 3; const u64 f = u64 (0x80u >> (bit1 & 7)); if (b < 8) mlo ^= f << (8 * b); else mhi ^= f << (8 * (b - 8)); }
      const uint32_t aa = (uint32_t (mlo >> 8) & 0xFF) << 16 | (uint32_t (mlo >> 16) & 0xFF) << 8 | (uint32_t (mlo >> 24) & 0xFF);
      msg_bits = uint32_t (mbits);
      msg_type = df >= 24 ? 24 : df;
      addr = ap ? crc : acq ? aa : (aa | non_icao_flag_words (mlo, mhi, int (df)));
      ocrc = crc;
      oscore = score;
      oph = p.mode == 2000 ? 0u : phase;
      // demod_2000's sig_level is wiped by decode_mode_S_message's memset (ref: src/demod-2000.c:966, src/dump1090.c:2385)
      sig = p.mode == 2000 ? 0.0 : (double (acc) / 65535.0 / 65535.0) / double (slen);
    }
    const u64 sigbits = u64 (__double_as_longlong (sig));
    uint4 o0, o1, o2; uint2 o3;
    o0.x = uint32_t (ts); o0.y = uint32_t (ts >> 32); o0.z = uint32_t (mlo); o0.w = uint32_t (mlo >> 32);
    o1.x = uint32_t (mhi); o1.y = (uint32_t (mhi >> 32) & 0xFFFFu) | (msg_bits << 16) | (msg_type << 24); o1.z = addr; o1.w = ocrc;
    o2.x = uint32_t (uint16_t (oscore)) | (eb << 16) | (oph << 24); o2.y = uint32_t ((p.first_sample >> p.block_shift) + blk);
    o2.z = uint32_t (sigbits); o2.w = uint32_t (sigbits >> 32);
    o3.x = uint32_t (j); o3.y = 0;
    uint32_t *dst = reinterpret_cast<uint32_t *> (out + slot);
    *reinterpret_cast<uint2 *> (dst) = make_uint2 (o0.x, o0.y);
    *reinterpret_cast<uint2 *> (dst + 2) = make_uint2 (o0.z, o0.w);
    *reinterpret_cast<uint2 *> (dst + 4) = make_uint2 (o1.x, o1.y);
    *reinterpret_cast<uint2 *> (dst + 6) = make_uint2 (o1.z, o1.w);
    *reinterpret_cast<uint2 *> (dst + 8) = make_uint2 (o2.x, o2.y);
    *reinterpret_cast<uint2 *> (dst + 10) = make_uint2 (o2.z, o2.w);
    *reinterpret_cast<uint2 *> (dst + 12) = o3;
  };
  uint32_t k = 0;
  for (uint32_t rb = 0; rb < nrec && k < nacc; rb += 16) {
    uint4 ra = { 0, 0, 0, 0 }, rc = { 0, 0, 0, 0 };
    if (rb + sub < nrec) { const uint4 *s4 = reinterpret_cast<const uint4 *> (recs + rb + sub); ra = s4[0]; rc = s4[1]; }
    const bool mine = (rc.z >> 16) & 1u;                                                  // flags bit 0: accepted
    uint32_t accm = uint32_t (__ballot (mine) >> (16 * row)) & 0xFFFFu;                    // the row's own records
    if (p.mode == 2000) {
      // demod_2000's messages carry no measured level (decode_mode_S_message's memset wipes it): nothing is summed, so
      // every accepted record of the batch becomes its message in its own lane
      if (mine) build (ra.x, ra.y, ra.z, ra.w, rc.x, rc.y, rc.z, rc.w, 0, k + uint32_t (__popc (accm & ((1u << sub) - 1u))));
      k += uint32_t (__popc (accm));
      continue;
    }
    for (; accm; accm &= accm - 1, k++) {
      const int src = 16 * row + __builtin_ctz (accm);
      const uint32_t d0 = uint32_t (__shfl (int (ra.x), src, 64)), d1 = uint32_t (__shfl (int (ra.y), src, 64)),
                     crc = uint32_t (__shfl (int (ra.z), src, 64)), d3 = uint32_t (__shfl (int (ra.w), src, 64));
      const uint32_t d4 = uint32_t (__shfl (int (rc.x), src, 64)), d5 = uint32_t (__shfl (int (rc.y), src, 64)),
                     d6 = uint32_t (__shfl (int (rc.z), src, 64)), d7 = uint32_t (__shfl (int (rc.w), src, 64));
      const uint32_t off = d0 & 0xFFFF, df = d0 >> 24;
      const uint64_t P = uint64_t (strip) * kStrip + off;
      const int slen = msg_bits_for_df (int (df)) * 12 / 5;
      u64 acc = 0;                                  // ref: src/demod-2400.c:486-497
      {
        // a lane takes four neighbouring samples at a time: five wide loads per lane cover the 268 samples of a long message
        constexpr int NQ = (kLongBits * 12 / 5 + 63) / 64;
        const long long n0 = (long long) P + 19 - (long long) p.ov;                     // first sample of the sum
        const bool lut = kEmitLut;
        if (!lut && CONV != kConvMagU16 && n0 >= 0 && uint64_t (n0) + 320 <= p.nsamples) {
          // inside the shard: the loads of a lane are issued together (dword-aligned 16-byte / word-aligned 8-byte loads)
          struct __attribute__ ((packed, aligned (4))) Q4 { uint32_t v[4]; };
          struct __attribute__ ((packed, aligned (2))) H4 { uint16_t v[4]; };
          uint32_t raw[NQ][4];
#pragma unroll
          for (int q = 0; q < NQ; q++) {
            const int x = 4 * (sub + 16 * q);
#pragma unroll
            for (int b = 0; b < 4; b++) raw[q][b] = 0;
            if (x < slen) {
              if (bps == 2) { const H4 h = *reinterpret_cast<const H4 *> (reinterpret_cast<const uint16_t *> (p.iq) + n0 + x);
#pragma unroll
                              for (int b = 0; b < 4; b++) raw[q][b] = h.v[b]; }
              else          { const Q4 w = *reinterpret_cast<const Q4 *> (reinterpret_cast<const uint32_t *> (p.iq) + n0 + x);
#pragma unroll
                              for (int b = 0; b < 4; b++) raw[q][b] = w.v[b]; }
            }
          }
#pragma unroll
          for (int q = 0; q < NQ; q++) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
              const int x = 4 * (sub + 16 * q) + b;
              const uint32_t r = raw[q][b];
              uint32_t m;
              if (CONV == kConvUc8Generic) m = mag_from_float (c_uc8_float[r & 0xFF], c_uc8_float[(r >> 8) & 0xFF], nullptr);
              else if (CONV == kConvSc16Int) m = mag_sc16_word (r) & 0xFFFFu;
              else {
                const float sc = CONV == kConvSc16Generic ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
                m = mag_from_float (float (int16_t (r & 0xFFFF)) * sc, float (int16_t (r >> 16)) * sc, nullptr);
              }
              if (x < slen) acc += u64 (m * m);
            }
          }
        } else if (lut && n0 >= 0 && uint64_t (n0) + 320 <= p.nsamples && (uint64_t (n0) & uint64_t (p.block_samples - 1)) + 320 <= p.block_samples
                   && !(reinterpret_cast<uintptr_t> (p.iq) & 3)) {
          // the byte table inside one block: sample i of the block is the table's value for byte i of the block.  Whole
          // dwords from the aligned address at or below the first byte; `sh` bytes of the first one lie in front of the sum
          const uint8_t *src8 = lut_byte_addr_of (p.iq, p.block_samples, n0);
          const int sh = int (reinterpret_cast<uintptr_t> (src8) & 3);
          const uint32_t *src32 = reinterpret_cast<const uint32_t *> (src8 - sh);
          uint32_t raw[NQ];
#pragma unroll
          for (int q = 0; q < NQ; q++) raw[q] = 4 * (sub + 16 * q) < slen + sh ? src32[sub + 16 * q] : 0u;
#pragma unroll
          for (int q = 0; q < NQ; q++) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
              const int x = 4 * (sub + 16 * q) + b - sh;
              const uint32_t m = s_lut[kEmitLut ? (raw[q] >> (8 * b)) & 0xFF : 0u];
              if (x >= 0 && x < slen) acc += u64 (m * m);
            }
          }
        } else if (CONV == kConvMagU16 && uint64_t (P) + 19 + 320 <= p.navail) {
          // magnitudes as the DC-block chain left them (the virtual stream V itself): four neighbours per load
          struct __attribute__ ((packed, aligned (2))) H4 { uint16_t v[4]; };
          const uint16_t *v16 = reinterpret_cast<const uint16_t *> (p.iq) + P + 19;
          H4 raw[NQ];
#pragma unroll
          for (int q = 0; q < NQ; q++) {
            const int x = 4 * (sub + 16 * q);
            if (x < slen) raw[q] = *reinterpret_cast<const H4 *> (v16 + x);
            else { raw[q].v[0] = raw[q].v[1] = raw[q].v[2] = raw[q].v[3] = 0; }
          }
#pragma unroll
          for (int q = 0; q < NQ; q++) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
              const uint32_t m = raw[q].v[b];
              if (4 * (sub + 16 * q) + b < slen) acc += u64 (m * m);
            }
          }
        } else {
          for (int q = sub; q < slen; q += 16) {
            const uint32_t m = mag_of_V<CONV> (p.iq, CONV == kConvMagU16 ? p.navail : p.nsamples, p.halo, p.ov, p.block_samples, p.lut256, (long long) P + 19 + q);
            acc += u64 (m * m);
          }
        }
      }
      for (int d = 8; d >= 1; d >>= 1) acc += __shfl_xor (acc, d, 64);                   // over the row
      if (sub == 0) build (d0, d1, crc, d3, d4, d5, d6, d7, acc, k);
    }
  }
  }
}

// ---- RAW output of the accepted messages: "*<hex>;\n" per message, back to back ----
// lower case = what raw_out_send() writes to port-30002 clients (ref: src/raw-sbs.c:86-97), upper case = what
// modeS_user_message() prints with --raw (ref: src/dump1090.c:3086-3097)
__global__ __launch_bounds__ (256)
void raw_len_kernel (const MsgOut *msgs, uint32_t n, uint32_t *len)
{
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) len[i] = 3u + 2u * (msgs[i].msg_bits / 8u);
}

__global__ __launch_bounds__ (256)
void raw_format_kernel (const MsgOut *msgs, uint32_t n, const uint64_t *offsets, int upper, char *text)
{
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const MsgOut &m = msgs[i];
  char *p = text + offsets[i];
  const int nb = m.msg_bits / 8;
  const char a = upper ? 'A' : 'a';
  *p++ = '*';
  for (int b = 0; b < nb; b++) {
    const int hi = m.msg[b] >> 4, lo = m.msg[b] & 15;
    *p++ = char (hi < 10 ? '0' + hi : a + hi - 10);
    *p++ = char (lo < 10 ? '0' + lo : a + lo - 10);
  }
  *p++ = ';';
  *p++ = '\n';
}

// lengths -> offsets (the strip scan kernels serve: one "strip" per message) -> text.  totals[0] receives the text length.
hipError_t launch_raw_format (const void *msgs, uint32_t n, uint32_t *len, uint64_t *chunk_sum, uint64_t *chunk_vp,
                              uint64_t *chunk_ovf, uint64_t *totals, uint64_t *offsets, int upper, char *text, int pass, hipStream_t stream)
{
  if (n == 0) return hipSuccess;
  const MsgOut *m = static_cast<const MsgOut *> (msgs);
  if (pass == 0) {
    hipLaunchKernelGGL (raw_len_kernel, dim3 ((n + 255) / 256), dim3 (256), 0, stream, m, n, len);
    hipError_t e = launch_count_scan (len, len, n, chunk_sum, chunk_vp, chunk_ovf, totals, stream);
    if (e != hipSuccess) return e;
    return launch_strip_offsets (len, n, chunk_sum, offsets, stream);
  }
  hipLaunchKernelGGL (raw_format_kernel, dim3 ((n + 255) / 256), dim3 (256), 0, stream, m, n, offsets, upper, text);
  return hipGetLastError ();
}

// `ov` magnitudes of the virtual stream starting at V[at]: with at = nsamples this is what fifo_enqueue keeps for the
// next block (ref: src/fifo.c:205-211), i.e. the halo of a following shard
template <int CONV>
__global__ __launch_bounds__ (256)
void tail_kernel (const ScanParams p, uint64_t at, uint16_t *out)
{
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.ov) return;
  out[i] = mag_of_V<CONV> (p.iq, CONV == kConvMagU16 ? p.navail : p.nsamples, p.halo, p.ov, p.block_samples, p.lut256, (long long) (at + i));
}

hipError_t launch_tail (const ScanParams &p, uint64_t at, uint16_t *out, hipStream_t stream)
{
  if (!p.ov) return hipSuccess;
  const dim3 grid ((p.ov + 255) / 256), block (256);
  switch (p.conv) {
    case kConvUc8Lut: case kConvUc8LutPower: hipLaunchKernelGGL (tail_kernel<kConvUc8Lut>, grid, block, 0, stream, p, at, out); break;
    case kConvUc8Generic:     hipLaunchKernelGGL (tail_kernel<kConvUc8Generic>, grid, block, 0, stream, p, at, out); break;
    case kConvSc16Int:        hipLaunchKernelGGL (tail_kernel<kConvSc16Int>, grid, block, 0, stream, p, at, out); break;
    case kConvSc16Generic:    hipLaunchKernelGGL (tail_kernel<kConvSc16Generic>, grid, block, 0, stream, p, at, out); break;
    case kConvSc16Q11Generic: hipLaunchKernelGGL (tail_kernel<kConvSc16Q11Generic>, grid, block, 0, stream, p, at, out); break;
    case kConvMagU16:         hipLaunchKernelGGL (tail_kernel<kConvMagU16>, grid, block, 0, stream, p, at, out); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError ();
}

// ------------------------------------------------------------------------------------------
// launchers (called from engine.cpp)
// ------------------------------------------------------------------------------------------
hipError_t upload_constants (const uint32_t *single_bit, const float *uc8_float)
{
  hipError_t e = upload_constants_tu (single_bit, uc8_float);
  if (e != hipSuccess) return e;
  e = upload_constants_2000 (single_bit, uc8_float);
  if (e != hipSuccess) return e;
  e = upload_constants_ac (single_bit, uc8_float);
  if (e != hipSuccess) return e;
  return upload_constants_8000 (single_bit, uc8_float);
}

// Small fills and copies as kernels: a hipMemsetAsync / hipMemcpyAsync on the stream of the ordered pass is a blit that
// queues behind the read-back of the range before (tens of MB on the copy stream) and held the pass up for as long as
// that copy lasted.  `dst` or `src` of the copy may be pinned host memory (a few KB over PCIe).
__global__ __launch_bounds__ (256)
void fill_u32_kernel (uint32_t *p, size_t n, uint32_t v)
{
  const size_t stride = size_t (gridDim.x) * blockDim.x;
  for (size_t i = size_t (blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}
__global__ __launch_bounds__ (256)
void copy_u32_kernel (uint32_t *dst, const uint32_t *src, size_t n)
{
  const size_t stride = size_t (gridDim.x) * blockDim.x;
  for (size_t i = size_t (blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
hipError_t launch_fill_u32 (uint32_t *p, size_t n, uint32_t v, hipStream_t stream)
{
  if (!n) return hipSuccess;
  const size_t want = (n + 255) / 256;
  hipLaunchKernelGGL (fill_u32_kernel, dim3 ((unsigned) (want < 1024 ? want : 1024)), dim3 (256), 0, stream, p, n, v);
  return hipGetLastError ();
}
hipError_t launch_copy_u32 (uint32_t *dst, const uint32_t *src, size_t n, hipStream_t stream)
{
  if (!n) return hipSuccess;
  const size_t want = (n + 255) / 256;
  hipLaunchKernelGGL (copy_u32_kernel, dim3 ((unsigned) (want < 256 ? want : 256)), dim3 (256), 0, stream, dst, src, n);
  return hipGetLastError ();
}

hipError_t launch_scan2400 (const ScanParams &p, hipStream_t stream)
{
  const dim3 block (64 * kWavesPerBlock);
  if (p.nstrips == 0) return hipSuccess;
  if (p.block_strips % 2) return hipErrorInvalidValue;            // a block is whole spans (block_strips 0: a single block)
#define SCAN(C)                                                                                    \
  do {                                                                                             \
    const uint32_t pair = p.phase_enhance ? 1u : uint32_t (ScanPair<C>::n);                        \
    const uint32_t nspans = ((p.strip_end ? p.strip_end : p.nstrips) - p.strip_begin + pair - 1) / pair; \
    const dim3 grid ((nspans + kWavesPerBlock - 1) / kWavesPerBlock);                              \
    if (p.phase_enhance) hipLaunchKernelGGL ((scan2400_kernel<C, true>), grid, block, 0, stream, p);  \
    else                 hipLaunchKernelGGL ((scan2400_kernel<C, false>), grid, block, 0, stream, p); \
  } while (0)
  switch (p.conv) {
    case kConvUc8Lut: case kConvUc8LutPower: SCAN (kConvUc8Lut); break;
    case kConvUc8Generic:     SCAN (kConvUc8Generic); break;
    case kConvSc16Int:        SCAN (kConvSc16Int); break;
    case kConvSc16Generic:    SCAN (kConvSc16Generic); break;
    case kConvSc16Q11Generic: SCAN (kConvSc16Q11Generic); break;
    case kConvMagU16:         SCAN (kConvMagU16); break;
    case kConvUc8LutR8: case kConvUc8LutRaw: {
      if (p.phase_enhance || !p.rank256 || !p.rankmag) return hipErrorInvalidValue;
      const uint32_t pair = uint32_t (ScanPair<kConvUc8LutR8>::n);
      const uint32_t nspans = ((p.strip_end ? p.strip_end : p.nstrips) - p.strip_begin + pair - 1) / pair;
      const dim3 grid ((nspans + kWavesPerBlock - 1) / kWavesPerBlock);
      if (p.conv == kConvUc8LutRaw) {
        // whole 16-byte chunks of input behind every 16th scan position: ov as demod_2400's, the input 16-byte aligned
        if (p.ov != kOv2400 || (reinterpret_cast<uintptr_t> (p.iq) & 15) || (p.block_samples & 15) || ((p.strip_begin * uint64_t (kStrip)) & 15))
          return hipErrorInvalidValue;
        hipLaunchKernelGGL ((scan2400_kernel<kConvUc8LutRaw, false>), grid, block, 0, stream, p);
      } else
        hipLaunchKernelGGL ((scan2400_kernel<kConvUc8LutR8, false>), grid, block, 0, stream, p);
      break;
    }
    default: return hipErrorInvalidValue;
  }
#undef SCAN
  return hipGetLastError ();
}

hipError_t launch_convert (int conv, const uint8_t *iq, uint16_t *mag, size_t n, const uint16_t *lut256,
                           unsigned long long *power_u64, hipStream_t stream)
{
  if (n == 0) return hipSuccess;
  const size_t want = (n + 255) / 256;
  const dim3 grid ((unsigned) (want < 65536 ? want : 65536)), block (256);
  switch (conv) {
    case kConvUc8Lut:         hipLaunchKernelGGL (convert_kernel<kConvUc8Lut>, grid, block, 0, stream, iq, mag, n, lut256, power_u64); break;
    case kConvUc8LutPower:    hipLaunchKernelGGL (convert_kernel<kConvUc8LutPower>, grid, block, 0, stream, iq, mag, n, lut256, power_u64); break;
    case kConvUc8Generic:     hipLaunchKernelGGL (convert_kernel<kConvUc8Generic>, grid, block, 0, stream, iq, mag, n, lut256, power_u64); break;
    case kConvSc16Int:        hipLaunchKernelGGL (convert_kernel<kConvSc16Int>, grid, block, 0, stream, iq, mag, n, lut256, power_u64); break;
    case kConvSc16Generic:    hipLaunchKernelGGL (convert_kernel<kConvSc16Generic>, grid, block, 0, stream, iq, mag, n, lut256, power_u64); break;
    case kConvSc16Q11Generic: hipLaunchKernelGGL (convert_kernel<kConvSc16Q11Generic>, grid, block, 0, stream, iq, mag, n, lut256, power_u64); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError ();
}


hipError_t launch_rssi2400 (int conv, const uint8_t *iq, uint64_t nsamples, const uint16_t *halo, uint32_t ov,
                            uint32_t block_samples, const uint16_t *lut256, const uint64_t *pos, const uint32_t *len,
                            uint64_t *sum_out, uint32_t nmsg, hipStream_t stream)
{
  if (nmsg == 0) return hipSuccess;
  const dim3 grid ((nmsg + 3) / 4), block (256);
#define RSSI(C) hipLaunchKernelGGL (rssi2400_kernel<C>, grid, block, 0, stream, iq, nsamples, halo, ov, block_samples, lut256, pos, len, sum_out, nmsg)
  switch (conv) {
    case kConvUc8Lut: case kConvUc8LutPower: RSSI (kConvUc8Lut); break;
    case kConvUc8Generic:     RSSI (kConvUc8Generic); break;
    case kConvSc16Int:        RSSI (kConvSc16Int); break;
    case kConvSc16Generic:    RSSI (kConvSc16Generic); break;
    case kConvSc16Q11Generic: RSSI (kConvSc16Q11Generic); break;
    case kConvMagU16:         RSSI (kConvMagU16); break;           // `nsamples` = magnitudes held by `iq`
    default: return hipErrorInvalidValue;
  }
#undef RSSI
  return hipGetLastError ();
}

// Step 1: per-chunk sums + exclusive scan; totals[0..2] = records, valid preambles, overflowed strips.
// chunk_sum / chunk_vp / chunk_ovf need nstrips/1024 + 1 entries each.
hipError_t launch_count_scan (const uint32_t *counts, const uint32_t *vp_counts, uint32_t nstrips, uint64_t *chunk_sum,
                              uint64_t *chunk_vp, uint64_t *chunk_ovf, uint64_t *totals, hipStream_t stream)
{
  if (nstrips == 0) return hipSuccess;
  const uint32_t nchunks = (nstrips + kScanChunk - 1) / kScanChunk;
  hipLaunchKernelGGL (chunk_sums_kernel, dim3 (nchunks), dim3 (256), 0, stream, counts, vp_counts, nstrips, chunk_sum, chunk_vp, chunk_ovf);
  hipLaunchKernelGGL (scan_chunk_sums_kernel, dim3 (1), dim3 (64), 0, stream, chunk_sum, chunk_vp, chunk_ovf, nchunks, totals);
  return hipGetLastError ();
}

// Step 2: gather the per-strip slots into one ordered list.
hipError_t launch_gather (const uint32_t *counts, uint32_t nstrips, const uint64_t *chunk_sum, const Record *records,
                          uint32_t cap, Record *packed, uint64_t *packed_pos, uint64_t *offsets, hipStream_t stream)
{
  if (nstrips == 0) return hipSuccess;
  hipLaunchKernelGGL (gather_kernel, dim3 ((nstrips + 3) / 4), dim3 (256), 0, stream, counts, nstrips, chunk_sum, records,
                      cap, packed, packed_pos, offsets);
  return hipGetLastError ();
}

// Float converter with explicit DC-block state (one convert_func call).  zI/zQ/mag_sq are n-float scratch
// buffers; z_io[2] holds z1_I, z1_Q on entry and exit (device memory); power_out (device, 1 float) may be null.
hipError_t launch_convert_float (int conv, const uint8_t *iq, uint16_t *mag, size_t n, float A, float B, float zI0, float zQ0,
                                 float *zI, float *zQ, float *mag_sq, float *z_final, float *power_out, hipStream_t stream)
{
  if (n == 0) return hipSuccess;
  const size_t want = (n + 255) / 256;
  const dim3 grid ((unsigned) (want < 16384 ? want : 16384)), block (256);
  const bool dc = !(A == 0.0f && B == 1.0f);
#define FLOATCONV(C)                                                                                                     \
  if (dc) {                                                                                                               \
    hipLaunchKernelGGL (dc_zscan_kernel<C>, dim3 (1), dim3 (64), 0, stream, iq, n, A, B, zI0, zQ0, zI, zQ, z_final);      \
    hipLaunchKernelGGL (dc_mag_kernel<C>, grid, block, 0, stream, iq, n, zI, zQ, mag, power_out ? mag_sq : nullptr);      \
  } else                                                                                                                  \
    hipLaunchKernelGGL (nodc_mag_kernel<C>, grid, block, 0, stream, iq, n, mag, power_out ? mag_sq : nullptr)
  switch (conv) {
    case kConvUc8Generic:     FLOATCONV (kConvUc8Generic); break;
    case kConvSc16Generic:    FLOATCONV (kConvSc16Generic); break;
    case kConvSc16Q11Generic: FLOATCONV (kConvSc16Q11Generic); break;
    default: return hipErrorInvalidValue;
  }
#undef FLOATCONV
  if (power_out) hipLaunchKernelGGL (power_seq_kernel, dim3 (1), dim3 (64), 0, stream, mag_sq, n, power_out);
  return hipGetLastError ();
}

hipError_t launch_accept2400 (const AcceptParams &p, hipStream_t stream)
{
  const uint32_t nb = (p.blk_end ? p.blk_end : p.nblocks) - p.blk_begin;
  if (p.nblocks == 0 || nb == 0) return hipSuccess;
  if (p.mode == 2000) hipLaunchKernelGGL (accept_kernel<2000>, dim3 (nb), dim3 (64), 0, stream, p);
  else                hipLaunchKernelGGL (accept_kernel<2400>, dim3 (nb), dim3 (64), 0, stream, p);
  return hipGetLastError ();
}

// delta -> snap_in for the next round; chunk_last/chunk_entry hold (nblocks/32 + 1) * 1024 words each
hipError_t launch_snapshots (const uint32_t *delta, const uint32_t *blk_wrote, uint32_t nblocks, const uint32_t *cache_in, uint32_t *chunk_last,
                             uint32_t *chunk_entry, uint32_t *snap_in, uint32_t *cache_out, uint32_t *dirty, const uint32_t *changed,
                             uint32_t *marked, uint32_t *marked_host, int first_round, hipStream_t stream)
{
  if (nblocks == 0) return hipSuccess;
  const uint32_t nchunks = (nblocks + kSnapChunk - 1) / kSnapChunk;
  hipLaunchKernelGGL (snap_chunk_last_kernel, dim3 (kIcaoSlots / 256, nchunks), dim3 (256), 0, stream, delta, blk_wrote, nblocks, chunk_last, dirty, changed);
  hipLaunchKernelGGL (snap_chunk_scan_kernel, dim3 (kIcaoSlots / 256), dim3 (256), 0, stream, chunk_last, nchunks, cache_in, chunk_entry, cache_out, changed, first_round);
  hipLaunchKernelGGL (snap_fill_kernel, dim3 (kIcaoSlots / 256, nchunks), dim3 (256), 0, stream, delta, blk_wrote, nblocks, chunk_entry, snap_in, dirty, changed, cache_in, first_round, marked, marked_host);
  return hipGetLastError ();
}

hipError_t launch_emit2400 (const EmitParams &p_in, hipStream_t stream)
{
  if (p_in.strip_end <= p_in.strip_begin) return hipSuccess;
  if (p_in.block_samples == 0 || (p_in.block_samples & (p_in.block_samples - 1))) return hipErrorInvalidValue;
  EmitParams p = p_in;
  p.block_shift = uint32_t (__builtin_ctz (p.block_samples));
  const uint32_t n = p.strip_end - p.strip_begin;
  const dim3 grid ((n + 4 * kEmitStrips - 1) / (4 * kEmitStrips)), block (256);
  switch (p.conv) {
    case kConvUc8Lut: case kConvUc8LutPower: hipLaunchKernelGGL (emit2400_kernel<kConvUc8Lut>, grid, block, 0, stream, p); break;
    case kConvUc8Generic:     hipLaunchKernelGGL (emit2400_kernel<kConvUc8Generic>, grid, block, 0, stream, p); break;
    case kConvSc16Int:        hipLaunchKernelGGL (emit2400_kernel<kConvSc16Int>, grid, block, 0, stream, p); break;
    case kConvSc16Generic:    hipLaunchKernelGGL (emit2400_kernel<kConvSc16Generic>, grid, block, 0, stream, p); break;
    case kConvSc16Q11Generic: hipLaunchKernelGGL (emit2400_kernel<kConvSc16Q11Generic>, grid, block, 0, stream, p); break;
    case kConvMagU16:         hipLaunchKernelGGL (emit2400_kernel<kConvMagU16>, grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError ();
}

}  // namespace modes
