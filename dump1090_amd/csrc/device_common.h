// Device-side helpers shared by the gfx950 kernels of libmodes_hip.so: sample converters,
// wave/DPP primitives and the staging of a strip of magnitudes into a wave-private LDS tile.
// "ref:" citations are relative to the reference's root.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

namespace modes {
static __constant__ uint32_t c_single_bit[kLongBits];     // ref: src/crc.c:35,56-61
static __constant__ float    c_uc8_float[256];            // (float)((b-127.5)/127.5), ref: src/convert.c:132

// Every translation unit that includes this header owns a copy of the two tables; its launcher file
// exposes one upload call built on this.
static inline hipError_t upload_constants_tu (const uint32_t *single_bit, const float *uc8_float)
{
  hipError_t e = hipMemcpyToSymbol (HIP_SYMBOL (c_single_bit), single_bit, sizeof (uint32_t) * kLongBits);
  if (e != hipSuccess) return e;
  return hipMemcpyToSymbol (HIP_SYMBOL (c_uc8_float), uc8_float, sizeof (float) * 256);
}

using u64 = unsigned long long;

// ------------------------------------------------------------------------------------------
// converters, one sample
// ------------------------------------------------------------------------------------------

// ref: src/convert.c:254-292.  The reference branches four ways on (I < Q, I < threshold(Q)); the four linear forms
// pair up under swapping I and Q, so with small = min, big = max one select on (I < Q) == (I < t) picks the pair:
//   mag = ((sel ? 12868 : 36646) * small + (sel ? 64692 : 54842) * big) >> 16
__device__ __forceinline__ uint32_t mag_sc16_abs (uint32_t I, uint32_t Q)      // I, Q = |i16|, |q16| in 0..32768
{
  const bool lt = I < Q;
  const uint32_t small = lt ? I : Q, big = lt ? Q : I;
  const uint32_t t = __umul24 (lt ? 106u : 618u, Q) >> 8;
  const bool sel = lt == (I < t);
  return (__umul24 (sel ? 12868u : 36646u, small) + __umul24 (sel ? 64692u : 54842u, big)) >> 16;
}
__device__ __forceinline__ uint16_t mag_sc16_int (int i16, int q16)
{
  return uint16_t (mag_sc16_abs (uint32_t (i16 < 0 ? -i16 : i16), uint32_t (q16 < 0 ? -q16 : q16)));
}
// one IQ pair as it lies in memory (I in the low half): |.| of both halves with two packed operations, then the
// selected linear form as ONE packed dot product: the four forms are (cI, cQ) pairs, chosen by (I < Q) and
// (I < threshold); result = (cI * |I| + cQ * |Q|) >> 16 (at most 91488 * 32768 < 2^32).  Returns the 32-bit sum:
// the magnitude is its high half.
typedef short short2v __attribute__ ((ext_vector_type (2)));
typedef unsigned short ushort2v __attribute__ ((ext_vector_type (2)));
__device__ __forceinline__ uint32_t mag_sc16_sum (uint32_t d)
{
  const short2v v = __builtin_bit_cast (short2v, d);
  const short2v a = __builtin_elementwise_max (v, short2v (-v));      // -32768 stays 0x8000 = 32768 unsigned
  const uint32_t u = __builtin_bit_cast (uint32_t, a);
  const uint32_t I = u & 0xFFFFu, Q = u >> 16;
  const bool lt = I < Q;
  const uint32_t t = __umul24 (lt ? 106u : 618u, Q) >> 8;
  const bool sel = lt == (I < t);
  const uint32_t k_sel = lt ? (12868u | (64692u << 16)) : (64692u | (12868u << 16));
  const uint32_t k_not = lt ? (36646u | (54842u << 16)) : (54842u | (36646u << 16));
  const uint32_t k = sel ? k_sel : k_not;
  return __builtin_amdgcn_udot2 (__builtin_bit_cast (ushort2v, u), __builtin_bit_cast (ushort2v, k), 0u, false);
}
__device__ __forceinline__ uint32_t mag_sc16_word (uint32_t d) { return mag_sc16_sum (d) >> 16; }
// magnitudes of two IQ pairs packed as they lie in a magnitude buffer (first sample in the low half)
__device__ __forceinline__ uint32_t mag_sc16_pair (uint32_t d0, uint32_t d1)
{
  return __builtin_amdgcn_perm (mag_sc16_sum (d1), mag_sc16_sum (d0), 0x07060302u);
}

// ref: src/convert.c:142-147 -- the tail shared by the three float converters, DC block already applied.
// sqrtf as the compiler lowers it carries the scaling that keeps v_sqrt_f32 exact for denormal inputs and a class test
// for 0 / inf: seventeen operations.  Here mag_sq lies in [0, 1], and below 2^-40 the magnitude is 0 whatever the
// root's last bit is (65535 * 2^-20 + 0.5 < 1), so the correctly rounded root is only needed for normal inputs: the
// hardware's root (1 ulp) and the two residual tests around it, the same correction the compiler emits.  The product
// 65535 * root is exact in double and the sum rounds nowhere near an integer, so one fused operation gives the
// reference's `root * 65535.0 + 0.5`.
__device__ __forceinline__ uint16_t mag_from_float (float fI, float fQ, float *mag_sq_out)
{
  float mag_sq = __fadd_rn (__fmul_rn (fI, fI), __fmul_rn (fQ, fQ));
  mag_sq = __builtin_fminf (mag_sq, 1.0f);                        // (never NaN: the inputs are finite)
  if (mag_sq_out) *mag_sq_out = mag_sq;
  float r = __builtin_amdgcn_sqrtf (mag_sq);
  const float rm = __uint_as_float (__float_as_uint (r) - 1u), rp = __uint_as_float (__float_as_uint (r) + 1u);
  const float em = __builtin_fmaf (-rm, r, mag_sq), ep = __builtin_fmaf (-rp, r, mag_sq);
  r = em <= 0.0f ? rm : r;
  r = ep > 0.0f ? rp : r;
  const double v = __builtin_fma (double (r), 65535.0, 0.5);
  return mag_sq < 0x1p-40f ? uint16_t (0) : uint16_t (int (v));
}

// ------------------------------------------------------------------------------------------
// wave helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id () { return int (__builtin_amdgcn_mbcnt_hi (~0u, __builtin_amdgcn_mbcnt_lo (~0u, 0u))); }


// ---- DPP cross-lane helpers (GFX9 row/wave controls; no LDS round trip) ----
template <int CTRL, int ROW_MASK = 0xF, bool BOUND = true>
__device__ __forceinline__ uint32_t dpp_mov (uint32_t v)
{
  return uint32_t (__builtin_amdgcn_update_dpp (0, int (v), CTRL, ROW_MASK, 0xF, BOUND));
}
// every lane of a 16-lane row receives the XOR / OR of the row (row_ror:1,2,4,8)
__device__ __forceinline__ uint32_t row_xor (uint32_t v)
{
  v ^= dpp_mov<0x121> (v); v ^= dpp_mov<0x122> (v); v ^= dpp_mov<0x124> (v); v ^= dpp_mov<0x128> (v);
  return v;
}
__device__ __forceinline__ uint32_t row_or (uint32_t v)
{
  v |= dpp_mov<0x121> (v); v |= dpp_mov<0x122> (v); v |= dpp_mov<0x124> (v); v |= dpp_mov<0x128> (v);
  return v;
}

// inclusive +scan over the 64 lanes: row_shr 1,2,4,8 inside each row, then row_bcast:15 / row_bcast:31
__device__ __forceinline__ int wave_incl_scan (int v, int lane)
{
  (void) lane;
  uint32_t u = uint32_t (v);
  u += dpp_mov<0x111> (u); u += dpp_mov<0x112> (u); u += dpp_mov<0x114> (u); u += dpp_mov<0x118> (u);
  u += dpp_mov<0x142, 0xA, false> (u);
  u += dpp_mov<0x143, 0xC, false> (u);
  return int (u);
}

__device__ __forceinline__ uint32_t wave_xor (uint32_t v)
{
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v ^= uint32_t (__shfl_xor (int (v), d, 64));
  return v;
}

__device__ __forceinline__ u64 funnel (u64 w, u64 nx, int k) { return (w >> k) | (nx << (64 - k)); }

// per-byte bit reversal of a 32-bit word (message bit k sits at mask bit k, MSB-first bytes wanted)
__device__ __forceinline__ uint32_t rev_in_bytes (uint32_t x) { return __builtin_bswap32 (__brev (x)); }

// ------------------------------------------------------------------------------------------
// 2.4 MS/s slicers.  A data bit that starts `f` fifths of a sample after m[0] is sliced at
// sample f/5 with the weights of row f%5 (ref: src/demod-2400.c:32-55, 339-412).
// ------------------------------------------------------------------------------------------
template <int F>
__device__ __forceinline__ int slice_const (const int *s)   // s[] = samples as ints, compile-time position
{
  constexpr int i = F / 5, r = F % 5;
  if (r == 0) return 5 * s[i] - 3 * s[i + 1] - 2 * s[i + 2];
  if (r == 1) return 4 * s[i] - s[i + 1] - 3 * s[i + 2];
  if (r == 2) return 3 * s[i] + s[i + 1] - 4 * s[i + 2];
  if (r == 3) return 2 * s[i] + 3 * s[i + 1] - 5 * s[i + 2];
  return s[i] + 5 * s[i + 1] - 5 * s[i + 2] - s[i + 3];
}
template <int R> __device__ __forceinline__ constexpr int corr_weight () { return R == 0 ? 26 : R == 1 ? 38 : R == 2 ? 38 : R == 3 ? 26 : 19; }

// correlate_check for start phase T over the first five bits (ref: src/demod-2400.c:86-129);
// also returns those five sliced bits (bit 4 = first data bit).
template <int T>
__device__ __forceinline__ int check_phase (const int *s, int *five_bits)
{
  const int a = slice_const<T> (s), b = slice_const<T + 12> (s), c = slice_const<T + 24> (s),
            d = slice_const<T + 36> (s), e = slice_const<T + 48> (s);
  *five_bits = ((a > 0) << 4) | ((b > 0) << 3) | ((c > 0) << 2) | ((d > 0) << 1) | (e > 0);
  return abs (a * corr_weight<T % 5> ()) + abs (b * corr_weight<(T + 12) % 5> ()) + abs (c * corr_weight<(T + 24) % 5> ()) +
         abs (d * corr_weight<(T + 36) % 5> ()) + abs (e * corr_weight<(T + 48) % 5> ());
}

// bytes sliced for a DF (ref: src/demod-2400.c:416-435): 7, 14 or 0 (= give up after byte 0)
__device__ __forceinline__ int slice_bytes_for_df (int df)
{
  const uint32_t short_set = (1u << 0) | (1u << 4) | (1u << 5) | (1u << 11);
  const uint32_t long_set = (1u << 16) | (1u << 17) | (1u << 18) | (1u << 20) | (1u << 21) | (1u << 24);
  return ((short_set >> df) & 1) ? kShortBytes : ((long_set >> df) & 1) ? kLongBytes : 0;
}
// ref: src/dump1090.c:1427-1432
__device__ __forceinline__ int msg_bits_for_df (int df)
{
  return (df == 16 || df == 17 || df == 19 || df == 20 || df == 21) ? kLongBits : kShortBits;
}

// Wave-cooperative search of a sorted error table (n <= 4096): two probes per lane.
__device__ __forceinline__ int table_find (const ErrEntry *tab, int n, uint32_t syn, int lane)
{
  if (n <= 0) return -1;
  const int step = (n + 63) >> 6;
  const int first = lane * step;
  const bool le = first < n && tab[first].syndrome <= syn;
  const u64 m = __ballot (le);
  if (!m) return -1;
  const int seg = 63 - __builtin_clzll (m);       // `le` is monotone in the lane index
  const int idx = seg * step + lane;
  const bool eq = lane < step && idx < n && tab[idx].syndrome == syn;
  const u64 h = __ballot (eq);
  return h ? seg * step + int (__builtin_ctzll (h)) : -1;
}

// ------------------------------------------------------------------------------------------
// staging: convert samples [a_lo, a_lo + count) into w.tile[0 .. count)
// ------------------------------------------------------------------------------------------
template <int CONV> struct RawWords { static constexpr int n = (CONV == kConvUc8Lut || CONV == kConvUc8LutPower || CONV == kConvUc8LutR8) ? 2 : (CONV == kConvUc8Generic || CONV == kConvMagU16) ? 4 : 8; };

// raw IQ of 8 consecutive samples starting at n0 (a multiple of 8, whole chunk inside the shard)
template <int CONV>
__device__ __forceinline__ void load_raw8 (const ScanParams &p, long long n0, uint32_t *d)
{
  if (CONV == kConvUc8Lut || CONV == kConvUc8LutPower || CONV == kConvUc8LutR8) {
    // as written: output sample i of a block reads input BYTE i of that block (ref: src/convert.c:19,29)
    // block_samples is a power of two (256 KiB / bytes per sample): byte offset = 2 * block start + offset in block
    const uint64_t in_blk = uint64_t (n0) & uint64_t (p.block_samples - 1);
    const uint2 v = *reinterpret_cast<const uint2 *> (p.iq + 2 * (uint64_t (n0) - in_blk) + in_blk);
    d[0] = v.x; d[1] = v.y;
  } else if (CONV == kConvUc8Generic || CONV == kConvMagU16) {
    const uint4 v = *reinterpret_cast<const uint4 *> (p.iq + 2 * n0);
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  } else {
    const uint4 *src = reinterpret_cast<const uint4 *> (p.iq + 4 * n0);
    const uint4 v0 = src[0], v1 = src[1];
    d[0] = v0.x; d[1] = v0.y; d[2] = v0.z; d[3] = v0.w; d[4] = v1.x; d[5] = v1.y; d[6] = v1.z; d[7] = v1.w;
  }
}

template <int CONV>
__device__ __forceinline__ uint4 convert_raw8 (const uint16_t *lut, const uint32_t *d)
{
  if (CONV == kConvMagU16) { uint4 pk; pk.x = d[0]; pk.y = d[1]; pk.z = d[2]; pk.w = d[3]; return pk; }
  if (CONV == kConvSc16Int) {
    uint4 pk;
    pk.x = mag_sc16_pair (d[0], d[1]); pk.y = mag_sc16_pair (d[2], d[3]);
    pk.z = mag_sc16_pair (d[4], d[5]); pk.w = mag_sc16_pair (d[6], d[7]);
    return pk;
  }
  uint16_t out[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if (CONV == kConvUc8Lut || CONV == kConvUc8LutPower)
      out[k] = lut[(d[k >> 2] >> (8 * (k & 3))) & 0xFF];
    else if (CONV == kConvUc8Generic) {
      const uint32_t iqp = d[k >> 1] >> (16 * (k & 1));
      out[k] = mag_from_float (c_uc8_float[iqp & 0xFF], c_uc8_float[(iqp >> 8) & 0xFF], nullptr);
    } else {
      const int I = int (int16_t (d[k] & 0xFFFF)), Q = int (int16_t (d[k] >> 16));
      if (CONV == kConvSc16Int) out[k] = uint16_t (mag_sc16_word (d[k]));
      else {
        const float sc = CONV == kConvSc16Generic ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
        out[k] = mag_from_float (float (I) * sc, float (Q) * sc, nullptr);
      }
    }
  }
  uint4 pk;
  pk.x = out[0] | (uint32_t (out[1]) << 16); pk.y = out[2] | (uint32_t (out[3]) << 16);
  pk.z = out[4] | (uint32_t (out[5]) << 16); pk.w = out[6] | (uint32_t (out[7]) << 16);
  return pk;
}


// kTileCount (a multiple of 8) samples starting at sample index a_lo (a multiple of 8; may be negative or run
// past the end: those read the halo / zeros).  `lut` = the as-written uc8 table in LDS.
// PAD: 16 bytes of padding behind every 256 samples of the tile (sample i sits at i + 8 * (i >> 8)): lanes that read
// the tile 256 samples apart -- the eight 32-step segments of the demod_8000 trigger kernel -- then hit different banks.
__device__ __forceinline__ constexpr int tile_pad (int i) { return i + 8 * (i >> 8); }
template <int CONV, int kTileCount, bool PAD = false>
__device__ __forceinline__ void stage_tile (uint16_t *tile, const uint16_t *lut, const ScanParams &p, long long a_lo, int lane)
{
  constexpr int kTileChunks = kTileCount / 8;
  constexpr int kTileIters = (kTileChunks + 63) / 64;
  const long long N = (long long) (CONV == kConvMagU16 ? p.navail : p.nsamples);
  constexpr int NW = RawWords<CONV>::n;
  if (a_lo >= 0 && a_lo + kTileCount <= N) {
    // interior strip: every load is issued before the first conversion
    uint32_t raw[kTileIters][NW];
#pragma unroll
    for (int it = 0; it < kTileIters; it++) {
      const int ch = it * 64 + lane;
      if (ch < kTileChunks) load_raw8<CONV> (p, a_lo + 8 * ch, raw[it]);
    }
#pragma unroll
    for (int it = 0; it < kTileIters; it++) {
      const int ch = it * 64 + lane;
      if (ch < kTileChunks) *reinterpret_cast<uint4 *> (&tile[PAD ? tile_pad (8 * ch) : 8 * ch]) = convert_raw8<CONV> (lut, raw[it]);
    }
    return;
  }
  // first / last strips of the shard: per-sample bounds
  for (int c = lane * 8; c < kTileCount; c += 64 * 8) {
    const long long n0 = a_lo + c;
    uint4 pk;
    if (n0 >= 0 && n0 + 8 <= N) {
      uint32_t d[NW];
      load_raw8<CONV> (p, n0, d);
      pk = convert_raw8<CONV> (lut, d);
    } else {
      uint16_t out[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const long long n = n0 + k;
        uint16_t m = 0;
        if (n < 0) {
          if (p.halo && n + (long long) p.ov >= 0) m = p.halo[n + (long long) p.ov];
        } else if (n < N) {
          if (CONV == kConvMagU16) m = reinterpret_cast<const uint16_t *> (p.iq)[n];
          else if (CONV == kConvUc8Lut || CONV == kConvUc8LutPower) {
            const uint64_t blk = uint64_t (n) / p.block_samples, i = uint64_t (n) % p.block_samples;
            m = lut[p.iq[blk * 2 * p.block_samples + i]];
          } else if (CONV == kConvUc8Generic) {
            m = mag_from_float (c_uc8_float[p.iq[2 * n]], c_uc8_float[p.iq[2 * n + 1]], nullptr);
          } else {
            const int16_t *s = reinterpret_cast<const int16_t *> (p.iq) + 2 * n;
            if (CONV == kConvSc16Int) m = mag_sc16_int (s[0], s[1]);
            else {
              const float sc = CONV == kConvSc16Generic ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
              m = mag_from_float (float (s[0]) * sc, float (s[1]) * sc, nullptr);
            }
          }
        }
        out[k] = m;
      }
      pk.x = out[0] | (uint32_t (out[1]) << 16); pk.y = out[2] | (uint32_t (out[3]) << 16);
      pk.z = out[4] | (uint32_t (out[5]) << 16); pk.w = out[6] | (uint32_t (out[7]) << 16);
    }
    *reinterpret_cast<uint4 *> (&tile[PAD ? tile_pad (c) : c]) = pk;
  }
}

// Position-aligned staging: tile[x] = magnitude of sample s0 + x for x in [0, kCount).  `tile` is 16-byte aligned with
// kTileLead uint16 of slack in front (a chunk that straddles s0 spills there) and room for 8 * ((kCount + 14) / 8)
// entries behind.  Global loads stay 8-sample aligned; the realignment happens in the LDS store.
constexpr int kTileLead = 8;
template <int kCount> struct TileChunks { static constexpr int n = (kCount + 14) / 8; };

template <int CONV>
__device__ __forceinline__ uint16_t mag_of_sample (const uint16_t *lut, const ScanParams &p, long long n, long long N)
{
  if (n < 0) return (p.halo && n + (long long) p.ov >= 0) ? p.halo[n + (long long) p.ov] : uint16_t (0);
  if (n >= N) return 0;
  if (CONV == kConvMagU16) return reinterpret_cast<const uint16_t *> (p.iq)[n];
  if (CONV == kConvUc8Lut || CONV == kConvUc8LutPower) {
    const uint64_t blk = uint64_t (n) / p.block_samples, i = uint64_t (n) % p.block_samples;
    return lut[p.iq[blk * 2 * p.block_samples + i]];
  }
  if (CONV == kConvUc8Generic) return mag_from_float (c_uc8_float[p.iq[2 * n]], c_uc8_float[p.iq[2 * n + 1]], nullptr);
  const int16_t *s = reinterpret_cast<const int16_t *> (p.iq) + 2 * n;
  if (CONV == kConvSc16Int) return mag_sc16_int (s[0], s[1]);
  const float sc = CONV == kConvSc16Generic ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
  return mag_from_float (float (s[0]) * sc, float (s[1]) * sc, nullptr);
}

// 16 bytes per lane from per-lane global addresses to consecutive LDS bytes behind a wave-uniform base, no register in
// between (global_load_lds_dwordx4); the caller waits with s_waitcnt vmcnt(0) in front of its barrier
__device__ __forceinline__ void glds16 (const uint8_t *src, uint8_t *lds_wave_base)
{
  __builtin_amdgcn_global_load_lds ((const __attribute__ ((address_space (1))) void *) src,
                                    (__attribute__ ((address_space (3))) void *) lds_wave_base, 16, 0, 0);
}

// Magnitudes that lie in HBM as they are wanted in the tile (kConvMagU16): `bytes` (a multiple of 16) from `src`
// (16-byte aligned, all readable) to `tile` (16-byte aligned), 1 KiB pieces dealt round the kWaves waves of the workgroup.
template <int kWaves>
__device__ __forceinline__ void stage_u16_dma (uint16_t *tile, const uint16_t *src, int bytes, int wv, int lane)
{
  const int pieces = bytes >> 10, tail16 = (bytes & 1023) >> 4;
  const uint8_t *s8 = reinterpret_cast<const uint8_t *> (src);
  uint8_t *t8 = reinterpret_cast<uint8_t *> (tile);
  for (int pc = wv; pc < pieces; pc += kWaves) glds16 (s8 + 1024 * pc + 16 * lane, t8 + 1024 * pc);
  if (tail16 && wv == pieces % kWaves && lane < tail16) glds16 (s8 + 1024 * pieces + 16 * lane, t8 + 1024 * pieces);
}

// kLanes = 64: a wave stages its own tile; kLanes = 256: the four waves of a workgroup stage a shared tile (lane = threadIdx.x)
template <int CONV, int kCount, int kLanes = 64>
__device__ __forceinline__ void stage_tile_pos (uint16_t *tile, const uint16_t *lut, const ScanParams &p, long long s0, int lane)
{
  constexpr int kChunks = TileChunks<kCount>::n;
  constexpr int kIters = (kChunks + kLanes - 1) / kLanes;
  constexpr int NW = RawWords<CONV>::n;
  const long long N = (long long) (CONV == kConvMagU16 ? p.navail : p.nsamples);
  const long long a_lo = (s0 >= 0 ? s0 : s0 - 7) / 8 * 8;
  const int shift = int (s0 - a_lo);                           // 0..7, the same for every wave of a launch
  if (a_lo >= 0 && a_lo + 8 * kChunks <= N) {
    uint32_t raw[kIters][NW];
#pragma unroll
    for (int it = 0; it < kIters; it++) {
      const int ch = it * kLanes + lane;
      if (ch < kChunks) load_raw8<CONV> (p, a_lo + 8 * ch, raw[it]);
    }
#pragma unroll
    for (int it = 0; it < kIters; it++) {
      const int ch = it * kLanes + lane;
      if (ch < kChunks) {
        const uint4 pk = convert_raw8<CONV> (lut, raw[it]);
        uint16_t *dst = tile + 8 * ch - shift;
        if (shift == 0) *reinterpret_cast<uint4 *> (dst) = pk;
        else if (!(shift & 1)) {
          uint32_t *d32 = reinterpret_cast<uint32_t *> (dst);
          d32[0] = pk.x; d32[1] = pk.y; d32[2] = pk.z; d32[3] = pk.w;
        } else {
          dst[0] = uint16_t (pk.x); dst[1] = uint16_t (pk.x >> 16); dst[2] = uint16_t (pk.y); dst[3] = uint16_t (pk.y >> 16);
          dst[4] = uint16_t (pk.z); dst[5] = uint16_t (pk.z >> 16); dst[6] = uint16_t (pk.w); dst[7] = uint16_t (pk.w >> 16);
        }
      }
    }
    return;
  }
  // first / last strips of the shard: per-sample bounds
  for (int c = lane * 8; c < (kCount + 7) / 8 * 8; c += kLanes * 8) {
    uint16_t out[8];
#pragma unroll
    for (int k = 0; k < 8; k++) out[k] = mag_of_sample<CONV> (lut, p, s0 + c + k, N);
    uint4 pk;
    pk.x = out[0] | (uint32_t (out[1]) << 16); pk.y = out[2] | (uint32_t (out[3]) << 16);
    pk.z = out[4] | (uint32_t (out[5]) << 16); pk.w = out[6] | (uint32_t (out[7]) << 16);
    *reinterpret_cast<uint4 *> (&tile[c]) = pk;
  }
}

// sample by sample, every bound checked (first / last tiles of a shard)
template <int CONV, int kLanes>
__device__ __forceinline__ void stage_span_slow (uint16_t *tile, const uint16_t *lut, const ScanParams &p, long long s0, int count, int lane)
{
  const long long N = (long long) (CONV == kConvMagU16 ? p.navail : p.nsamples);
  for (int c = lane * 8; c < (count + 7) / 8 * 8; c += kLanes * 8) {
    uint16_t out[8];
#pragma unroll
    for (int k = 0; k < 8; k++) out[k] = mag_of_sample<CONV> (lut, p, s0 + c + k, N);
    uint4 pk;
    pk.x = out[0] | (uint32_t (out[1]) << 16); pk.y = out[2] | (uint32_t (out[3]) << 16);
    pk.z = out[4] | (uint32_t (out[5]) << 16); pk.w = out[6] | (uint32_t (out[7]) << 16);
    *reinterpret_cast<uint4 *> (&tile[c]) = pk;
  }
}

// The same with a run-time length (count <= kMaxCount): the windows a workgroup of d8_window_kernel works on at a time
// lie as far apart as the triggers were.
template <int CONV, int kMaxCount, int kLanes>
__device__ __forceinline__ void stage_span_pos (uint16_t *tile, const uint16_t *lut, const ScanParams &p, long long s0, int count, int lane)
{
  constexpr int kIters = (TileChunks<kMaxCount>::n + kLanes - 1) / kLanes;
  constexpr int NW = RawWords<CONV>::n;
  const int nchunks = (count + 14) / 8;
  const long long N = (long long) (CONV == kConvMagU16 ? p.navail : p.nsamples);
  const long long a_lo = (s0 >= 0 ? s0 : s0 - 7) / 8 * 8;
  const int shift = int (s0 - a_lo);
  if (a_lo >= 0 && a_lo + 8 * nchunks <= N) {
    uint32_t raw[kIters][NW];
#pragma unroll
    for (int it = 0; it < kIters; it++) {
      const int ch = it * kLanes + lane;
      if (ch < nchunks) load_raw8<CONV> (p, a_lo + 8 * ch, raw[it]);
    }
#pragma unroll
    for (int it = 0; it < kIters; it++) {
      const int ch = it * kLanes + lane;
      if (ch < nchunks) {
        const uint4 pk = convert_raw8<CONV> (lut, raw[it]);
        uint16_t *dst = tile + 8 * ch - shift;
        if (shift == 0) *reinterpret_cast<uint4 *> (dst) = pk;
        else if (!(shift & 1)) {
          uint32_t *d32 = reinterpret_cast<uint32_t *> (dst);
          d32[0] = pk.x; d32[1] = pk.y; d32[2] = pk.z; d32[3] = pk.w;
        } else {
          dst[0] = uint16_t (pk.x); dst[1] = uint16_t (pk.x >> 16); dst[2] = uint16_t (pk.y); dst[3] = uint16_t (pk.y >> 16);
          dst[4] = uint16_t (pk.z); dst[5] = uint16_t (pk.z >> 16); dst[6] = uint16_t (pk.w); dst[7] = uint16_t (pk.w >> 16);
        }
      }
    }
    return;
  }
  stage_span_slow<CONV, kLanes> (tile, lut, p, s0, count, lane);
}

// ------------------------------------------------------------------------------------------
// kConvUc8LutR8: the tile holds one byte per sample, the rank of its magnitude among the byte table's values
// (comparisons of ranks are comparisons of magnitudes; rankmag[rank] gives the magnitude back).
// ------------------------------------------------------------------------------------------
constexpr int kTileLead8 = 16;

// rank of a magnitude that is not read through the table (the halo in front of the shard): exact only if it is one of
// the table's values, which a halo formed by this converter always is
__device__ __forceinline__ uint32_t rank_of_mag (const uint16_t *rankmag, uint32_t m, const ScanParams &p)
{
  int lo = 0, hi = 256;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (rankmag[mid] < m) lo = mid + 1; else hi = mid; }
  if (lo == 256 || rankmag[lo] != m) { if (p.inexact) atomicOr (p.inexact, 1u); if (lo == 256) lo = 255; }
  return uint32_t (lo);
}

__device__ __forceinline__ uint32_t rank_of_sample (const uint8_t *rank, const uint16_t *rankmag, const ScanParams &p, long long n, long long N)
{
  if (n < 0) {
    const uint32_t m = (p.halo && n + (long long) p.ov >= 0) ? p.halo[n + (long long) p.ov] : 0u;
    return m ? rank_of_mag (rankmag, m, p) : 0u;                   // the table maps byte 0 to magnitude 0: rank 0
  }
  if (n >= N) return 0u;
  const uint64_t blk = uint64_t (n) / p.block_samples, i = uint64_t (n) % p.block_samples;
  return rank[p.iq[blk * 2 * p.block_samples + i]];
}

template <int kCount, int kLanes>
__device__ __forceinline__ void stage_tile_rank8 (uint8_t *tile, const uint8_t *rank, const uint16_t *rankmag, const ScanParams &p, long long s0, int lane)
{
  constexpr int kChunks = TileChunks<kCount>::n;
  constexpr int kIters = (kChunks + kLanes - 1) / kLanes;
  const long long N = (long long) p.nsamples;
  const long long a_lo = (s0 >= 0 ? s0 : s0 - 7) / 8 * 8;
  const int shift = int (s0 - a_lo);                           // 0..7, the same for every wave of a launch
  if (a_lo >= 0 && a_lo + 8 * kChunks <= N) {
    uint32_t raw[kIters][2];
#pragma unroll
    for (int it = 0; it < kIters; it++) {
      const int ch = it * kLanes + lane;
      if (ch < kChunks) load_raw8<kConvUc8LutR8> (p, a_lo + 8 * ch, raw[it]);
    }
#pragma unroll
    for (int it = 0; it < kIters; it++) {
      const int ch = it * kLanes + lane;
      if (ch < kChunks) {
        uint32_t r[8];
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] = rank[(raw[it][k >> 2] >> (8 * (k & 3))) & 0xFF];
        const uint32_t w0 = r[0] | (r[1] << 8) | (r[2] << 16) | (r[3] << 24), w1 = r[4] | (r[5] << 8) | (r[6] << 16) | (r[7] << 24);
        uint8_t *dst = tile + 8 * ch - shift;
        if (shift == 0) *reinterpret_cast<uint2 *> (dst) = make_uint2 (w0, w1);
        else if (!(shift & 3)) { reinterpret_cast<uint32_t *> (dst)[0] = w0; reinterpret_cast<uint32_t *> (dst)[1] = w1; }
        else if (!(shift & 1)) {
          uint16_t *d16 = reinterpret_cast<uint16_t *> (dst);
          d16[0] = uint16_t (w0); d16[1] = uint16_t (w0 >> 16); d16[2] = uint16_t (w1); d16[3] = uint16_t (w1 >> 16);
        } else {
#pragma unroll
          for (int k = 0; k < 8; k++) dst[k] = uint8_t (r[k]);
        }
      }
    }
    return;
  }
  // first / last strips of the shard: per-sample bounds
  for (int c = lane * 8; c < (kCount + 7) / 8 * 8; c += kLanes * 8) {
    uint32_t w0 = 0, w1 = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) w0 |= rank_of_sample (rank, rankmag, p, s0 + c + k, N) << (8 * k);
#pragma unroll
    for (int k = 0; k < 4; k++) w1 |= rank_of_sample (rank, rankmag, p, s0 + c + 4 + k, N) << (8 * k);
    *reinterpret_cast<uint2 *> (&tile[c]) = make_uint2 (w0, w1);
  }
}

// ------------------------------------------------------------------------------------------
// kConvUc8LutRaw: the tile holds the input bytes themselves.  tile[x] = the byte the as-written converter reads for
// sample a0 + x (a0 a multiple of 16, the tile 16-byte aligned): one global_load_lds_dwordx4 moves 1 KiB (64 lanes x 16
// bytes) from per-lane source addresses to consecutive LDS bytes without touching a register.
// ------------------------------------------------------------------------------------------
// byte the as-written converter reads for sample n (ref: src/convert.c:19,29): byte n % B of block n / B
__device__ __forceinline__ const uint8_t *lut_byte_addr_of (const uint8_t *iq, uint32_t block_samples, long long n)
{
  const uint64_t in_blk = uint64_t (n) & uint64_t (block_samples - 1);       // block_samples is a power of two
  return iq + 2 * (uint64_t (n) - in_blk) + in_blk;
}
__device__ __forceinline__ const uint8_t *lut_byte_addr (const ScanParams &p, long long n) { return lut_byte_addr_of (p.iq, p.block_samples, n); }

// a byte whose table value is the magnitude m (the halo in front of a shard): exact only if m is one of the table's values
__device__ __forceinline__ uint32_t byte_of_mag (const uint16_t *lut, uint32_t m, const ScanParams &p)
{
  for (int b = 0; b < 256; b++) if (lut[b] == m) return uint32_t (b);
  if (p.inexact) atomicOr (p.inexact, 1u);
  return 0u;
}

__device__ __forceinline__ uint32_t raw_byte_of_sample (const uint16_t *lut, const ScanParams &p, long long n, long long N)
{
  if (n < 0) {
    const uint32_t m = (p.halo && n + (long long) p.ov >= 0) ? p.halo[n + (long long) p.ov] : 0u;
    return m ? byte_of_mag (lut, m, p) : 0u;                       // the table maps byte 0 to magnitude 0
  }
  if (n >= N) return 0u;
  return *lut_byte_addr (p, n);
}

// kBytes (a multiple of 16) bytes of tile from sample a0 on, by the kWaves waves of a workgroup
template <int kBytes, int kWaves>
__device__ __forceinline__ void stage_tile_raw (uint8_t *tile, const uint16_t *lut, const ScanParams &p, long long a0, int wv, int lane, int tid)
{
  const long long N = (long long) p.nsamples;
  constexpr int kPieces = kBytes / 1024, kTailLanes = (kBytes % 1024) / 16;
  if (a0 >= 0 && a0 + kBytes <= N) {
#pragma unroll
    for (int pc = 0; pc < (kPieces + kWaves - 1) / kWaves; pc++) {
      const int piece = pc * kWaves + wv;                          // wave-uniform
      if (piece < kPieces) glds16 (lut_byte_addr (p, a0 + 1024ll * piece + 16 * lane), tile + 1024 * piece);
    }
    if (kTailLanes && wv == kPieces % kWaves && lane < kTailLanes)
      glds16 (lut_byte_addr (p, a0 + 1024ll * kPieces + 16 * lane), tile + 1024 * kPieces);
    return;
  }
  // first / last workgroups of the shard: per-sample bounds
  for (int c = tid * 4; c < kBytes; c += 64 * kWaves * 4) {
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) w |= raw_byte_of_sample (lut, p, a0 + c + k, N) << (8 * k);
    *reinterpret_cast<uint32_t *> (tile + c) = w;
  }
}

// Reads of the magnitude tile: `mag` for arithmetic, `ord` where only the order of two samples matters.
template <bool R8> struct TileRef;
template <> struct TileRef<false> {
  const uint16_t *p; const uint16_t *rankmag;
  __device__ __forceinline__ int mag (int i) const { return p[i]; }
  __device__ __forceinline__ uint32_t ord (int i) const { return p[i]; }
  __device__ __forceinline__ TileRef at (int o) const { return TileRef { p + o, rankmag }; }
};
template <> struct TileRef<true> {
  const uint8_t *p; const uint16_t *rankmag;
  __device__ __forceinline__ int mag (int i) const { return rankmag[p[i]]; }
  __device__ __forceinline__ uint32_t ord (int i) const { return p[i]; }
  __device__ __forceinline__ TileRef at (int o) const { return TileRef { p + o, rankmag }; }
};

// ------------------------------------------------------------------------------------------
// one magnitude of the virtual stream V[x] (x = scan position domain), any converter, slow path
// ------------------------------------------------------------------------------------------
template <int CONV>
__device__ __forceinline__ uint16_t mag_of_V (const uint8_t *iq, uint64_t nsamples, const uint16_t *halo, uint32_t ov,
                                              uint32_t block_samples, const uint16_t *lut256, long long x)
{
  if (CONV == kConvMagU16) return (x >= 0 && uint64_t (x) < nsamples) ? reinterpret_cast<const uint16_t *> (iq)[x] : uint16_t (0);
  const long long n = x - (long long) ov;
  if (n < 0) return (halo && x >= 0) ? halo[x] : uint16_t (0);
  if (uint64_t (n) >= nsamples) return 0;
  if (CONV == kConvUc8Lut || CONV == kConvUc8LutPower) {
    const uint64_t blk = uint64_t (n) / block_samples, i = uint64_t (n) % block_samples;
    return lut256[iq[blk * 2 * block_samples + i]];
  }
  if (CONV == kConvUc8Generic) return mag_from_float (c_uc8_float[iq[2 * n]], c_uc8_float[iq[2 * n + 1]], nullptr);
  const int16_t *s = reinterpret_cast<const int16_t *> (iq) + 2 * n;
  if (CONV == kConvSc16Int) return mag_sc16_int (s[0], s[1]);
  const float sc = CONV == kConvSc16Generic ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
  return mag_from_float (float (s[0]) * sc, float (s[1]) * sc, nullptr);
}


}  // namespace modes
