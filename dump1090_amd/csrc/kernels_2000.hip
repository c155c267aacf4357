// gfx950 scan kernel of the 2.0 MS/s path (demod_2000, ref: src/demod-2000.c:612-1027).
//
// scan2000: IQ bytes in HBM -> magnitudes (LDS only) -> GT/LT bit planes -> preamble test ->
//           Manchester slicing straight from the bit planes -> error budget -> SNR squelch ->
//           CRC syndrome -> error-table diagnose -> one record per valid preamble.
//
// A data bit of demod_2000 is `m[2i] > m[2i+1]`, an encoding error is `m[2i] == m[2i+1]`
// (ref: src/demod-2000.c:782-836): both are bits of the same GT/LT planes the preamble test uses,
// so a lane slices a whole 112-bit frame with a handful of 64-bit shifts instead of 224 loads.
//
// One wavefront owns one strip of kStrip scan positions; LDS buffers are wave-private except the
// CRC byte table.  "ref:" citations are relative to the reference's root.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"
#include "device_common.h"

namespace modes {

namespace {

// 7.8 KB per wave without the table: five workgroups of four waves per CU (four with the 8.9 KB of round 1)
template <int TILE, bool LUT>
struct WaveLds2000 {
  uint16_t tile[TILE];                      // magnitudes V[P0 - shift ...], shift >= 1
  u64      gt[kPlaneWords2000 + 2];         // bit x: V[P0+x] > V[P0+x+1]
  u64      lt[kPlaneWords2000 + 2];         // bit x: V[P0+x] < V[P0+x+1]
  u64      ev[2][kWords];                   // per position: [0] valid Mode S preamble, [1] Mode A/C reply
  uint16_t q1[kQ1Cap];                      // positions that pass the adjacent-sample relations; stage C is its last reader,
                                            // after it the first kQ2Cap2000 entries hold the event positions: off | vp << 12 | modeA << 13
  uint16_t lut[LUT ? 256 : 1];              // as-written uc8 table
};
static_assert (kQ2Cap2000 <= kQ1Cap, "the event queue lives in the candidate queue");

constexpr int kWaves2000 = 4;

// even bits of x (0, 2, 4 ...) packed into the low 32 bits
__device__ __forceinline__ uint32_t compress_even (u64 x)
{
  x &= 0x5555555555555555ull;
  x = (x | (x >> 1)) & 0x3333333333333333ull;
  x = (x | (x >> 2)) & 0x0F0F0F0F0F0F0F0Full;
  x = (x | (x >> 4)) & 0x00FF00FF00FF00FFull;
  x = (x | (x >> 8)) & 0x0000FFFF0000FFFFull;
  x = (x | (x >> 16));
  return uint32_t (x);
}

__device__ __forceinline__ u64 plane_at (const u64 *plane, int w, int sh)
{
  const u64 a = plane[w], b = plane[w + 1];
  return sh ? (a >> sh) | (b << (64 - sh)) : a;
}

// index of the n-th (1-based) set bit of the 128-bit value e[0..3]; caller guarantees it exists
__device__ __forceinline__ int nth_set_bit (const uint32_t *e, int n)
{
  int base = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int c = __popc (e[k]);
    if (n <= c) {
      uint32_t v = e[k];
      for (int q = 1; q < n; q++) v &= v - 1;
      return base + __builtin_ctz (v);
    }
    n -= c;
    base += 32;
  }
  return 127;
}

__device__ __forceinline__ int len_by_type_dev (int df)    // ref: src/dump1090.c:1427-1432
{
  return (df == 16 || df == 17 || df == 19 || df == 20 || df == 21) ? kLongBits : kShortBits;
}

// crc_checksum (ref: src/crc.c:67-82) over the first `nbytes` bytes held little-endian in w[0..3]
__device__ __forceinline__ uint32_t crc_bytes (const uint32_t *tab, const uint32_t *w, int nbytes)
{
  uint32_t rem = 0;
  for (int i = 0; i < nbytes - 3; i++) {
    const uint32_t b = (w[i >> 2] >> (8 * (i & 3))) & 0xFF;
    rem = ((rem << 8) ^ tab[b ^ ((rem >> 16) & 0xFF)]) & 0xFFFFFFu;
  }
  const int n = nbytes;
  const uint32_t b3 = (w[(n - 3) >> 2] >> (8 * ((n - 3) & 3))) & 0xFF, b2 = (w[(n - 2) >> 2] >> (8 * ((n - 2) & 3))) & 0xFF,
                 b1 = (w[(n - 1) >> 2] >> (8 * ((n - 1) & 3))) & 0xFF;
  return rem ^ (b3 << 16) ^ (b2 << 8) ^ b1;
}

// crc_checksum_diagnose (ref: src/crc.c:314-339): binary search of the sorted table; -1 = not found
__device__ __forceinline__ int table_lookup (const ErrEntry *tab, int n, uint32_t syn)
{
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (tab[mid].syndrome < syn) lo = mid + 1; else hi = mid;
  }
  return (lo < n && tab[lo].syndrome == syn) ? lo : -1;
}

// Mode A/C reply at m[0] (ref: src/demod-2000.c:123-321).  Returns the 13-bit code (0 = none) and the
// value (fs + fn) whose square gives the signal level.
__device__ int detect_mode_A_dev (const uint16_t *m, int *level_out)
{
  // bit of the reply -> bit of the code; "xx" slots raise a flag instead (ref: src/demod-2000.c:26-87)
  const uint32_t kBit[24] = { 0, 0x10, 0x1000, 0x20, 0x2000, 0x40, 0x4000, 0x40000000, 0x100, 0x1, 0x200, 0x2, 0x400, 0x4, 0,
                              0x08000000, 0x04000000, 0x80, 0x02000000, 0x01000000, 0x00800000, 0x00400000, 0x00200000,
                              0x00100000 };
  int m0 = m[0], m1 = m[1], m2, m3, n0, n1, n2, n3;
  if (m0 >= m1) return 0;
  m2 = m[2]; m3 = m[3];
  if (m2 <= m0 || m2 < m3) { m3 = m2; m2 = m0; }
  if (m3 >= m1 || m0 > m2 || m3 > m2) return 0;
  const int f1s = (m1 + m2) - ((m0 + m3) << 1), f1n = (m0 + m3) >> 1;
  if (f1s < 0x07FF || f1s < (f1n << 2)) return 0;

  int phase = (m2 * 20) / (m1 + m2);
  int idx = (phase + 812) / 20;
  n0 = m[idx++]; n1 = m[idx++];
  if (n0 >= n1) return 0;
  n2 = m[idx++];
  if ((phase + 812) % 20) n3 = m[idx++];
  else { n3 = n2; n2 = n0; }
  if (n3 >= n1 || n0 > n2 || n3 > n2) return 0;
  const int f2s = (n1 + n2) - ((n0 + n3) << 1), f2n = (n0 + n3) >> 1;
  if (f2s < 0x07FF || f2s < (f2n << 2)) return 0;

  const int fs = (f1s + f2s) >> 1, fn = (f1n + f2n) >> 1, lolo = fn + (fs >> 2), level = fn + (fs >> 1);
  bool last_one = true;
  int last_bit = f1s, last_space = 0;
  uint32_t code = 0, err = 0;
  phase += 29;
  for (int j = 1; j < 48; phase += 29, j++) {
    idx = 1 + phase / 20;
    int sample = m[idx] - fn;
    if (phase % 20) sample += m[idx + 1] - fn;
    if (j & 1) { last_space = sample; continue; }
    int bit = j >> 1;
    if (sample >= level) {
      code |= kBit[bit--];
      const uint32_t gap = bit == 0 ? 0x80000000u : bit == 14 ? 0x20000000u : kBit[bit];
      if (last_one ? (last_space >= (sample >> 1) || last_space >= last_bit) : (last_space >= (sample >> 1))) err |= gap;
      last_one = true;
    } else {
      const uint32_t gap = bit == 0 ? 0x80000000u : bit == 14 ? 0x20000000u : kBit[bit];
      if (last_one ? (last_space >= last_bit) : (last_space >= lolo)) err |= gap;
      last_one = false;
    }
    last_bit = sample >> 1;
  }
  if (code < 3 || (code & 0xFFFF8808u) || err) return 0;
  *level_out = fs + fn;
  return int (code);
}

// What the bit loop of demod_2000 leaves behind (ref: src/demod-2000.c:768-937), computed from the 112 data
// bits M (bit i of the frame = bit i of M) and the 112 "samples equal" flags E instead of bit by bit.
struct Slice2000 {
  uint32_t W[4];          // message bytes, byte j of W[k] = msg[4k + j]; bytes the slicer never stored are 0
  int msg_len, errors, demod_inc, nbytes_written;
};

__device__ __forceinline__ void finish_slice (uint32_t *M, uint32_t *E, Slice2000 &r)
{
  M[3] &= 0xFFFF; E[3] &= 0xFFFF;
  M[0] |= E[0] & 1;                                                  // an undecided first bit is guessed 1 (ref: :829-835)
  const int nE_ty = __popc (E[0] & 0x1F);
  const uint32_t the_err = __brev (E[0] & 0x1F) >> 24;               // bit i of the type field at 0x80 >> i (ref: :852-854)
  const int df5 = int (__brev (M[0] & 0x1F) >> 27);
  const int msg_len0 = len_by_type_dev (df5);                        // ref: :842-850
  const bool e5_55 = ((E[0] & ~0x1Fu) | (E[1] & 0x00FFFFFFu)) != 0;
  const int scan_len = (nE_ty == 0 && !e5_55) ? msg_len0 : kLongBits;
  if (scan_len == kShortBits) { E[1] &= 0x00FFFFFFu; E[2] = 0; E[3] = 0; }
  const int total = __popc (E[0]) + __popc (E[1]) + __popc (E[2]) + __popc (E[3]);
  int demod_inc = nE_ty == 0 ? 1 : 0;
  int msg_len, errors, errors_ty = nE_ty, last_i = scan_len - 1;
  uint32_t byte0 = __brev (M[0] & 0xFF) >> 24;
  if (total < 4) { msg_len = msg_len0; errors = total; }
  else {                                                             // ref: :858-889, the fourth error ends the frame
    last_i = nth_set_bit (E, 4);
    if (last_i < kShortBits) { msg_len = 0; errors = 4; }
    else {
      const int errors56 = __popc (E[0]) + __popc (E[1] & 0x00FFFFFFu);
      msg_len = kShortBits; errors = errors56;
      if (nE_ty == 1 && the_err == 0x80) { byte0 ^= 0x80; errors_ty = 0; }
    }
  }
  if (msg_len > 0) {                                                 // ref: :894-901
    const int L = len_by_type_dev (int (byte0 >> 3));
    if (msg_len > L) msg_len = L; else if (msg_len < L) msg_len = 0;
  }
  if (msg_len && errors_ty == 1 && (the_err & 0x78)) {               // ref: :908-937
    const uint32_t valid_df = 0x017F0831u;
    if (!((valid_df >> (byte0 >> 3)) & 1)) {
      const uint32_t t = byte0 ^ the_err;
      if ((valid_df >> (t >> 3)) & 1) { byte0 = t; errors--; demod_inc++; }
    }
  }
  r.nbytes_written = (last_i + 1) >> 3;                              // one byte per complete group of 8 processed bits
#pragma unroll
  for (int k = 0; k < 4; k++) r.W[k] = rev_in_bytes (M[k]);
  r.W[0] = (r.W[0] & ~0xFFu) | byte0;
#pragma unroll
  for (int k = 0; k < 4; k++) {                                      // unwritten bytes: the reference keeps stale ones, here 0
    const int keep = r.nbytes_written - 4 * k;
    if (keep <= 0) r.W[k] = 0; else if (keep < 4) r.W[k] &= (1u << (8 * keep)) - 1;
  }
  r.msg_len = msg_len; r.errors = errors; r.demod_inc = demod_inc;
}

// Squelch, CRC and the state-independent part of decode_mode_S_message (ref: src/demod-2000.c:943-954,
// src/dump1090.c:2372-2578); packs the 32-byte record.
__device__ __forceinline__ void make_record (const ScanParams &p, const uint32_t *crc_tab, const Slice2000 &sl, uint32_t sig,
                                             uint32_t noise, int off, uint8_t info, uint4 *lo4, uint4 *hi4)
{
  info |= uint8_t (sl.demod_inc);
  int df = int ((sl.W[0] & 0xFF) >> 3), cls = kClsAP, nerr = 0, b0 = 0, b1 = 0;
  uint32_t crc = 0;
  if (sl.msg_len) {
    while (sig > 65535 || noise > 65535) { sig >>= 1; noise >>= 1; }
    const uint16_t snr = uint16_t (p.log10_lut[sig] - p.log10_lut[noise]);
    if (2 * int (snr) > 40 && sl.errors <= 3) {
      info |= kInfoDecode;
      const int bits = len_by_type_dev (df);
      if (bits / 8 > sl.nbytes_written) info |= kInfoStale;
      crc = crc_bytes (crc_tab, sl.W, bits / 8);
      bool reject = false;
      const bool es = df == 17 || df == 18, acq = df == 11;
      if (es || acq) {                                               // ref: src/dump1090.c:2471-2545
        cls = acq ? kClsDF11 : kClsES;
        const uint32_t key = acq ? (crc & 0xFFFF80u) : crc;
        if (key) {
          const bool shortt = bits == kShortBits;
          const ErrEntry *tab = shortt ? p.short_tab : p.long_tab;
          const int tn = shortt ? p.short_n : p.long_n;
          const uint32_t h = (key ^ (key >> 12)) & 0xFFFF;
          const uint32_t *bloom = shortt ? p.bloom_short : p.bloom_long;
          int hit = -1;
          if (tn > 0 && ((bloom[h >> 5] >> (h & 31)) & 1)) hit = table_lookup (tab, tn, key);
          if (hit < 0) reject = true;
          else {
            const ErrEntry te = tab[hit];
            if (acq && te.nerr > 1) reject = true;
            else { nerr = te.nerr; b0 = te.bit0; b1 = te.bit1; }
          }
        }
      } else if (df == 0 || df == 4 || df == 5 || df == 16 || df == 20 || df == 21 || df >= 24) cls = kClsAP;
      else reject = true;                                            // rc -3 (ref: src/dump1090.c:2573-2577)
      if (reject) info |= kInfoStaticReject;
    }
  }
  lo4->x = uint32_t (off) | (uint32_t (info) << 16) | (uint32_t (df) << 24);
  lo4->y = uint32_t (cls) | (uint32_t (uint8_t (nerr)) << 8) | (uint32_t (uint8_t (b0)) << 16) | (uint32_t (uint8_t (b1)) << 24);
  lo4->z = crc;
  lo4->w = sl.W[0];
  hi4->x = sl.W[1]; hi4->y = sl.W[2];
  hi4->z = (sl.W[3] & 0xFFFF);                                           // msg[12..13], flags = 0, pad[0] = 0
  hi4->w = uint32_t (sl.msg_len) << 8 | (uint32_t (sl.errors & 0xFF) << 16);  // pad[2] = msg_len, pad[3] = errors
}

__device__ __forceinline__ uint32_t clamped_scale_dev (uint32_t v, uint32_t scale)   // ref: src/demod-2000.c:505-512
{
  const uint32_t s = v * scale / 16384u;
  return s > 65535u ? 65535u : s;
}

// demod_2000's retry: apply_phase_correction() on a copy of the frame, then the same bit loop
// (ref: src/demod-2000.c:528-605, 757-759).  pp = preamble start; pp[-1 .. 16+224) readable.
// The correction only ever rescales one sample of each pair from the decision of the neighbouring pair,
// so it is streamed: no 241-sample copy.
__device__ void slice_corrected (const uint16_t *pp, Slice2000 &sl, uint32_t *sig_out, uint32_t *noise_out)
{
  const uint32_t on_time = uint32_t (pp[0]) + pp[2] + pp[7] + pp[9];
  const uint32_t early = (uint32_t (pp[-1]) + pp[6]) << 1, late = (uint32_t (pp[3]) + pp[10]) << 1;
  uint32_t M[4] = { 0, 0, 0, 0 }, E[4] = { 0, 0, 0, 0 };
  uint32_t sig = on_time, noise = uint32_t (pp[1]) + pp[3] + pp[4] + pp[6] + pp[8];
  const uint16_t *q = pp + 16;
  if (on_time == 0 && early == 0 && late == 0) {
    for (int i = 0; i < kLongBits; i++) {
      const uint32_t a = q[2 * i], b = q[2 * i + 1];
      if (a > b) M[i >> 5] |= 1u << (i & 31); else if (a == b) E[i >> 5] |= 1u << (i & 31);
      if (i < kShortBits) { sig += a > b ? a : b; noise += a > b ? b : a; }
    }
  } else if (early > late) {
    const uint32_t up = (16384u + 16384u * early / (early + on_time)) & 0xFFFF,
                   down = (16384u - 16384u * early / (early + on_time)) & 0xFFFF;
    // backwards: the second sample of pair i is rescaled from the decision of pair i+1
    bool next_one = false;                       // pair 112 does not exist: the last sample is scaled up
    for (int i = kLongBits - 1; i >= 0; i--) {
      const uint32_t a = q[2 * i];
      const uint32_t b = clamped_scale_dev (q[2 * i + 1], (i == kLongBits - 1) ? up : (next_one ? down : up));
      if (a > b) M[i >> 5] |= 1u << (i & 31); else if (a == b) E[i >> 5] |= 1u << (i & 31);
      if (i < kShortBits) { sig += a > b ? a : b; noise += a > b ? b : a; }
      next_one = a > b;
    }
  } else {
    const uint32_t up = (16384u + 16384u * late / (late + on_time)) & 0xFFFF,
                   down = (16384u - 16384u * late / (late + on_time)) & 0xFFFF;
    // forwards: the first sample of pair i is rescaled from the decision of pair i-1
    bool prev_one = false;
    for (int i = 0; i < kLongBits; i++) {
      const uint32_t a = clamped_scale_dev (q[2 * i], i == 0 ? up : (prev_one ? up : down));
      const uint32_t b = q[2 * i + 1];
      if (a > b) M[i >> 5] |= 1u << (i & 31); else if (a == b) E[i >> 5] |= 1u << (i & 31);
      if (i < kShortBits) { sig += a > b ? a : b; noise += a > b ? b : a; }
      prev_one = a > b;
    }
  }
  finish_slice (M, E, sl);
  *sig_out = sig; *noise_out = noise;
}

}  // namespace

template <int CONV, bool PE>
__global__ __launch_bounds__ (64 * kWaves2000)
void scan2000_kernel (const ScanParams p)
{
  constexpr int kLook = PE ? kLook2000PE : kLook2000;
  constexpr int kTile = kStrip + kLook + kBack2000 + 16;
  constexpr bool kLut = CONV == kConvUc8Lut || CONV == kConvUc8LutPower;
  __shared__ WaveLds2000<kTile, kLut> lds[kWaves2000];
  __shared__ uint32_t crc_tab[256];
  const int lane = lane_id ();
  const int wv = __builtin_amdgcn_readfirstlane (int (threadIdx.x >> 6));    // wave-uniform: what derives from it is scalar arithmetic
  WaveLds2000<kTile, kLut> &w = lds[wv];
  uint16_t *const q2 = w.q1;
  crc_tab[threadIdx.x & 255] = p.crc_table[threadIdx.x & 255];
  __syncthreads ();
  const uint32_t strip = p.strip_begin + blockIdx.x * kWaves2000 + wv;
  if (strip >= (p.strip_end ? p.strip_end : p.nstrips)) return;

  if (kLut) for (int i = lane; i < 256; i += 64) w.lut[i] = p.lut256 ? p.lut256[i] : uint16_t (0);
  if (lane < kWords) { w.ev[0][lane] = 0; w.ev[1][lane] = 0; }

  // ---- stage A: magnitudes of V[P0 - 1 .. P0 + kStrip + kLook] into LDS ----
  const long long P0 = (long long) strip * kStrip;
  const long long s0 = P0 - (long long) (CONV == kConvMagU16 ? 0u : p.ov);
  const long long s1 = s0 - 1;
  const long long a_lo = (s1 >= 0 ? s1 : s1 - 15) / 16 * 16;
  const int shift = int (s0 - a_lo);                          // 1..16: V[P0 + x] == tile[x + shift]
  stage_tile<CONV, kTile> (w.tile, w.lut, p, a_lo, lane);
  __builtin_amdgcn_s_waitcnt (0xc07f);
  __builtin_amdgcn_wave_barrier ();
  const uint16_t *V = w.tile + shift;

  // ---- stage B: GT/LT planes over [0, kStrip + kLook2000), written to LDS ----
  {
    uint32_t lt_lo = 0, lt_hi = 0, gt_lo = 0, gt_hi = 0;
#pragma unroll 4
    for (int s = 0; s < kPlaneWords2000; s++) {
      const int x = 64 * s + lane;
      const uint32_t a = V[x], b = V[x + 1];
      const u64 lt = __ballot (a < b), gt = __ballot (a > b);
      if (lane == s) { lt_lo = uint32_t (lt); lt_hi = uint32_t (lt >> 32); gt_lo = uint32_t (gt); gt_hi = uint32_t (gt >> 32); }
    }
    if (lane < kPlaneWords2000) { w.lt[lane] = (u64 (lt_hi) << 32) | lt_lo; w.gt[lane] = (u64 (gt_hi) << 32) | gt_lo; }
    else if (lane < kPlaneWords2000 + 2) { w.lt[lane] = 0; w.gt[lane] = 0; }
  }
  __builtin_amdgcn_s_waitcnt (0xc07f);
  __builtin_amdgcn_wave_barrier ();

  // adjacent-sample relations of the preamble (ref: src/demod-2000.c:707-716):
  //   p0 > p1, p1 < p2, p2 > p3, p7 > p8, p8 < p9
  u64 any = 0, rising = 0;
  if (lane < kWords) {
    const u64 G = w.gt[lane], Gn = w.gt[lane + 1], L = w.lt[lane], Ln = w.lt[lane + 1];
    any = G & funnel (L, Ln, 1) & funnel (G, Gn, 2) & funnel (G, Gn, 7) & funnel (L, Ln, 8);
    rising = L;                                                // a Mode A/C reply starts with m0 < m1 (ref: :139)
    const long long base = P0 + 64ll * lane, N = (long long) p.nsamples;
    u64 lim = ~0ull;
    if (base >= N) lim = 0;
    else if (base + 64 > N) lim = (~0ull) >> (64 - int (N - base));
    any &= lim; rising &= lim;
  }

  bool overflow = false;

  // ---- Mode A/C replies (Modes.mode_AC only; ref: src/demod-2000.c:677-701): lane l walks the rising edges of word l ----
  if (p.mode_ac) {
    u64 hits = 0;
    for (u64 mm = rising; mm; mm &= mm - 1) {
      const int b = __builtin_ctzll (mm);
      int level;
      if (detect_mode_A_dev (V + 64 * lane + b, &level)) hits |= 1ull << b;
    }
    if (lane < kWords) w.ev[1][lane] = hits;
  }

  // ---- queue 1 ----
  const int mycnt = __popcll (any);
  const int incl = wave_incl_scan (mycnt, lane);
  const int n1 = __builtin_amdgcn_readlane (incl, 63);          // scalar: the loops below branch on the scalar unit
  if (n1 > kQ1Cap) overflow = true;
  if (!overflow) {
    int slot = incl - mycnt;
    for (u64 mm = any; mm; mm &= mm - 1) w.q1[slot++] = uint16_t (64 * lane + __builtin_ctzll (mm));
  }
  __builtin_amdgcn_s_waitcnt (0xc07f);
  __builtin_amdgcn_wave_barrier ();

  // ---- stage C: the remaining preamble relations and the quiet-zone tests, one lane per candidate ----
  for (int base = 0; base < n1 && !overflow; base += 64) {
    const int qi = base + lane;
    bool pass = false;
    int off = 0;
    if (qi < n1) {
      off = w.q1[qi];
      const uint16_t *pp = V + off;
      const int p0 = pp[0], p2 = pp[2], p3 = pp[3], p4 = pp[4], p5 = pp[5], p6 = pp[6], p7 = pp[7], p9 = pp[9];
      pass = p3 < p0 && p4 < p0 && p5 < p0 && p6 < p0 && p9 > p6;                        // ref: :707-716
      if (pass) {
        const int high = (p0 + p2 + p7 + p9) / 6;                                          // ref: :728
        pass = p4 < high && p5 < high && pp[11] < high && pp[12] < high && pp[13] < high && pp[14] < high;   // ref: :729-748
      }
    }
    if (pass) atomicOr (&w.ev[0][off >> 6], 1ull << (off & 63));
  }
  __builtin_amdgcn_s_waitcnt (0xc07f);
  __builtin_amdgcn_wave_barrier ();

  // ---- queue 2: every position with a valid preamble and/or a Mode A/C reply, ascending ----
  int n2 = 0;
  uint32_t nvp = 0;
  {
    const u64 vpm = lane < kWords ? w.ev[0][lane] : 0ull, mam = lane < kWords ? w.ev[1][lane] : 0ull;
    const u64 both = vpm | mam;
    const int c = __popcll (both);
    const int inc = wave_incl_scan (c, lane);
    n2 = __builtin_amdgcn_readlane (inc, 63);
    nvp = uint32_t (wave_incl_scan (__popcll (vpm), lane));
    nvp = uint32_t (__builtin_amdgcn_readlane (int (nvp), 63));
    if (n2 > kQ2Cap2000) overflow = true;
    if (!overflow) {
      int slot = inc - c;
      for (u64 mm = both; mm; mm &= mm - 1) {
        const int b = __builtin_ctzll (mm);
        q2[slot++] = uint16_t ((64 * lane + b) | (int ((vpm >> b) & 1) << 12) | (int ((mam >> b) & 1) << 13));
      }
    }
  }
  __builtin_amdgcn_s_waitcnt (0xc07f);
  __builtin_amdgcn_wave_barrier ();

  // ---- stage D: one lane per event position; up to four records each ----
  //   Mode A/C reply | Mode S slice | retry at the same position | retry behind an accepted, repaired message
  Record *out = p.records + size_t (strip) * p.cap;
  uint32_t nrec = 0;
  // the Mode A/C reply at an event position (ref: src/demod-2000.c:677-701)
  auto mode_a_record = [&] (const uint16_t *pp, int off, uint4 *lo, uint4 *hi) {
    int level = 0;
    const int code = detect_mode_A_dev (pp, &level);
    *lo = uint4 { uint32_t (off) | (uint32_t (kKindModeA) << 22) | (32u << 24), uint32_t (kClsModeA), uint32_t (level),
                  uint32_t ((code >> 8) & 0xFF) | (uint32_t (code & 0xFF) << 8) };
    *hi = uint4 { 0, 0, 0, 0 };
  };
  // the plain slice of a valid preamble, straight from the bit planes (ref: :777-836); returns the slice for the retries
  auto primary_record = [&] (const uint16_t *pp, int off, uint8_t info, Slice2000 &sl, uint4 *lo, uint4 *hi) {
    const int s = off + 16, w0 = s >> 6, sh = s & 63;
    uint32_t M[4], E[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u64 g = plane_at (w.gt, w0 + k, sh), l = plane_at (w.lt, w0 + k, sh);
      M[k] = compress_even (g);
      E[k] = compress_even (~(g | l));
    }
    finish_slice (M, E, sl);
    uint32_t sig = 0, noise = 0;
    if (sl.msg_len) {                                                 // ref: :777-811
      sig = uint32_t (pp[0]) + pp[2] + pp[7] + pp[9]; noise = uint32_t (pp[1]) + pp[3] + pp[4] + pp[6] + pp[8];
      const uint16_t *q = pp + 16;
      for (int i = 0; i < kShortBits; i++) {
        const uint32_t a = q[2 * i], b = q[2 * i + 1];
        sig += a > b ? a : b; noise += a > b ? b : a;
      }
    }
    make_record (p, crc_tab, sl, sig, noise, off, info, lo, hi);
  };
  for (int base = 0; base < n2 && !overflow; base += 64) {
    const int ci = base + lane;
    const bool act = ci < n2;
    if constexpr (!PE) {
      // at most two records per event, and how many is known from the event alone: they go from registers to their slots
      const uint32_t ev = act ? uint32_t (q2[ci]) : 0u;
      const int cnt = int (((ev >> 12) & 1u) + ((ev >> 13) & 1u));
      const int inc = wave_incl_scan (cnt, lane);
      const uint32_t total = nrec + uint32_t (__builtin_amdgcn_readlane (inc, 63));
      if (total > p.cap) { overflow = true; break; }
      uint4 *dst = reinterpret_cast<uint4 *> (out + nrec + (inc - cnt));
      const int off = ev & 0xFFF;
      const uint16_t *pp = V + off;
      if (ev & 0x2000) {
        uint4 lo, hi;
        mode_a_record (pp, off, &lo, &hi);
        if (p.write_records) { dst[0] = lo; dst[1] = hi; }
        dst += 2;
      }
      if (ev & 0x1000) {
        uint4 lo, hi;
        Slice2000 sl;
        primary_record (pp, off, 0, sl, &lo, &hi);
        if (p.write_records) { dst[0] = lo; dst[1] = hi; }
      }
      nrec = total;
      continue;
    }
    uint4 lo4[4], hi4[4];
    int cnt = 0;
    if (act) {
      const uint32_t ev = q2[ci];
      const int off = ev & 0xFFF;
      const uint16_t *pp = V + off;
      if (ev & 0x2000) { mode_a_record (pp, off, &lo4[cnt], &hi4[cnt]); cnt++; }
      if (ev & 0x1000) {
        uint8_t info = 0;
        bool oop = false;
        if (PE && p.phase_enhance)                                        // ref: src/demod-2000.c:479-503
          oop = pp[3] > pp[2] / 3 || pp[10] > pp[9] / 3 || pp[6] > pp[7] / 3 || pp[-1] > pp[0] / 3;
        if (oop) info |= kInfoOutOfPhase;
        Slice2000 sl;
        primary_record (pp, off, info, sl, &lo4[cnt], &hi4[cnt]);
        const uint32_t pinfo = (lo4[cnt].x >> 16) & 0xFF, pnerr = (lo4[cnt].y >> 8) & 0xFF;
        cnt++;
        if (PE && oop) {
          const uint64_t P = uint64_t (P0) + uint64_t (off);
          const uint64_t blk = P / p.block_samples, jb = P % p.block_samples;
          const uint64_t mlen = min (uint64_t (p.block_samples), p.nsamples - blk * p.block_samples);
          if (jb != 0) {                                                  // `j &&` (ref: :1016): retry where the first try failed
            Slice2000 s2; uint32_t sg, ns;
            slice_corrected (pp, s2, &sg, &ns);
            make_record (p, crc_tab, s2, sg, ns, off, uint8_t (kKindRetrySame << 6), &lo4[cnt], &hi4[cnt]);
            cnt++;
          }
          // an accepted message that needed a repair re-enters the loop in correction mode one sample before the
          // position the skip-ahead lands on (ref: :997, 1016-1021)
          const bool may_fix = (pinfo & kInfoDecode) && !(pinfo & kInfoStaticReject) && pnerr > 0;
          const uint64_t step = 2u * (8u + uint32_t (sl.msg_len)) - 1u;
          if (may_fix && jb + step < mlen) {
            Slice2000 s3; uint32_t sg, ns;
            slice_corrected (pp + step, s3, &sg, &ns);
            make_record (p, crc_tab, s3, sg, ns, off, uint8_t (kKindRetryAfter << 6), &lo4[cnt], &hi4[cnt]);
            cnt++;
          }
        }
      }
    }
    const int inc = wave_incl_scan (cnt, lane);
    const uint32_t total = nrec + uint32_t (__builtin_amdgcn_readlane (inc, 63));
    if (total > p.cap) { overflow = true; break; }
    if (p.write_records) {
      uint4 *dst = reinterpret_cast<uint4 *> (out + nrec + (inc - cnt));
      for (int k = 0; k < cnt; k++) { dst[2 * k] = lo4[k]; dst[2 * k + 1] = hi4[k]; }
    }
    nrec = total;
  }

  if (lane == 0) {
    p.counts[strip] = overflow ? 0xFFFFFFFFu : nrec;
    p.vp_counts[strip] = nvp;
  }
}

hipError_t upload_constants_2000 (const uint32_t *single_bit, const float *uc8_float)
{
  return upload_constants_tu (single_bit, uc8_float);
}

hipError_t launch_scan2000 (const ScanParams &p, hipStream_t stream)
{
  const uint32_t nlaunch = (p.strip_end ? p.strip_end : p.nstrips) - p.strip_begin;
  const dim3 grid ((nlaunch + kWaves2000 - 1) / kWaves2000), block (64 * kWaves2000);
  if (p.nstrips == 0 || nlaunch == 0) return hipSuccess;
#define SCAN(C)                                                                              \
  do {                                                                                        \
    if (p.phase_enhance) hipLaunchKernelGGL ((scan2000_kernel<C, true>), grid, block, 0, stream, p);   \
    else                 hipLaunchKernelGGL ((scan2000_kernel<C, false>), grid, block, 0, stream, p);  \
  } while (0)
  switch (p.conv) {
    case kConvUc8Lut: case kConvUc8LutPower: SCAN (kConvUc8Lut); break;
    case kConvUc8Generic:     SCAN (kConvUc8Generic); break;
    case kConvSc16Int:        SCAN (kConvSc16Int); break;
    case kConvSc16Generic:    SCAN (kConvSc16Generic); break;
    case kConvSc16Q11Generic: SCAN (kConvSc16Q11Generic); break;
    case kConvMagU16:         SCAN (kConvMagU16); break;
    default: return hipErrorInvalidValue;
  }
#undef SCAN
  return hipGetLastError ();
}

}  // namespace modes
