// gfx950 scan kernel of demod_2400_AC (Mode A/C replies at 2.4 MS/s; ref: src/demod-2400AC.c:25-228).
//
// Every position of a block is an independent question ("does an F1/F2 bracket with clean reply slots start
// here?"); only the 69-sample skip behind a hit is ordered, and that is applied afterwards on the short hit
// list.  One wavefront owns one strip of kStrip positions; the strip's magnitudes live in a wave-private LDS tile.
// "ref:" citations are relative to the reference's root.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"
#include "device_common.h"

namespace modes {

namespace {

constexpr int kWavesAC = 4;
constexpr int kTileAC = kStrip + kLookAC + 16 + 16;

struct WaveLdsAC {
  uint16_t tile[kTileAC];
  uint16_t q1[kStrip / 2 + 64];             // F1 holds at x only if V[x+2] <= V[x+1], at x + 2 only if V[x+1] < V[x+2]: at most 2 of any 4 positions
  uint16_t lut[256];
};

}  // namespace

template <int CONV>
__global__ __launch_bounds__ (64 * kWavesAC)
void scanac2400_kernel (const ScanParams p)
{
  __shared__ WaveLdsAC lds[kWavesAC];
  const int lane = lane_id ();
  const int wv = __builtin_amdgcn_readfirstlane (int (threadIdx.x >> 6));    // wave-uniform: what derives from it is scalar arithmetic
  WaveLdsAC &w = lds[wv];
  const uint32_t strip = blockIdx.x * kWavesAC + wv;
  if (strip >= p.nstrips) return;
  for (int i = lane; i < 256; i += 64) w.lut[i] = p.lut256 ? p.lut256[i] : uint16_t (0);

  const long long P0 = (long long) strip * kStrip;
  const long long s0 = P0 - (long long) (CONV == kConvMagU16 ? 0u : p.ov);
  const long long s1 = s0 - 1;
  const long long a_lo = (s1 >= 0 ? s1 : s1 - 15) / 16 * 16;
  const int shift = int (s0 - a_lo);                          // 1..16
  stage_tile<CONV, kTileAC> (w.tile, w.lut, p, a_lo, lane);
  __builtin_amdgcn_s_waitcnt (0xc07f);
  __builtin_amdgcn_wave_barrier ();
  const uint16_t *V = w.tile + shift;

  // (the bulk engine's blocks are a power of two samples long: no 64-bit division there; a caller's mag_buf may be any length)
  const bool pow2 = (p.block_samples & (p.block_samples - 1)) == 0;
  const uint64_t blk = pow2 ? uint64_t (P0) >> __builtin_ctz (p.block_samples) : uint64_t (P0) / p.block_samples;
  const uint64_t jb0 = uint64_t (P0) - blk * p.block_samples;
  const uint64_t mlen = min (uint64_t (p.block_samples), p.nsamples - blk * p.block_samples);
  const uint32_t noise_level = p.noise_level[blk];            // ref: :34-35

  // ---- F1: rising edge, quiet two samples later, 6 dB over the noise (ref: :58-75) ----
  // One ballot per 64-position word, kept by the lane of the word's number (nothing is written inside the loop, so the
  // LDS reads of several words are in flight together); then every lane lists the positions of its own word.
  // The loop is `for (f1_sample = 1; f1_sample < mlen; ...)`: only positions x of the strip with xlo <= x < xhi count.
  const uint32_t nl2 = noise_level * 2;
  u64 mine = 0;
#pragma unroll 8
  for (int s = 0; s < kWords; s++) {
    const int x = 64 * s + lane;
    const uint32_t a = V[x - 1], b = V[x], c = V[x + 1], d = V[x + 2];
    const u64 bm = __ballot (a < b && !(d > b || d > c) && !(nl2 > (b + c) / 2));
    if (lane == s) mine = bm;
  }
  {
    const long long left = (long long) mlen - (long long) jb0;
    const int xhi = left >= kStrip ? kStrip : left <= 0 ? 0 : int (left), xlo = jb0 == 0 ? 1 : 0;
    const int base = 64 * lane;
    u64 lim = ~0ull;
    if (lane >= kWords || base >= xhi) lim = 0;
    else if (base + 64 > xhi) lim = (~0ull) >> (64 - (xhi - base));
    if (xlo > base) lim &= (~0ull) << (xlo - base);
    mine &= lim;
  }
  const int mycnt = __popcll (mine);
  const int incl1 = wave_incl_scan (mycnt, lane);
  const int n1 = __builtin_amdgcn_readlane (incl1, 63);        // <= kStrip / 2 (see q1)
  {
    int slot = incl1 - mycnt;
    for (u64 mm = mine; mm; mm &= mm - 1) w.q1[slot++] = uint16_t (64 * lane + __builtin_ctzll (mm));
  }
  __builtin_amdgcn_s_waitcnt (0xc07f);
  __builtin_amdgcn_wave_barrier ();

  // ---- F2: the bracket's second pulse at +87*14 clocks with the same tests (ref: :105-124); survivors (few) are
  //      compacted in place before the 20-slot loop, so that loop runs with full lanes ----
  int n2 = 0;
  for (int base = 0; base < n1; base += 64) {
    const int qi = base + lane;
    bool ok = false;
    int off = 0;
    if (qi < n1) {
      off = w.q1[qi];
      const uint32_t f1 = uint32_t (jb0) + uint32_t (off);     // index into the block's mag_buf
      const uint16_t *m = V + off;                             // m[k] == data[f1 + k]
      // sub-sample position of F1 from the energy split of its two samples (ref: :97-108)
      const float pa = __fmul_rn (float (m[0]), float (m[0])), pb = __fmul_rn (float (m[1]), float (m[1]));
      const float frac = __fdiv_rn (pb, __fadd_rn (pa, pb));
      const float pos = __fmul_rn (25.0f, __fadd_rn (float (f1), __fmul_rn (frac, frac)));
      const uint32_t f1_clock = uint32_t (__dadd_rn (double (pos), 0.5));
      const uint32_t f2 = (f1_clock + 87 * 14) / 25;
      const uint16_t *n = m + (int (f2) - int (f1));
      ok = n[-1] < n[0] && !(n[2] > n[0] || n[2] > n[1]);                                    // ref: :112-124
      const uint32_t f2_level = (uint32_t (n[0]) + n[1]) / 2;
      ok = ok && !(noise_level * 2 > f2_level);
    }
    const u64 bm = __ballot (ok);
    if (ok) w.q1[n2 + __popcll (bm & ((1ull << lane) - 1))] = uint16_t (off);                // n2 + rank <= qi: in place
    n2 += __popcll (bm);
  }
  __builtin_amdgcn_s_waitcnt (0xc07f);
  __builtin_amdgcn_wave_barrier ();

  Record *out = p.records + size_t (strip) * p.cap;
  uint32_t nrec = 0;
  bool overflow = false;
  for (int base = 0; base < n2 && !overflow; base += 64) {
    const int qi = base + lane;
    bool hit = false;
    uint32_t code = 0, f2_clock = 0;
    int off = 0;
    if (qi < n2) {
      off = w.q1[qi];
      const uint32_t f1 = uint32_t (jb0) + uint32_t (off);
      const uint16_t *m = V + off;
      const uint32_t f1_level = (uint32_t (m[0]) + m[1]) / 2;
      const float pa = __fmul_rn (float (m[0]), float (m[0])), pb = __fmul_rn (float (m[1]), float (m[1]));
      const float frac = __fdiv_rn (pb, __fadd_rn (pa, pb));
      const float pos = __fmul_rn (25.0f, __fadd_rn (float (f1), __fmul_rn (frac, frac)));
      const uint32_t f1_clock = uint32_t (__dadd_rn (double (pos), 0.5));
      f2_clock = f1_clock + 87 * 14;
      const uint32_t f2 = f2_clock / 25;
      const uint16_t *n = m + (int (f2) - int (f1));
      const uint32_t f2_level = (uint32_t (n[0]) + n[1]) / 2;
      {
        const uint32_t top = f1_level > f2_level ? f1_level : f2_level;
        const float mid = __fsqrt_rn (float (noise_level * top));                            // ref: :132
        const uint32_t sig_thr = uint32_t (__dadd_rn (__dmul_rn (double (mid), 1.41421356237309504880), 0.5));
        const uint32_t noise_thr = uint32_t (__dadd_rn (__ddiv_rn (double (mid), 1.41421356237309504880), 0.5));
        uint32_t bits = 0, noisy = 0, unsure = 0, clock = f1_clock;
        for (int b = 0; b < 20; b++, clock += 87) {                                          // ref: :145-173
          const uint16_t *q = m + (int (clock / 25) - int (f1));
          bits <<= 1; noisy <<= 1; unsure <<= 1;
          if (q[2] >= sig_thr) noisy |= 1;
          if (q[0] >= sig_thr || q[1] >= sig_thr) bits |= 1;
          else if (q[0] > noise_thr && q[1] > noise_thr) unsure |= 1;
        }
        if ((bits & 0x80020) == 0x80020 && !(bits & 0x0101B) && !noisy && !unsure) {         // ref: :177-187
          hit = true;
          // reply slot -> code bit (ref: :191-204)
          const uint32_t in[13] = { 0x40000, 0x20000, 0x10000, 0x08000, 0x04000, 0x02000, 0x00800, 0x00400, 0x00200, 0x00100,
                                    0x00080, 0x00040, 0x00004 };
          const uint32_t ob[13] = { 0x0010, 0x1000, 0x0020, 0x2000, 0x0040, 0x4000, 0x0100, 0x0001, 0x0200, 0x0002, 0x0400,
                                    0x0004, 0x0080 };
#pragma unroll
          for (int k = 0; k < 13; k++) if (bits & in[k]) code |= ob[k];
        }
      }
    }
    const u64 bm = __ballot (hit);
    const uint32_t total = nrec + uint32_t (__popcll (bm));
    if (total > p.cap) { overflow = true; break; }
    if (hit) {
      uint4 lo4, hi4 = { 0, 0, 0, 0 };
      lo4.x = uint32_t (off) | (uint32_t (kKindModeA) << 22) | (32u << 24);
      lo4.y = uint32_t (kClsModeA);
      lo4.z = f2_clock;
      lo4.w = ((code >> 8) & 0xFF) | ((code & 0xFF) << 8);
      uint4 *dst = reinterpret_cast<uint4 *> (out + nrec + __popcll (bm & ((1ull << lane) - 1)));
      dst[0] = lo4; dst[1] = hi4;
    }
    nrec = total;
  }
  if (lane == 0) { p.counts[strip] = overflow ? 0xFFFFFFFFu : nrec; p.vp_counts[strip] = 0; }
}

// noise_level of every block from its mean_power (mean_level is never set by the reference; ref: :34-35, src/fifo.h:58)
__global__ void noise_level_kernel (const double *mean_power, uint32_t nblocks, uint32_t *noise_level)
{
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  const double mp = mean_power[b];
  const double sd = __dsqrt_rn (mp);                          // sqrt (mean_power - mean_level * mean_level), mean_level == 0
  const double v = __dadd_rn (__dmul_rn (__dadd_rn (mp, sd), 65535.0), 0.5);
  // out-of-range conversions: what the reference's x86-64 build does (64-bit truncating convert, low half kept)
  noise_level[b] = (v < 9.2e18 && v > -9.2e18) ? uint32_t (uint64_t ((long long) v)) : 0u;
}

// mean_power of every block as the selected converter reports it (ref: src/convert.c:146,153-154: a float running sum
// over the block's samples, returned as double; src/convert.c:60-108: an integer sum / 65535^2).  One lane per block,
// samples in order, because the float sum depends on the order.
template <int CONV>
__global__ void block_power_kernel (const ScanParams p, uint32_t nblocks, double *mean_power)
{
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  const uint64_t n0 = uint64_t (b) * p.block_samples, n1 = min (n0 + p.block_samples, p.nsamples);
  if (CONV == kConvUc8LutPower) {
    unsigned long long acc = 0;
    for (uint64_t n = n0; n < n1; n++) { const uint32_t m = p.lut256[p.iq[2 * n0 + (n - n0)]]; acc += m * m; }
    mean_power[b] = double (acc) / 65535.0 / 65535.0;
  } else if (CONV == kConvSc16Generic || CONV == kConvSc16Q11Generic || CONV == kConvUc8Generic) {
    float acc = 0.0f;
    for (uint64_t n = n0; n < n1; n++) {
      float fI, fQ, sq;
      if (CONV == kConvUc8Generic) { fI = c_uc8_float[p.iq[2 * n]]; fQ = c_uc8_float[p.iq[2 * n + 1]]; }
      else {
        const int16_t *s = reinterpret_cast<const int16_t *> (p.iq) + 2 * n;
        const float sc = CONV == kConvSc16Generic ? (1.0f / 32768.0f) : (1.0f / 2048.0f);
        fI = float (s[0]) * sc; fQ = float (s[1]) * sc;
      }
      (void) mag_from_float (fI, fQ, &sq);
      acc = __fadd_rn (acc, sq);
    }
    mean_power[b] = double (acc);
  } else
    mean_power[b] = 0.0;                                       // converters without power measurement set *out_power = 0
}

hipError_t launch_block_power (const ScanParams &p, uint32_t nblocks, double *mean_power, hipStream_t stream)
{
  if (nblocks == 0) return hipSuccess;
  const dim3 grid ((nblocks + 63) / 64), block (64);
  switch (p.conv) {
    case kConvUc8LutPower:    hipLaunchKernelGGL (block_power_kernel<kConvUc8LutPower>, grid, block, 0, stream, p, nblocks, mean_power); break;
    case kConvUc8Generic:     hipLaunchKernelGGL (block_power_kernel<kConvUc8Generic>, grid, block, 0, stream, p, nblocks, mean_power); break;
    case kConvSc16Generic:    hipLaunchKernelGGL (block_power_kernel<kConvSc16Generic>, grid, block, 0, stream, p, nblocks, mean_power); break;
    case kConvSc16Q11Generic: hipLaunchKernelGGL (block_power_kernel<kConvSc16Q11Generic>, grid, block, 0, stream, p, nblocks, mean_power); break;
    default:                  hipLaunchKernelGGL (block_power_kernel<kConvSc16Int>, grid, block, 0, stream, p, nblocks, mean_power); break;
  }
  return hipGetLastError ();
}

hipError_t upload_constants_ac (const uint32_t *single_bit, const float *uc8_float)
{
  return upload_constants_tu (single_bit, uc8_float);
}

hipError_t launch_noise_level (const double *mean_power, uint32_t nblocks, uint32_t *noise_level, hipStream_t stream)
{
  if (nblocks == 0) return hipSuccess;
  hipLaunchKernelGGL (noise_level_kernel, dim3 ((nblocks + 255) / 256), dim3 (256), 0, stream, mean_power, nblocks, noise_level);
  return hipGetLastError ();
}

hipError_t launch_scanac2400 (const ScanParams &p, hipStream_t stream)
{
  const dim3 grid ((p.nstrips + kWavesAC - 1) / kWavesAC), block (64 * kWavesAC);
  if (p.nstrips == 0) return hipSuccess;
  switch (p.conv) {
    case kConvUc8Lut: case kConvUc8LutPower:
      hipLaunchKernelGGL (scanac2400_kernel<kConvUc8Lut>, grid, block, 0, stream, p); break;
    case kConvUc8Generic:     hipLaunchKernelGGL (scanac2400_kernel<kConvUc8Generic>, grid, block, 0, stream, p); break;
    case kConvSc16Int:        hipLaunchKernelGGL (scanac2400_kernel<kConvSc16Int>, grid, block, 0, stream, p); break;
    case kConvSc16Generic:    hipLaunchKernelGGL (scanac2400_kernel<kConvSc16Generic>, grid, block, 0, stream, p); break;
    case kConvSc16Q11Generic: hipLaunchKernelGGL (scanac2400_kernel<kConvSc16Q11Generic>, grid, block, 0, stream, p); break;
    case kConvMagU16:         hipLaunchKernelGGL (scanac2400_kernel<kConvMagU16>, grid, block, 0, stream, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError ();
}

}  // namespace modes
