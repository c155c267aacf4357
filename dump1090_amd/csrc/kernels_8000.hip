// gfx950 kernels of the 8 MS/s path (demod_8000, ref: src/demod-8000.c:143-565).
//
// The reference walks the magnitude stream once, one bit period (8 samples) per step:
//   * dbuf[j] = m[j] - m[j+4]                                  (4-sample difference, ref: :208-215)
//   * phase[i] = sum of |dbuf| over 56 taps 8 samples apart     (sliding, one sum per sample phase, ref: :235-271)
//   * a slow average of phase[0] decides when a match window opens (ref: :275-302)
//   * a window records 168 steps, then the peak of the match is searched for a message with 16 one-bit shifts of
//     a short and a long candidate (ref: :331-420), and the walk resumes 55 steps back (ref: :493-496).
// Restated as functions of the step index n, which is what lets it run wide:
//   * phase[i](n) is a difference of two column prefix sums                      -> d8_trig_kernel, d8_window_kernel
//   * the average is a recurrence over n that no backtrack disturbs (the restore puts back exactly the value it had
//     at that step); it only matters where 0 < max <= 4576, because av <= 50000000 >> 14 -> d8_acc_kernel (rare)
//   * window openings are the chain s' = first trigger at or after s + 113        -> d8_chain_* kernels
//   * each completed window is independent work                                  -> d8_window_kernel
// What needs the ICAO cache (which candidate wins, accept) is left to the ordered pass over the windows' records.
// "ref:" citations are relative to the reference's root.
#include <limits.h>
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"
#include "device_common.h"

// waves per SIMD the register allocation of the two big kernels aims for
#ifndef D8WIN_WAVES
#define D8WIN_WAVES 7     // 71 registers, nothing spilled; 7 x 22 KB of LDS per CU
#endif
#ifndef D8_WIN_GRIDY
#define D8_WIN_GRIDY 16     // workgroups per chunk of the window kernel
#endif
#ifndef D8TRIG_WAVES
#define D8TRIG_WAVES 5     // measured against 3, 4, 6, 7, 8 with the shared tile (1.83 ms; 6: 1.87, 4: 2.19)
#endif
namespace modes {

namespace {

#ifndef D8_WG_WAVES
#define D8_WG_WAVES 4
#endif
constexpr int kW8 = D8_WG_WAVES;                                           // waves per workgroup
constexpr int kTrigRows = kD8StripSteps + 56;                    // 312 rows of 8 differences per strip
constexpr int kTrigTile = 2520;                                  // 8 * 312 + 4 + 15 alignment, rounded to 8
constexpr int kWinRows = 241;                                    // rows s-4 .. s+236 of a window
constexpr int kWinTile = 1952;                                   // 8 * 241 + 4 + 15, rounded to 8

// D'[g]: the difference stream in the look-ahead domain of the reference's `dbuf = d8m_dbuf + D8M_LOOK_BACK`
// (ref: :208-222): D'[g] = D[g - 544].  tile[x - x0] = V[x], the magnitudes in scan-position coordinates.
template <bool PAD = false>
__device__ __forceinline__ int dprime_tile (const D8Params &q, const uint16_t *tile, long long x0, long long g)
{
  if (g < kD8LookAhead) return (q.dprev && g >= -kD8LookBack) ? q.dprev[g + kD8LookBack] : 0;
  const uint64_t P = uint64_t (g - kD8LookAhead);
  if (P >= q.T) return 0;
  const uint64_t B = q.sp.block_samples;
  const uint64_t blk = q.bshift >= 63 ? 0 : P >> q.bshift;
  const uint64_t j = P - blk * B;
  const uint64_t mlen = min (B, q.T - blk * B);
  const int i = int ((long long) P - x0);
  int v = tile[PAD ? tile_pad (i) : i];
  if (j + 4 < mlen) v -= tile[PAD ? tile_pad (i + 4) : i + 4];     // the last 4 of a block stay raw (ref: :210-215)
  return v;
}

// The four strips of a workgroup share one tile: neighbouring strips overlap by 56 rows, which are then converted once
// per workgroup instead of once per strip (13 % fewer conversions).  tile[kTileLead + i] = V[xlo of the first strip + i].
constexpr int kTrigGroupTile = kTrigTile + (D8_WG_WAVES - 1) * 8 * kD8StripSteps;
struct TrigLds {
  __attribute__ ((aligned (16))) uint16_t tile[kTileLead + 8 * TileChunks<kTrigGroupTile>::n];
  uint16_t lut[256];
};

struct WinLds {
  uint16_t tile[kWinTile];                  // magnitudes; differences are formed from them where they are needed
  int32_t  match[kD8Win];
  uint8_t  par[kD8Win + 8];
  uint16_t lut[256];
};

// d8_window_kernel: the four waves of a workgroup work on four neighbouring windows of the chunk at a time.  Openings
// are at least kD8Net steps apart and mostly not much more, so neighbouring windows overlap by half: where the four fit
// (nearly always) they share ONE position-aligned tile staged by all 256 threads, else every wave stages its own
// quarter of it.
constexpr int kWinNeed = 1944;                                   // samples a window reads: 8 * 241 + 8 differences, each m[j] - m[j+4]
constexpr int kWinQuarter = 1960;                                // 8 * TileChunks<kWinNeed>::n plus the slack in front of the next one
constexpr int kWinGroupTile = kW8 * kWinQuarter;
struct WinPerWave {
  int32_t  match[kD8Win];
  uint8_t  par[kD8Win + 8];
};
struct WinGroupLds {
  __attribute__ ((aligned (16))) uint16_t tile[kTileLead + 8 * TileChunks<kWinGroupTile>::n];
  WinPerWave per[kW8];
  uint16_t lut[256];
};

// The windows at a block edge (d8_window_edge_kernel): their differences come from three places (this call's magnitudes,
// the differences carried in, what the previous call left in the buffer), so they are formed once, by all lanes side by
// side, and the window reads them here.
constexpr int kWinDiffs = kWinRows * 8 + 8;
struct WinEdgeLds {
  WinLds  base;
  int32_t diff[kWinDiffs];
};
// what a wave of either kernel works with
struct WinView { uint16_t *tile; int32_t *match; uint8_t *par; uint16_t *lut; int32_t *diff; };
__device__ __forceinline__ WinView view_of (WinGroupLds *l, int wv) { return { l->tile + kTileLead, l->per[wv].match, l->per[wv].par, l->lut, nullptr }; }
__device__ __forceinline__ WinView view_of (WinLds *l, int wv) { return { l[wv].tile, l[wv].match, l[wv].par, l[wv].lut, nullptr }; }
__device__ __forceinline__ WinView view_of (WinEdgeLds *l, int wv) { return { l[wv].base.tile, l[wv].base.match, l[wv].base.par, l[wv].base.lut, l[wv].diff }; }

// |a - b| + c in one VALU operation (v_sad_u32); the compiler does not form it from the C expression
__device__ __forceinline__ uint32_t sad_u32 (uint32_t a, uint32_t b, uint32_t c)
{
  uint32_t d;
  asm ("v_sad_u32 %0, %1, %2, %3" : "=v" (d) : "v" (a), "v" (b), "v" (c));
  return d;
}

__device__ __forceinline__ int len_by_type8 (int df) { return (df == 16 || df == 17 || df == 19 || df == 20 || df == 21) ? kLongBits : kShortBits; }

__device__ __forceinline__ uint32_t crc_words (const uint32_t *tab, const uint32_t *w, int nbytes)
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

__device__ __forceinline__ int table_lookup8 (const ErrEntry *tab, int n, uint32_t syn)
{
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (tab[mid].syndrome < syn) lo = mid + 1; else hi = mid;
  }
  return (lo < n && tab[lo].syndrome == syn) ? lo : -1;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// per step: the 8 sliding sums, their maximum, and whether the search would open a window there
// ---------------------------------------------------------------------------------------------
template <int CONV, bool ARRAYS>
__device__ __forceinline__ void d8_trig_body (const D8Params &q, uint64_t strip_limit, TrigLds *lds, uint32_t wg_begin)
{
  const int lane = lane_id ();
  const int wv = __builtin_amdgcn_readfirstlane (int (threadIdx.x >> 6));    // wave-uniform: what derives from it is scalar arithmetic
  TrigLds &w = lds[0];
  const uint64_t nstrips = (q.nsteps + kD8StripSteps - 1) / kD8StripSteps;
  const uint64_t strip0 = (uint64_t (blockIdx.x) + wg_begin) * kW8, strip = strip0 + wv;
  if (strip0 >= nstrips || strip0 >= strip_limit) return;                   // (the whole workgroup)
  if (threadIdx.x < 256) w.lut[threadIdx.x] = q.sp.lut256 ? q.sp.lut256[threadIdx.x] : uint16_t (0);
  __syncthreads ();
  {
    // scan position of the first strip's first difference row, and the sample there
    const long long xlo0 = 8 * ((long long) strip0 * kD8StripSteps + 1) - kD8LookAhead;
    const long long slo0 = xlo0 - (long long) (CONV == kConvMagU16 ? 0u : q.sp.ov);
    stage_tile_pos<CONV, kTrigGroupTile, 64 * kW8> (w.tile + kTileLead, w.lut, q.sp, slo0, int (threadIdx.x));
  }
  __syncthreads ();
  if constexpr (!ARRAYS && CONV != kConvMagU16) {
    // the magnitude plane: the workgroup's own 4 x 2048 positions as they lie in the tile, 16 bytes per thread and go
    // (the rows it shares with the next workgroup are that one's; what the last workgroup leaves open in front of T is
    // filled by d8_vmag_fill_kernel)
    if (q.vmag) {
      const long long xlo0 = 8 * ((long long) strip0 * kD8StripSteps + 1) - kD8LookAhead;     // a multiple of 8
#pragma unroll
      for (int it = 0; it < kW8 * 8 * kD8StripSteps / (8 * 64 * kW8); it++) {
        const int i8 = 8 * (it * 64 * kW8 + int (threadIdx.x));
        const long long x = xlo0 + i8;
        if (x >= 0 && uint64_t (x) + 8 <= q.vcap)
          *reinterpret_cast<uint4 *> (q.vmag + x) = *reinterpret_cast<const uint4 *> (w.tile + kTileLead + i8);
      }
    }
  }
  if (strip >= nstrips || strip >= strip_limit) return;
  const uint16_t *tile = w.tile + kTileLead + 8 * kD8StripSteps * wv;       // tile[i] = V[x0 + i] for this wave's strip

  const long long n0 = (long long) strip * kD8StripSteps;
  const long long g0 = 8 * (n0 + 1);                                        // first difference of row 0
  const long long x0 = g0 - kD8LookAhead;                                   // its scan position (may be < 0)

  // `plain`: every difference of the strip is m[j] - m[j+4] inside one block of the stream (see d8_window_kernel)
  const long long Plo = g0 - kD8LookAhead, Phi = Plo + kTrigRows * 8;
  bool plain = Plo >= 0 && uint64_t (Phi) + 4 < q.T;
  if (plain && q.bshift < 63) plain = (uint64_t (Plo) >> q.bshift) == ((uint64_t (Phi) + 4) >> q.bshift);

  // Lane = segment * 8 + column: eight segments of 32 steps slide their own 56-row window sum (ref: :235-271), the
  // eight sums of a step sit in eight neighbouring lanes.
  static_assert (kD8StripSteps == 256, "eight segments of 32 steps");
  const int col = lane & 7, seg = lane >> 3, t0 = seg * 32;
  // The two bit rows are shifted in step by step, newest at the bottom: `over` takes the sign of kD8Ambig - max (max above
  // the ambiguity bound), `zero` the sign of max - 1 (sums stay below 2^23).  Reversed and masked to the steps that exist
  // behind the loop; an ambiguous step is one that is neither.
  uint32_t over = 0, zero = 0;
  auto step_out = [&] (int u, uint32_t psum) {
    uint32_t mxv = psum;
    mxv = max (mxv, dpp_mov<0xB1> (mxv));                // quad_perm [1,0,3,2]
    mxv = max (mxv, dpp_mov<0x4E> (mxv));                // quad_perm [2,3,0,1]
    mxv = max (mxv, dpp_mov<0x141> (mxv));               // row_half_mirror: the other quad of the eight
    over = __builtin_amdgcn_alignbit (over, uint32_t (kD8Ambig) - mxv, 31);
    zero = __builtin_amdgcn_alignbit (zero, mxv - 1u, 31);
    if constexpr (ARRAYS) {
      const long long n = n0 + t0 + u;
      if (uint64_t (n) < q.nsteps && col == 0) { q.p0[n] = psum; q.mx[n] = mxv; }
    }
  };
  if (plain) {
    const uint16_t *tp = tile + 8 * t0 + col;          // row r of this lane's segment: samples 8 (t0 + r) + col and + 4 of the tile
#define TP(r) tp[8 * (r)]
#define TQ(r) tp[8 * (r) + 4]
    uint32_t psum = 0;
#pragma unroll
    for (int r = 0; r < 56; r++) psum = sad_u32 (TP (r), TQ (r), psum);                   // v_sad_u32: |a - b| + c in one operation
#pragma unroll
    for (int u = 0; u < 32; u++) {
      step_out (u, psum);
      psum = sad_u32 (TP (u + 56), TQ (u + 56), psum) - sad_u32 (TP (u), TQ (u), 0u);
    }
#undef TP
#undef TQ
  } else {
    auto absd = [&] (int row) -> uint32_t { const int d = dprime_tile (q, tile, x0, g0 + 8 * row + col); return uint32_t (d < 0 ? -d : d); };
    uint32_t psum = 0;
    for (int r = 0; r < 56; r++) psum += absd (t0 + r);
    for (int u = 0; u < 32; u++) {
      step_out (u, psum);
      if (u + 1 < 32) psum += absd (t0 + u + 56) - absd (t0 + u);
    }
  }
  const long long left = (long long) q.nsteps - (n0 + t0);
  const uint32_t exist = left >= 32 ? ~0u : left <= 0 ? 0u : (1u << int (left)) - 1u;
  const uint32_t on_bits = __brev (over) & exist, am_bits = ~__brev (over | zero) & exist;
  // word (n0 >> 6) + seg / 2 of the bit arrays, half seg & 1: every lane of a segment holds the same 32 bits
  if (col == 0) {
    reinterpret_cast<uint32_t *> (q.trig)[(n0 >> 5) + seg] = on_bits;
    reinterpret_cast<uint32_t *> (q.amb)[(n0 >> 5) + seg] = am_bits;
  }
  uint32_t namb = col == 0 ? uint32_t (__popc (am_bits)) : 0u;
  for (int d = 32; d >= 1; d >>= 1) namb += uint32_t (__shfl_xor (int (namb), d, 64));
  if (lane == 0 && namb) atomicAdd (q.amb_count, namb);
}

template <int CONV>
__global__ __launch_bounds__ (64 * kW8) __attribute__ ((amdgpu_waves_per_eu (D8TRIG_WAVES, D8TRIG_WAVES)))
void d8_trig_kernel (const D8Params q, uint64_t strip_limit, uint32_t wg_begin)
{
  __shared__ TrigLds lds[1];
  d8_trig_body<CONV, false> (q, strip_limit, lds, wg_begin);
}

// the same, and the per-step sums and maxima written out (drop-in boundary, the slow average's steps)
template <int CONV>
__global__ __launch_bounds__ (64 * kW8)
void d8_trig_arrays_kernel (const D8Params q, uint64_t strip_limit, uint32_t wg_begin)
{
  __shared__ TrigLds lds[1];
  d8_trig_body<CONV, true> (q, strip_limit, lds, wg_begin);
}

// positions [from, to) of the magnitude plane, one by one (what the last workgroup of d8_trig_kernel leaves open)
template <int CONV>
__global__ __launch_bounds__ (256)
void d8_vmag_fill_kernel (const D8Params q, long long from, long long to)
{
  const long long x = from + (long long) blockIdx.x * 256 + threadIdx.x;
  if (x < to && uint64_t (x) < q.vcap)
    q.vmag[x] = mag_of_V<CONV> (q.sp.iq, q.sp.nsamples, q.sp.halo, q.sp.ov, q.sp.block_samples, q.sp.lut256, x);
}

// ---------------------------------------------------------------------------------------------
// the slow average, only when some step needs it (ref: :275-288, 302).  One wave: all lanes fetch, lane 0 recurs.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__ (64)
void d8_acc_kernel (const D8Params q, uint64_t limit, int32_t *acc_out)
{
  __shared__ uint32_t p0s[1024], mxs[1024];
  __shared__ u64 ambs[16], trs[16];
  const int lane = threadIdx.x;
  int acc = q.acc0;
  for (uint64_t base = 0; base < limit; base += 1024) {
    const int cnt = int (min (uint64_t (1024), limit - base));
    for (int i = lane; i < cnt; i += 64) { p0s[i] = q.p0[base + i]; mxs[i] = q.mx[base + i]; }
    if (lane < 16) { ambs[lane] = q.amb[(base >> 6) + lane]; trs[lane] = 0; }
    __syncthreads ();
    if (lane == 0) {
      for (int i = 0; i < cnt; i++) {
        acc += int (p0s[i]);
        if (acc > kD8SaneMax || acc < -kD8SaneMax) acc = 0;
        const int av = acc >> 14;
        acc -= av;
        if ((ambs[i >> 6] >> (i & 63)) & 1)
          if (int (mxs[i]) * 2 > av * 3) trs[i >> 6] |= 1ull << (i & 63);
      }
    }
    __syncthreads ();
    if (lane < 16 && trs[lane]) q.trig[(base >> 6) + lane] |= trs[lane];
    __syncthreads ();
  }
  if (lane == 0 && acc_out) *acc_out = acc;
}

// ---------------------------------------------------------------------------------------------
// window openings: s' = first trigger at or after s + 113
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ long long first_trig (const u64 *bm, int nwords, long long rel)
{
  int wi = int (rel >> 6);
  if (wi >= nwords) return -1;
  u64 word = bm[wi] & (~0ull << (rel & 63));
  while (!word) { if (++wi >= nwords) return -1; word = bm[wi]; }
  return (long long) wi * 64 + __builtin_ctzll (word);
}

// the same inside one chunk, where every position fits an int (the hops of the chain kernels are dependent: fewer
// operations per hop is what makes them shorter)
__device__ __forceinline__ int first_trig32 (const u64 *bm, int nwords, int rel)
{
  const uint32_t *bm32 = reinterpret_cast<const uint32_t *> (bm);           // the bitmap in 32-bit words: one shift, one v_ffbl per hop
  const int n32 = 2 * nwords;
  int wi = rel >> 5;
  if (wi >= n32) return -1;
  uint32_t word = bm32[wi] >> (rel & 31);
  if (word) return rel + __builtin_ctz (word);
  while (++wi < n32) { word = bm32[wi]; if (word) return wi * 32 + __builtin_ctz (word); }
  return -1;
}

// A hop of the chain is a dependent LDS read and a dozen dependent operations (260 ns), and a chunk is 580 of them: the
// chunk is cut into kD8Subs parts whose tables (exit offset and number of openings for every entry offset) are formed
// side by side, kD8Subs x 113 threads of 36 hops, and composed afterwards -- kD8Subs dependent look-ups.
constexpr int kD8Subs = 16, kD8SubSteps = kD8ChunkSteps / kD8Subs, kD8ChainThreads = 1024, kD8WalkThreads = 256;
constexpr int kD8SubBytes = 2 * kD8Subs * 128;                  // per chunk: exits and counts of its parts, what D8Params::table holds behind [nchunks][128]
static_assert (kD8SubBytes == D8_TABLE_BYTES_PER_CHUNK - 128, "the hosts size D8Params::table by D8_TABLE_BYTES_PER_CHUNK");
static_assert (kD8SubSteps % 64 == 0 && kD8SubSteps / kD8Net < 255, "a part is whole bitmap words; its count fits a byte");

template <bool COUNT>
__device__ __forceinline__ void d8_sub_tables (const u64 *bm, int len, uint8_t (*sub_tab)[128], uint8_t (*sub_cnt)[128])
{
  for (int item = threadIdx.x; item < kD8Subs * 128; item += kD8ChainThreads) {
    const int j = item >> 7, r = item & 127;
    if (r >= kD8Net) continue;
    const int sublen = max (0, min (kD8SubSteps, len - j * kD8SubSteps));
    const int nw = (sublen + 63) >> 6;
    int pos = r, exit_off = 0, cnt = 0;
    for (;;) {
      if (pos >= sublen) { exit_off = min (pos - sublen, kD8Net - 1); break; }
      const int s = first_trig32 (bm + j * (kD8SubSteps / 64), nw, pos);
      if (s < 0) { exit_off = 0; break; }                      // nothing more in this part: the search goes on at the start of the next
      cnt++;
      pos = s + kD8Net;
    }
    sub_tab[j][r] = uint8_t (exit_off);
    if (COUNT) sub_cnt[j][r] = uint8_t (cnt);
  }
}

// table[chunk][r] = where the search stands, relative to the end of the chunk, when it entered at offset r
__global__ __launch_bounds__ (kD8ChainThreads)
void d8_chain_table_kernel (const D8Params q, uint32_t c_begin)
{
  __shared__ u64 bm[kD8ChunkSteps / 64];
  __shared__ __attribute__ ((aligned (16))) uint8_t sub_tab[2 * kD8Subs][128];         // exits, then counts
  const uint32_t c = c_begin + blockIdx.x;
  const long long n_lo = (long long) c * kD8ChunkSteps, n_hi = min (n_lo + (long long) kD8ChunkSteps, (long long) q.nsteps);
  const int nwords = int ((n_hi - n_lo + 63) >> 6);
  for (int i = threadIdx.x; i < kD8ChunkSteps / 64; i += kD8ChainThreads) bm[i] = i < nwords ? q.trig[(n_lo >> 6) + i] : 0;
  __syncthreads ();
  d8_sub_tables<true> (bm, int (n_hi - n_lo), sub_tab, sub_tab + kD8Subs);
  __syncthreads ();
  // the walk kernel reads them back (they lie behind the chunks' own tables)
  if (threadIdx.x < kD8SubBytes / 16)
    reinterpret_cast<uint4 *> (q.table + size_t (q.nchunks) * 128 + size_t (c) * kD8SubBytes)[threadIdx.x] = reinterpret_cast<const uint4 *> (&sub_tab[0][0])[threadIdx.x];
  const int r = threadIdx.x;
  if (r >= kD8Net) return;
  uint32_t e = uint32_t (r);
#pragma unroll
  for (int j = 0; j < kD8Subs; j++) e = sub_tab[j][e];
  q.table[size_t (c) * 128 + r] = uint8_t (e);
}

// The chain through the chunks is one dependent table look-up per chunk.  Groups of kD8Group chunks compose their
// tables side by side (this kernel), one thread follows the groups' tables (d8_chain_resolve_kernel) and every chunk's
// workgroup of the walk kernel follows its own group up to itself.
constexpr int kD8Group = 64;
__device__ __forceinline__ uint8_t *group_tables (const D8Params &q) { return q.table + size_t (q.nchunks) * D8_TABLE_BYTES_PER_CHUNK; }

__global__ __launch_bounds__ (128)
void d8_chain_group_kernel (const D8Params q, uint32_t g_begin)
{
  __shared__ __attribute__ ((aligned (16))) uint8_t tab[kD8Group * 128];
  const uint32_t g = g_begin + blockIdx.x;
  const uint32_t c0 = g * kD8Group, n = min (uint32_t (kD8Group), q.nchunks - c0);
  const uint4 *src = reinterpret_cast<const uint4 *> (q.table + size_t (c0) * 128);
  for (uint32_t i = threadIdx.x; i < n * 8; i += 128) reinterpret_cast<uint4 *> (tab)[i] = src[i];
  __syncthreads ();
  const uint32_t r = threadIdx.x;
  uint32_t e = r < uint32_t (kD8Net) ? r : 0u;
  for (uint32_t c = 0; c < n; c++) e = tab[c * 128 + e];
  group_tables (q)[size_t (g) * 128 + r] = uint8_t (e);
}

// One thread: what happens before step 0 (a window carried in, triggers among the last steps of the previous
// call, ref: :493-496 re-scans them), then the entry offset of every group of chunks (entry[kD8Group * g]).
// pre[0] = count, pre[1..2] = window starts (< 0) found there.
__global__ __launch_bounds__ (256)
void d8_chain_resolve_kernel (const D8Params q, u64 trig_tail, long long search0, long long *pre, uint32_t g_begin, uint32_t g_end)
{
  // groups [g_begin, g_end) of the chain: the first call (g_begin == 0) also settles what happens before step 0, a later
  // one goes on from the entry the call before left at its last group boundary
  __shared__ __attribute__ ((aligned (16))) uint8_t tab[256 * 128];
  uint32_t e = 0;
  if (threadIdx.x == 0 && g_begin) e = q.entry[size_t (g_begin) * kD8Group];
  if (threadIdx.x == 0 && !g_begin) {
    long long pos = search0;
    int npre = 0;
    if (q.pending < 0) { pre[1 + npre++] = q.pending; pos = q.pending + kD8Net; }
    while (pos < 0) {                                        // steps -64 .. -1 of the previous call
      const int b = int (pos + 64);
      const u64 word = b >= 0 ? trig_tail & (~0ull << b) : trig_tail;
      if (!word) { pos = 0; break; }
      const long long s = (long long) __builtin_ctzll (word) - 64;
      pre[1 + npre++] = s;
      pos = s + kD8Net;
    }
    pre[0] = npre;
    e = uint32_t (min (pos, (long long) (kD8Net - 1)));
    q.entry[0] = uint8_t (e);
  }
  const uint32_t ngroups = g_end;
  for (uint32_t base = g_begin; base < ngroups; base += 256) {
    const uint32_t n = min (256u, ngroups - base);
    const uint4 *src = reinterpret_cast<const uint4 *> (group_tables (q) + size_t (base) * 128);
    for (uint32_t i = threadIdx.x; i < n * 8; i += 256) reinterpret_cast<uint4 *> (tab)[i] = src[i];
    __syncthreads ();
    if (threadIdx.x == 0)
      for (uint32_t g = 0; g < n; g++) { e = tab[g * 128 + e]; if ((base + g + 1) * kD8Group <= q.nchunks) q.entry[(base + g + 1) * kD8Group] = uint8_t (e); }
    __syncthreads ();
  }
}

// Where the chain leaves the shard for every offset it can enter it at: exit_of[e] = start (< 0, relative to the end
// of the shard) of the window still open when the samples end, or 1 = none.  What a rank needs to tell the ranks behind
// it before anybody knows where the chain really enters (modes_engine_d8_chain_map).  One workgroup, thread = entry
// offset; chunks that end a window's length before the end of the shard go through their tables, the rest is walked.
__global__ __launch_bounds__ (128)
void d8_chain_exit_kernel (const D8Params q, int32_t *exit_of)
{
  __shared__ __attribute__ ((aligned (16))) uint8_t tab[128 * 128];
  __shared__ u64 bm[kD8ChunkSteps / 64 + 8];
  const int r = threadIdx.x;
  uint32_t e = r < kD8Net ? uint32_t (r) : 0u;
  const uint64_t nsafe64 = q.nsteps >= uint64_t (kD8Win) ? (q.nsteps - kD8Win) / kD8ChunkSteps : 0;
  const uint32_t nsafe = uint32_t (min (nsafe64, uint64_t (q.nchunks)));
  for (uint32_t base = 0; base < nsafe; base += 128) {
    const uint32_t n = min (128u, nsafe - base);
    const uint4 *src = reinterpret_cast<const uint4 *> (q.table + size_t (base) * 128);
    for (uint32_t i = threadIdx.x; i < n * 8; i += 128) reinterpret_cast<uint4 *> (tab)[i] = src[i];
    __syncthreads ();
    for (uint32_t c = 0; c < n; c++) e = tab[c * 128 + e];
    __syncthreads ();
  }
  const long long n_lo = (long long) nsafe * kD8ChunkSteps, len = (long long) q.nsteps - n_lo;     // < one chunk + one window
  const int nwords = int ((len + 63) >> 6);
  for (int i = threadIdx.x; i < nwords; i += 128) bm[i] = q.trig[(n_lo >> 6) + i];
  __syncthreads ();
  int32_t out = 1;
  long long pos = e;
  while (pos < len) {
    const long long s = first_trig (bm, nwords, pos);
    if (s < 0) break;
    if (s + kD8Win > len) { out = int32_t (s - len); break; }    // the samples end inside this window: nothing opens behind it
    pos = s + kD8Net;
  }
  if (r < kD8Net) exit_of[r] = out;
}

// per chunk: the window openings themselves; the last one may still be open when the stream ends
// The parts' tables of the table kernel give the entry offset and the place in the list of every part by composition,
// then one lane per part writes its openings.
__global__ __launch_bounds__ (kD8WalkThreads)
void d8_chain_walk_kernel (const D8Params q, const long long *pre, long long *pending_out, uint32_t c_begin)
{
  __shared__ u64 bm[kD8ChunkSteps / 64];
  __shared__ __attribute__ ((aligned (16))) uint8_t sub_tab[2 * kD8Subs][128];
  uint8_t (*const sub_cnt)[128] = sub_tab + kD8Subs;
  __shared__ int s_entry[kD8Subs], s_off[kD8Subs], s_whole[kD8Subs], s_cut[kD8Subs];
  __shared__ __attribute__ ((aligned (16))) uint8_t before[(kD8Group - 1) * 128];      // the tables of the chunks of this group in front of this one
  const uint32_t c = c_begin + blockIdx.x;
  const uint32_t c_first = c / kD8Group * kD8Group;
  const long long n_lo = (long long) c * kD8ChunkSteps, n_hi = min (n_lo + (long long) kD8ChunkSteps, (long long) q.nsteps);
  const int nwords = int ((n_hi - n_lo + 63) >> 6);
  const int len = int (n_hi - n_lo);
  for (int i = threadIdx.x; i < kD8ChunkSteps / 64; i += kD8WalkThreads) bm[i] = i < nwords ? q.trig[(n_lo >> 6) + i] : 0;
  for (int i = threadIdx.x; i < kD8SubBytes / 16; i += kD8WalkThreads)
    reinterpret_cast<uint4 *> (&sub_tab[0][0])[i] = reinterpret_cast<const uint4 *> (q.table + size_t (q.nchunks) * 128 + size_t (c) * kD8SubBytes)[i];
  for (uint32_t i = threadIdx.x; i < (c - c_first) * 8; i += kD8WalkThreads)
    reinterpret_cast<uint4 *> (before)[i] = reinterpret_cast<const uint4 *> (q.table + size_t (c_first) * 128)[i];
  __syncthreads ();
  int64_t *ws = q.win_start + size_t (c) * kD8WinCap;
  if (threadIdx.x == 0) {
    int n = 0;
    if (c == 0)
      for (int k = 0; k < int (pre[0]); k++) {
        const long long s = pre[1 + k];
        if (s + kD8Win <= (long long) q.nsteps) ws[n++] = s; else atomicMin (pending_out, s - (long long) q.nsteps);
      }
    uint32_t e = q.entry[c_first];
    for (uint32_t cc = 0; cc < c - c_first; cc++) e = before[cc * 128 + e];
    for (int j = 0; j < kD8Subs; j++) { s_entry[j] = int (e); s_off[j] = n; n += sub_cnt[j][e]; e = sub_tab[j][e]; }
  }
  __syncthreads ();
  // openings at or behind `last_whole` are cut off by the end of the stream
  const long long room = (long long) q.nsteps - kD8Win - n_lo;            // s <= room: the window is whole
  const int last_whole = room >= len ? len : room < 0 ? -1 : int (room);
  if (threadIdx.x < kD8Subs) {
    const int j = threadIdx.x;
    const int sublen = max (0, min (kD8SubSteps, len - j * kD8SubSteps));
    const int nw = (sublen + 63) >> 6;
    int pos = s_entry[j], i = s_off[j], cut = INT_MAX;
    while (pos < sublen) {
      const int s = first_trig32 (bm + j * (kD8SubSteps / 64), nw, pos);
      if (s < 0) break;
      const int S = j * kD8SubSteps + s;                                 // relative to the chunk
      if (S <= last_whole) { if (i < kD8WinCap) ws[i] = n_lo + S; i++; }
      else { cut = S; break; }                                           // the stream ends inside this window: nothing opens after it
      pos = s + kD8Net;
    }
    s_whole[j] = i - s_off[j];
    s_cut[j] = cut;
  }
  __syncthreads ();
  if (threadIdx.x == 0) {
    int n = s_off[0], cut = INT_MAX;
    for (int j = 0; j < kD8Subs; j++) { n += s_whole[j]; cut = min (cut, s_cut[j]); }
    // the first window the stream ends in stays open (a chunk shorter than a window may lie behind it and see triggers too)
    if (cut != INT_MAX) atomicMin (pending_out, n_lo + cut - (long long) q.nsteps);
    q.win_count[c] = uint32_t (n);
  }
}

// ---------------------------------------------------------------------------------------------
// one completed window per wave (ref: :331-420): match curve, peak, the two search strings, 32 candidates
// ---------------------------------------------------------------------------------------------
// PLAIN: only the windows whose differences are all m[j] - m[j+4] inside one block of the stream (all but a few per
// block); !PLAIN: only the others.  Two kernels, so that the rare window's block edges, carried-in differences and
// stale buffer contents cost the common one neither registers nor branches.
template <int CONV, bool PLAIN, typename Lds>
__device__ __forceinline__ void d8_window_body (const D8Params &q, Lds *lds, uint32_t *crc_tab)
{
  constexpr bool GROUP = std::is_same<Lds, WinGroupLds>::value;              // the workgroup's four windows at a time share a tile
  const int lane0 = lane_id ();
  const int wv = __builtin_amdgcn_readfirstlane (int (threadIdx.x >> 6));    // wave-uniform: what derives from it is scalar arithmetic
  const WinView w = view_of (lds, wv);
  int32_t *const diff = w.diff;
  crc_tab[threadIdx.x & 255] = q.sp.crc_table[threadIdx.x & 255];
  const uint32_t chunk = q.chunk0 + blockIdx.x;
  const uint32_t nwin = min (q.win_count[chunk], uint32_t (kD8WinCap));
  if (GROUP) for (int i = int (threadIdx.x); i < 256; i += 64 * kW8) w.lut[i] = q.sp.lut256 ? q.sp.lut256[i] : uint16_t (0);
  else for (int i = lane0; i < 256; i += 64) w.lut[i] = q.sp.lut256 ? q.sp.lut256[i] : uint16_t (0);
  __syncthreads ();

  // The block whose demod_8000 call completes the window that opens at step s, and how far that call's buffer is filled
  // (ref: :203-215); `plain`: every difference of the window is m[j] - m[j+4] inside one block of the stream (no block
  // edge, no carried-in differences, nothing beyond what the call wrote).
  const uint64_t B = q.sp.block_samples;
  auto block_of = [&] (long long s) -> uint64_t { return q.bshift >= 63 ? 0 : uint64_t (8 * (s + kD8Win - 1)) >> q.bshift; };   // by sptr of the last recorded step
  auto fill_of = [&] (uint64_t blkc) -> long long { return (long long) (blkc * B + min (B, q.T - blkc * B)) + kD8LookAhead; };
  auto plain_of = [&] (long long s) -> bool {
    const long long gbase = 8 * (s - 4), Plo = gbase - kD8LookAhead, Phi = Plo + kWinRows * 8 + 8;
    bool plain = Plo >= 0 && uint64_t (Phi) + 4 < q.T && gbase + kWinRows * 8 + 8 < fill_of (block_of (s));
    if (plain && q.bshift < 63) plain = (uint64_t (Plo) >> q.bshift) == ((uint64_t (Phi) + 4) >> q.bshift);
    return plain;
  };

  // The windows of a chunk go round the waves of its workgroups.  Lane j looks at the j-th window of this wave: which of
  // them are this kernel's is settled for all of them at once, and each one's opening step is then a scalar.
  // With a shared tile the workgroup goes through the groups of four in step (every wave of the workgroup takes part in the staging
  // of a group, whether its own window of it is this kernel's or not); otherwise the waves each go their own way.
  const uint32_t stride = gridDim.y * kW8;
  const int64_t *const starts = q.win_start + size_t (chunk) * kD8WinCap;
  for (uint32_t kb0 = blockIdx.y * kW8; kb0 < nwin; kb0 += 64 * stride) {
  const uint32_t kb = kb0 + uint32_t (wv);
  const uint32_t kl = kb + uint32_t (lane0) * stride;
  long long s_lane = 0;
  bool mine = false;
  if (kl < nwin) { s_lane = starts[kl]; mine = plain_of (s_lane) == PLAIN; }
  u64 todo = __ballot (mine);
  const int npick = int (min (64u, (nwin - kb0 + stride - 1) / stride));
  int pick = -1;
  while (true) {
    bool my = true;
    if constexpr (GROUP) { if (++pick >= npick) break; my = (todo >> pick) & 1; }
    else { if (!todo) break; pick = __builtin_ctzll (todo); todo &= todo - 1; }
    const uint32_t k = kb + uint32_t (pick) * stride;
    const long long s = (long long) (u64 (uint32_t (__builtin_amdgcn_readlane (int (uint32_t (u64 (s_lane))), pick)))
                                     | u64 (uint32_t (__builtin_amdgcn_readlane (int (uint32_t (u64 (s_lane) >> 32)), pick))) << 32);
    // the lane number is made opaque per window: what derives from it (shuffle addresses, masks, LDS offsets) is then
    // formed where it is used instead of being held in registers across the whole loop, which cost two waves per SIMD
    int lane = lane0;
    asm volatile ("" : "+v" (lane));
    const long long gbase = 8 * (s - 4);                                     // dw[0] = D'[gbase]
    const long long xlo = gbase - kD8LookAhead;
    const long long slo = xlo - (long long) (CONV == kConvMagU16 ? 0u : q.sp.ov);
    const long long a_lo = (slo >= 0 ? slo : slo - 15) / 16 * 16;
    const long long x0 = xlo - (slo - a_lo);
    const uint64_t blkc = block_of (s);
    const long long fill_g = fill_of (blkc);
    const long long Plo = gbase - kD8LookAhead;

    int tbase;                                                               // tile index of the sample behind D'[gbase]
    if constexpr (GROUP) {
      // the group's span: from the first of its windows' opening steps to the last (listed in order but for what a
      // chunk takes over from the one before, so the four are looked at)
      const uint32_t kf = kb0 + uint32_t (pick) * stride;
      long long s_min = starts[kf], s_max = s_min;
#pragma unroll
      for (uint32_t j = 1; j < uint32_t (kW8); j++) {
        const long long sj = starts[min (kf + j, nwin - 1)];
        s_min = min (s_min, sj); s_max = max (s_max, sj);
      }
      const long long ov = (long long) (CONV == kConvMagU16 ? 0u : q.sp.ov);
      const long long xlo_min = 8 * (s_min - 4) - kD8LookAhead;
      const long long span = 8 * (s_max - s_min) + kWinNeed;
      const bool shared = span <= kWinGroupTile;
      tbase = shared ? int (xlo - xlo_min) : wv * kWinQuarter;
      __syncthreads ();                                                      // every wave is through with the tile of the group before
#pragma unroll 1
      for (int r = 0; r < (shared ? 1 : kW8); r++) {                         // all 256 threads stage: the span, or one window after the other
        const long long x_r = shared ? xlo_min : 8 * ((long long) starts[min (kf + uint32_t (r), nwin - 1)] - 4) - kD8LookAhead;
        const int cnt = shared ? int (span) : kWinNeed;
        if constexpr (CONV == kConvMagU16) {
          // the tile is a copy of what lies in HBM (x_r is a multiple of 8): LDS-DMA, no register and no VALU work
          const uint16_t *src = reinterpret_cast<const uint16_t *> (q.sp.iq) + x_r;
          const int bytes = (2 * cnt + 15) & ~15;
          if (x_r >= 0 && uint64_t (x_r) + uint64_t (bytes / 2) <= q.sp.navail && !(reinterpret_cast<uintptr_t> (src) & 15))
            stage_u16_dma<kW8> (w.tile + r * kWinQuarter, src, bytes, wv, lane0);
          else stage_span_slow<CONV, 64 * kW8> (w.tile + r * kWinQuarter, w.lut, q.sp, x_r, cnt, int (threadIdx.x));
        } else stage_span_pos<CONV, kWinGroupTile, 64 * kW8> (w.tile + r * kWinQuarter, w.lut, q.sp, x_r - ov, cnt, int (threadIdx.x));
      }
      if constexpr (CONV == kConvMagU16) asm volatile ("s_waitcnt vmcnt(0)" ::: "memory");       // the DMA writes land before the barrier lets anyone read
      __syncthreads ();
      if (!my) continue;
    } else {
      __builtin_amdgcn_s_waitcnt (0xc07f);
      __builtin_amdgcn_wave_barrier ();
      stage_tile<CONV, kWinTile> (w.tile, w.lut, q.sp, a_lo, lane);
      __builtin_amdgcn_s_waitcnt (0xc07f);
      __builtin_amdgcn_wave_barrier ();
      tbase = int (Plo - x0);
    }
    auto dval_edge = [&] (int idx) -> int {
      const long long g = gbase + idx;
      if (g < fill_g) return dprime_tile (q, w.tile, x0, g);
      // beyond what this call wrote: the buffer still holds what the previous call put at that index, or the
      // zeros it was allocated with (ref: :106-123, 208-215, 509)
      int d = 0;
      const uint64_t jj = uint64_t (g - kD8LookAhead) - blkc * B;
      if (blkc >= 1 && jj < B && q.bshift < 63) {
        const long long xp = (long long) ((blkc - 1) * B + jj);
        const uint16_t va = mag_of_V<CONV> (q.sp.iq, q.sp.nsamples, q.sp.halo, q.sp.ov, q.sp.block_samples, q.sp.lut256, xp);
        d = va;
        if (jj + 4 < B) d -= mag_of_V<CONV> (q.sp.iq, q.sp.nsamples, q.sp.halo, q.sp.ov, q.sp.block_samples, q.sp.lut256, xp + 4);
      }
      return d;
    };
    if constexpr (!PLAIN) {
      for (int i = lane; i < kWinDiffs; i += 64) diff[i] = dval_edge (i);
      __builtin_amdgcn_s_waitcnt (0xc07f);
      __builtin_amdgcn_wave_barrier ();
    }
    auto dval = [&] (int idx) -> int {
      if constexpr (PLAIN) return int (w.tile[tbase + idx]) - int (w.tile[tbase + idx + 4]);
      else return diff[idx];
    };

    // phase sums of steps s .. s+167 (ref: :235-271): ph_i(t) = sum over rows t .. t+55 of |D'| in column i, rows counted
    // from difference row 5.  Lane = segment * 8 + column: eight segments of 21 steps slide their own window sum, the
    // eight columns of a step sit in eight neighbouring lanes.
    const int col = lane & 7, t0 = (lane >> 3) * 21;
    int start_phase = 0;
    // one step: the best phase (the first maximum wins, ref: :305-307) and the match along the opening step's phase
    auto step_out = [&] (int t, int u, uint32_t psum) {
      uint32_t key = (psum << 3) | uint32_t (7 - col);
      key = max (key, dpp_mov<0xB1> (key));              // quad_perm [1,0,3,2]
      key = max (key, dpp_mov<0x4E> (key));              // quad_perm [2,3,0,1]
      key = max (key, dpp_mov<0x141> (key));             // row_half_mirror: the other quad of the eight
      const int best = 7 - int (key & 7u);
      if (u == 0) start_phase = __builtin_amdgcn_readfirstlane (best);      // lane 0: step 0 of the window
      if (col == 0) w.par[t] = uint8_t (best);
      if (col == start_phase) w.match[t] = int (psum);
    };
    if constexpr (PLAIN) {
      const uint16_t *tp = w.tile + tbase + 5 * 8 + col + 8 * t0;            // row r of this lane's segment: tp[8 r], tp[8 r + 4]
      uint32_t psum = 0;
#pragma unroll 8
      for (int r = 0; r < 56; r++) psum = sad_u32 (tp[8 * r], tp[8 * r + 4], psum);         // v_sad_u32: |a - b| + c in one operation
#pragma unroll 7
      for (int u = 0; u < 21; u++) {
        step_out (t0 + u, u, psum);
        psum = sad_u32 (tp[8 * (u + 56)], tp[8 * (u + 56) + 4], psum) - sad_u32 (tp[8 * u], tp[8 * u + 4], 0u);
      }
    } else {
      auto absd = [&] (int row) -> uint32_t { const int d = dval (5 * 8 + 8 * row + col); return uint32_t (d < 0 ? -d : d); };
      uint32_t psum = 0;
      for (int r = 0; r < 56; r++) psum += absd (t0 + r);
      for (int u = 0; u < 21; u++) {
        step_out (t0 + u, u, psum);
        if (u + 1 < 21) psum += absd (t0 + u + 56) - absd (t0 + u);
      }
    }
    __builtin_amdgcn_s_waitcnt (0xc07f);
    __builtin_amdgcn_wave_barrier ();

    // pick_peak (ref: :529-553): the last index holding the maximum
    uint32_t k0 = 0, k1 = 0;
    for (int i = lane; i < kD8Win - kShortBits; i += 64) {
      const uint32_t a = uint32_t (w.match[i]), b = a + uint32_t (w.match[i + kShortBits]);
      k0 = max (k0, a * 128u + uint32_t (i));
      k1 = max (k1, b * 128u + uint32_t (i));
    }
    for (int d = 32; d >= 1; d >>= 1) { k0 = max (k0, uint32_t (__shfl_xor (int (k0), d, 64))); k1 = max (k1, uint32_t (__shfl_xor (int (k1), d, 64))); }
    const int peak_short = int (k0 & 127), peak_long = int (k1 & 127);
    const long long win_start = 8 * (s + 1);
    const long long g_short = win_start + w.par[peak_short] + (peak_short - 4) * 8;
    const long long g_long = win_start + w.par[peak_long] + (peak_long - 4) * 8;

    // the two bit strings (ref: :352-364): bit = !(sum of three differences < 0), first bit first
    u64 S[2][2];
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      const long long g = pass ? g_long : g_short;
      const int nbits = pass ? 128 : 72;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int kb = h * 64 + lane;
        bool bit = false;
        if (kb < nbits) {
          const int o = int (g + 8 * kb - gbase);
          bit = (dval (o - 1) + dval (o) + dval (o + 1)) >= 0;
        }
        S[pass][h] = __ballot (bit);
      }
    }

    // 32 candidates: lane = pass * 16 + shift (ref: :368-379)
    bool keep = false;
    uint4 lo4 = { 0, 0, 0, 0 }, hi4 = { 0, 0, 0, 0 };
    if (lane < kD8Cand) {
      const int pass = lane >> 4, sh = lane & 15;
      const u64 A = S[pass][0], Bw = S[pass][1];
      const u64 x0b = sh ? (A >> sh) | (Bw << (64 - sh)) : A;
      const u64 x1b = Bw >> sh;
      uint32_t W[4];
      W[0] = rev_in_bytes (uint32_t (x0b)); W[1] = rev_in_bytes (uint32_t (x0b >> 32));
      W[2] = rev_in_bytes (uint32_t (x1b)); W[3] = rev_in_bytes (uint32_t (x1b >> 32)) & 0xFFFF;
      if (!pass) { W[1] &= 0x00FFFFFF; W[2] = 0; W[3] = 0; }                 // memcpy (best_msg, msg, 7) (ref: :374)
      const int valid_bits = pass ? kLongBits : kShortBits;
      const int df = int ((W[0] & 0xFF) >> 3);
      const int bits = len_by_type8 (df);
      keep = valid_bits >= bits;                                              // ref: src/dump1090.c:3286-3292
      if (keep) keep = bits == kLongBits ? (W[0] | W[1] | W[2] | W[3]) != 0 : (W[0] | (W[1] & 0x00FFFFFF)) != 0;
      int cls = kClsAP, nerr = 0, b0 = 0, b1 = 0;
      uint32_t crc = 0;
      if (keep) {
        crc = crc_words (crc_tab, W, bits / 8);
        const bool es = df == 17 || df == 18, acq = df == 11;
        if (es || acq) {
          cls = acq ? kClsDF11 : kClsES;
          const uint32_t key = acq ? (crc & 0xFFFF80u) : crc;
          if (key) {
            const bool shortt = bits == kShortBits;
            const ErrEntry *tab = shortt ? q.sp.short_tab : q.sp.long_tab;
            const int tn = shortt ? q.sp.short_n : q.sp.long_n;
            const uint32_t h = (key ^ (key >> 12)) & 0xFFFF;
            const uint32_t *bloom = shortt ? q.sp.bloom_short : q.sp.bloom_long;
            int hit = -1;
            if (tn > 0 && ((bloom[h >> 5] >> (h & 31)) & 1)) hit = table_lookup8 (tab, tn, key);
            if (hit < 0) keep = false;
            else {
              const ErrEntry te = tab[hit];
              if (acq && te.nerr > 1) keep = false;
              else { nerr = te.nerr; b0 = te.bit0; b1 = te.bit1; }
            }
          }
        } else if (df == 0 || df == 4 || df == 5 || df == 16 || df == 24) cls = kClsAP;
        else if (df == 20 || df == 21) cls = kClsAP2;
        else keep = false;
      }
      // an address only ever enters the ICAO cache through an error-free DF17/18 or DF11 with IID 0
      // (ref: src/dump1090.c:2676-2687): remember every address that could, for the record filter below
      if (keep && q.announce && nerr == 0 && (cls == kClsES ? crc == 0 : cls == kClsDF11 && crc == 0)) {
        uint32_t aa = ((W[0] >> 8) & 0xFF) << 16 | ((W[0] >> 16) & 0xFF) << 8 | (W[0] >> 24);
        atomicOr (&q.announce[aa >> 5], 1u << (aa & 31));
      }
      lo4.x = uint32_t (lane) << 16 | (uint32_t (df) << 24);                  // Record.phase = candidate index
      lo4.y = uint32_t (cls) | (uint32_t (uint8_t (nerr)) << 8) | (uint32_t (uint8_t (b0)) << 16) | (uint32_t (uint8_t (b1)) << 24);
      lo4.z = crc; lo4.w = W[0];
      hi4.x = W[1]; hi4.y = W[2]; hi4.z = W[3] & 0xFFFF; hi4.w = 0;
    }
    const u64 km = __ballot (keep);
    const size_t slot = size_t (chunk) * kD8WinCap + k;
    if (keep) {
      uint4 *dst = reinterpret_cast<uint4 *> (q.records + slot * kD8Cand + __popcll (km & ((1ull << lane) - 1)));
      dst[0] = lo4; dst[1] = hi4;
    }
    if (lane == 0) {
      D8Window h;
      h.s = s; h.g_short = g_short; h.g_long = g_long; h.nrec = uint32_t (__popcll (km)); h.pad = 0;
      q.windows[slot] = h;
    }
  }
  }
}

template <int CONV>
__global__ __launch_bounds__ (64 * kW8) __attribute__ ((amdgpu_waves_per_eu (D8WIN_WAVES, D8WIN_WAVES)))
void d8_window_kernel (const D8Params q)
{
  // the float converters hold too many registers through a staging pass of four rounds of 256 threads, and the magnitudes
  // of the drop-in boundary have nothing to convert: those keep a tile per wave
  constexpr bool kGroup = CONV != kConvSc16Generic && CONV != kConvSc16Q11Generic;
  __shared__ std::conditional_t<kGroup, WinGroupLds, WinLds> lds[kGroup ? 1 : kW8];
  __shared__ uint32_t crc_tab[256];
  d8_window_body<CONV, true> (q, lds, crc_tab);
}

template <int CONV>
__global__ __launch_bounds__ (64 * kW8)
void d8_window_edge_kernel (const D8Params q)
{
  __shared__ WinEdgeLds lds[kW8];
  __shared__ uint32_t crc_tab[256];
  d8_window_body<CONV, false> (q, lds, crc_tab);
}

// the last kD8Overlap differences of the stream, for the next call (ref: :509)
template <int CONV>
__global__ __launch_bounds__ (256)
void d8_tail_kernel (const D8Params q, int32_t *dnext)
{
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= kD8Overlap) return;
  // next call's D'[g'] for g' in [-1384, 544) is this call's D'[g' + T]
  const long long g = (long long) q.T + idx - kD8LookBack;
  int d;
  if (g < kD8LookAhead) d = (q.dprev && g >= -kD8LookBack) ? q.dprev[g + kD8LookBack] : 0;
  else {
    const uint64_t P = uint64_t (g - kD8LookAhead);
    const uint64_t B = q.sp.block_samples;
    const uint64_t blk = q.bshift >= 63 ? 0 : P >> q.bshift;
    const uint64_t j = P - blk * B, mlen = min (B, q.T - blk * B);
    d = mag_of_V<CONV> (q.sp.iq, CONV == kConvMagU16 ? q.sp.navail : q.sp.nsamples, q.sp.halo, q.sp.ov, q.sp.block_samples, q.sp.lut256, (long long) P);
    if (j + 4 < mlen)
      d -= mag_of_V<CONV> (q.sp.iq, CONV == kConvMagU16 ? q.sp.navail : q.sp.nsamples, q.sp.halo, q.sp.ov, q.sp.block_samples, q.sp.lut256, (long long) P + 4);
  }
  dnext[idx] = d;
}

// addresses the ICAO cache holds when the run starts: they can match an address/parity message from the first step on
__global__ void d8_seed_kernel (uint32_t *announce, const uint32_t *addr, uint32_t n)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicOr (&announce[addr[i] >> 5], 1u << (addr[i] & 31));
}

hipError_t launch_d8_seed (uint32_t *announce, const uint32_t *addr, uint32_t n, hipStream_t stream)
{
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL (d8_seed_kernel, dim3 ((n + 255) / 256), dim3 (256), 0, stream, announce, addr, n);
  return hipGetLastError ();
}

// index just behind the last step that needs the running average (0 = none)
__global__ __launch_bounds__ (256)
void d8_last_amb_kernel (const uint64_t *amb, uint64_t w_begin, uint64_t w_end, unsigned long long *limit, uint32_t *flag_host)
{
  const uint64_t i = w_begin + uint64_t (blockIdx.x) * 256 + threadIdx.x;
  if (i < w_end) {
    const uint64_t wv = amb[i];
    if (wv) {
      const unsigned long long v = (unsigned long long) (i * 64 + 64 - __builtin_clzll (wv));
      atomicMax (limit, v);
      if (flag_host) *flag_host = 1u;                          // (pinned host memory: what the host looks at between two ranges)
    }
  }
}

// words [w_begin, w_end) of the plane; flag_host (may be null): a word in pinned host memory, cleared by the caller, set
// when there is such a step
hipError_t launch_d8_last_amb (const uint64_t *amb, uint64_t w_begin, uint64_t w_end, unsigned long long *limit, uint32_t *flag_host, hipStream_t stream)
{
  if (w_end > w_begin)
    hipLaunchKernelGGL (d8_last_amb_kernel, dim3 ((unsigned) ((w_end - w_begin + 255) / 256)), dim3 (256), 0, stream, amb, w_begin, w_end, limit, flag_host);
  return hipGetLastError ();
}

// A record whose score can never reach 0 whatever the run does to the ICAO cache is dead weight for the ordered pass:
// address/parity formats and DF11 with IID != 0 need their address in the cache (ref: src/dump1090.c:3302-3307,
// 3332-3338, 3355-3359), and only announced addresses (plus the cache the run started with) can be there.
__device__ __forceinline__ bool record_relevant (const Record &r, const uint32_t *announce)
{
  if (!announce) return true;
  uint32_t a;
  if (r.cls == kClsAP || r.cls == kClsAP2) a = r.crc;
  else if (r.cls == kClsDF11 && (r.crc & 0x7F) != 0) {
    a = (uint32_t (r.msg[1]) << 16) | (uint32_t (r.msg[2]) << 8) | r.msg[3];
    if (r.nerr >= 1 && r.bit0 >= 8 && r.bit0 <= 31) a ^= 1u << (31 - r.bit0);
  } else return true;
  return a < (1u << 24) && ((announce[a >> 5] >> (a & 31)) & 1);
}

// ---- compaction of the per-window slots into two dense arrays, window order kept ----
// One thread per window slot; chunk_rec[c] (records that survive) and chunk_rec[nchunks + c] (windows left with something
// to score) are summed with one atomic per wave: the launcher clears them first.
constexpr int kCountBlocks = (kD8WinCap + 255) / 256;
__global__ __launch_bounds__ (256)
void d8_count_kernel (const D8Params q, uint32_t *chunk_rec)
{
  const uint32_t c = q.chunk0 + blockIdx.x, nwin = min (q.win_count[c], uint32_t (kD8WinCap));
  const uint32_t k = blockIdx.y * 256 + threadIdx.x;
  uint32_t live = 0;
  if (k < nwin) {
    D8Window &h = q.windows[size_t (c) * kD8WinCap + k];
    Record *src = q.records + (size_t (c) * kD8WinCap + k) * kD8Cand;
    // the verdict is kept in the record: the gather must see what was counted, whatever later windows announce meanwhile
    for (uint32_t r = 0; r < h.nrec; r++) { const bool rel = record_relevant (src[r], q.announce); src[r].flags = rel; live += rel; }
    h.pad = live;                                              // records that survive the filter
  }
  uint32_t recs = live, wins = live != 0;
  for (int d = 32; d >= 1; d >>= 1) { recs += uint32_t (__shfl_xor (int (recs), d, 64)); wins += uint32_t (__shfl_xor (int (wins), d, 64)); }
  if ((threadIdx.x & 63) == 0 && wins) { atomicAdd (&chunk_rec[c], recs); atomicAdd (&chunk_rec[q.nchunks + c], wins); }
}

// Where each chunk of a range of chunks puts its live windows and records in the range's packed arrays (one wave).
// tot[0..3] = live windows, records, largest window count of a chunk, 1 if the range does not fit `cap_w` windows /
// `cap_r` records (or a chunk overflowed its window slots): the gather then leaves the arrays alone.  tot_host gets a
// copy (pinned host memory the host reads once the range's event has fired).
__global__ __launch_bounds__ (64)
void d8_offsets_kernel (const uint32_t *chunk_rec, const uint32_t *win_count, uint32_t nchunks, uint32_t c0, uint32_t n,
                        uint64_t *win_off, uint64_t *rec_off, uint64_t cap_w, uint64_t cap_r, uint64_t *tot, uint64_t *tot_host)
{
  const int lane = threadIdx.x;
  uint64_t wrun = 0, rrun = 0;
  uint32_t mx = 0;
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t c = c0 + base + lane;
    const bool have = base + lane < n;
    const uint64_t r = have ? chunk_rec[c] : 0, w = have ? chunk_rec[nchunks + c] : 0;
    if (have) mx = max (mx, win_count[c]);
    uint64_t ri = r, wi = w;
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t ro = __shfl_up (ri, d, 64), wo = __shfl_up (wi, d, 64);
      if (lane >= d) { ri += ro; wi += wo; }
    }
    if (have) { rec_off[c] = rrun + ri - r; win_off[c] = wrun + wi - w; }
    rrun += __shfl (ri, 63, 64); wrun += __shfl (wi, 63, 64);
  }
  for (int d = 32; d >= 1; d >>= 1) mx = max (mx, uint32_t (__shfl_xor (int (mx), d, 64)));
  if (lane == 0) {
    const uint64_t over = (wrun > cap_w || rrun > cap_r || mx > uint32_t (kD8WinCap)) ? 1 : 0;
    tot[0] = wrun; tot[1] = rrun; tot[2] = mx; tot[3] = over;
    tot_host[0] = wrun; tot_host[1] = rrun; tot_host[2] = mx; tot_host[3] = over;
  }
}

hipError_t launch_d8_offsets (const uint32_t *chunk_rec, const uint32_t *win_count, uint32_t nchunks, uint32_t c0, uint32_t n,
                              uint64_t *win_off, uint64_t *rec_off, uint64_t cap_w, uint64_t cap_r, uint64_t *tot, uint64_t *tot_host,
                              hipStream_t stream)
{
  hipLaunchKernelGGL (d8_offsets_kernel, dim3 (1), dim3 (64), 0, stream, chunk_rec, win_count, nchunks, c0, n, win_off, rec_off, cap_w, cap_r, tot, tot_host);
  return hipGetLastError ();
}

__global__ __launch_bounds__ (256)
void d8_gather_kernel (const D8Params q, const uint64_t *win_off, const uint64_t *rec_off, D8Window *win_out, Record *rec_out,
                       const uint64_t *range_tot)
{
  __shared__ uint32_t pre[kD8WinCap + 1], wpre[kD8WinCap + 1];
  if (range_tot && range_tot[3]) return;                        // the range does not fit where it should go
  const uint32_t c = q.chunk0 + blockIdx.x, nwin = min (q.win_count[c], uint32_t (kD8WinCap));
  if (threadIdx.x == 0) {
    uint32_t run = 0, wrun = 0;
    for (uint32_t k = 0; k < nwin; k++) {
      const uint32_t live = q.windows[size_t (c) * kD8WinCap + k].pad;
      pre[k] = run; wpre[k] = wrun; run += live; wrun += live != 0;
    }
  }
  __syncthreads ();
  for (uint32_t k = threadIdx.x; k < nwin; k += 256) {
    D8Window h = q.windows[size_t (c) * kD8WinCap + k];
    if (!h.pad) continue;
    const uint64_t ro = rec_off[c] + pre[k];
    const Record *src = q.records + (size_t (c) * kD8WinCap + k) * kD8Cand;
    uint32_t n = 0;
    for (uint32_t r = 0; r < h.nrec; r++)
      if (src[r].flags) {
        const uint4 *s4 = reinterpret_cast<const uint4 *> (src + r);
        uint4 *d4 = reinterpret_cast<uint4 *> (rec_out + ro + n);
        d4[0] = s4[0]; d4[1] = s4[1];
        n++;
      }
    h.nrec = n;
    h.pad = uint32_t (ro);
    win_out[win_off[c] + wpre[k]] = h;
  }
}

hipError_t launch_d8_count (const D8Params &q, uint32_t nch, uint32_t *chunk_rec, hipStream_t stream)
{
  if (!nch) return hipSuccess;
  hipError_t he = hipMemsetAsync (chunk_rec + q.chunk0, 0, nch * sizeof (uint32_t), stream);
  if (he == hipSuccess) he = hipMemsetAsync (chunk_rec + q.nchunks + q.chunk0, 0, nch * sizeof (uint32_t), stream);
  if (he != hipSuccess) return he;
  hipLaunchKernelGGL (d8_count_kernel, dim3 (nch, kCountBlocks), dim3 (256), 0, stream, q, chunk_rec);
  return hipGetLastError ();
}

hipError_t launch_d8_gather (const D8Params &q, uint32_t nch, const uint64_t *win_off, const uint64_t *rec_off, D8Window *win_out,
                             Record *rec_out, hipStream_t stream, const uint64_t *range_tot)
{
  if (nch) hipLaunchKernelGGL (d8_gather_kernel, dim3 (nch), dim3 (256), 0, stream, q, win_off, rec_off, win_out, rec_out, range_tot);
  return hipGetLastError ();
}

hipError_t upload_constants_8000 (const uint32_t *single_bit, const float *uc8_float)
{
  return upload_constants_tu (single_bit, uc8_float);
}

#define D8_DISPATCH(KERNEL, GRID, BLOCK, ...)                                                                      \
  switch (q.sp.conv) {                                                                                              \
    case kConvUc8Lut: case kConvUc8LutPower: hipLaunchKernelGGL (KERNEL<kConvUc8Lut>, GRID, BLOCK, 0, stream, __VA_ARGS__); break;   \
    case kConvUc8Generic:     hipLaunchKernelGGL (KERNEL<kConvUc8Generic>, GRID, BLOCK, 0, stream, __VA_ARGS__); break;     \
    case kConvSc16Int:        hipLaunchKernelGGL (KERNEL<kConvSc16Int>, GRID, BLOCK, 0, stream, __VA_ARGS__); break;        \
    case kConvSc16Generic:    hipLaunchKernelGGL (KERNEL<kConvSc16Generic>, GRID, BLOCK, 0, stream, __VA_ARGS__); break;    \
    case kConvSc16Q11Generic: hipLaunchKernelGGL (KERNEL<kConvSc16Q11Generic>, GRID, BLOCK, 0, stream, __VA_ARGS__); break; \
    case kConvMagU16:         hipLaunchKernelGGL (KERNEL<kConvMagU16>, GRID, BLOCK, 0, stream, __VA_ARGS__); break;         \
    default: return hipErrorInvalidValue;                                                                           \
  }

// step_limit: only the strips that hold steps below it (0 = all); the steps of chunks [c_begin, c_begin + c_count) only
// (c_count 0: to the end)
hipError_t launch_d8_trig (const D8Params &q, uint64_t step_limit, hipStream_t stream, uint32_t c_begin, uint32_t c_count)
{
  if (q.nsteps == 0) return hipSuccess;
  static_assert (kD8ChunkSteps % (kD8StripSteps * kW8) == 0, "a chunk is whole workgroups of the trigger kernel");
  uint64_t nstrips = (q.nsteps + kD8StripSteps - 1) / kD8StripSteps;
  const bool to_end = !c_count || uint64_t (c_begin) + c_count >= q.nchunks;
  if (!to_end) nstrips = min (nstrips, (uint64_t (c_begin) + c_count) * (kD8ChunkSteps / kD8StripSteps));
  if (step_limit) nstrips = min (nstrips, (step_limit + kD8StripSteps - 1) / kD8StripSteps);
  const uint32_t wg_begin = c_begin * uint32_t (kD8ChunkSteps / (kD8StripSteps * kW8));
  const uint64_t wg_end = (nstrips + kW8 - 1) / kW8;
  if (wg_end <= wg_begin) return hipSuccess;
  const dim3 grid ((unsigned) (wg_end - wg_begin)), block (64 * kW8);
  if (q.write_arrays) { D8_DISPATCH (d8_trig_arrays_kernel, grid, block, q, nstrips, wg_begin); }
  else {
    D8_DISPATCH (d8_trig_kernel, grid, block, q, nstrips, wg_begin);
    if (q.vmag && q.sp.conv != kConvMagU16 && !step_limit && to_end) {
      // the strips' own positions end at 2048 * (workgroups) - 536: the rest of [0, T) one by one
      const long long own_end = 8ll * kD8StripSteps * kW8 * (long long) wg_end + 8 - kD8LookAhead;
      const long long from = own_end > 0 ? own_end : 0, to = (long long) q.T;
      if (to > from) { D8_DISPATCH (d8_vmag_fill_kernel, dim3 ((unsigned) ((to - from + 255) / 256)), dim3 (256), q, from, to); }
    }
  }
  return hipGetLastError ();
}

hipError_t launch_d8_acc (const D8Params &q, uint64_t limit, int32_t *acc_out, hipStream_t stream)
{
  hipLaunchKernelGGL (d8_acc_kernel, dim3 (1), dim3 (64), 0, stream, q, limit, acc_out);
  return hipGetLastError ();
}

// chunks [c_begin, c_begin + c_count) (c_count 0: to the end)
hipError_t launch_d8_chain_table (const D8Params &q, hipStream_t stream, uint32_t c_begin, uint32_t c_count)
{
  const uint32_t n = c_count ? c_count : q.nchunks - c_begin;
  if (n) hipLaunchKernelGGL (d8_chain_table_kernel, dim3 (n), dim3 (kD8ChainThreads), 0, stream, q, c_begin);
  return hipGetLastError ();
}

hipError_t launch_d8_chain_exit (const D8Params &q, int32_t *exit_of, hipStream_t stream)
{
  hipLaunchKernelGGL (d8_chain_exit_kernel, dim3 (1), dim3 (128), 0, stream, q, exit_of);
  return hipGetLastError ();
}

// behind launch_d8_chain_table
// The chain through chunks [c_begin, c_begin + c_count) (c_count 0: to the end).  Ranges are taken in order, the first
// from chunk 0: a range composes the groups of 64 chunks that end inside it (or with the stream), follows them from the
// entry the range before left at its last group boundary, and walks its own chunks.
hipError_t launch_d8_chain (const D8Params &q, unsigned long long trig_tail, long long search0, long long *pre,
                            long long *pending_out, hipStream_t stream, uint32_t c_begin, uint32_t c_count)
{
  const uint32_t c_end = c_count ? c_begin + c_count : q.nchunks;
  const bool to_end = c_end >= q.nchunks;
  const uint32_t g_begin = c_begin / kD8Group;                                          // groups before it are settled
  const uint32_t g_end = to_end ? (q.nchunks + kD8Group - 1) / kD8Group : c_end / kD8Group;
  if (g_end > g_begin) hipLaunchKernelGGL (d8_chain_group_kernel, dim3 (g_end - g_begin), dim3 (128), 0, stream, q, g_begin);
  if (g_end > g_begin || !c_begin)
    hipLaunchKernelGGL (d8_chain_resolve_kernel, dim3 (1), dim3 (256), 0, stream, q, (u64) trig_tail, search0, pre, g_begin, g_end > g_begin ? g_end : g_begin);
  if (c_end > c_begin) hipLaunchKernelGGL (d8_chain_walk_kernel, dim3 (c_end - c_begin), dim3 (kD8WalkThreads), 0, stream, q, pre, pending_out, c_begin);
  return hipGetLastError ();
}

// `which`: 1 = the windows inside one block, 2 = the windows at a block edge, 3 = both
hipError_t launch_d8_windows (const D8Params &q, uint32_t nch, hipStream_t stream, int which)
{
  if (nch == 0) return hipSuccess;
  const dim3 grid (nch, D8_WIN_GRIDY), block (64 * kW8);
  if (which & 1) {
    if (q.vmag && q.sp.conv != kConvMagU16) {
      // the plane d8_trig_kernel left: the tiles are copies, nothing is converted a second time
      D8Params qm = q;
      qm.sp.conv = kConvMagU16; qm.sp.iq = reinterpret_cast<const uint8_t *> (q.vmag); qm.sp.navail = q.vcap;     // (all of it readable; a plain window reads nothing at or beyond T)
      qm.sp.halo = nullptr; qm.sp.ov = 0;
      hipLaunchKernelGGL (d8_window_kernel<kConvMagU16>, grid, block, 0, stream, qm);
    } else { D8_DISPATCH (d8_window_kernel, grid, block, q); }
  }
  // a few windows per block edge: fewer, longer-lived waves (each looks at 64 windows per pass)
  const dim3 edge_grid (nch, 4);
  if (which & 2) { D8_DISPATCH (d8_window_edge_kernel, edge_grid, block, q); }
  return hipGetLastError ();
}

hipError_t launch_d8_tail (const D8Params &q, int32_t *dnext, hipStream_t stream)
{
  const dim3 grid ((kD8Overlap + 255) / 256), block (256);
  D8_DISPATCH (d8_tail_kernel, grid, block, q, dnext);
  return hipGetLastError ();
}

}  // namespace modes
