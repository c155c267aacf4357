// --strip as a streaming pass on the device (ref: src/dump1090.c:3423-3448, strip_mode_loop): uc8 IQ pairs with both
// |I - 127| and |Q - 127| below `level` are "quiet"; a quiet pair is dropped when it is the 33rd or later of a run of
// quiet pairs (4 * MODES_PREAMBLE_US = 32 are kept), everything else is copied in order.
//
// The reference counts the run in one variable; here  drop[i] = quiet[i] & quiet[i-1] & ... & quiet[i-32]  is formed
// from bit planes (one lane = 64 pairs, five doubling steps and one shift), the kept pairs of a 4096-pair tile are
// counted, the counts scanned, and a second pass compacts every tile through LDS into its place of the output.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

#include "../../include/modes_hip.h"
#include "common.h"
#include "device_common.h"

namespace modes { int set_error (int code, const char *fmt, ...); }
using namespace modes;

namespace {

constexpr int kTilePairs = 4096;                 // pairs per wave: 64 lanes x 64 pairs
constexpr int kKeepRun = 32;                     // ref: 4 * MODES_PREAMBLE_US

// quiet bits of pairs [64 lane, 64 lane + 64) of the tile that starts at pair t0; pairs beyond n read as not quiet
__device__ __forceinline__ u64 quiet_word (const uint8_t *iq, long long p0, long long n, int level)
{
  u64 q = 0;
  if (p0 + 64 <= n) {
    const uint4 *src = reinterpret_cast<const uint4 *> (iq + 2 * p0);
#pragma unroll
    for (int v = 0; v < 8; v++) {
      const uint4 x = src[v];
      const uint32_t w[4] = { x.x, x.y, x.z, x.w };
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int I = int ((w[k >> 1] >> (16 * (k & 1))) & 0xFF), Q = int ((w[k >> 1] >> (16 * (k & 1) + 8)) & 0xFF);
        const bool quiet = abs (I - 127) < level && abs (Q - 127) < level;
        q |= u64 (quiet) << (8 * v + k);
      }
    }
  } else {
    for (int k = 0; k < 64 && p0 + k < n; k++) {
      const int I = iq[2 * (p0 + k)], Q = iq[2 * (p0 + k) + 1];
      if (abs (I - 127) < level && abs (Q - 127) < level) q |= 1ull << k;
    }
  }
  return q;
}

// drop bits of the lane's 64 pairs: bit i = pairs i-32 .. i all quiet.  `prev` = quiet bits of the 64 pairs before.
__device__ __forceinline__ u64 drop_word (u64 q, u64 prev)
{
  // r_k[i] = AND of quiet[i-k+1 .. i] over the 128-bit window (prev : q)
  u64 lo = prev, hi = q;
#pragma unroll
  for (int sh = 1; sh <= 16; sh <<= 1) {         // runs of 2, 4, 8, 16, 32
    const u64 nhi = hi & ((hi << sh) | (lo >> (64 - sh)));
    lo = lo & (lo << sh);                        // bits of `lo` below sh lose their history: they are never used (only bits >= 31 are)
    hi = nhi;
  }
  // 33 = 32 + 1: AND of the run of 32 ending at i with quiet[i-32]
  return hi & ((q << 32) | (prev >> 32));
}

__device__ __forceinline__ u64 lane_drop (const uint8_t *iq, long long n, int level, long long tile0, int lane, u64 carry_bits, u64 *q_out)
{
  const long long p0 = tile0 + 64ll * lane;
  const u64 q = p0 < n ? quiet_word (iq, p0, n, level) : 0ull;
  // the 64 pairs in front: the lane below, or the tile below (recomputed), or what the caller carried in
  const uint32_t plo = dpp_mov<0x138, 0xF, true> (uint32_t (q)), phi = dpp_mov<0x138, 0xF, true> (uint32_t (q >> 32));   // wave_shr:1
  u64 prev = (u64 (phi) << 32) | plo;
  if (lane == 0) prev = tile0 == 0 ? carry_bits : quiet_word (iq, tile0 - 64, n, level);
  *q_out = q;
  return drop_word (q, prev);
}

__global__ __launch_bounds__ (256)
void strip_count_kernel (const uint8_t *iq, long long n, int level, u64 carry_bits, uint32_t *tile_keep)
{
  const int lane = threadIdx.x & 63;
  const long long tile = (long long) blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long tile0 = tile * kTilePairs;
  if (tile0 >= n) return;
  u64 q;
  const u64 drop = lane_drop (iq, n, level, tile0, lane, carry_bits, &q);
  const long long p0 = tile0 + 64ll * lane;
  const int valid = p0 >= n ? 0 : (n - p0 >= 64 ? 64 : int (n - p0));
  const u64 vmask = valid == 64 ? ~0ull : ((1ull << valid) - 1);
  int keep = __popcll (~drop & vmask);
  for (int d = 32; d >= 1; d >>= 1) keep += __shfl_xor (keep, d, 64);
  if (lane == 0) tile_keep[tile] = uint32_t (keep);
}

__global__ __launch_bounds__ (256)
void strip_write_kernel (const uint8_t *iq, long long n, int level, u64 carry_bits, const uint64_t *tile_off, uint8_t *out)
{
  __shared__ uint16_t s_pairs[4][kTilePairs];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long tile = (long long) blockIdx.x * 4 + wv;
  const long long tile0 = tile * kTilePairs;
  if (tile0 >= n) return;
  u64 q;
  const u64 drop = lane_drop (iq, n, level, tile0, lane, carry_bits, &q);
  const long long p0 = tile0 + 64ll * lane;
  const int valid = p0 >= n ? 0 : (n - p0 >= 64 ? 64 : int (n - p0));
  const u64 vmask = valid == 64 ? ~0ull : ((1ull << valid) - 1);
  u64 keep = ~drop & vmask;
  const int mine = __popcll (keep);
  const int incl = wave_incl_scan (mine, lane);
  const int total = __builtin_amdgcn_readlane (incl, 63);
  int slot = incl - mine;
  const uint16_t *src = reinterpret_cast<const uint16_t *> (iq) + p0;
  while (keep) {
    const int b = __builtin_ctzll (keep);
    keep &= keep - 1;
    s_pairs[wv][slot++] = src[b];
  }
  __builtin_amdgcn_s_waitcnt (0xc07f);
  __builtin_amdgcn_wave_barrier ();
  uint16_t *dst = reinterpret_cast<uint16_t *> (out) + tile_off[tile];
  for (int i = lane; i < total; i += 64) dst[i] = s_pairs[wv][i];
}

// exclusive scan of the tile counts: one workgroup, 1024 tiles at a time (the counts are small: <= 4096 each)
__global__ __launch_bounds__ (1024)
void strip_scan_kernel (const uint32_t *tile_keep, uint32_t ntiles, uint64_t *tile_off, uint64_t *total)
{
  __shared__ uint64_t buf[1024];
  __shared__ uint64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads ();
  for (uint32_t base = 0; base < ntiles; base += 1024) {
    const uint32_t i = base + threadIdx.x;
    const uint64_t v = i < ntiles ? tile_keep[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads ();
    for (int d = 1; d < 1024; d <<= 1) {
      const uint64_t o = threadIdx.x >= (unsigned) d ? buf[threadIdx.x - d] : 0;
      __syncthreads ();
      buf[threadIdx.x] += o;
      __syncthreads ();
    }
    if (i < ntiles) tile_off[i] = carry + buf[threadIdx.x] - v;
    __syncthreads ();
    if (threadIdx.x == 1023) carry += buf[1023];
    __syncthreads ();
  }
  if (threadIdx.x == 0) *total = carry;
}

// quiet pairs in a row at the end of the buffer, saturated at 32 (any longer run behaves like 32)
__global__ void strip_tail_kernel (const uint8_t *iq, long long n, int level, uint64_t run_in, uint64_t *run_out)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint64_t c = 0;
  long long i = n - 1;
  for (; i >= 0 && c < uint64_t (kKeepRun); i--) {
    const int I = iq[2 * i], Q = iq[2 * i + 1];
    if (abs (I - 127) < level && abs (Q - 127) < level) c++; else break;
  }
  if (i < 0 && c < uint64_t (kKeepRun)) c += run_in;        // the whole buffer was quiet: the run started before it
  *run_out = c > uint64_t (kKeepRun) ? uint64_t (kKeepRun) : c;
}

struct Scratch { uint32_t *keep = nullptr; uint64_t *off = nullptr, *small = nullptr; size_t cap = 0; };
std::mutex g_mu;
Scratch g_scr[64];

}  // namespace

#define TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return set_error (MODES_E_HIP, "%s: %s", #expr, hipGetErrorString (e_)); } while (0)

extern "C" __attribute__ ((visibility ("default")))
int modes_strip_device (const void *d_iq, size_t nbytes, int level, uint64_t *run_io, void *d_out, size_t *out_bytes, void *stream_v)
{
  hipStream_t stream = static_cast<hipStream_t> (stream_v);
  if (!out_bytes || (!d_iq && nbytes) || (!d_out && nbytes)) return set_error (MODES_E_ARG, "null argument");
  if (reinterpret_cast<uintptr_t> (d_iq) & 15) return set_error (MODES_E_ARG, "modes_strip_device: d_iq must be 16-byte aligned");
  if (reinterpret_cast<uintptr_t> (d_out) & 1) return set_error (MODES_E_ARG, "modes_strip_device: d_out must be 2-byte aligned");
  *out_bytes = 0;
  const long long n = (long long) (nbytes / 2);             // a trailing odd byte is not a pair (ref: the loop ends at the first EOF)
  if (n == 0) return MODES_OK;
  int dev = 0;
  TRY (hipGetDevice (&dev));
  if (dev < 0 || dev >= 64) return set_error (MODES_E_ARG, "device %d", dev);
  const uint64_t run_in = run_io ? *run_io : 0;
  // what the run carried in looks like to the first pairs: the last min(run, 32) pairs in front were quiet
  const uint64_t r = run_in > uint64_t (kKeepRun) ? uint64_t (kKeepRun) : run_in;
  const u64 carry_bits = r == 0 ? 0ull : (~0ull << (64 - r));
  const uint32_t ntiles = uint32_t ((n + kTilePairs - 1) / kTilePairs);
  std::lock_guard<std::mutex> lk (g_mu);
  Scratch &s = g_scr[dev];
  if (ntiles > s.cap) {
    if (s.keep) (void) hipFree (s.keep);
    if (s.off) (void) hipFree (s.off);
    s.keep = nullptr; s.off = nullptr; s.cap = 0;
    const size_t want = ntiles + ntiles / 4 + 64;
    TRY (hipMalloc (reinterpret_cast<void **> (&s.keep), want * sizeof (uint32_t)));
    TRY (hipMalloc (reinterpret_cast<void **> (&s.off), want * sizeof (uint64_t)));
    s.cap = want;
  }
  if (!s.small) TRY (hipMalloc (reinterpret_cast<void **> (&s.small), 2 * sizeof (uint64_t)));
  const uint8_t *iq = static_cast<const uint8_t *> (d_iq);
  const dim3 grid ((ntiles + 3) / 4), block (256);
  hipLaunchKernelGGL (strip_count_kernel, grid, block, 0, stream, iq, n, level, carry_bits, s.keep);
  hipLaunchKernelGGL (strip_scan_kernel, dim3 (1), dim3 (1024), 0, stream, s.keep, ntiles, s.off, s.small);
  hipLaunchKernelGGL (strip_write_kernel, grid, block, 0, stream, iq, n, level, carry_bits, s.off, static_cast<uint8_t *> (d_out));
  hipLaunchKernelGGL (strip_tail_kernel, dim3 (1), dim3 (64), 0, stream, iq, n, level, run_in, s.small + 1);
  TRY (hipGetLastError ());
  uint64_t res[2] = { 0, 0 };
  TRY (hipMemcpyAsync (res, s.small, sizeof res, hipMemcpyDeviceToHost, stream));
  TRY (hipStreamSynchronize (stream));
  *out_bytes = size_t (res[0]) * 2;
  if (run_io) *run_io = res[1];
  return MODES_OK;
}
