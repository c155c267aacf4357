// Host side of libmodes_hip.so: device set-up, the bulk engine, the C ABI of include/modes_hip.h.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <mutex>
#include <string>
#include <chrono>
#include <vector>

#include "../../include/modes_hip.h"
#include "accept.h"
#include "common.h"
#include "kernels.h"
#include "tables.h"

using namespace modes;

namespace {
thread_local std::string g_err;
}

namespace {


int fail (int code, const char *fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start (ap, fmt);
  vsnprintf (buf, sizeof buf, fmt, ap);
  va_end (ap);
  g_err = buf;
  return code;
}

}  // namespace

namespace modes {
// error reporting shared with dropin.cpp
int set_error (int code, const char *fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start (ap, fmt);
  vsnprintf (buf, sizeof buf, fmt, ap);
  va_end (ap);
  g_err = buf;
  return code;
}
}  // namespace modes

namespace {

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return fail (MODES_E_HIP, "%s: %s", #expr, hipGetErrorString (e_));  \
  } while (0)

// Per-process, per-device constant tables.
struct DeviceTables {
  int device = -1;
  int fix_bits = -1;
  ErrEntry *short_tab = nullptr, *long_tab = nullptr;
  int short_n = 0, long_n = 0;
  uint16_t *lut256 = nullptr;
  uint8_t *rank256 = nullptr;          // byte -> rank of its table value; rank -> value (see kConvUc8LutR8)
  uint16_t *rankmag = nullptr;
  uint32_t *bloom_short = nullptr, *bloom_long = nullptr;
  uint16_t *log10_lut = nullptr;
  uint32_t *crc_table = nullptr;
};

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;
  int reserve (size_t n)
  {
    if (n <= cap) return MODES_OK;
    if (p) (void) hipFree (p);
    p = nullptr; cap = 0;
    size_t want = n + n / 8 + 64;
    hipError_t e = hipMalloc (reinterpret_cast<void **> (&p), want * sizeof (T));
    if (e != hipSuccess) return fail (MODES_E_NOMEM, "hipMalloc(%zu bytes): %s", want * sizeof (T), hipGetErrorString (e));
    cap = want;
    return MODES_OK;
  }
  void release () { if (p) (void) hipFree (p); p = nullptr; cap = 0; }
};

// grow-only pinned host memory: read-backs land there in one DMA (pageable targets are staged piecewise by the runtime)
template <typename T>
struct PinBuf {
  T *p = nullptr;
  size_t cap = 0;
  int reserve (size_t n)
  {
    if (n <= cap) return MODES_OK;
    if (p) (void) hipHostFree (p);
    p = nullptr; cap = 0;
    size_t want = n + n / 4 + 1024;
    hipError_t e = hipHostMalloc (reinterpret_cast<void **> (&p), want * sizeof (T), hipHostMallocDefault);
    if (e != hipSuccess) return fail (MODES_E_NOMEM, "hipHostMalloc(%zu bytes): %s", want * sizeof (T), hipGetErrorString (e));
    cap = want;
    return MODES_OK;
  }
  void release () { if (p) (void) hipHostFree (p); p = nullptr; cap = 0; }
};

uint32_t trailing_samples (uint32_t rate)     // ref: src/dump1090.c:734
{
  return uint32_t ((8 + 112 + 16) * 1E-6 * rate);
}

int bytes_per_sample (int fmt) { return fmt == INPUT_UC8 ? 2 : 4; }

// converters_table walk of convert_init (ref: src/convert.c:294-328)
int select_converter (int fmt, bool dc, bool power)
{
  static const struct { int fmt; bool dc, pw; } rows[6] = {
    { INPUT_UC8, false, false }, { INPUT_UC8, false, true }, { INPUT_UC8, true, true },
    { INPUT_SC16, false, false }, { INPUT_SC16, true, true }, { INPUT_SC16Q11, true, true } };
  for (int i = 0; i < 6; i++) {
    if (rows[i].fmt != fmt) continue;
    if (dc && !rows[i].dc) continue;
    if (power && !rows[i].pw) continue;
    return i;
  }
  return -1;
}

}  // namespace

struct D8PlanesKey {
  const void *iq = nullptr; size_t nbytes = 0; uint64_t first_sample = 0; const void *halo = nullptr; bool have_in = false; int32_t acc0 = 0;
  bool operator== (const D8PlanesKey &o) const
  { return iq == o.iq && nbytes == o.nbytes && first_sample == o.first_sample && halo == o.halo && have_in == o.have_in && acc0 == o.acc0; }
};

struct modes_engine {
  modes_engine_config cfg;
  int conv = -1;
  int mode = 2400;                          // which demodulator: 2000 / 2400
  uint32_t ov = 0, block_samples = 0;
  CrcTables crc;
  DeviceTables tabs;
  IcaoCache cache;
  // device buffers
  DevBuf<Record> records, packed;
  DevBuf<uint32_t> counts, vp_counts, rssi_len, snap_in, delta, chunk_last, chunk_entry, acc_counts, blk_stats, small_u32;
  DevBuf<modes_message> d_messages;
  DevBuf<modes_fields> d_fields;
  PinBuf<modes_fields> pin_fields;
  DevBuf<uint64_t> chunk_sum, chunk_vp, chunk_ovf, packed_pos, offsets, rssi_pos, rssi_sum, vp_bitmap;
  DevBuf<uint8_t> staging;                  // for run_host
  DevBuf<uint16_t> tail;
  uint32_t tail_half = 0;                   // which half of `tail` the last run wrote
  DevBuf<double> block_power;
  DevBuf<uint32_t> noise_level;
  size_t nblocks_power = 0;
  // HIP events around the dominant kernel(s) of the last run, recorded on the launch stream
  hipEvent_t ev[4] = { nullptr, nullptr, nullptr, nullptr };
  float kernel_ms[2] = { 0, 0 };
  // 8 MS/s
  DevBuf<uint64_t> d8_trig, d8_amb, d8_off;
  DevBuf<uint16_t> d8_vmag;            // the magnitude plane d8_trig_kernel leaves for the window kernel (2 B per sample)
  DevBuf<uint32_t> d8_p0, d8_mx, d8_win_count, d8_chunk_rec, d8_announce;
  DevBuf<uint8_t> d8_table, d8_entry;
  DevBuf<int64_t> d8_win_start, d8_small;
  DevBuf<uint32_t> dirty, blk_wrote;
  hipStream_t post_stream = nullptr;          // ordered pass and emit of one range of blocks while the next is scanned
  hipStream_t copy_stream = nullptr;          // read-back of a range's messages
  hipStream_t emit_stream = nullptr;          // a range's emit, between its ordered pass and its read-back
  std::vector<hipEvent_t> emit_events;        // one per range: its messages are formed
  std::vector<hipEvent_t> range_events;       // two per range: around its scan
  size_t last_nm = 0;                         // messages of the last run: what the next run's buffers start with
  bool have_last_nm = false;
  bool no_rank_tile = false;                  // a halo held a magnitude the byte table does not produce: scan with the magnitude tile
  // what the host reads once per accept round, in one copy: 32 totals (u64), 64 `changed` words (one per round), the
  // ICAO cache behind the range's last block
  DevBuf<uint32_t> rb;
  PinBuf<uint32_t> pin_rb;             // where a range's totals and exit cache land, and the words the rounds report in
  DevBuf<uint32_t> pass_words;         // [2][kPassWords]
  hipEvent_t round_ev[2] = { nullptr, nullptr }, fin_ev = nullptr;
  // MODES_TIMING: device-side marks of a ranged decode (name, event), printed against the first scan's start
  std::vector<std::pair<std::string, hipEvent_t>> marks;
  bool marking = false;
  PinBuf<uint32_t> pin_entry;          // the ICAO cache a range is entered with, where a kernel can fetch it
  DevBuf<D8Window> d8_windows, d8_win_packed;
  DevBuf<Record> d8_records, d8_rec_packed;
  PinBuf<D8Window> pin_wins;
  PinBuf<Record> pin_recs;
  PinBuf<uint32_t> pin_bits;                  // announce bitmap as the run starts with it
  PinBuf<uint64_t> pin_rpos;                  // 8 MS/s: positions and lengths of the messages to measure, staged for the device
  PinBuf<uint32_t> pin_rlen;
  std::vector<uint32_t> d8_lvl_msg;           // 8 MS/s: which message each measured level belongs to
  // 8 MS/s: how the last completed run went, range by range -- the ICAO cache, the messages and the counters behind each
  // range.  A run over the same windows (a hand-over round in which only the entry cache moved) stops walking at the
  // first range boundary where its cache is the last run's: from there on it would repeat that run step for step.
  struct D8Trace {
    bool valid = false;
    std::vector<uint32_t> lo;                               // the ranges of chunks
    std::vector<uint32_t> cache_after;                      // [range][kIcaoSlots]
    std::vector<size_t> msg_end;                            // messages behind each range
    std::vector<Pass8000Stats> ps_after;
    std::vector<uint64_t> cand_after;
    std::vector<uint32_t> final_cache;
    uint64_t serial = 0;                                    // which message list is this run's
  } d8_trace;
  std::vector<modes_message> d8_prev_msgs;    // the message list of the run before (swapped out of `messages` when a run starts)
  uint64_t d8_msgs_serial = 0, d8_prev_serial = 0, d8_serials = 0;   // 0: not the list of a traced run
  PinBuf<uint64_t> pin_sums, pin_tot;         // pin_tot: four totals per range of chunks, written by the device
  DevBuf<uint64_t> d8_tot;
  PinBuf<D8Window> pin_wins2;                 // a range that outgrew its share of pin_wins / pin_recs
  PinBuf<Record> pin_recs2;
  uint64_t d8_last_chunks = 0, d8_last_nwin = 0, d8_last_nrec = 0, d8_last_msgs = 0;   // the last run: sizes the next run's arrays                  // 8 MS/s: signal sums of the accepted messages, range by range
  std::vector<hipEvent_t> d8_gathered;        // one per range of chunks: its packed windows / records are in pinned memory
  hipEvent_t d8_chain_done = nullptr;
  std::vector<hipEvent_t> d8_trig_ev, d8_chain_ev;      // pipelined run_8000: around every range's planes / behind its chain
  std::vector<hipEvent_t> d8_lvl_ev;                             // [range]: the range's signal levels are in pinned memory
  std::vector<hipEvent_t> d8_edge_ev;                            // [range]: the range's windows at block edges are done   // the window list stands / the windows at block edges are done
  std::vector<hipEvent_t> d8_events;          // three per range of chunks: windows begin / end, counts (then records) on the host
  unsigned long long *d_stats = nullptr;
  uint32_t cap = 0;
  // float converters with the DC block on (kernels_dc.hip): V = [overlap | magnitudes] of the shard in HBM
  DevBuf<uint16_t> dc_mag;
  DevBuf<float> dc_sq;                 // squared magnitudes of a filter_dc run with compute_power (what mean_power adds up)
  DevBuf<DcChunk> dc_desc;
  DevBuf<float> dc_zs, dc_ze;
  DevBuf<uint32_t> dc_unm;
  DevBuf<double> dc_lin, dc_wtab;
  DevBuf<float> dc_zc, dc_ckpt;
  DevBuf<long long> dc_closed;
  DevBuf<unsigned long long> dc_ticks;
  hipEvent_t dc_power_done = nullptr, dc_power_go = nullptr;     // the block powers of a filter_dc run are formed behind the scan, beside the ordered pass
  bool dc_power_pending = false, dc_power_unqueued = false;
  const float *dc_power_sq = nullptr; uint64_t dc_power_n = 0;
  hipStream_t power_stream = nullptr;       // lowest priority: the block powers give way to the scan kernels
  hipEvent_t dc_ev[4] = { nullptr, nullptr, nullptr, nullptr };   // around the chain launch / around each repair launch
  float dc_A = 0.0f, dc_B = 1.0f, dc_z[2] = { 0.0f, 0.0f };
  uint64_t dc_lead = 0, dc_warm = 0;
  modes_dc_info dc_info;
  // RAW text of the last run's messages
  DevBuf<uint32_t> raw_len;
  DevBuf<uint64_t> raw_off, raw_cs, raw_cv, raw_co;
  DevBuf<char> raw_text;
  char *pin_text = nullptr;
  size_t pin_text_cap = 0;
  // results: the device accept path lands its messages in pinned host memory (one DMA, no staging copies);
  // the host-ordered paths build `messages`
  std::vector<modes_message> messages;
  modes_message *pin_msgs = nullptr;
  size_t pin_cap = 0, pin_n = 0;
  bool in_pinned = false;
  unsigned long long *pin_small = nullptr;   // 32 words: totals and statistics read back once per round; word 32: `changed`
  uint32_t *pin_cache = nullptr;             // the ICAO cache behind the last block, read back with them
  modes_stats stats;
  // what modes_engine_rerun_accept() needs: the last scan's geometry (its records are still in `records`)
  ScanParams last_sp;
  uint64_t last_first_sample = 0;
  bool have_scan = false;
  uint64_t scan_launches = 0, accept_passes = 0;      // since the engine was created
  // 8 MS/s: state carried in from the shard before (one run), and what this run leaves for the next
  modes_d8_state d8_in, d8_out;
  bool d8_in_set = false, d8_out_acc_valid = false;
  DevBuf<int32_t> d8_dprev, d8_dnext;
  PinBuf<int32_t> pin_d8;
  uint64_t d8_amb_steps = 0, d8_limit = 0;
  unsigned long long d8_trig_tail_in = 0;
  D8Params d8_q;
  // trigger planes / chunk tables of the last 8 MS/s call and what they were formed from (see run_8000)
  D8PlanesKey d8_planes_key;
  int32_t d8_planes_dprev[1928];
  unsigned long long d8_planes_limit = 0;
  bool d8_planes_valid = false, d8_reuse = false, d8_planes_timed = true;
  // the windows (and what they announced) of the last full run: formed from planes generation d8_win_gen, entered with
  // this open window / trigger history.  A run on the same planes with the same entry only repeats what hangs on the
  // ICAO cache: count, gather, the host's walk.
  uint64_t d8_planes_gen = 0, d8_win_gen = ~0ull;
  long long d8_win_pending = 0;
  unsigned long long d8_win_tail = 0;
  DevBuf<int32_t> d8_exit_of;
  bool d8_have_run = false, d8_passthrough = false;
};

namespace {

int setup_tables (modes_engine *e)
{
  DeviceTables &t = e->tabs;
  e->crc.build (e->cfg.fix_bits);
  float uc8f[256];
  build_uc8_float_lut (uc8f);
  HIP_TRY (upload_constants (e->crc.single_bit (), uc8f));
  uint16_t lut[256];
  build_uc8_byte_lut (lut);
  HIP_TRY (hipMalloc (reinterpret_cast<void **> (&t.lut256), sizeof lut));
  HIP_TRY (hipMemcpy (t.lut256, lut, sizeof lut, hipMemcpyHostToDevice));
  {
    // the table's distinct values in ascending order: comparing two samples' ranks is comparing their magnitudes
    std::vector<uint16_t> vals (lut, lut + 256);
    std::sort (vals.begin (), vals.end ());
    vals.erase (std::unique (vals.begin (), vals.end ()), vals.end ());
    uint8_t rank[256];
    uint16_t rankmag[256];
    for (int b = 0; b < 256; b++) rank[b] = uint8_t (std::lower_bound (vals.begin (), vals.end (), lut[b]) - vals.begin ());
    for (int r = 0; r < 256; r++) rankmag[r] = r < int (vals.size ()) ? vals[r] : uint16_t (0xFFFF);
    HIP_TRY (hipMalloc (reinterpret_cast<void **> (&t.rank256), sizeof rank));
    HIP_TRY (hipMemcpy (t.rank256, rank, sizeof rank, hipMemcpyHostToDevice));
    HIP_TRY (hipMalloc (reinterpret_cast<void **> (&t.rankmag), sizeof rankmag));
    HIP_TRY (hipMemcpy (t.rankmag, rankmag, sizeof rankmag, hipMemcpyHostToDevice));
  }
  for (int bits : { kShortBits, kLongBits }) {
    std::vector<ErrEntry> d = e->crc.device_table (bits);
    ErrEntry *p = nullptr;
    if (d.size () > 4096) return fail (MODES_E_ARG, "error table too large (%zu)", d.size ());
    if (!d.empty ()) {
      HIP_TRY (hipMalloc (reinterpret_cast<void **> (&p), d.size () * sizeof (ErrEntry)));
      HIP_TRY (hipMemcpy (p, d.data (), d.size () * sizeof (ErrEntry), hipMemcpyHostToDevice));
    }
    std::vector<uint32_t> bloom (2048, 0);
    for (const ErrEntry &x : d) { const uint32_t h = (x.syndrome ^ (x.syndrome >> 12)) & 0xFFFF; bloom[h >> 5] |= 1u << (h & 31); }
    uint32_t *pb = nullptr;
    HIP_TRY (hipMalloc (reinterpret_cast<void **> (&pb), bloom.size () * sizeof (uint32_t)));
    HIP_TRY (hipMemcpy (pb, bloom.data (), bloom.size () * sizeof (uint32_t), hipMemcpyHostToDevice));
    if (bits == kShortBits) { t.short_tab = p; t.short_n = int (d.size ()); t.bloom_short = pb; }
    else                    { t.long_tab = p;  t.long_n = int (d.size ()); t.bloom_long = pb; }
  }
  HIP_TRY (hipMalloc (reinterpret_cast<void **> (&e->d_stats), 32 * sizeof (unsigned long long)));
  HIP_TRY (hipHostMalloc (reinterpret_cast<void **> (&e->pin_small), 40 * sizeof (unsigned long long), hipHostMallocDefault));
  HIP_TRY (hipHostMalloc (reinterpret_cast<void **> (&e->pin_cache), kIcaoSlots * sizeof (uint32_t), hipHostMallocDefault));
  if (e->cfg.sample_rate == 2000000) {
    std::vector<uint16_t> l10 (65536);
    build_log10_lut (l10.data ());
    HIP_TRY (hipMalloc (reinterpret_cast<void **> (&t.log10_lut), l10.size () * sizeof (uint16_t)));
    HIP_TRY (hipMemcpy (t.log10_lut, l10.data (), l10.size () * sizeof (uint16_t), hipMemcpyHostToDevice));
  }
  HIP_TRY (hipMalloc (reinterpret_cast<void **> (&t.crc_table), 256 * sizeof (uint32_t)));
  HIP_TRY (hipMemcpy (t.crc_table, e->crc.byte_table (), 256 * sizeof (uint32_t), hipMemcpyHostToDevice));
  return MODES_OK;
}

int fill_scan_params (modes_engine *e, const void *d_iq, size_t nbytes, const uint16_t *d_halo, ScanParams *sp)
{
  const uint64_t nsamples = nbytes / bytes_per_sample (e->cfg.input_format);
  const uint64_t nstrips64 = (nsamples + kStrip - 1) / kStrip;
  if (nstrips64 > 0x7FFFFFFFull) return fail (MODES_E_ARG, "shard too large");
  const uint32_t nstrips = uint32_t (nstrips64);
  int rc;
  if ((rc = e->records.reserve (size_t (nstrips) * e->cap)) != MODES_OK) return rc;
  if ((rc = e->counts.reserve (nstrips)) != MODES_OK) return rc;
  if ((rc = e->vp_counts.reserve (nstrips)) != MODES_OK) return rc;
  if ((rc = e->vp_bitmap.reserve (size_t (nstrips) * kWords)) != MODES_OK) return rc;
  if (e->block_samples & (e->block_samples - 1)) return fail (MODES_E_ARG, "block length %u is not a power of two", e->block_samples);
  memset (sp, 0, sizeof *sp);
  sp->iq = static_cast<const uint8_t *> (d_iq);
  sp->nsamples = nsamples;
  sp->halo = d_halo;
  sp->ov = e->ov;
  sp->block_samples = e->block_samples;
  sp->block_strips = e->block_samples % kStrip == 0 ? e->block_samples / kStrip : 0;
  sp->conv = e->conv;
  sp->phase_enhance = e->cfg.phase_enhance;
  sp->short_tab = e->tabs.short_tab; sp->short_n = e->tabs.short_n;
  sp->long_tab = e->tabs.long_tab;   sp->long_n = e->tabs.long_n;
  sp->lut256 = e->tabs.lut256;
  sp->bloom_short = e->tabs.bloom_short; sp->bloom_long = e->tabs.bloom_long;
  sp->records = e->records.p;
  sp->counts = e->counts.p;
  sp->cap = e->cap;
  sp->nstrips = nstrips;
  sp->vp_counts = e->vp_counts.p;
  sp->write_records = 1;
  sp->vp_bitmap = e->vp_bitmap.p;
  sp->debug_stop = 0;
  sp->log10_lut = e->tabs.log10_lut;
  sp->crc_table = e->tabs.crc_table;
  sp->mode_ac = e->cfg.mode_ac;
  return MODES_OK;
}


// Float converter with the DC block on over the whole shard (ref: src/convert.c:112-245).  Fills e->dc_mag with the
// virtual stream V = [overlap | magnitudes] and switches `sp` to read it.  See kernels_dc.hip for the decomposition:
// independent chunks, each warmed up from a bracket that must have closed by the chunk's first sample; a chunk whose
// bracket is still open -- or whose start state differs from the end state of the chunk before -- is re-run from that
// exact end state, in order, so the result is the sequential one whatever the warm-up length.
int queue_dc_power (modes_engine *e, hipStream_t stream);
int prepare_dc (modes_engine *e, const void *d_iq, size_t nbytes, const uint16_t *d_halo, hipStream_t stream, ScanParams *sp)
{
  const int bps = bytes_per_sample (e->cfg.input_format);
  const uint64_t n = nbytes / bps;
  const uint32_t ov = e->ov, ovr = (ov + 7u) & ~7u;
  if (reinterpret_cast<uintptr_t> (d_iq) & 15) return fail (MODES_E_ARG, "filter_dc: d_iq must be 16-byte aligned");
  const uint64_t lead = e->dc_lead;
  e->dc_lead = 0;                                                     // one run only
  if (lead % 256) return fail (MODES_E_ARG, "filter_dc: the lead-in must be a multiple of 256 samples");
  int rc;
  if ((rc = e->dc_mag.reserve (size_t (n) + ov + 1024)) != MODES_OK) return rc;
  uint16_t *V = e->dc_mag.p + 8;
  const bool own_halo = lead >= ovr;                                  // the lead-in covers the overlap: the chain forms it
  const uint64_t back = own_halo ? ovr : 0;                           // samples in front of the shard whose magnitude is kept
  if (!own_halo) {
    if (d_halo) HIP_TRY (hipMemcpyAsync (V, d_halo, ov * sizeof (uint16_t), hipMemcpyDeviceToDevice, stream));
    else        HIP_TRY (hipMemsetAsync (V, 0, ov * sizeof (uint16_t), stream));
  }
  HIP_TRY (hipMemsetAsync (V + ov + n, 0, 512 * sizeof (uint16_t), stream));
  memset (&e->dc_info, 0, sizeof e->dc_info);
  sp->iq = reinterpret_cast<const uint8_t *> (V);
  sp->conv = kConvMagU16;
  sp->navail = n + ov;
  sp->halo = nullptr;
  if (n == 0) return MODES_OK;

  const uint64_t n_eff = n + back, lead_eff = (lead - back) / kDcBatch * kDcBatch;   // chains start on the batch grid of the shard
  uint64_t K = n_eff / 65536;
  if (K < 1) K = 1;
  uint64_t kmax = 512;                                                // two workgroups (four chains each, 65 KB of LDS) per CU
  if (const char *ev = getenv ("MODES_DC_CHUNKS")) kmax = uint64_t (atoll (ev)) ? uint64_t (atoll (ev)) : kmax;
  if (K > kmax) K = kmax;
  const uint64_t L = ((n_eff + K - 1) / K + kDcBatch - 1) / kDcBatch * kDcBatch;
  K = (n_eff + L - 1) / L;
  // Warm-up: the bracket a chunk starts from is (where the filter is in real arithmetic) +- |z| / m, m = DC_A / 2^-24:
  // RN (z * B) = z - k ulp with k an integer, so the float filter's leak is off by up to half an ulp per step, which
  // the filter itself turns into at most 0.5 ulp / DC_A <= |z| / m.  A bracket that wide closes after about
  // ln (2 / m * 2^24) / DC_A samples (13.5 time constants at 2.4 MS/s).  Chunks still open at their first sample are
  // repaired below (only up to where they closed), so the length trades warm-up steps against repairs: 14 (measured on the bench recording and on the same without frames: 9 to 14 within 4 %, below 9 whole chunks never close).
  double warm_tau = 14.0;
  if (const char *ev = getenv ("MODES_DC_WARM_TAU")) if (atof (ev) > 0) warm_tau = atof (ev);
  uint64_t warm = e->dc_warm ? e->dc_warm : uint64_t (std::ceil (warm_tau / double (e->dc_A)));
  warm = (warm + kDcBatch - 1) / kDcBatch * kDcBatch;
  const float bound = e->conv == kConvSc16Q11Generic ? 16.0f : 1.0f;  // |f| never exceeds it, so neither does |z1|

  // where the filter is in real arithmetic at every segment boundary of [origin, n_eff), origin = -lead_eff
  const uint64_t total = lead_eff + n_eff, nseg = (total + kDcSeg - 1) / kDcSeg;
  std::vector<double> est (2 * (nseg + 1)), half (2 * (nseg + 1));     // value and half-width of the bracket, per channel
  {
    if ((rc = e->dc_lin.reserve (2 * nseg)) != MODES_OK) return rc;
    if (!e->dc_wtab.p) {                                               // B^(1023 - j) for the kernel's Horner steps, and B^1024
      if ((rc = e->dc_wtab.reserve (1025)) != MODES_OK) return rc;
      std::vector<double> w (1025);
      for (int j = 0; j < 1024; j++) w[j] = std::pow (double (e->dc_B), double (1023 - j));
      w[1024] = std::pow (double (e->dc_B), 1024.0);
      HIP_TRY (hipMemcpy (e->dc_wtab.p, w.data (), w.size () * sizeof (double), hipMemcpyHostToDevice));
    }
    HIP_TRY (launch_dc_linear (e->conv, static_cast<const uint8_t *> (d_iq) - (back + lead_eff) * bps, total, double (e->dc_A), e->dc_wtab.p,
                               e->dc_lin.p, stream));
    std::vector<double> S (2 * nseg);
    HIP_TRY (hipMemcpyAsync (S.data (), e->dc_lin.p, 2 * nseg * sizeof (double), hipMemcpyDeviceToHost, stream));
    HIP_TRY (hipStreamSynchronize (stream));
    const double Bseg = std::pow (double (e->dc_B), double (kDcSeg));
    const double rel = 1.25 / (double (e->dc_A) * 16777216.0);
    for (int ch = 0; ch < 2; ch++) {
      double E = lead == 0 ? double (e->dc_z[ch]) : 0.0;               // the stream's own state, or unknown
      double U = lead == 0 ? 0.0 : double (bound);                     // what is not known about it
      double M = std::fabs (E);                                        // the largest |z1| of the recent past, decaying like z1 itself
      for (uint64_t j = 0; j <= nseg; j++) {
        est[2 * j + ch] = E;
        half[2 * j + ch] = rel * M + U + 1e-12 * double (bound);
        if (j == nseg) break;
        E = E * Bseg + S[2 * j + ch];
        U *= Bseg;
        M = std::max (M * Bseg, std::fabs (E));
      }
    }
  }
  auto lo_of = [&] (uint64_t j, int ch) { const float v = float (est[2 * j + ch] - half[2 * j + ch]); return std::max (-bound, std::nextafterf (v, -INFINITY)); };
  auto hi_of = [&] (uint64_t j, int ch) { const float v = float (est[2 * j + ch] + half[2 * j + ch]); return std::min (bound, std::nextafterf (v, INFINITY)); };

  auto mid_of = [&] (uint64_t j, int ch) { return std::min (bound, std::max (-bound, float (est[2 * j + ch]))); };
  auto same = [] (float a, float b) { return memcmp (&a, &b, sizeof a) == 0; };

  // Every chunk but the first starts ONE chain per channel where the filter is in real arithmetic, `warm` samples early;
  // it is right when it enters the chunk with the very state the chunk before ended on (checked below).  The first
  // chunk starts from the stream's own state, or -- behind a lead-in, where nothing earlier can vouch for it -- from a
  // bracket that must have closed by the shard's first sample.
  std::vector<DcChunk> desc (K);
  for (uint64_t c = 0; c < K; c++) {
    DcChunk &d = desc[c];
    const uint64_t c0 = c * L;
    d.chunk = uint32_t (c); d.slot = uint32_t (c); d.stop = 0;
    uint64_t j = 0;                                                    // segment boundary the chain starts on
    if (c0 + lead_eff > warm) j = (c0 + lead_eff - warm) / kDcSeg;
    d.start = (long long) (j * kDcSeg) - (long long) lead_eff;
    if (j == 0 && lead == 0) { d.zIlo = d.zIhi = e->dc_z[0]; d.zQlo = d.zQhi = e->dc_z[1]; }       // the stream's own state
    else if (c == 0) { d.zIlo = lo_of (j, 0); d.zIhi = hi_of (j, 0); d.zQlo = lo_of (j, 1); d.zQhi = hi_of (j, 1); }
    else { d.zIlo = d.zIhi = mid_of (j, 0); d.zQlo = d.zQhi = mid_of (j, 1); }
  }
  // slots 0 .. K-1: the chunks; K .. 2K-1: repairs of a chunk's first samples; 2K: a chunk re-run as a whole
  const uint32_t nck = uint32_t (L / kDcCkpt + 1);
  if ((rc = e->dc_desc.reserve (2 * K + 1)) != MODES_OK) return rc;
  if ((rc = e->dc_zs.reserve (4 * (2 * K + 1))) != MODES_OK) return rc;
  if ((rc = e->dc_ze.reserve (2 * (2 * K + 1))) != MODES_OK) return rc;
  if ((rc = e->dc_zc.reserve (2 * (2 * K + 1))) != MODES_OK) return rc;
  if ((rc = e->dc_closed.reserve (2 * K + 1)) != MODES_OK) return rc;
  if ((rc = e->dc_unm.reserve (2 * K + 1 + 16)) != MODES_OK) return rc;
  if ((rc = e->dc_ckpt.reserve (2 * size_t (K) * nck)) != MODES_OK) return rc;
  DcParams q;
  memset (&q, 0, sizeof q);
  q.iq = static_cast<const uint8_t *> (d_iq) - back * bps;
  q.n = n_eff; q.chunk_len = L; q.A = e->dc_A; q.B = e->dc_B;
  q.desc = e->dc_desc.p; q.mag = V + ov - back;
  if ((e->cfg.compute_power || e->cfg.mode_ac) && e->mode == 2400) {      // mean_power is only read by demod_2400's noise statistics and demod_2400_AC
    if ((rc = e->dc_sq.reserve (size_t (n_eff) + 1024)) != MODES_OK) return rc;
    q.mag_sq = e->dc_sq.p;
  }
  q.z_start = e->dc_zs.p; q.z_end = e->dc_ze.p; q.unmerged = e->dc_unm.p; q.closed_at = e->dc_closed.p; q.z_close = e->dc_zc.p;
  q.ckpt = e->dc_ckpt.p; q.nck = nck;
  q.rounds = reinterpret_cast<unsigned long long *> (e->dc_unm.p + ((2 * K + 2) & ~1ull));
  HIP_TRY (hipMemsetAsync (q.rounds, 0, 6 * sizeof (unsigned long long), stream));
  const bool profile = getenv ("MODES_DC_PROFILE") != nullptr;
  if (profile) {
    if ((rc = e->dc_ticks.reserve (16 * K)) != MODES_OK) return rc;
    HIP_TRY (hipMemsetAsync (e->dc_ticks.p, 0, 16 * K * sizeof (unsigned long long), stream));
    q.ticks = e->dc_ticks.p;
  }
  if (!e->dc_ev[0]) for (hipEvent_t &ev : e->dc_ev) HIP_TRY (hipEventCreate (&ev));
  HIP_TRY (hipMemcpyAsync (e->dc_desc.p, desc.data (), K * sizeof (DcChunk), hipMemcpyHostToDevice, stream));
  HIP_TRY (hipEventRecord (e->dc_ev[0], stream));
  HIP_TRY (launch_dc_chain (e->conv, q, uint32_t (K), stream));
  HIP_TRY (hipEventRecord (e->dc_ev[1], stream));
  // read-backs land in vectors that outlive every wait below: an early return synchronises first (SyncOnExit)
  struct SyncOnExit { hipStream_t s; ~SyncOnExit () { (void) hipStreamSynchronize (s); } } sync_guard { stream };
  std::vector<float> zs (4 * K), ze (2 * K);
  std::vector<uint32_t> unm (K);
  unsigned long long rounds[6] = { 0, 0, 0, 0, 0, 0 };
  HIP_TRY (hipMemcpyAsync (zs.data (), e->dc_zs.p, 4 * K * sizeof (float), hipMemcpyDeviceToHost, stream));
  HIP_TRY (hipMemcpyAsync (ze.data (), e->dc_ze.p, 2 * K * sizeof (float), hipMemcpyDeviceToHost, stream));
  HIP_TRY (hipMemcpyAsync (unm.data (), e->dc_unm.p, K * sizeof (uint32_t), hipMemcpyDeviceToHost, stream));
  HIP_TRY (hipMemcpyAsync (rounds, q.rounds, sizeof rounds, hipMemcpyDeviceToHost, stream));
  HIP_TRY (hipStreamSynchronize (stream));
  (void) hipEventElapsedTime (&e->dc_info.chain_ms, e->dc_ev[0], e->dc_ev[1]);
  e->dc_info.chunks = uint32_t (K); e->dc_info.chunk_samples = L; e->dc_info.warm_samples = warm;
  e->dc_info.guess_rounds = rounds[0]; e->dc_info.plain_rounds = rounds[1];
  if (profile) {
    std::vector<unsigned long long> tk (16 * K);
    HIP_TRY (hipMemcpy (tk.data (), e->dc_ticks.p, 16 * K * sizeof (unsigned long long), hipMemcpyDeviceToHost));
    std::vector<uint64_t> order (K);
    for (uint64_t c = 0; c < K; c++) order[c] = c;
    std::sort (order.begin (), order.end (), [&] (uint64_t a, uint64_t b) { return tk[16 * a] > tk[16 * b]; });
    for (uint64_t i = 0; i < K && i < 6; i++) {
      const uint64_t c = order[i];
      fprintf (stderr, "  slowest: chunk %4llu: %9llu ticks, %7llu rounds, %6llu plain, chain %llu, barriers %llu; rounds per wave %llu %llu %llu %llu (plain steps, or with DC_COUNT_PASSES passes: I %llu Q %llu); wave 0 at barriers inside the chunk %llu; helper wave 4: work %llu, barriers %llu (inside the chunk %llu); start %lld (chunk begins %llu)\n", (unsigned long long) c, tk[16 * c], tk[16 * c + 1], tk[16 * c + 2], tk[16 * c + 3], tk[16 * c + 4],
               tk[16 * c + 5], tk[16 * c + 7], tk[16 * c + 9], tk[16 * c + 11], tk[16 * c + 6], tk[16 * c + 10], tk[16 * c + 12], tk[16 * c + 13], tk[16 * c + 14], tk[16 * c + 15], desc[c].start, (unsigned long long) (c * L));
    }
    uint64_t missed = 0;
    for (uint64_t c = 1; c < K; c++) missed += !(same (zs[4 * c], ze[2 * (c - 1)]) && same (zs[4 * c + 2], ze[2 * (c - 1) + 1]));
    fprintf (stderr, "dc profile (wave 0 sums over %llu chunks): chain %llu barrier %llu total %llu ticks, longest %llu; warm %llu, %llu chunks entered with another state than the chunk before ended on\n",
             (unsigned long long) K, rounds[2], rounds[3], rounds[4], rounds[5], (unsigned long long) warm, (unsigned long long) missed);
  }
  for (uint64_t c = 0; c < K; c++) e->dc_info.chain_steps += 2 * uint64_t ((long long) std::min (n_eff, (c + 1) * L) - desc[c].start);
  q.rounds = nullptr;
  q.ticks = nullptr;
  if (unm[0]) return fail (MODES_E_ARG, "filter_dc: a lead-in of %llu samples does not settle the DC filter (about %llu needed)",
                           (unsigned long long) lead, (unsigned long long) warm);
  // A chunk entered with another state than the chunk before ended on is stepped again from that end state, all such
  // chunks in one launch, each only until it has met the trajectory of the first pass (the checkpoints): from there on
  // the first pass's values stand.  One that never meets it is stepped to its end and hands on a new end state.
  std::vector<DcChunk> fix;
  std::vector<long long> fix_idx (K, -1);
  for (uint64_t c = 1; c < K; c++) {
    const float *st = &zs[4 * c], *pe = &ze[2 * (c - 1)];
    if (same (st[0], pe[0]) && same (st[2], pe[1])) continue;
    DcChunk d;
    d.start = (long long) (c * L); d.chunk = uint32_t (c); d.slot = uint32_t (K + fix.size ()); d.stop = -1;
    d.zIlo = d.zIhi = pe[0]; d.zQlo = d.zQhi = pe[1];
    fix_idx[c] = (long long) fix.size ();
    fix.push_back (d);
  }
  std::vector<float> fe (2 * fix.size () + 2);
  std::vector<long long> fclosed (fix.size () + 1);
  if (!fix.empty ()) {
    HIP_TRY (hipMemcpyAsync (e->dc_desc.p + K, fix.data (), fix.size () * sizeof (DcChunk), hipMemcpyHostToDevice, stream));
    q.desc = e->dc_desc.p + K;
    HIP_TRY (hipEventRecord (e->dc_ev[2], stream));
    HIP_TRY (launch_dc_chain (e->conv, q, uint32_t (fix.size ()), stream));
    HIP_TRY (hipEventRecord (e->dc_ev[3], stream));
    HIP_TRY (hipMemcpyAsync (fe.data (), e->dc_ze.p + 2 * K, 2 * fix.size () * sizeof (float), hipMemcpyDeviceToHost, stream));
    HIP_TRY (hipMemcpyAsync (fclosed.data (), e->dc_closed.p + K, fix.size () * sizeof (long long), hipMemcpyDeviceToHost, stream));
    HIP_TRY (hipStreamSynchronize (stream));
    { float ms = 0; (void) hipEventElapsedTime (&ms, e->dc_ev[2], e->dc_ev[3]); e->dc_info.chain_ms += ms;      // chain_ms: every launch of the chain kernel
      if (profile) {
        long long far = 0;
        for (size_t i = 0; i < fix.size (); i++) far = std::max (far, (fclosed[i] < 0 ? (long long) L : fclosed[i] - fix[i].start));
        fprintf (stderr, "dc profile: first pass %.3f ms; %zu chunks stepped again in %.3f ms, the longest for %lld samples\n", e->dc_info.chain_ms - ms, fix.size (), ms, far);
      } }
    e->dc_info.prefix_reruns = uint32_t (fix.size ());
  }
  // In order: what a chunk really ends on.  Whatever was checked or repaired against an end state that has changed since
  // is stepped again as a whole from the right one.
  std::vector<float> true_end (ze);
  for (uint64_t c = 1; c < K; c++) {
    const float *te = &true_end[2 * (c - 1)];
    if (same (te[0], ze[2 * (c - 1)]) && same (te[1], ze[2 * (c - 1) + 1])) {      // built on what the chunk before does end on
      const long long i = fix_idx[c];
      if (i >= 0 && fclosed[i] < 0) { true_end[2 * c] = fe[2 * i]; true_end[2 * c + 1] = fe[2 * i + 1]; }   // never met the first pass
      continue;
    }
    DcChunk d;
    d.start = (long long) (c * L); d.chunk = uint32_t (c); d.slot = uint32_t (2 * K); d.stop = 0;
    d.zIlo = d.zIhi = te[0]; d.zQlo = d.zQhi = te[1];
    HIP_TRY (hipMemcpyAsync (e->dc_desc.p + 2 * K, &d, sizeof d, hipMemcpyHostToDevice, stream));
    q.desc = e->dc_desc.p + 2 * K;
    HIP_TRY (hipEventRecord (e->dc_ev[2], stream));
    HIP_TRY (launch_dc_chain (e->conv, q, 1, stream));
    HIP_TRY (hipEventRecord (e->dc_ev[3], stream));
    HIP_TRY (hipMemcpyAsync (&true_end[2 * c], e->dc_ze.p + 2 * (2 * K), 2 * sizeof (float), hipMemcpyDeviceToHost, stream));
    HIP_TRY (hipStreamSynchronize (stream));
    { float ms = 0; (void) hipEventElapsedTime (&ms, e->dc_ev[2], e->dc_ev[3]); e->dc_info.chain_ms += ms; }
    e->dc_info.rerun_chunks++;
  }
  ze = true_end;
  e->dc_z[0] = ze[2 * (K - 1)]; e->dc_z[1] = ze[2 * (K - 1) + 1];
  if (q.mag_sq) {
    // mag_buf.mean_power of every block: the converter's float running sum over the block's squared magnitudes, in
    // sample order (ref: src/convert.c:146,153-154) -- formed from the values the chain kernel left, repairs included
    const uint32_t nblocks = uint32_t ((n + e->block_samples - 1) / e->block_samples);
    if ((rc = e->block_power.reserve (nblocks)) != MODES_OK) return rc;
    // (a bandwidth-bound pass over 4 bytes per sample that nothing in the decode reads: queued by queue_dc_power () on a
    // stream of its own BEHIND the scan kernels -- beside them it took their bandwidth -- where it runs under the ordered
    // pass, the emit and the read-back, which are bound by latency; the caller's stream waits for it before the run is over)
    if (!e->copy_stream) HIP_TRY (hipStreamCreateWithFlags (&e->copy_stream, hipStreamNonBlocking));
    if (!e->dc_power_done) HIP_TRY (hipEventCreateWithFlags (&e->dc_power_done, hipEventDisableTiming));
    if (!e->dc_power_go) HIP_TRY (hipEventCreateWithFlags (&e->dc_power_go, hipEventDisableTiming));
    e->dc_power_sq = q.mag_sq + back; e->dc_power_n = n;
    e->dc_power_unqueued = true;
    e->nblocks_power = nblocks;
    if (getenv ("MODES_DC_POWER_EARLY")) { if ((rc = queue_dc_power (e, stream)) != MODES_OK) return rc; }     // (experiments: beside the scan kernels)
  }
  return MODES_OK;
}

// The block powers of a filter_dc run (prepare_dc): behind everything queued on `stream` so far.
int queue_dc_power (modes_engine *e, hipStream_t stream)
{
  if (!e->dc_power_unqueued) return MODES_OK;
  e->dc_power_unqueued = false;
  if (!e->power_stream) {
    int least = 0, greatest = 0;
    HIP_TRY (hipDeviceGetStreamPriorityRange (&least, &greatest));
    HIP_TRY (hipStreamCreateWithPriority (&e->power_stream, hipStreamNonBlocking, getenv ("MODES_DC_POWER_PRIO") ? atoi (getenv ("MODES_DC_POWER_PRIO")) : least));
  }
  HIP_TRY (hipEventRecord (e->dc_power_go, stream));
  HIP_TRY (hipStreamWaitEvent (e->power_stream, e->dc_power_go, 0));
  HIP_TRY (launch_dc_block_power (e->dc_power_sq, e->dc_power_n, e->block_samples, e->nblocks_power, e->block_power.p, e->power_stream));
  HIP_TRY (hipEventRecord (e->dc_power_done, e->power_stream));
  e->dc_power_pending = true;
  return MODES_OK;
}

constexpr int kRetryMagnitudeTile = -1000;     // internal: scan again without the rank tile (never leaves run_device_inner)
constexpr uint32_t kRbInexactWord = 6;         // low word of totals[3] in modes_engine::rb

// What the scan kernel is launched with: for the as-written byte table converter behind the ordered pass on the
// device, the instance whose LDS tile holds ranks instead of magnitudes (kConvUc8LutR8: half the LDS, 6 instead of 4
// waves per SIMD with two strips per wave).  The halo is the one input that does not come through the table; a value
// the table cannot produce is reported through `inexact` and the run is repeated with the magnitude tile.
ScanParams scan_variant (modes_engine *e, const ScanParams &sp)
{
  ScanParams q = sp;
  const bool off = getenv ("MODES_NO_RANK_TILE") != nullptr;          // tests: compare against the magnitude tile
  if (e->mode == 2400 && !off && !e->no_rank_tile && !e->cfg.host_replay && !e->cfg.phase_enhance && e->rb.p &&
      (sp.conv == kConvUc8Lut || sp.conv == kConvUc8LutPower) && sp.block_strips) {
    // the input bytes by LDS-DMA where whole 16-byte chunks line up with the scan positions (they do for hipMalloc'ed
    // recordings); MODES_RANK_TILE_GATHER=1 keeps the register-staged rank tile (tests, comparisons)
    const bool raw = !getenv ("MODES_RANK_TILE_GATHER") && sp.ov == kOv2400 && !(reinterpret_cast<uintptr_t> (sp.iq) & 15) && !(sp.block_samples & 15);
    q.conv = raw ? kConvUc8LutRaw : kConvUc8LutR8;
    q.rank256 = e->tabs.rank256; q.rankmag = e->tabs.rankmag;
    q.inexact = e->rb.p + kRbInexactWord;
  }
  return q;
}

// Blocks [blk0, blk0 + nblk) of a shard and their strips: the unit the ordered accept pass works on.
struct RangeView { uint32_t blk0, nblk, strip0, nstrips; };

// Room for `need` messages in d_messages and pin_msgs; the first `keep` stay (the caller has waited for every copy
// into them).
int reserve_messages (modes_engine *e, size_t need, size_t keep)
{
  if (need > e->d_messages.cap) {
    DevBuf<modes_message> bigger;
    int rc = bigger.reserve (need);
    if (rc != MODES_OK) return rc;
    if (keep) HIP_TRY (hipMemcpy (bigger.p, e->d_messages.p, keep * sizeof (modes_message), hipMemcpyDeviceToDevice));
    e->d_messages.release ();
    e->d_messages = bigger;
  }
  if (need > e->pin_cap) {
    modes_message *np = nullptr;
    const size_t want = need + need / 4 + 1024;
    HIP_TRY (hipHostMalloc (reinterpret_cast<void **> (&np), want * sizeof (modes_message), hipHostMallocDefault));
    if (keep) memcpy (np, e->pin_msgs, keep * sizeof (modes_message));
    if (e->pin_msgs) (void) hipHostFree (e->pin_msgs);
    e->pin_msgs = np; e->pin_cap = want;
  }
  return MODES_OK;
}

// a timing mark on `stream` (only while MODES_TIMING is set and a ranged decode is collecting them)
static void mark (modes_engine *e, hipStream_t stream, const char *fmt, unsigned a = 0, unsigned b = 0)
{
  if (!e->marking) return;
  char name[64];
  snprintf (name, sizeof name, fmt, a, b);
  hipEvent_t ev = nullptr;
  if (hipEventCreate (&ev) != hipSuccess) return;
  (void) hipEventRecord (ev, stream);
  e->marks.emplace_back (name, ev);
}

constexpr uint32_t kRbTotals = 0, kRbChanged = 64, kRbRounds = 64, kRbMarked = kRbChanged + kRbRounds, kRbCache = kRbMarked + kRbRounds, kRbWords = kRbCache + kIcaoSlots;
// per range parity, on the device: the rounds' `changed` and `marked` words and the ICAO cache behind the range
constexpr uint32_t kPassWords = 2 * kRbRounds + kIcaoSlots;

int reserve_accept_buffers (modes_engine *e, const ScanParams &sp)
{
  const uint32_t nblocks = uint32_t ((sp.nsamples + e->block_samples - 1) / e->block_samples);
  const uint32_t nchunks = (nblocks + 31) / 32, nsc = (sp.nstrips + 1023) / 1024;
  int rc;
  if ((rc = e->snap_in.reserve (size_t (nblocks) * kIcaoSlots)) != MODES_OK) return rc;
  if ((rc = e->delta.reserve (size_t (nblocks) * kIcaoSlots)) != MODES_OK) return rc;
  if ((rc = e->chunk_last.reserve (size_t (nchunks + 1) * kIcaoSlots)) != MODES_OK) return rc;
  if ((rc = e->chunk_entry.reserve (size_t (nchunks + 1) * kIcaoSlots)) != MODES_OK) return rc;
  if ((rc = e->acc_counts.reserve (sp.nstrips)) != MODES_OK) return rc;
  if ((rc = e->blk_stats.reserve (size_t (nblocks) * kBlkStats)) != MODES_OK) return rc;
  if ((rc = e->small_u32.reserve (2 * kIcaoSlots + 16)) != MODES_OK) return rc;
  if ((rc = e->rb.reserve (kRbWords)) != MODES_OK) return rc;
  if ((rc = e->pin_rb.reserve (kRbWords + 2 * kRbRounds)) != MODES_OK) return rc;       // the totals and the exit cache; the rounds' `marked` words of two ranges
  if ((rc = e->pass_words.reserve (2 * kPassWords)) != MODES_OK) return rc;
  if (!e->fin_ev) HIP_TRY (hipEventCreateWithFlags (&e->fin_ev, hipEventDisableTiming));
  if ((rc = e->pin_entry.reserve (kIcaoSlots)) != MODES_OK) return rc;
  if ((rc = e->dirty.reserve (nblocks)) != MODES_OK) return rc;
  if ((rc = e->blk_wrote.reserve (nblocks)) != MODES_OK) return rc;
  if ((rc = e->chunk_sum.reserve (nsc + 1)) != MODES_OK) return rc;
  if ((rc = e->chunk_vp.reserve (nsc + 1)) != MODES_OK) return rc;
  if ((rc = e->chunk_ovf.reserve (nsc + 1)) != MODES_OK) return rc;
  return e->offsets.reserve (2 * size_t (sp.nstrips));
}

// The ordered accept pass on the device for one range of blocks whose scan is complete on `stream`: fixed-point rounds
// over the per-block ICAO snapshots, then emit.  Three steps, so that a caller with several ranges can queue the next
// range's rounds before it turns to this range's totals and messages:
//   pass_begin   the range's entry cache (the host's, or what the range before left on the device), round 0
//   pass_settle  the host follows the rounds until no block's entry cache changed
//   pass_finish  totals, statistics, message offsets (on the emit stream: none of it stands between two rounds or
//                between this range's rounds and the next's), then emit and read-back
// A round on the pass's stream is the accept kernel and the three snapshot kernels, nothing else: the pass is a chain of
// small dependent launches, and every launch in it is 5 us the ranges behind it wait for.
struct PassState {
  RangeView rg;
  hipStream_t stream;
  AcceptParams ap;
  uint32_t par;                 // which of the two sets of round words / exit caches the range uses
  uint32_t queued, rounds;
  uint32_t *d_changed, *d_cache_out;
  volatile uint32_t *marked_host;
};

int pass_queue_round (modes_engine *e, const ScanParams &sp, PassState &ps)
{
  (void) sp;
  const uint32_t r = ps.queued, slot = r % kRbRounds;               // each round has its own `changed` word
  const RangeView &rg = ps.rg;
  if (r && slot == 0) {
    HIP_TRY (launch_fill_u32 (ps.d_changed, 2 * kRbRounds, 0u, ps.stream));
    for (uint32_t i = 0; i < kRbRounds; i++) ps.marked_host[i] = 0;     // (the host has seen every round queued so far: pass_settle)
  }
  ps.ap.changed = ps.d_changed + slot;
  ps.ap.first_round = r == 0;
  ps.ap.dirty = r ? e->dirty.p : nullptr;                          // from the second round on
  HIP_TRY (launch_accept2400 (ps.ap, ps.stream));
  HIP_TRY (launch_snapshots (e->delta.p + size_t (rg.blk0) * kIcaoSlots, e->blk_wrote.p + rg.blk0, rg.nblk, e->small_u32.p, e->chunk_last.p, e->chunk_entry.p,
                             e->snap_in.p + size_t (rg.blk0) * kIcaoSlots, ps.d_cache_out, e->dirty.p + rg.blk0,
                             ps.ap.changed, ps.d_changed + kRbRounds + slot, const_cast<uint32_t *> (ps.marked_host) + slot, ps.ap.first_round, ps.stream));
  HIP_TRY (hipEventRecord (e->round_ev[r & 1], ps.stream));
  mark (e, ps.stream, "  blocks %u..: round %u ends", rg.blk0, r);
  ps.queued = r + 1;
  return MODES_OK;
}

// prev: the range before this one, whose exit cache is on the device (null: the range starts from e->cache)
int pass_begin (modes_engine *e, const ScanParams &sp, hipStream_t stream, const RangeView &rg, uint32_t par, const PassState *prev, PassState &ps)
{
  const uint32_t B = e->block_samples;
  ps.rg = rg; ps.stream = stream; ps.par = par & 1; ps.queued = ps.rounds = 0;
  ps.d_changed = e->pass_words.p + ps.par * kPassWords;
  ps.d_cache_out = ps.d_changed + 2 * kRbRounds;
  ps.marked_host = e->pin_rb.p + kRbWords + ps.par * kRbRounds;
  uint32_t *d_cache_in = e->small_u32.p;
  mark (e, stream, "  blocks %u..: pass begins", rg.blk0);
  // (kernels, not blits: see launch_fill_u32)
  if (prev) HIP_TRY (launch_copy_u32 (d_cache_in, prev->d_cache_out, kIcaoSlots, stream));
  else {
    memcpy (e->pin_entry.p, e->cache.raw (), kIcaoSlots * sizeof (uint32_t));
    HIP_TRY (launch_copy_u32 (d_cache_in, e->pin_entry.p, kIcaoSlots, stream));
  }
  HIP_TRY (launch_fill_u32 (e->acc_counts.p + rg.strip0, rg.nstrips, 0u, stream));
  HIP_TRY (launch_fill_u32 (ps.d_changed, 2 * kRbRounds, 0u, stream));
  for (uint32_t i = 0; i < kRbRounds; i++) ps.marked_host[i] = 0;
  // (no per-block state is prepared: in the first round every block starts from d_cache_in, writes its delta row only
  // if it wrote a slot, and says so in blk_wrote)
  AcceptParams &ap = ps.ap;
  memset (&ap, 0, sizeof ap);
  ap.records = e->records.p; ap.counts = e->counts.p; ap.cap = e->cap; ap.nstrips = sp.nstrips;
  ap.strips_per_block = B / kStrip; ap.nblocks = uint32_t ((sp.nsamples + B - 1) / B); ap.nsamples = sp.nsamples; ap.block_samples = B;
  ap.blk_begin = rg.blk0; ap.blk_end = rg.blk0 + rg.nblk;
  ap.snap_in = e->snap_in.p; ap.delta = e->delta.p; ap.acc_counts = e->acc_counts.p; ap.blk_stats = e->blk_stats.p;
  ap.vp_bitmap = e->mode == 2400 ? e->vp_bitmap.p : nullptr;
  ap.mode = e->mode;
  ap.cache_in = d_cache_in; ap.blk_wrote = e->blk_wrote.p; ap.first_round = 1;
  if (!e->round_ev[0]) for (hipEvent_t &ev : e->round_ev) HIP_TRY (hipEventCreateWithFlags (&ev, hipEventDisableTiming));
  return pass_queue_round (e, sp, ps);
}

// The host never stands between two rounds: round r + 1 is queued before the host waits for what round r found, so the
// device goes from round to round without the host's wake-up and launch latency in between (about 0.1 ms a round); the
// one round queued too many -- behind the round that found nothing changed -- looks at no block (none is dirty) and
// leaves every result as it is.
int pass_settle (modes_engine *e, const ScanParams &sp, PassState &ps)
{
  int rc;
  for (;;) {
    // (a round that wraps the `changed` words clears them first: not while the round before may still set its own)
    const bool ahead = (ps.rounds + 1) % kRbRounds != 0;
    if (ahead && ps.queued < ps.rounds + 2 && (rc = pass_queue_round (e, sp, ps)) != MODES_OK) return rc;
    HIP_TRY (hipEventSynchronize (e->round_ev[ps.rounds & 1]));
    const bool changed = ps.marked_host[ps.rounds % kRbRounds] != 0;      // a block has to run again
    ps.rounds++;
    if (!changed) break;
    // a write set can only change because an earlier block's did: block k is final after round k + 1 at the latest
    if (ps.rounds > ps.rg.nblk + 2) return fail (MODES_E_HIP, "accept pass did not reach its fixed point in %u rounds", ps.rounds);
    if (ps.queued < ps.rounds + 1 && (rc = pass_queue_round (e, sp, ps)) != MODES_OK) return rc;
  }
  e->stats.replay_rounds += ps.rounds;
  return MODES_OK;
}

// fin: the stream the totals, the emit kernel (and, without a copy stream, the read-back) run on; with streams of their
// own the emit and the read-back leave the pass's stream as soon as the round that settled the range has ended (not the
// round queued behind it): pass, emit and read-back of consecutive ranges form a pipeline of three stages beside the scan.
// settled_ev: recorded behind the settled round (taken when the range settled, before the next range's rounds reuse the
// round events).
int pass_finish (modes_engine *e, const ScanParams &sp, uint64_t first_sample, PassState &ps, bool last, hipStream_t fin,
                 hipStream_t copy_stream, hipEvent_t emitted)
{
  const RangeView &rg = ps.rg;
  int rc;
  uint32_t *acc_counts = e->acc_counts.p + rg.strip0;
  uint64_t *d_totals = reinterpret_cast<uint64_t *> (e->rb.p + kRbTotals);
  const uint32_t rounds = ps.rounds;
  // totals of the range's scan (candidates, valid preambles, overflowed strips), of the accepted messages, the block
  // statistics, and where each strip's messages go
  HIP_TRY (launch_count_scan (e->counts.p + rg.strip0, e->vp_counts.p + rg.strip0, rg.nstrips, e->chunk_sum.p, e->chunk_vp.p, e->chunk_ovf.p, d_totals, fin));
  HIP_TRY (launch_count_scan (acc_counts, e->vp_counts.p + rg.strip0, rg.nstrips, e->chunk_sum.p, e->chunk_vp.p, e->chunk_ovf.p, d_totals + 4, fin));
  HIP_TRY (launch_blk_stats_sum (e->blk_stats.p + size_t (rg.blk0) * kBlkStats, rg.nblk, reinterpret_cast<unsigned long long *> (d_totals + 8), fin));
  HIP_TRY (launch_strip_offsets (acc_counts, rg.nstrips, e->chunk_sum.p, e->offsets.p + rg.strip0, fin));
  HIP_TRY (launch_copy_u32 (e->pin_rb.p, e->rb.p, kRbChanged, fin));                                  // into pinned host memory
  HIP_TRY (launch_copy_u32 (e->pin_rb.p + kRbCache, ps.d_cache_out, kIcaoSlots, fin));
  HIP_TRY (hipEventRecord (e->fin_ev, fin));
  HIP_TRY (hipEventSynchronize (e->fin_ev));
  const uint32_t *rbh = e->pin_rb.p;
  memcpy (e->cache.raw (), rbh + kRbCache, kIcaoSlots * sizeof (uint32_t));
  const unsigned long long *st = reinterpret_cast<const unsigned long long *> (rbh + kRbTotals), *tot = st + 4, *sum = st + 8;
  if (st[2]) return fail (MODES_E_OVERFLOW, "%llu strips overflowed their candidate slots (cap %u)", st[2], e->cap);
  if (st[3]) return fail (kRetryMagnitudeTile, "the halo holds magnitudes the byte table does not produce");
  e->stats.candidates += st[0];
  const size_t nm = size_t (tot[0]);
  e->stats.demodulated += sum[kStDemodulated]; e->stats.out_of_phase += sum[kStOutOfPhase];
  e->stats.CRC_good += sum[kStGood]; e->stats.CRC_bad += sum[kStBad];
  e->stats.CRC_single_bit_fix += sum[kStFix1]; e->stats.CRC_two_bits_fix += sum[kStFix2];
  e->stats.CRC_fixed += sum[kStFix1] + sum[kStFix2];
  e->stats.demod_rejected_unknown += sum[kStUnknown];
  e->stats.demod_modeac += sum[kStModeAC];
  if (e->mode == 2000) e->stats.valid_preamble += sum[kStVp];     // every valid preamble is a record there
  else e->stats.valid_preamble += st[1] - sum[kStSkippedVp];
  if (sum[kStAccepted] != nm) return fail (MODES_E_HIP, "accepted-count mismatch (%llu vs %zu)", (unsigned long long) sum[kStAccepted], nm);
  (void) rounds;

  const size_t base = e->pin_n;
  if (nm) {
    if (base + nm > e->d_messages.cap || base + nm > e->pin_cap) {
      // the list outgrew what the run set out with: wait for the read-backs in flight, then move to larger buffers
      if (base) {
        HIP_TRY (hipStreamSynchronize (fin));
        if (copy_stream) HIP_TRY (hipStreamSynchronize (copy_stream));
      }
      const uint32_t nb_all = ps.ap.nblocks, nb_done = rg.blk0 + rg.nblk;
      const size_t guess = size_t (double (base + nm) * double (nb_all) / double (nb_done) * 1.25) + 1024;
      if ((rc = reserve_messages (e, guess, base)) != MODES_OK) return rc;
    }
    EmitParams ep;
    memset (&ep, 0, sizeof ep);
    ep.records = e->records.p; ep.counts = e->counts.p; ep.acc_counts = e->acc_counts.p; ep.offsets = e->offsets.p;
    ep.cap = e->cap; ep.nstrips = sp.nstrips;
    ep.iq = sp.iq; ep.nsamples = sp.nsamples; ep.navail = sp.navail; ep.halo = sp.halo; ep.ov = sp.ov;
    ep.block_samples = sp.block_samples; ep.conv = sp.conv; ep.lut256 = sp.lut256;
    ep.first_sample = first_sample; ep.sample_rate = e->cfg.sample_rate;
    ep.messages = e->d_messages.p + base;             // offsets[] count from the range's first strip
    ep.mode = e->mode;
    ep.strip_begin = rg.strip0; ep.strip_end = rg.strip0 + rg.nstrips;
    mark (e, fin, "  blocks %u..: emit begins", rg.blk0);
    HIP_TRY (launch_emit2400 (ep, fin));
    mark (e, fin, "  blocks %u..: emit ends", rg.blk0);
    hipStream_t cs = fin;
    if (copy_stream) {
      HIP_TRY (hipEventRecord (emitted, fin));
      HIP_TRY (hipStreamWaitEvent (copy_stream, emitted, 0));
      cs = copy_stream;
    }
    HIP_TRY (hipMemcpyAsync (e->pin_msgs + base, e->d_messages.p + base, nm * sizeof (modes_message), hipMemcpyDeviceToHost, cs));
    mark (e, cs, "  blocks %u..: read-back ends", rg.blk0);
    e->pin_n = base + nm;
  }
  e->in_pinned = true;
  if (last) {
    HIP_TRY (hipStreamSynchronize (ps.stream));
    HIP_TRY (hipStreamSynchronize (fin));
    if (copy_stream) HIP_TRY (hipStreamSynchronize (copy_stream));
  }
  return MODES_OK;
}

// one range, begin to finish, everything on `stream`: enters with e->cache = the ICAO cache in front of the range and
// leaves it as it stands behind the range; statistics add up in e->stats, the range's messages follow the e->pin_n
// already there
int accept_range (modes_engine *e, const ScanParams &sp, uint64_t first_sample, hipStream_t stream, const RangeView &rg, bool last)
{
  PassState ps;
  int rc = pass_begin (e, sp, stream, rg, 0, nullptr, ps);
  if (rc == MODES_OK) rc = pass_settle (e, sp, ps);
  if (rc == MODES_OK) rc = pass_finish (e, sp, first_sample, ps, last, stream, nullptr, nullptr);
  return rc;
}

// The whole shard as one range, behind a scan already queued on `stream`.
int run_accept_on_device (modes_engine *e, const ScanParams &sp, uint64_t first_sample, hipStream_t stream, bool scan_timed = true)
{
  int rc = reserve_accept_buffers (e, sp);
  if (rc != MODES_OK) return rc;
  const uint32_t nblocks = uint32_t ((sp.nsamples + e->block_samples - 1) / e->block_samples);
  const RangeView all = { 0, nblocks, 0, sp.nstrips };
  e->pin_n = 0;
  rc = accept_range (e, sp, first_sample, stream, all, true);
  e->accept_passes++;
  e->kernel_ms[0] = e->kernel_ms[1] = 0;
  if (scan_timed) (void) hipEventElapsedTime (&e->kernel_ms[0], e->ev[0], e->ev[1]);     // ev[] lie around this run's scan
  return rc;
}

// A long shard in ranges of blocks: the scan of range r + 1 runs (on the caller's stream) while the ordered pass, the
// emit and the read-back of range r run on a second stream.  The ICAO cache is handed from range to range like it is
// from block to block; each range's messages are final when its pass ends, so their copy to the host overlaps the
// scan as well.  Same records, same order, same list as the single pass.
int scan_and_accept_in_ranges (modes_engine *e, ScanParams &sp, uint64_t first_sample, hipStream_t stream, uint32_t nranges)
{
  const uint32_t B = e->block_samples, spb = B / kStrip;
  const uint32_t nblocks = uint32_t ((sp.nsamples + B - 1) / B);
  int rc = reserve_accept_buffers (e, sp);
  if (rc != MODES_OK) return rc;
  if (!e->post_stream) HIP_TRY (hipStreamCreateWithFlags (&e->post_stream, hipStreamNonBlocking));
  if (!e->copy_stream) HIP_TRY (hipStreamCreateWithFlags (&e->copy_stream, hipStreamNonBlocking));
  if (!e->emit_stream) HIP_TRY (hipStreamCreateWithFlags (&e->emit_stream, hipStreamNonBlocking));
  while (e->emit_events.size () < size_t (nranges)) {
    hipEvent_t ev;
    HIP_TRY (hipEventCreateWithFlags (&ev, hipEventDisableTiming));
    e->emit_events.push_back (ev);
  }
  while (e->range_events.size () < 2 * size_t (nranges)) {
    hipEvent_t ev;
    HIP_TRY (hipEventCreate (&ev));
    e->range_events.push_back (ev);
  }
  if (e->last_nm) if ((rc = reserve_messages (e, e->last_nm + e->last_nm / 8 + 1024, 0)) != MODES_OK) return rc;
  std::vector<RangeView> rgs;
  // Ranges shrink towards the end (weights R, R-1, .. 1): what is left to do when the last scan ends -- the last
  // range's pass, emit and read-back -- is what the overlap cannot hide.
  const uint64_t wsum = uint64_t (nranges) * (nranges + 1) / 2;
  // (equal and growing ranges were measured too -- within 0.05 ms of these on the
  // message-heaviest input, up to 0.5 ms worse elsewhere)
  auto cut = [&] (uint32_t r) -> uint32_t {       // first block of range r
    return uint32_t (uint64_t (nblocks) * (uint64_t (r) * (2 * nranges - r + 1) / 2) / wsum);
  };
  for (uint32_t r = 0; r < nranges; r++) {
    const uint32_t b0 = cut (r), b1 = r + 1 == nranges ? nblocks : cut (r + 1);
    const uint32_t s0 = b0 * spb, s1 = b1 * spb < sp.nstrips ? b1 * spb : sp.nstrips;
    if (b1 > b0) rgs.push_back (RangeView { b0, b1 - b0, s0, s1 - s0 });
  }
  nranges = uint32_t (rgs.size ());
  auto queue_scan = [&] (uint32_t r) -> hipError_t {
    ScanParams q = scan_variant (e, sp);
    q.strip_begin = rgs[r].strip0; q.strip_end = rgs[r].strip0 + rgs[r].nstrips;
    hipError_t he = hipEventRecord (e->range_events[2 * r], stream);
    if (he == hipSuccess) he = e->mode == 2000 ? launch_scan2000 (q, stream) : launch_scan2400 (q, stream);
    if (he == hipSuccess) he = hipEventRecord (e->range_events[2 * r + 1], stream);
    return he;
  };
  // on any way out nothing of this run is left in flight on either stream
  struct Drain {
    modes_engine *e; hipStream_t s;
    ~Drain () {
      (void) hipStreamSynchronize (e->post_stream); (void) hipStreamSynchronize (e->emit_stream); (void) hipStreamSynchronize (e->copy_stream);
      if (e->power_stream) (void) hipStreamSynchronize (e->power_stream);
      (void) hipStreamSynchronize (s);
    }
  } drain = { e, stream };
  e->pin_n = 0;
  HIP_TRY (hipMemsetAsync (e->rb.p + kRbInexactWord, 0, 2 * sizeof (uint32_t), stream));
  // every range's scan is queued at once, back to back on the caller's stream: none waits for the host, which is busy
  // with the rounds of the ranges before
  for (uint32_t r = 0; r < nranges; r++) HIP_TRY (queue_scan (r));
  HIP_TRY (launch_tail (sp, sp.nsamples, e->tail.p + e->tail_half * ((e->ov + 7u) & ~7u), stream));   // what a following shard needs as d_halo
  { const int prc = queue_dc_power (e, stream); if (prc != MODES_OK) return prc; }
  e->scan_launches++;
  const bool timing = getenv ("MODES_TIMING") != nullptr;              // the host's way through the ranges on stderr
  e->marking = timing;
  const auto t_start = std::chrono::steady_clock::now ();
  auto since = [&] { return std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now () - t_start).count (); };
  // the next range's rounds are queued as soon as this range has settled, before the host turns to this range's totals
  // and messages: the pass's stream goes from range to range without waiting for the host's dealings with the emit stream
  PassState pass[2];
  HIP_TRY (hipStreamWaitEvent (e->post_stream, e->range_events[1], 0));
  if ((rc = pass_begin (e, sp, e->post_stream, rgs[0], 0, nullptr, pass[0])) != MODES_OK) return rc;
  for (uint32_t r = 0; r < nranges; r++) {
    PassState &ps = pass[r & 1];
    const double t_in = since ();
    if ((rc = pass_settle (e, sp, ps)) != MODES_OK) return rc;
    const double t_settled = since ();
    HIP_TRY (hipStreamWaitEvent (e->emit_stream, e->round_ev[(ps.rounds - 1) & 1], 0));       // (before the next range's rounds record the event again)
    if (r + 1 < nranges) {
      HIP_TRY (hipStreamWaitEvent (e->post_stream, e->range_events[2 * (r + 1) + 1], 0));
      if ((rc = pass_begin (e, sp, e->post_stream, rgs[r + 1], r + 1, &ps, pass[(r + 1) & 1])) != MODES_OK) return rc;
    }
    if ((rc = pass_finish (e, sp, first_sample, ps, r + 1 == nranges, e->emit_stream, e->copy_stream, e->emit_events[r])) != MODES_OK) return rc;
    if (timing) fprintf (stderr, "  range %u (%u blocks): host turned to it at %.3f ms, settled (%u rounds) at %.3f ms, messages queued%s at %.3f ms, %zu messages so far\n", r, rgs[r].nblk, t_in,
                         ps.rounds, t_settled, r + 1 == nranges ? " and everything read back" : "", since (), e->pin_n);
  }
  HIP_TRY (hipStreamSynchronize (stream));
  if (timing) {
    float t = 0, t0 = 0;
    for (uint32_t r = 0; r < nranges; r++) {
      (void) hipEventElapsedTime (&t0, e->range_events[0], e->range_events[2 * r]);
      (void) hipEventElapsedTime (&t, e->range_events[0], e->range_events[2 * r + 1]);
      fprintf (stderr, "  scan of range %u: %.3f .. %.3f ms after the first scan began\n", r, t0, t);
    }
    for (auto &m : e->marks) {
      (void) hipEventSynchronize (m.second);
      (void) hipEventElapsedTime (&t, e->range_events[0], m.second);
      fprintf (stderr, "%s at %.3f ms\n", m.first.c_str (), t);
      (void) hipEventDestroy (m.second);
    }
    e->marks.clear ();
    fprintf (stderr, "  ranged decode: %.3f ms on the host\n", since ());
  }
  e->marking = false;
  e->accept_passes++;
  float ms = 0;
  for (uint32_t r = 0; r < nranges; r++) {
    float t = 0;
    (void) hipEventElapsedTime (&t, e->range_events[2 * r], e->range_events[2 * r + 1]);
    ms += t;
  }
  e->kernel_ms[0] = ms; e->kernel_ms[1] = 0;
  return MODES_OK;
}

}  // namespace

extern "C" {

const char *modes_last_error (void) { return g_err.c_str (); }

int modes_engine_create (const modes_engine_config *cfg, modes_engine **out)
{
  if (!cfg || !out) return fail (MODES_E_ARG, "null argument");
  *out = nullptr;
  if (cfg->sample_rate != 2400000 && cfg->sample_rate != 2000000 && cfg->sample_rate != 8000000)
    return fail (MODES_E_ARG, "sample rate %u: the reference demodulates 2.0, 2.4 and 8.0 MS/s", cfg->sample_rate);
  const int conv = select_converter (cfg->input_format, cfg->filter_dc != 0, cfg->compute_power != 0);
  if (conv < 0) return fail (MODES_E_ARG, "no suitable converter for format=%d power=%d DC=%d", cfg->input_format,
                             cfg->compute_power, cfg->filter_dc);
  int ndev = 0;
  if (hipGetDeviceCount (&ndev) != hipSuccess || ndev <= 0) return fail (MODES_E_NODEVICE, "no HIP device");
  if (cfg->device < 0 || cfg->device >= ndev) return fail (MODES_E_ARG, "device %d of %d", cfg->device, ndev);
  HIP_TRY (hipSetDevice (cfg->device));

  modes_engine *e = new modes_engine ();
  e->cfg = *cfg;
  e->conv = conv;
  e->ov = trailing_samples (cfg->sample_rate);
  e->block_samples = kAsyncBufSize / bytes_per_sample (cfg->input_format);
  e->mode = cfg->sample_rate == 2000000 ? 2000 : cfg->sample_rate == 8000000 ? 8000 : 2400;
  e->cap = e->mode == 2000 ? uint32_t (cfg->phase_enhance ? kCap2000PE : kCap2000) : cfg->phase_enhance ? 128 : 32;
  if (const char *ev = getenv ("MODES_STRIP_SLOTS")) if (atoi (ev) > 0) e->cap = uint32_t (atoi (ev));   // test hook: start small, grow on demand
  memset (&e->stats, 0, sizeof e->stats);
  memset (&e->dc_info, 0, sizeof e->dc_info);
  if (cfg->filter_dc) {                       // ref: src/convert.c:349-354
    e->dc_B = float (exp (-2.0 * M_PI * 1.0 / double (cfg->sample_rate)));
    e->dc_A = float (1.0 - e->dc_B);
  }
  int rc = setup_tables (e);
  if (rc != MODES_OK) { modes_engine_destroy (e); return rc; }
  rc = e->tail.reserve (2 * size_t (e->ov) + 16);       // two halves: a run writes the one the previous run did not (its halo may be that one)
  if (rc != MODES_OK) { modes_engine_destroy (e); return rc; }
  for (hipEvent_t &ev : e->ev)
    if (hipEventCreate (&ev) != hipSuccess) { modes_engine_destroy (e); return fail (MODES_E_HIP, "hipEventCreate failed"); }
  *out = e;
  return MODES_OK;
}

void modes_engine_destroy (modes_engine *e)
{
  if (!e) return;
  e->records.release (); e->packed.release (); e->counts.release (); e->rssi_len.release ();
  e->chunk_sum.release (); e->chunk_vp.release (); e->chunk_ovf.release (); e->vp_counts.release (); e->packed_pos.release (); e->offsets.release (); e->rssi_pos.release ();
  e->snap_in.release (); e->delta.release (); e->chunk_last.release (); e->chunk_entry.release (); e->acc_counts.release ();
  e->blk_stats.release (); e->small_u32.release (); e->d_messages.release ();
  e->rssi_sum.release (); e->staging.release (); e->tail.release (); e->vp_bitmap.release ();
  e->block_power.release (); e->noise_level.release ();
  e->raw_len.release (); e->raw_off.release (); e->raw_cs.release (); e->raw_cv.release (); e->raw_co.release (); e->raw_text.release ();
  if (e->pin_text) (void) hipHostFree (e->pin_text);
  e->dc_lin.release (); e->dc_wtab.release (); e->dc_ticks.release (); e->dc_zc.release (); e->dc_ckpt.release (); e->dc_closed.release ();
  for (hipEvent_t ev : e->dc_ev) if (ev) (void) hipEventDestroy (ev);
  if (e->dc_power_done) (void) hipEventDestroy (e->dc_power_done);
  if (e->dc_power_go) (void) hipEventDestroy (e->dc_power_go);
  if (e->power_stream) (void) hipStreamDestroy (e->power_stream);
  for (hipEvent_t ev : e->round_ev) if (ev) (void) hipEventDestroy (ev);
  if (e->fin_ev) (void) hipEventDestroy (e->fin_ev);
  e->dc_mag.release (); e->dc_sq.release (); e->dc_desc.release (); e->dc_zs.release (); e->dc_ze.release (); e->dc_unm.release ();
  e->d8_trig.release (); e->d8_amb.release (); e->d8_off.release (); e->d8_p0.release (); e->d8_mx.release ();
  e->d8_win_count.release (); e->d8_chunk_rec.release (); e->d8_announce.release (); e->d8_table.release (); e->d8_entry.release ();
  e->d8_win_start.release (); e->d8_small.release (); e->d8_windows.release (); e->d8_win_packed.release ();
  e->d8_records.release (); e->d8_rec_packed.release (); e->dirty.release (); e->blk_wrote.release (); e->rb.release (); e->pass_words.release (); e->pin_rb.release (); e->pin_entry.release (); e->pin_sums.release (); e->pin_tot.release (); e->d8_tot.release (); e->pin_wins2.release (); e->pin_recs2.release (); e->pin_rpos.release (); e->pin_rlen.release ();
  e->d8_dprev.release (); e->d8_dnext.release (); e->pin_d8.release (); e->d8_exit_of.release ();
  e->d_fields.release (); e->pin_fields.release ();
  if (e->post_stream) (void) hipStreamDestroy (e->post_stream);
  if (e->copy_stream) (void) hipStreamDestroy (e->copy_stream);
  if (e->emit_stream) (void) hipStreamDestroy (e->emit_stream);
  for (hipEvent_t ev : e->emit_events) if (ev) (void) hipEventDestroy (ev);
  for (hipEvent_t ev : e->range_events) if (ev) (void) hipEventDestroy (ev);
  e->pin_wins.release (); e->pin_recs.release (); e->pin_bits.release ();
  for (hipEvent_t ev : e->d8_events) if (ev) (void) hipEventDestroy (ev);
  for (hipEvent_t ev : e->d8_gathered) if (ev) (void) hipEventDestroy (ev);
  if (e->d8_chain_done) (void) hipEventDestroy (e->d8_chain_done);
  for (hipEvent_t ev : e->d8_edge_ev) (void) hipEventDestroy (ev);
  for (hipEvent_t ev : e->d8_trig_ev) (void) hipEventDestroy (ev);
  for (hipEvent_t ev : e->d8_chain_ev) (void) hipEventDestroy (ev);
  for (hipEvent_t ev : e->d8_lvl_ev) (void) hipEventDestroy (ev);
  if (e->tabs.short_tab) (void) hipFree (e->tabs.short_tab);
  if (e->tabs.long_tab) (void) hipFree (e->tabs.long_tab);
  if (e->tabs.lut256) (void) hipFree (e->tabs.lut256);
  if (e->tabs.rank256) (void) hipFree (e->tabs.rank256);
  if (e->tabs.rankmag) (void) hipFree (e->tabs.rankmag);
  if (e->tabs.bloom_short) (void) hipFree (e->tabs.bloom_short);
  if (e->tabs.bloom_long) (void) hipFree (e->tabs.bloom_long);
  if (e->tabs.log10_lut) (void) hipFree (e->tabs.log10_lut);
  if (e->tabs.crc_table) (void) hipFree (e->tabs.crc_table);
  if (e->d_stats) (void) hipFree (e->d_stats);
  if (e->pin_small) (void) hipHostFree (e->pin_small);
  if (e->pin_cache) (void) hipHostFree (e->pin_cache);
  if (e->pin_msgs) (void) hipHostFree (e->pin_msgs);
  for (hipEvent_t ev : e->ev) if (ev) (void) hipEventDestroy (ev);
  delete e;
}

int modes_engine_set_icao (modes_engine *e, const uint32_t *slots)
{
  if (!e) return fail (MODES_E_ARG, "null engine");
  if (slots) memcpy (e->cache.raw (), slots, sizeof (uint32_t) * kIcaoSlots); else e->cache.clear ();
  return MODES_OK;
}

int modes_engine_get_icao (modes_engine *e, uint32_t *slots)
{
  if (!e || !slots) return fail (MODES_E_ARG, "null argument");
  memcpy (slots, e->cache.raw (), sizeof (uint32_t) * kIcaoSlots);
  return MODES_OK;
}

static int scan_and_decode (modes_engine *e, ScanParams &sp, uint64_t first_sample, hipStream_t stream);

static int run_device_inner (modes_engine *e, const void *d_iq, size_t nbytes, uint64_t first_sample,
                             const uint16_t *d_halo, void *stream_v)
{
  if (!e || (!d_iq && nbytes)) return fail (MODES_E_ARG, "null argument");
  hipStream_t stream = static_cast<hipStream_t> (stream_v);
  // a run of several blocks starts on a block boundary; a single block (a live receiver's buffer) may start anywhere
  if (first_sample % e->block_samples && nbytes / bytes_per_sample (e->cfg.input_format) > e->block_samples)
    return fail (MODES_E_ARG, "first_sample must be a multiple of the block length");
  HIP_TRY (hipSetDevice (e->cfg.device));
  e->messages.clear ();
  e->in_pinned = false; e->pin_n = 0;
  e->have_scan = false;
  memset (&e->stats, 0, sizeof e->stats);

  ScanParams sp;
  int rc = fill_scan_params (e, d_iq, nbytes, d_halo, &sp);
  if (rc != MODES_OK) return rc;
  e->nblocks_power = 0;
  if (e->cfg.filter_dc && (rc = prepare_dc (e, d_iq, nbytes, d_halo, stream, &sp)) != MODES_OK) return rc;
  if (sp.nstrips == 0) return MODES_OK;
  // A strip that holds more candidates than its slots is reported, never cut short: scan again with four times the
  // slots (up to one per position and retry kind) until everything fits.
  const uint32_t cap_max = e->mode == 2000 ? 3u * kStrip : uint32_t (kStrip);
  std::vector<uint32_t> cache_in (e->cache.raw (), e->cache.raw () + kIcaoSlots);      // an overflowed attempt may have touched it
  e->tail_half ^= 1;                        // once per run: a rescan writes the same half again
  if (e->mode == 2400 && e->cfg.compute_power && !e->cfg.filter_dc && !e->cfg.mode_ac) {
    // Modes.measure_noise: demod_2400 reads mag_buf.mean_power of every block (ref: src/demod-2400.c:524-530)
    const uint32_t nblocks = uint32_t ((sp.nsamples + e->block_samples - 1) / e->block_samples);
    if ((rc = e->block_power.reserve (nblocks)) != MODES_OK) return rc;
    HIP_TRY (launch_block_power (sp, nblocks, e->block_power.p, stream));
    e->nblocks_power = nblocks;
  }
  for (;;) {
    rc = scan_and_decode (e, sp, first_sample, stream);
    if (rc == kRetryMagnitudeTile) e->no_rank_tile = true;        // for this engine, from now on
    else {
      if (rc != MODES_E_OVERFLOW || e->cap >= cap_max) return rc;
      e->cap = e->cap * 4 < cap_max ? e->cap * 4 : cap_max;
    }
    e->stats.scan_retries++;
    const uint64_t retries = e->stats.scan_retries;
    if ((rc = e->records.reserve (size_t (sp.nstrips) * e->cap)) != MODES_OK) return rc;
    sp.records = e->records.p; sp.cap = e->cap;
    e->messages.clear ();
    e->in_pinned = false; e->pin_n = 0;
    memset (&e->stats, 0, sizeof e->stats);
    e->stats.scan_retries = retries;
    memcpy (e->cache.raw (), cache_in.data (), kIcaoSlots * sizeof (uint32_t));
  }
}

static int scan_and_decode (modes_engine *e, ScanParams &sp, uint64_t first_sample, hipStream_t stream)
{
  int rc;
  const bool device_accept = !e->cfg.host_replay && !e->cfg.phase_enhance;
  if (device_accept) {
    // What the ranges hide is the read-back of the message list (the ordered pass and the emit are latency bound and
    // fill the wave slots the scan would use, so running them beside the scan gains nothing): three or four ranges
    // for a long shard, none for a short one or when the last run's list was small.
    const uint32_t nblocks = uint32_t ((sp.nsamples + e->block_samples - 1) / e->block_samples);
    // (measured, tools/ranges_sweep.sh on 4 GiB: three ranges up to some 0.5 M messages, four above -- the more there is
    // to emit and read back per range, the more a further range hides)
    uint32_t nranges = nblocks >= 1024 ? (e->mode == 2000 || (e->have_last_nm && e->last_nm >= 600000) ? 4u : 3u) : nblocks >= 512 ? 2u : 1u;
    if (e->have_last_nm && e->last_nm < 65536) nranges = 1;
    if (const char *ev = getenv ("MODES_RANGES")) { const int v = atoi (ev); if (v >= 1 && v <= 64) nranges = uint32_t (v) < nblocks ? uint32_t (v) : nblocks; }
    if (nranges > 1 && sp.block_strips) {
      rc = scan_and_accept_in_ranges (e, sp, first_sample, stream, nranges);
      if (rc == MODES_OK) { e->last_sp = sp; e->last_first_sample = first_sample; e->have_scan = true; e->last_nm = e->pin_n; e->have_last_nm = true; }
      return rc;
    }
  }
  if (device_accept) {
    if ((rc = reserve_accept_buffers (e, sp)) != MODES_OK) return rc;
    HIP_TRY (hipMemsetAsync (e->rb.p + kRbInexactWord, 0, 2 * sizeof (uint32_t), stream));
  }
  HIP_TRY (hipEventRecord (e->ev[0], stream));
  HIP_TRY (e->mode == 2000 ? launch_scan2000 (sp, stream) : launch_scan2400 (scan_variant (e, sp), stream));
  e->scan_launches++;
  HIP_TRY (hipEventRecord (e->ev[1], stream));
  HIP_TRY (launch_tail (sp, sp.nsamples, e->tail.p + e->tail_half * ((e->ov + 7u) & ~7u), stream));   // what a following shard needs as d_halo
  if ((rc = queue_dc_power (e, stream)) != MODES_OK) return rc;

  if (device_accept) {
    rc = run_accept_on_device (e, sp, first_sample, stream);
    if (rc == MODES_OK) { e->last_sp = sp; e->last_first_sample = first_sample; e->have_scan = true; e->last_nm = e->pin_n; e->have_last_nm = true; }
    return rc;
  }

  // totals, then compact the per-strip slots into one ordered list
  const uint32_t nchunks = (sp.nstrips + 1023) / 1024;
  if ((rc = e->chunk_sum.reserve (nchunks + 1)) != MODES_OK) return rc;
  if ((rc = e->chunk_vp.reserve (nchunks + 1)) != MODES_OK) return rc;
  if ((rc = e->chunk_ovf.reserve (nchunks + 1)) != MODES_OK) return rc;
  if ((rc = e->offsets.reserve (sp.nstrips)) != MODES_OK) return rc;
  uint64_t *d_totals = reinterpret_cast<uint64_t *> (e->d_stats);
  HIP_TRY (launch_count_scan (e->counts.p, e->vp_counts.p, sp.nstrips, e->chunk_sum.p, e->chunk_vp.p, e->chunk_ovf.p,
                              d_totals, stream));

  unsigned long long st[3];
  HIP_TRY (hipMemcpyAsync (st, d_totals, sizeof st, hipMemcpyDeviceToHost, stream));
  HIP_TRY (hipStreamSynchronize (stream));
  (void) hipEventElapsedTime (&e->kernel_ms[0], e->ev[0], e->ev[1]);
  e->kernel_ms[1] = 0;
  if (st[2]) return fail (MODES_E_OVERFLOW, "%llu strips overflowed their candidate slots (cap %u)", st[2], e->cap);
  const size_t nrec = size_t (st[0]);
  e->stats.valid_preamble = st[1];
  e->stats.candidates = nrec;

  if ((rc = e->packed.reserve (nrec + 1)) != MODES_OK) return rc;
  if ((rc = e->packed_pos.reserve (nrec + 1)) != MODES_OK) return rc;
  HIP_TRY (launch_gather (e->counts.p, sp.nstrips, e->chunk_sum.p, e->records.p, e->cap, e->packed.p, e->packed_pos.p,
                          e->offsets.p, stream));

  std::vector<Record> recs (nrec);
  std::vector<uint64_t> pos (nrec);
  std::vector<uint64_t> vp (size_t (sp.nstrips) * kWords);
  HIP_TRY (hipMemcpyAsync (vp.data (), e->vp_bitmap.p, vp.size () * sizeof (uint64_t), hipMemcpyDeviceToHost, stream));
  if (nrec) {
    HIP_TRY (hipMemcpyAsync (recs.data (), e->packed.p, nrec * sizeof (Record), hipMemcpyDeviceToHost, stream));
    HIP_TRY (hipMemcpyAsync (pos.data (), e->packed_pos.p, nrec * sizeof (uint64_t), hipMemcpyDeviceToHost, stream));
  }
  HIP_TRY (hipStreamSynchronize (stream));

  if (e->mode == 2000) {
    // ordered walk of demod_2000 on the host (phase_enhance retries / compare mode); all slicing was done by the kernel
    Pass2000Stats ps;
    uint64_t good = 0, bad = 0, fix1 = 0, fix2 = 0, unknown = 0;
    Pass2000Hooks hooks;
    hooks.decode = [&] (const Record &r, uint64_t, uint8_t, modes_message *out) {
      const int rc = (r.phase & kInfoStaticReject) ? -2 : accept_record (r, e->cache, out);
      if (rc < 0) { bad++; if (rc == -1) unknown++; }
      else if (out->error_bits == 1) fix1++;
      else if (out->error_bits == 2) fix2++;
      else good++;
      return rc;
    };
    hooks.emit = [&] (const modes_message &m, const Record &) { e->messages.push_back (m); };
    accept_pass_2000 (recs.data (), pos.data (), nrec, e->block_samples, sp.nsamples, first_sample, e->cfg.sample_rate,
                      e->cfg.phase_enhance != 0, hooks, ps);
    e->stats.valid_preamble = ps.valid_preamble; e->stats.demodulated = ps.demodulated;
    e->stats.out_of_phase = ps.out_of_phase; e->stats.demod_modeac = ps.demod_modeac;
    e->stats.CRC_good = good; e->stats.CRC_bad = bad; e->stats.CRC_single_bit_fix = fix1; e->stats.CRC_two_bits_fix = fix2;
    e->stats.CRC_fixed = fix1 + fix2; e->stats.demod_rejected_unknown = unknown;
    e->stats.replay_rounds = 0;
    return MODES_OK;
  }

  // ordered accept pass
  AcceptStats as;
  std::vector<uint64_t> mpos;
  accept_pass_2400 (recs.data (), pos.data (), nrec, e->block_samples, first_sample, e->cfg.sample_rate, e->cache, as,
                    e->messages, mpos, vp.data (), sp.nsamples);
  e->stats.valid_preamble -= as.skipped_preambles;
  e->stats.demodulated = as.demodulated; e->stats.out_of_phase = as.out_of_phase;
  e->stats.CRC_good = as.CRC_good; e->stats.CRC_bad = as.CRC_bad; e->stats.CRC_fixed = as.CRC_fixed;
  e->stats.CRC_single_bit_fix = as.CRC_single_bit_fix; e->stats.CRC_two_bits_fix = as.CRC_two_bits_fix;
  e->stats.demod_rejected_unknown = as.rejected_unknown;
  e->stats.replay_rounds = 0;

  // signal level of the accepted messages (ref: src/demod-2400.c:486-497)
  const size_t nm = e->messages.size ();
  if (nm) {
    std::vector<uint32_t> len (nm);
    std::vector<uint64_t> sums (nm);
    for (size_t i = 0; i < nm; i++) len[i] = uint32_t (e->messages[i].msg_bits) * 12 / 5;
    if ((rc = e->rssi_pos.reserve (nm)) != MODES_OK) return rc;
    if ((rc = e->rssi_len.reserve (nm)) != MODES_OK) return rc;
    if ((rc = e->rssi_sum.reserve (nm)) != MODES_OK) return rc;
    HIP_TRY (hipMemcpyAsync (e->rssi_pos.p, mpos.data (), nm * sizeof (uint64_t), hipMemcpyHostToDevice, stream));
    HIP_TRY (hipMemcpyAsync (e->rssi_len.p, len.data (), nm * sizeof (uint32_t), hipMemcpyHostToDevice, stream));
    HIP_TRY (launch_rssi2400 (sp.conv, sp.iq, sp.conv == kConvMagU16 ? sp.navail : sp.nsamples, sp.halo, sp.ov, sp.block_samples,
                              sp.lut256, e->rssi_pos.p, e->rssi_len.p, e->rssi_sum.p, uint32_t (nm), stream));
    HIP_TRY (hipMemcpyAsync (sums.data (), e->rssi_sum.p, nm * sizeof (uint64_t), hipMemcpyDeviceToHost, stream));
    HIP_TRY (hipStreamSynchronize (stream));
    for (size_t i = 0; i < nm; i++) {
      const double signal_power = double (sums[i]) / 65535.0 / 65535.0;
      e->messages[i].sig_level = signal_power / double (len[i]);
    }
  }
  return MODES_OK;
}

// demod_2400_AC behind demod_2400 on every block (BASELINE.json configs[4]; ref: src/demod-2400AC.c:25-228).
// The kernel answers "does a reply start here" for every position; the 69-sample skip behind a hit is applied here on
// the hit list, and each block's replies are queued behind the block's Mode S messages.
static int run_mode_ac_2400 (modes_engine *e, const void *d_iq, size_t nbytes, uint64_t first_sample,
                             const uint16_t *d_halo, hipStream_t stream)
{
  ScanParams sp;
  int rc = fill_scan_params (e, d_iq, nbytes, d_halo, &sp);
  if (rc != MODES_OK) return rc;
  if (sp.nstrips == 0) return MODES_OK;
  const uint32_t B = e->block_samples;
  const uint32_t nblocks = uint32_t ((sp.nsamples + B - 1) / B);
  if ((rc = e->noise_level.reserve (nblocks)) != MODES_OK) return rc;
  if (e->cfg.filter_dc) {
    // the Mode S pass before this one left the magnitudes of the whole shard (DC block applied) and, with
    // compute_power, every block's mean_power: read those (prepare_dc)
    if (!e->dc_mag.p) return fail (MODES_E_ARG, "demod_2400_AC: no filter_dc magnitudes to read");
    sp.iq = reinterpret_cast<const uint8_t *> (e->dc_mag.p + 8);
    sp.conv = kConvMagU16; sp.navail = sp.nsamples + e->ov; sp.halo = nullptr;
    // (the float converters report their power sum whether or not it was asked for: ref: src/convert.c:146,153-154)
    if (e->nblocks_power != nblocks) return fail (MODES_E_ARG, "demod_2400_AC: the filter_dc run left no block powers");
  } else {
    if ((rc = e->block_power.reserve (nblocks)) != MODES_OK) return rc;
    HIP_TRY (launch_block_power (sp, nblocks, e->block_power.p, stream));
    e->nblocks_power = nblocks;
  }
  HIP_TRY (launch_noise_level (e->block_power.p, nblocks, e->noise_level.p, stream));
  sp.noise_level = e->noise_level.p;
  uint64_t *d_totals = reinterpret_cast<uint64_t *> (e->d_stats);
  unsigned long long st[3];
  for (;;) {                                   // as the Mode S scan: a strip out of slots means scanning again with more
    HIP_TRY (launch_scanac2400 (sp, stream));
    HIP_TRY (launch_count_scan (e->counts.p, e->vp_counts.p, sp.nstrips, e->chunk_sum.p, e->chunk_vp.p, e->chunk_ovf.p,
                                d_totals, stream));
    HIP_TRY (hipMemcpyAsync (st, d_totals, sizeof st, hipMemcpyDeviceToHost, stream));
    HIP_TRY (hipStreamSynchronize (stream));
    if (!st[2]) break;
    if (sp.cap >= uint32_t (kStrip)) return fail (MODES_E_OVERFLOW, "%llu strips overflowed their Mode A/C slots", st[2]);
    sp.cap = sp.cap * 4 < uint32_t (kStrip) ? sp.cap * 4 : uint32_t (kStrip);
    e->cap = sp.cap;
    e->stats.scan_retries++;
    if ((rc = e->records.reserve (size_t (sp.nstrips) * sp.cap)) != MODES_OK) return rc;
    sp.records = e->records.p;
  }
  const size_t nrec = size_t (st[0]);
  if (!nrec) return MODES_OK;
  if ((rc = e->packed.reserve (nrec + 1)) != MODES_OK) return rc;
  if ((rc = e->packed_pos.reserve (nrec + 1)) != MODES_OK) return rc;
  if ((rc = e->offsets.reserve (sp.nstrips)) != MODES_OK) return rc;
  HIP_TRY (launch_gather (e->counts.p, sp.nstrips, e->chunk_sum.p, e->records.p, e->cap, e->packed.p, e->packed_pos.p,
                          e->offsets.p, stream));
  std::vector<Record> recs (nrec);
  std::vector<uint64_t> pos (nrec);
  HIP_TRY (hipMemcpyAsync (recs.data (), e->packed.p, nrec * sizeof (Record), hipMemcpyDeviceToHost, stream));
  HIP_TRY (hipMemcpyAsync (pos.data (), e->packed_pos.p, nrec * sizeof (uint64_t), hipMemcpyDeviceToHost, stream));
  HIP_TRY (hipStreamSynchronize (stream));

  if (e->in_pinned) { e->messages.assign (e->pin_msgs, e->pin_msgs + e->pin_n); e->in_pinned = false; }
  std::vector<modes_message> ac;
  accept_pass_2400ac (recs.data (), pos.data (), nrec, B, first_sample, e->cfg.sample_rate, ac);
  e->stats.valid_preamble += ac.size ();                       // ref: src/demod-2400AC.c:215-216
  e->stats.demod_modeac += ac.size ();
  // per block: Mode S messages first, then the block's replies
  std::vector<modes_message> merged;
  merged.reserve (e->messages.size () + ac.size ());
  size_t a = 0, s = 0;
  while (a < ac.size () || s < e->messages.size ()) {
    if (a == ac.size () || (s < e->messages.size () && e->messages[s].block <= ac[a].block)) merged.push_back (e->messages[s++]);
    else merged.push_back (ac[a++]);
  }
  e->messages.swap (merged);
  return MODES_OK;
}

// demod_8000 over a whole recording (ref: src/demod-8000.c:143-523); see kernels_8000.hip for the decomposition.
// chain_map (modes_engine_d8_chain_map): stop behind the trigger planes and the chunk tables and report where the
// window chain leaves the shard for every state it can be entered in; nothing is decoded and the state set for the run
// stays set.
static int run_8000 (modes_engine *e, const void *d_iq, size_t nbytes, uint64_t first_sample, const uint16_t *d_halo,
                     hipStream_t stream, int32_t *chain_map = nullptr, bool allow_pipeline = true)
{
  if (first_sample % e->block_samples && nbytes / bytes_per_sample (e->cfg.input_format) > e->block_samples)
    return fail (MODES_E_ARG, "first_sample must be a multiple of the block length");
  const auto t_entry = std::chrono::steady_clock::now ();
  auto since_entry = [&] { return std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now () - t_entry).count (); };
  const bool timing = getenv ("MODES_TIMING") != nullptr;
  HIP_TRY (hipSetDevice (e->cfg.device));
  e->d8_prev_msgs.swap (e->messages);
  e->d8_prev_serial = e->d8_msgs_serial;
  e->d8_msgs_serial = 0;
  e->messages.clear ();
  e->in_pinned = false; e->pin_n = 0;
  memset (&e->stats, 0, sizeof e->stats);
  const uint32_t B = e->block_samples;
  const uint64_t nsamples = nbytes / bytes_per_sample (e->cfg.input_format);
  const uint64_t nfull = nsamples / B, rest = nsamples - nfull * B;
  const uint64_t T = nfull * B + (rest & ~7ull);                  // `mlen &= ~7` per block (ref: :203)
  D8Params q;
  memset (&q, 0, sizeof q);
  ScanParams &sp = q.sp;
  sp.iq = static_cast<const uint8_t *> (d_iq); sp.nsamples = nsamples; sp.halo = d_halo; sp.ov = e->ov;
  sp.block_samples = B; sp.conv = e->conv;
  sp.short_tab = e->tabs.short_tab; sp.short_n = e->tabs.short_n; sp.long_tab = e->tabs.long_tab; sp.long_n = e->tabs.long_n;
  sp.lut256 = e->tabs.lut256; sp.bloom_short = e->tabs.bloom_short; sp.bloom_long = e->tabs.bloom_long;
  sp.crc_table = e->tabs.crc_table;
  if (e->cfg.filter_dc) {
    int rc_dc = prepare_dc (e, d_iq, nbytes, d_halo, stream, &sp);
    if (rc_dc != MODES_OK) return rc_dc;
  }
  q.T = T; q.nsteps = T / 8;
  q.bshift = 0; while ((1u << q.bshift) < B) q.bshift++;
  q.acc0 = 0; q.pending = 1;
  // what the shard before left (ref: src/demod-8000.c:80-89,186-220): one run only
  const bool have_in = e->d8_in_set;
  if (!chain_map) e->d8_in_set = false;
  e->d8_out_acc_valid = false;
  e->d8_amb_steps = 0;
  e->d8_have_run = false;
  unsigned long long trig_tail_in = 0;
  if (first_sample != 0 && !have_in)
    return fail (MODES_E_ARG, "demod_8000: a shard that does not start the stream needs modes_engine_set_d8_state() from the shard before it");
  if (have_in) {
    int rc_in;
    if ((rc_in = e->d8_dprev.reserve (kD8Overlap)) != MODES_OK) return rc_in;
    HIP_TRY (hipMemcpyAsync (e->d8_dprev.p, e->d8_in.dprev, kD8Overlap * sizeof (int32_t), hipMemcpyHostToDevice, stream));
    q.dprev = e->d8_dprev.p;
    q.acc0 = e->d8_in.acc; q.pending = e->d8_in.pending; trig_tail_in = e->d8_in.trig_tail;
  }
  if (!chain_map) {
    e->tail_half ^= 1;
    HIP_TRY (launch_tail (sp, sp.nsamples, e->tail.p + e->tail_half * ((e->ov + 7u) & ~7u), stream));   // what a following shard needs as d_halo
  }
  if (q.nsteps == 0) {
    if (have_in) { e->d8_out = e->d8_in; e->d8_out_acc_valid = e->d8_in.acc_valid != 0; e->d8_have_run = true; e->d8_passthrough = true; }
    return MODES_OK;
  }
  e->d8_passthrough = false;
  const uint32_t nchunks = uint32_t ((q.nsteps + kD8ChunkSteps - 1) / kD8ChunkSteps);
  q.nchunks = nchunks;
  const size_t nwords = size_t (q.nsteps / 64 + 64);
  int rc;
  if ((rc = e->d8_trig.reserve (nwords)) != MODES_OK) return rc;
  if ((rc = e->d8_amb.reserve (nwords)) != MODES_OK) return rc;
  if ((rc = e->d8_table.reserve (D8_TABLE_BYTES (nchunks))) != MODES_OK) return rc;
  if ((rc = e->d8_entry.reserve (nchunks + 1)) != MODES_OK) return rc;
  if ((rc = e->d8_win_start.reserve (size_t (nchunks) * kD8WinCap)) != MODES_OK) return rc;
  if ((rc = e->d8_win_count.reserve (nchunks)) != MODES_OK) return rc;
  if ((rc = e->d8_chunk_rec.reserve (2 * size_t (nchunks))) != MODES_OK) return rc;
  if ((rc = e->d8_announce.reserve ((1u << 24) / 32)) != MODES_OK) return rc;
  if ((rc = e->d8_windows.reserve (size_t (nchunks) * kD8WinCap)) != MODES_OK) return rc;
  if ((rc = e->d8_records.reserve (size_t (nchunks) * kD8WinCap * kD8Cand)) != MODES_OK) return rc;
  if ((rc = e->d8_small.reserve (16)) != MODES_OK) return rc;
  if ((rc = e->d8_off.reserve (2 * size_t (nchunks) + 2)) != MODES_OK) return rc;
  // the ranges of chunks the run is decoded in (below)
  uint32_t nranges = nchunks >= 256 ? 8u : nchunks >= 16 ? 4u : 1u;
  if (const char *ev = getenv ("MODES_RANGES")) { const int v = atoi (ev); if (v >= 1 && v <= 64) nranges = uint32_t (v) < nchunks ? uint32_t (v) : nchunks; }
  std::vector<uint32_t> lo (nranges + 1);
  {
    // equal ranges (a small first or last range, or more ranges, were measured in the pipelined run too: nothing in it)
    for (uint32_t r = 0; r < nranges; r++) lo[r] = uint32_t (uint64_t (nchunks) * r / nranges);
    lo[nranges] = nchunks;
  }
  q.trig = e->d8_trig.p; q.amb = e->d8_amb.p; q.table = e->d8_table.p; q.entry = e->d8_entry.p;
  // (with the DC block on the magnitudes are in HBM already: prepare_dc made `sp` a kConvMagU16 stream)
  if (sp.conv != kConvMagU16 && !getenv ("MODES_D8_NO_PLANE")) {
    const size_t vcap = size_t ((T + 7) & ~7ull) + 1024;         // the window kernel's tiles are fetched in whole 16-byte pieces
    if ((rc = e->d8_vmag.reserve (vcap)) != MODES_OK) return rc;
    q.vmag = e->d8_vmag.p; q.vcap = vcap;
  }
  q.win_start = e->d8_win_start.p; q.win_count = e->d8_win_count.p; q.windows = e->d8_windows.p; q.records = e->d8_records.p;
  // d8_small: [0..3] pre-window list, [4] pending window, [5] amb count (low half) , [6] last ambiguous step + 1
  long long *d_pre = reinterpret_cast<long long *> (e->d8_small.p);
  long long *d_pending = d_pre + 4;
  q.amb_count = reinterpret_cast<uint32_t *> (d_pre + 5);
  unsigned long long *d_limit = reinterpret_cast<unsigned long long *> (d_pre + 6);
  // The trigger planes and the chunk tables are functions of the samples, the differences carried in and (on near-silent
  // input) the running average carried in -- not of where the window chain enters.  A caller that promises the samples
  // are those of the call before (modes_engine_d8_chain_map, modes_engine_d8_reuse_planes) gets them reused.
  D8PlanesKey key;
  key.iq = d_iq; key.nbytes = nbytes; key.first_sample = first_sample; key.halo = d_halo; key.have_in = have_in; key.acc0 = q.acc0;
  const bool reuse = e->d8_reuse && !e->cfg.filter_dc && e->d8_planes_valid && key == e->d8_planes_key &&
                     (!have_in || !memcmp (e->d8_planes_dprev, e->d8_in.dprev, sizeof e->d8_planes_dprev));
  e->d8_reuse = false;
  e->d8_planes_valid = false;
  unsigned long long limit = 0;
  // Same planes and same entry of the window chain as the run before (a hand-over round in which only the ICAO cache
  // moved): the windows, their candidate records and what they announced stand.
  const bool same_windows = reuse && !chain_map && e->d8_win_gen == e->d8_planes_gen && e->d8_win_pending == (long long) q.pending &&
                            e->d8_win_tail == trig_tail_in;
  if (!chain_map) e->d8_win_gen = ~0ull;
  // Pipelined: the trigger planes and the chain are formed range by range too, in front of each range's windows, so the
  // host's ordered walk starts behind the first range instead of behind the whole recording's planes, and nothing waits
  // for the host between the planes and the windows.  It rests on the bet that no step needs the running average (none
  // does unless the input is near-silent); a range that shows such a step ends the bet and the run starts again the
  // plain way.
  const bool pipelined = allow_pipeline && !reuse && !chain_map && nranges > 1 && !getenv ("MODES_D8_NO_PIPELINE");
  std::vector<uint32_t> cache_at_entry;
  if (pipelined) cache_at_entry.assign (e->cache.raw (), e->cache.raw () + kIcaoSlots);
  if (reuse) {
    const long long init[5] = { 0, 0, 0, 0, 1 };                  // the pre-window list and the open window; counts and planes stay
    if (!same_windows) HIP_TRY (hipMemcpyAsync (d_pre, init, sizeof init, hipMemcpyHostToDevice, stream));
    limit = e->d8_planes_limit;
    e->kernel_ms[1] = 0;
  } else {
    const long long init[8] = { 0, 0, 0, 0, 1, 0, 0, 0 };
    HIP_TRY (hipMemcpyAsync (d_pre, init, sizeof init, hipMemcpyHostToDevice, stream));
    HIP_TRY (hipMemsetAsync (q.trig, 0, nwords * sizeof (uint64_t), stream));
    HIP_TRY (hipMemsetAsync (q.amb, 0, nwords * sizeof (uint64_t), stream));
  }
  if (!reuse && !pipelined) {
    HIP_TRY (hipEventRecord (e->ev[2], stream));
    HIP_TRY (launch_d8_trig (q, 0, stream));
    HIP_TRY (hipEventRecord (e->ev[3], stream));
    HIP_TRY (launch_d8_last_amb (q.amb, 0, nwords, d_limit, nullptr, stream));
    HIP_TRY (hipMemcpyAsync (&limit, d_limit, sizeof limit, hipMemcpyDeviceToHost, stream));
    HIP_TRY (hipStreamSynchronize (stream));
    if (timing) fprintf (stderr, "  %.3f ms: trigger planes done\n", since_entry ());
    if (limit && have_in && !e->d8_in.acc_valid)
      return fail (MODES_E_ARG, "demod_8000: this shard holds steps whose trigger depends on the running average (near-silent input): "
                                "the state handed over must carry acc (modes_engine_get_d8_state with want_acc)");
    if (limit) {
      // some steps sit in the band where the trigger depends on the running average: form it, in order, up to the last one
      if ((rc = e->d8_p0.reserve (size_t (q.nsteps) + 64)) != MODES_OK) return rc;
      if ((rc = e->d8_mx.reserve (size_t (q.nsteps) + 64)) != MODES_OK) return rc;
      q.p0 = e->d8_p0.p; q.mx = e->d8_mx.p; q.write_arrays = 1;
      HIP_TRY (launch_d8_trig (q, limit, stream));                  // only the strips up to the last such step
      q.write_arrays = 0;
      HIP_TRY (launch_d8_acc (q, limit, nullptr, stream));
    }
    HIP_TRY (launch_d8_chain_table (q, stream));
  }
  if (!reuse) e->d8_planes_gen++;
  e->d8_planes_key = key; e->d8_planes_limit = limit;
  if (have_in) memcpy (e->d8_planes_dprev, e->d8_in.dprev, sizeof e->d8_planes_dprev);
  e->d8_planes_timed = !reuse;
  if (chain_map) {
    if (limit && first_sample != 0)
      return fail (MODES_E_ARG, "modes_engine_d8_chain_map: this shard holds steps whose trigger depends on the running average "
                                "(near-silent input), which depends on every shard before it");
    // map[i]: the window left open behind the shard when it is entered with pending = (i ? -i : 1)
    if ((rc = e->d8_exit_of.reserve (128)) != MODES_OK) return rc;
    int32_t exit_of[kD8Net];
    HIP_TRY (launch_d8_chain_exit (q, e->d8_exit_of.p, stream));
    HIP_TRY (hipMemcpyAsync (exit_of, e->d8_exit_of.p, sizeof exit_of, hipMemcpyDeviceToHost, stream));
    HIP_TRY (hipStreamSynchronize (stream));
    for (int i = 0; i < MODES_D8_CHAIN_STATES; i++) {
      // what d8_chain_resolve_kernel does in front of step 0 (ref: src/demod-8000.c:493-496 re-scans the last steps)
      long long pos = 0;
      if (i) pos = -(long long) i + kD8Net;
      while (pos < 0) {
        const int b = int (pos + 64);
        const unsigned long long word = b >= 0 ? trig_tail_in & (~0ull << b) : trig_tail_in;
        if (!word) { pos = 0; break; }
        pos = (long long) __builtin_ctzll (word) - 64 + kD8Net;
      }
      chain_map[i] = exit_of[pos < kD8Net - 1 ? pos : kD8Net - 1];
    }
    e->d8_planes_valid = true;                                     // the run that follows on the same samples starts from them
    e->d8_reuse = true;
    return MODES_OK;
  }
  if (!pipelined) e->d8_planes_valid = true;                       // (pipelined: when the last range's are there, at the end)
  // addresses the run can know: what the cache starts with, plus every address the window kernel sees announced
  q.announce = e->d8_announce.p;
  {
    // the bitmap is cleared on the device; only the addresses the starting cache holds travel (4 KB)
    if (!same_windows) HIP_TRY (hipMemsetAsync (q.announce, 0, ((1u << 24) / 32) * sizeof (uint32_t), stream));
    if ((rc = e->pin_bits.reserve (kIcaoSlots)) != MODES_OK) return rc;
    if ((rc = e->small_u32.reserve (2 * kIcaoSlots + 16)) != MODES_OK) return rc;
    uint32_t nseed = 0;
    for (uint32_t i = 0; i < kIcaoSlots; i++)                      // the non-ICAO flag (bit 24) never matches a 24-bit test
      if (e->cache.raw ()[i] && e->cache.raw ()[i] < (1u << 24)) e->pin_bits.p[nseed++] = e->cache.raw ()[i] & 0xFFFFFFu;
    if (nseed) {
      HIP_TRY (hipMemcpyAsync (e->small_u32.p, e->pin_bits.p, nseed * sizeof (uint32_t), hipMemcpyHostToDevice, stream));
      HIP_TRY (launch_d8_seed (q.announce, e->small_u32.p, nseed, stream));
    }
  }
  if (!same_windows && !pipelined) HIP_TRY (launch_d8_chain (q, trig_tail_in, 0ll, d_pre, d_pending, stream));
  auto queue_hand_over = [&] (hipStream_t hs) -> int {
    // what this run leaves for the shard behind it: the last 1928 differences, the open window, the last 64 trigger
    // decisions (read back with the first range's counts below); acc only on request (modes_engine_get_d8_state)
    int rc_out;
    if ((rc_out = e->d8_dnext.reserve (kD8Overlap)) != MODES_OK) return rc_out;
    if ((rc_out = e->pin_d8.reserve (kD8Overlap + 16)) != MODES_OK) return rc_out;
    HIP_TRY (launch_d8_tail (q, e->d8_dnext.p, hs));
    HIP_TRY (hipMemcpyAsync (e->pin_d8.p, e->d8_dnext.p, kD8Overlap * sizeof (int32_t), hipMemcpyDeviceToHost, hs));
    HIP_TRY (hipMemcpyAsync (e->pin_d8.p + kD8Overlap, d_pending, sizeof (long long), hipMemcpyDeviceToHost, hs));
    const size_t w_last = size_t ((q.nsteps - 1) >> 6);
    HIP_TRY (hipMemcpyAsync (e->pin_d8.p + kD8Overlap + 2, q.trig + (w_last ? w_last - 1 : 0), (w_last ? 2 : 1) * sizeof (uint64_t), hipMemcpyDeviceToHost, hs));
    HIP_TRY (hipMemcpyAsync (e->pin_d8.p + kD8Overlap + 6, q.amb_count, sizeof (uint32_t), hipMemcpyDeviceToHost, hs));
    return MODES_OK;
  };
  if (!pipelined && (rc = queue_hand_over (stream)) != MODES_OK) return rc;
  e->d8_q = q; e->d8_limit = limit; e->d8_trig_tail_in = trig_tail_in;
  // The completed windows in ranges of chunks.  Everything the device does for a range is queued up front on the
  // caller's stream, all ranges back to back: windows, count (address filter), offsets (one wave: where each chunk's
  // live windows and records go), gather -- which writes the packed windows and records straight into pinned host
  // memory -- and an event.  The host waits for a range's event, walks its records in order (the part of demod_8000
  // that needs the live ICAO cache) while the device is busy with the ranges behind it, and queues the signal levels of
  // the messages it accepted on a second stream.  A record of range r can only be accepted through an address announced
  // in ranges <= r (or cached at the start), and count(r) runs behind windows(r), so the address filter sees all it needs.
  // The pinned arrays are sized from the last run (or a default per chunk); a range that does not fit is gathered again
  // into arrays of the right size.
  if (e->d8_events.size () < 3 * size_t (nranges)) {
    const size_t have = e->d8_events.size ();
    e->d8_events.resize (3 * size_t (nranges), nullptr);
    for (size_t i = have; i < e->d8_events.size (); i++) HIP_TRY (hipEventCreate (&e->d8_events[i]));
  }
  if (!e->d8_chain_done) HIP_TRY (hipEventCreateWithFlags (&e->d8_chain_done, hipEventDisableTiming));
  while (e->d8_edge_ev.size () < nranges) {
    hipEvent_t ev;
    HIP_TRY (hipEventCreateWithFlags (&ev, hipEventDisableTiming));
    e->d8_edge_ev.push_back (ev);
  }
  while (e->d8_gathered.size () < nranges) {
    hipEvent_t ev;
    HIP_TRY (hipEventCreateWithFlags (&ev, hipEventDisableTiming));
    e->d8_gathered.push_back (ev);
  }
  if (!e->post_stream) HIP_TRY (hipStreamCreateWithFlags (&e->post_stream, hipStreamNonBlocking));
  if (!e->copy_stream) HIP_TRY (hipStreamCreateWithFlags (&e->copy_stream, hipStreamNonBlocking));
  hipStream_t side = e->post_stream, lvl = e->copy_stream;       // gathers (queued up front) / signal levels (queued range by range)
  struct Drain { hipStream_t a, b, c; ~Drain () { (void) hipStreamSynchronize (a); (void) hipStreamSynchronize (b); (void) hipStreamSynchronize (c); } } drain = { side, lvl, stream };
  // per-chunk budget of the pinned arrays
  uint64_t bw = 192, br = 256;
  if (e->d8_last_chunks) {
    bw = e->d8_last_nwin * 3 / 2 / e->d8_last_chunks + 32;
    br = e->d8_last_nrec * 3 / 2 / e->d8_last_chunks + 32;
  }
  if (const char *ev = getenv ("MODES_D8_BUDGET")) if (atoi (ev) > 0) bw = br = uint64_t (atoi (ev));      // test hook
  if ((rc = e->pin_wins.reserve (size_t (nchunks) * bw)) != MODES_OK) return rc;
  if ((rc = e->pin_recs.reserve (size_t (nchunks) * br + 1)) != MODES_OK) return rc;
  if ((rc = e->d8_tot.reserve (4 * size_t (nranges))) != MODES_OK) return rc;
  if ((rc = e->pin_tot.reserve (4 * size_t (nranges) + 4)) != MODES_OK) return rc;      // + needs_average, limit_host
  // pipelined: set by the device when a range holds a step that needs the running average
  volatile uint32_t *const needs_average = reinterpret_cast<volatile uint32_t *> (e->pin_tot.p + 4 * size_t (nranges));
  *needs_average = 0;
  // the few windows at block edges are slow ones (a few waves that wait a lot): the first range's in front of its
  // other windows, every other range's in a launch of its own on the stream of the signal levels beside them (idle
  // until the first walk is over) -- 0.1-0.2 ms each there, through long before the range's other windows are.  (All of
  // them in one launch took 0.7-1.3 ms beside the window kernels and the second range waited for it: the host stood
  // still for 0.6 ms and never made that up.)
  if (pipelined) {
    // caller's stream: planes (0), planes (1), windows (0), planes (2), windows (1), ... -- the windows of a range read the
    // magnitude plane a little into the next range's; second stream: chain (0), chain (1), edge windows / count / gather (0),
    // chain (2), ... -- a range's chain (small kernels that leave the GPU nearly empty) runs beside the next range's planes.
    while (e->d8_trig_ev.size () < 2 * size_t (nranges)) {
      hipEvent_t ev;
      HIP_TRY (hipEventCreate (&ev));
      e->d8_trig_ev.push_back (ev);
    }
    while (e->d8_chain_ev.size () < nranges) {
      hipEvent_t ev;
      HIP_TRY (hipEventCreateWithFlags (&ev, hipEventDisableTiming));
      e->d8_chain_ev.push_back (ev);
    }
    constexpr uint64_t kChunkWords = kD8ChunkSteps / 64;
    auto queue_planes = [&] (uint32_t r) -> int {
      const uint32_t c0 = lo[r], n = lo[r + 1] - c0;
      const bool last = r + 1 == nranges;
      HIP_TRY (hipEventRecord (e->d8_trig_ev[2 * r], stream));
      HIP_TRY (launch_d8_trig (q, 0, stream, c0, last ? 0 : n));
      HIP_TRY (hipEventRecord (e->d8_trig_ev[2 * r + 1], stream));
      const uint64_t w1 = last ? uint64_t (nwords) : std::min (uint64_t (nwords), uint64_t (lo[r + 1]) * kChunkWords);
      // (the first range's such steps -- the stream's first steps nearly always are -- get their average below)
      HIP_TRY (launch_d8_last_amb (q.amb, uint64_t (c0) * kChunkWords, w1, d_limit, r ? const_cast<uint32_t *> (needs_average) : nullptr, stream));
      return MODES_OK;
    };
    auto queue_chain = [&] (uint32_t r, hipEvent_t planes_done) -> int {
      const uint32_t c0 = lo[r], n = lo[r + 1] - c0;
      const bool last = r + 1 == nranges;
      HIP_TRY (hipStreamWaitEvent (side, planes_done, 0));
      HIP_TRY (launch_d8_chain_table (q, side, c0, n));
      HIP_TRY (launch_d8_chain (q, trig_tail_in, 0ll, d_pre, d_pending, side, c0, last ? 0 : n));
      HIP_TRY (hipEventRecord (e->d8_chain_ev[r], side));
      return last ? queue_hand_over (side) : MODES_OK;
    };
    auto queue_windows = [&] (uint32_t k) -> int {
      const uint32_t c0 = lo[k], n = lo[k + 1] - c0;
      q.chunk0 = c0;
      HIP_TRY (hipStreamWaitEvent (stream, e->d8_chain_ev[k], 0));
      HIP_TRY (hipEventRecord (e->d8_events[3 * k], stream));
      HIP_TRY (launch_d8_windows (q, n, stream, 1));                 // (a stream of their own beside the planes: measured, no gain)
      HIP_TRY (hipEventRecord (e->d8_events[3 * k + 1], stream));
      HIP_TRY (launch_d8_windows (q, n, side, 2));                 // the few windows at block edges, beside the others
      HIP_TRY (hipStreamWaitEvent (side, e->d8_events[3 * k + 1], 0));
      HIP_TRY (launch_d8_count (q, n, e->d8_chunk_rec.p, side));
      HIP_TRY (launch_d8_offsets (e->d8_chunk_rec.p, e->d8_win_count.p, nchunks, c0, n, e->d8_off.p, e->d8_off.p + nchunks + 1,
                                  uint64_t (n) * bw, uint64_t (n) * br, e->d8_tot.p + 4 * k, e->pin_tot.p + 4 * k, side));
      HIP_TRY (launch_d8_gather (q, n, e->d8_off.p, e->d8_off.p + nchunks + 1, e->pin_wins.p + size_t (c0) * bw, e->pin_recs.p + size_t (c0) * br,
                                 side, e->d8_tot.p + 4 * k));
      HIP_TRY (hipEventRecord (e->d8_gathered[k], side));
      return MODES_OK;
    };
    // the first range's planes, and whether its first steps need the running average (the host looks while the second
    // range's planes are being formed)
    volatile uint32_t *const limit_host = needs_average + 2;
    limit_host[0] = limit_host[1] = 0;
    if ((rc = queue_planes (0)) != MODES_OK) return rc;
    HIP_TRY (launch_copy_u32 (const_cast<uint32_t *> (limit_host), reinterpret_cast<const uint32_t *> (d_limit), 2, stream));
    HIP_TRY (hipEventRecord (e->d8_chain_done, stream));
    if ((rc = queue_planes (1)) != MODES_OK) return rc;
    HIP_TRY (hipEventSynchronize (e->d8_chain_done));
    limit = uint64_t (limit_host[0]) | uint64_t (limit_host[1]) << 32;
    if (timing) fprintf (stderr, "  %.3f ms: planes of range 0 done, %llu steps up to the last that needs the running average\n", since_entry (), limit);
    hipEvent_t planes0 = e->d8_trig_ev[1];
    if (limit) {
      if (have_in && !e->d8_in.acc_valid)
        return fail (MODES_E_ARG, "demod_8000: this shard holds steps whose trigger depends on the running average (near-silent input): "
                                  "the state handed over must carry acc (modes_engine_get_d8_state with want_acc)");
      if ((rc = e->d8_p0.reserve (size_t (q.nsteps) + 64)) != MODES_OK) return rc;
      if ((rc = e->d8_mx.reserve (size_t (q.nsteps) + 64)) != MODES_OK) return rc;
      q.p0 = e->d8_p0.p; q.mx = e->d8_mx.p; q.write_arrays = 1;
      HIP_TRY (launch_d8_trig (q, limit, stream));                  // only the strips up to the last such step
      q.write_arrays = 0;
      HIP_TRY (launch_d8_acc (q, limit, nullptr, stream));
      HIP_TRY (hipEventRecord (e->d8_chain_done, stream));
      planes0 = e->d8_chain_done;
      e->d8_planes_limit = limit; e->d8_limit = limit; e->d8_q = q;
    }
    if ((rc = queue_chain (0, planes0)) != MODES_OK) return rc;
    if ((rc = queue_chain (1, e->d8_trig_ev[3])) != MODES_OK) return rc;
    if ((rc = queue_windows (0)) != MODES_OK) return rc;
    for (uint32_t r = 2; r < nranges; r++) {
      if ((rc = queue_planes (r)) != MODES_OK) return rc;
      if ((rc = queue_chain (r, e->d8_trig_ev[2 * r + 1])) != MODES_OK) return rc;
      if ((rc = queue_windows (r - 1)) != MODES_OK) return rc;
    }
    if ((rc = queue_windows (nranges - 1)) != MODES_OK) return rc;
  }
  if (!same_windows && !pipelined) {
    q.chunk0 = 0;
    HIP_TRY (launch_d8_windows (q, lo[1], stream, 2));
    if (nranges > 1) {
      HIP_TRY (hipEventRecord (e->d8_chain_done, stream));
      HIP_TRY (hipStreamWaitEvent (lvl, e->d8_chain_done, 0));     // (the stream of the signal levels: idle until the first walk is over)
      for (uint32_t r = 1; r < nranges; r++) {
        q.chunk0 = lo[r];
        HIP_TRY (launch_d8_windows (q, lo[r + 1] - lo[r], lvl, 2));
        HIP_TRY (hipEventRecord (e->d8_edge_ev[r], lvl));
      }
    }
  }
  for (uint32_t r = 0; r < nranges && !pipelined; r++) {
    const uint32_t c0 = lo[r], n = lo[r + 1] - c0;
    q.chunk0 = c0;
    HIP_TRY (hipEventRecord (e->d8_events[3 * r], stream));
    if (!same_windows) HIP_TRY (launch_d8_windows (q, n, stream, 1));
    HIP_TRY (hipEventRecord (e->d8_events[3 * r + 1], stream));
    // Count, offsets and gather of the range run on the second stream beside the next range's window kernel, which
    // leaves the caller's stream to the window kernels alone.  The address filter of count (r) may then also see what
    // windows of range r + 1 have announced by the time it runs: a record more is kept now and then (its address is not
    // in the cache when the walk gets there, so it changes nothing but `candidates`).
    HIP_TRY (hipStreamWaitEvent (side, e->d8_events[3 * r + 1], 0));
    if (r >= 1 && !same_windows) HIP_TRY (hipStreamWaitEvent (side, e->d8_edge_ev[r], 0));
    HIP_TRY (launch_d8_count (q, n, e->d8_chunk_rec.p, side));
    HIP_TRY (launch_d8_offsets (e->d8_chunk_rec.p, e->d8_win_count.p, nchunks, c0, n, e->d8_off.p, e->d8_off.p + nchunks + 1,
                                uint64_t (n) * bw, uint64_t (n) * br, e->d8_tot.p + 4 * r, e->pin_tot.p + 4 * r, side));
    HIP_TRY (launch_d8_gather (q, n, e->d8_off.p, e->d8_off.p + nchunks + 1, e->pin_wins.p + size_t (c0) * bw, e->pin_recs.p + size_t (c0) * br,
                               side, e->d8_tot.p + 4 * r));
    HIP_TRY (hipEventRecord (e->d8_gathered[r], side));
  }
  Pass8000Stats ps;
  if (timing) fprintf (stderr, "  %.3f ms: everything queued\n", since_entry ());
  // the messages whose signal level is measured, over all ranges: pin_rpos / pin_rlen (position, samples summed),
  // lvl_msg (index in e->messages), pin_sums (what the device found); range r's part starts where range r - 1's ended
  std::vector<uint32_t> &lvl_msg = e->d8_lvl_msg;
  auto level_room = [&] (size_t n) -> int {
    if (n <= e->pin_rpos.cap && n <= e->pin_rlen.cap && n <= e->pin_sums.cap && n <= e->rssi_pos.cap && n <= e->rssi_len.cap && n <= e->rssi_sum.cap) return MODES_OK;
    // (first run, or more messages than the run before) nothing of an earlier range may be in flight in what grows
    hipError_t he = hipStreamSynchronize (lvl);
    if (he != hipSuccess) return fail (MODES_E_HIP, "hipStreamSynchronize: %s", hipGetErrorString (he));
    const size_t want = 2 * n;
    int rc2;
    PinBuf<uint64_t> sums; PinBuf<uint32_t> lens; PinBuf<uint64_t> poss;
    if ((rc2 = sums.reserve (want)) != MODES_OK || (rc2 = lens.reserve (want)) != MODES_OK || (rc2 = poss.reserve (want)) != MODES_OK) {
      sums.release (); lens.release (); poss.release ();
      return rc2;
    }
    const size_t keep = std::min (n, std::min (e->pin_sums.cap, e->pin_rlen.cap));
    if (keep) { memcpy (sums.p, e->pin_sums.p, keep * sizeof (uint64_t)); memcpy (lens.p, e->pin_rlen.p, keep * sizeof (uint32_t)); }
    e->pin_sums.release (); e->pin_sums = sums;
    e->pin_rlen.release (); e->pin_rlen = lens;
    e->pin_rpos.release (); e->pin_rpos = poss;
    if ((rc2 = e->rssi_pos.reserve (want)) != MODES_OK) return rc2;
    if ((rc2 = e->rssi_len.reserve (want)) != MODES_OK) return rc2;
    return e->rssi_sum.reserve (want);
  };
  size_t nsums = 0;
  // The levels go into the messages range by range as they arrive, in front of the wait for the next range's windows
  // (140 K divisions and scattered stores: 0.15 ms if left to the end, when nothing else is going on).
  std::vector<size_t> lvl_end;                                     // nsums behind every range that queued levels
  size_t lvl_applied = 0, lvl_applied_k = 0;
  auto apply_levels = [&] (bool all) {
    while (lvl_applied_k < lvl_end.size ()) {
      if (!all && hipEventQuery (e->d8_lvl_ev[lvl_applied_k]) != hipSuccess) { (void) hipGetLastError (); break; }
      const size_t end = lvl_end[lvl_applied_k++];
      for (size_t i = lvl_applied; i < end; i++)
        e->messages[lvl_msg[i]].sig_level = (double (e->pin_sums.p[i]) / 65535.0 / 65535.0) / double (e->pin_rlen.p[i]);
      lvl_applied = end;
    }
  };
  uint64_t all_win = 0, all_rec = 0;
  double host_ms = 0.0;
  e->kernel_ms[0] = 0;
  e->messages.reserve (e->d8_last_msgs + e->d8_last_msgs / 8 + 64);
  // the last run's trace, if this run walks the same windows in the same ranges; the trace of this run
  modes_engine::D8Trace prev;
  prev.valid = false;
  if (same_windows && e->d8_trace.valid && e->d8_trace.lo == lo && e->d8_trace.serial && e->d8_trace.serial == e->d8_prev_serial &&
      !getenv ("MODES_D8_NO_EARLY_STOP"))
    prev = std::move (e->d8_trace);
  e->d8_trace = modes_engine::D8Trace ();
  modes_engine::D8Trace cur;
  cur.lo = lo;
  cur.cache_after.resize (size_t (nranges) * kIcaoSlots); cur.msg_end.resize (nranges); cur.ps_after.resize (nranges); cur.cand_after.resize (nranges);
  auto ps_add = [] (Pass8000Stats a, const Pass8000Stats &b, const Pass8000Stats &c) {     // a + b - c
    static_assert (sizeof (Pass8000Stats) % sizeof (uint64_t) == 0, "counters only");
    uint64_t *x = reinterpret_cast<uint64_t *> (&a);
    const uint64_t *y = reinterpret_cast<const uint64_t *> (&b), *z = reinterpret_cast<const uint64_t *> (&c);
    for (size_t i = 0; i < sizeof (Pass8000Stats) / sizeof (uint64_t); i++) x[i] += y[i] - z[i];
    return a;
  };
  bool stopped_early = false;
  for (uint32_t r = 0; r < nranges; r++) {
    if (r) {
      // behind range r - 1: the trace, and -- over the same windows as the last run -- whether the two runs have met
      const uint32_t d = r - 1;
      memcpy (cur.cache_after.data () + size_t (d) * kIcaoSlots, e->cache.raw (), kIcaoSlots * sizeof (uint32_t));
      cur.msg_end[d] = e->messages.size (); cur.ps_after[d] = ps; cur.cand_after[d] = e->stats.candidates;
      if (prev.valid && !memcmp (prev.cache_after.data () + size_t (d) * kIcaoSlots, e->cache.raw (), kIcaoSlots * sizeof (uint32_t))) {
        // same cache, same windows, and every record this run keeps that the last one dropped asks for an address the
        // cache does not hold: ranges r .. end go as they went.  (The device finishes their count / gather unobserved.)
        const size_t n_now = e->messages.size (), p_from = prev.msg_end[d];
        e->messages.insert (e->messages.end (), e->d8_prev_msgs.begin () + p_from, e->d8_prev_msgs.end ());
        for (uint32_t k = r; k < nranges; k++) {
          memcpy (cur.cache_after.data () + size_t (k) * kIcaoSlots, prev.cache_after.data () + size_t (k) * kIcaoSlots, kIcaoSlots * sizeof (uint32_t));
          cur.msg_end[k] = prev.msg_end[k] - p_from + n_now;
          cur.ps_after[k] = ps_add (ps, prev.ps_after[k], prev.ps_after[d]);
          cur.cand_after[k] = e->stats.candidates + prev.cand_after[k] - prev.cand_after[d];
        }
        ps = cur.ps_after[nranges - 1];
        e->stats.candidates = cur.cand_after[nranges - 1];
        memcpy (e->cache.raw (), prev.final_cache.data (), kIcaoSlots * sizeof (uint32_t));
        e->stats.replay_rounds = r;                             // ranges walked before the runs met
        stopped_early = true;
        break;
      }
    }
    const uint32_t c0 = lo[r], c1 = lo[r + 1];
    apply_levels (false);
    const auto t_wait0 = std::chrono::steady_clock::now ();
    HIP_TRY (hipEventSynchronize (e->d8_gathered[r]));
    if (timing) fprintf (stderr, "  %.3f ms: range %u: waited %.3f ms for the device", since_entry (), r, std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now () - t_wait0).count ());
    if (pipelined && *needs_average) {
      // the bet is lost: nothing of this run stands; the engine as it was entered, and the plain way
      if (timing) fprintf (stderr, "\n  %.3f ms: a step of range %u needs the running average: once more, planes first\n", since_entry (), r);
      HIP_TRY (hipStreamSynchronize (side)); HIP_TRY (hipStreamSynchronize (lvl)); HIP_TRY (hipStreamSynchronize (stream));
      memcpy (e->cache.raw (), cache_at_entry.data (), kIcaoSlots * sizeof (uint32_t));
      e->messages.swap (e->d8_prev_msgs);
      e->d8_msgs_serial = e->d8_prev_serial;
      e->d8_in_set = have_in;
      e->tail_half ^= 1;
      const int rc_again = run_8000 (e, d_iq, nbytes, first_sample, d_halo, stream, chain_map, false);
      e->stats.scan_retries++;
      return rc_again;
    }
    float wms = 0;
    if (pipelined) {
      if (r == 0) e->kernel_ms[1] = 0;
      (void) hipEventElapsedTime (&wms, e->d8_trig_ev[2 * r], e->d8_trig_ev[2 * r + 1]);
      e->kernel_ms[1] += wms;
    } else if (r == 0 && e->d8_planes_timed) (void) hipEventElapsedTime (&e->kernel_ms[1], e->ev[2], e->ev[3]);
    (void) hipEventElapsedTime (&wms, e->d8_events[3 * r], e->d8_events[3 * r + 1]);
    e->kernel_ms[0] += wms;
    const uint64_t *tot = e->pin_tot.p + 4 * r;
    const uint64_t nwin = tot[0], nrec = tot[1];
    if (tot[2] > uint64_t (kD8WinCap)) return fail (MODES_E_OVERFLOW, "a chunk of range %u holds %llu windows (cap %d)", r, (unsigned long long) tot[2], kD8WinCap);
    e->stats.candidates += nrec;
    all_win += nwin; all_rec += nrec;
    if (!nwin) continue;
    const D8Window *wins = e->pin_wins.p + size_t (c0) * bw;
    const Record *recs = e->pin_recs.p + size_t (c0) * br;
    if (tot[3]) {
      // the range outgrew its share of the pinned arrays: gather it again (the offsets are right) into arrays of its size
      e->stats.scan_retries++;
      if ((rc = e->d8_win_packed.reserve (nwin)) != MODES_OK) return rc;
      if ((rc = e->d8_rec_packed.reserve (nrec + 1)) != MODES_OK) return rc;
      if ((rc = e->pin_wins2.reserve (nwin)) != MODES_OK) return rc;
      if ((rc = e->pin_recs2.reserve (nrec + 1)) != MODES_OK) return rc;
      q.chunk0 = c0;
      HIP_TRY (launch_d8_gather (q, c1 - c0, e->d8_off.p, e->d8_off.p + nchunks + 1, e->d8_win_packed.p, e->d8_rec_packed.p, side));
      HIP_TRY (hipMemcpyAsync (e->pin_wins2.p, e->d8_win_packed.p, nwin * sizeof (D8Window), hipMemcpyDeviceToHost, side));
      if (nrec) HIP_TRY (hipMemcpyAsync (e->pin_recs2.p, e->d8_rec_packed.p, nrec * sizeof (Record), hipMemcpyDeviceToHost, side));
      HIP_TRY (hipStreamSynchronize (side));
      wins = e->pin_wins2.p; recs = e->pin_recs2.p;
    }
    const auto t_host0 = std::chrono::steady_clock::now ();
    // every live window yields at most one message: room for all of them, trimmed behind the walk
    if ((rc = level_room (nsums + nwin)) != MODES_OK) return rc;
    const size_t m0 = e->messages.size ();
    e->messages.resize (m0 + nwin);
    if (lvl_msg.size () < nsums + nwin) lvl_msg.resize (nsums + nwin + (nsums + nwin) / 4);
    {
      modes_message *const mout = e->messages.data () + m0;
      uint64_t *const rp = e->pin_rpos.p + nsums;
      uint32_t *const rl = e->pin_rlen.p + nsums, *const lm = lvl_msg.data () + nsums;
      size_t nm = 0, nr = 0;
      IcaoCache &cache = e->cache;
      accept_pass_8000_with (wins, nwin, recs, B, q.bshift, T, e->ov, first_sample, e->cfg.sample_rate,
        [&] (const Record &rr, int) { return score_record (rr, cache); },
        [&] (const Record &rr, uint64_t, modes_message *out) { return accept_record (rr, cache, out); },
        [&] (modes_message &m, const Record &, uint64_t blkc, int m_start, int slen) {
          if (!m.addr) return;                                      // ref: src/demod-8000.c:487
          if (m_start >= 0) {
            rp[nr] = blkc * B + uint64_t (m_start) - 19;            // the shared kernel sums from position + 19
            rl[nr] = uint32_t (slen);
            lm[nr] = uint32_t (m0 + nm);
            nr++;
          }
          nm++;                                                     // the message stays where it was built
        },
        [&] { return mout + nm; }, ps, &cache);
      e->messages.resize (m0 + nm);
      host_ms += std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now () - t_host0).count ();
      if (timing) fprintf (stderr, ", walked %llu windows / %llu records in %.3f ms", (unsigned long long) nwin, (unsigned long long) nrec,
                           std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now () - t_host0).count ());
      // signal level of the range's messages (ref: :469-484): queued behind the walk, read when all ranges are done
      if (nr) {
        HIP_TRY (hipMemcpyAsync (e->rssi_pos.p + nsums, rp, nr * sizeof (uint64_t), hipMemcpyHostToDevice, lvl));
        HIP_TRY (hipMemcpyAsync (e->rssi_len.p + nsums, rl, nr * sizeof (uint32_t), hipMemcpyHostToDevice, lvl));
        HIP_TRY (launch_rssi2400 (sp.conv, sp.iq, sp.conv == kConvMagU16 ? sp.navail : sp.nsamples, sp.halo, sp.ov, sp.block_samples,
                                  sp.lut256, e->rssi_pos.p + nsums, e->rssi_len.p + nsums, e->rssi_sum.p + nsums, uint32_t (nr), lvl));
        HIP_TRY (hipMemcpyAsync (e->pin_sums.p + nsums, e->rssi_sum.p + nsums, nr * sizeof (uint64_t), hipMemcpyDeviceToHost, lvl));
        nsums += nr;
        while (e->d8_lvl_ev.size () <= lvl_end.size ()) {
          hipEvent_t ev;
          HIP_TRY (hipEventCreateWithFlags (&ev, hipEventDisableTiming));
          e->d8_lvl_ev.push_back (ev);
        }
        HIP_TRY (hipEventRecord (e->d8_lvl_ev[lvl_end.size ()], lvl));
        lvl_end.push_back (nsums);
      }
    }
    const auto t_lvl0 = t_host0;
    if (timing) fprintf (stderr, ", levels queued in %.3f ms\n", std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now () - t_lvl0).count ());
  }
  if (!stopped_early) {
    const uint32_t d = nranges - 1;
    memcpy (cur.cache_after.data () + size_t (d) * kIcaoSlots, e->cache.raw (), kIcaoSlots * sizeof (uint32_t));
    cur.msg_end[d] = e->messages.size (); cur.ps_after[d] = ps; cur.cand_after[d] = e->stats.candidates;
  }
  const auto t_end0 = std::chrono::steady_clock::now ();
  HIP_TRY (hipStreamSynchronize (lvl));
  HIP_TRY (hipStreamSynchronize (stream));
  if (pipelined) { HIP_TRY (hipStreamSynchronize (side)); e->d8_planes_valid = true; }
  if (timing) fprintf (stderr, "  %.3f ms: end: %.3f ms for the streams to drain\n", since_entry (), std::chrono::duration<double, std::milli> (std::chrono::steady_clock::now () - t_end0).count ());
  e->d8_win_gen = e->d8_planes_gen; e->d8_win_pending = (long long) q.pending; e->d8_win_tail = trig_tail_in;
  if (!stopped_early) { e->d8_last_chunks = nchunks; e->d8_last_nwin = all_win; e->d8_last_nrec = all_rec; }   // (what sizes the next run's arrays)
  e->d8_last_msgs = e->messages.size ();
  apply_levels (true);
  cur.final_cache.assign (e->cache.raw (), e->cache.raw () + kIcaoSlots);
  cur.serial = e->d8_msgs_serial = ++e->d8_serials;
  cur.valid = true;
  e->d8_trace = std::move (cur);
  {
    // the hand-over state (the copies were queued in front of the first range: long done)
    modes_d8_state &o = e->d8_out;
    memcpy (o.dprev, e->pin_d8.p, kD8Overlap * sizeof (int32_t));
    long long pend; memcpy (&pend, e->pin_d8.p + kD8Overlap, sizeof pend);
    o.pending = pend;
    uint64_t tw[2] = { 0, 0 };
    const size_t w_last = size_t ((q.nsteps - 1) >> 6);
    if (w_last) memcpy (tw, e->pin_d8.p + kD8Overlap + 2, 2 * sizeof (uint64_t)); else memcpy (&tw[1], e->pin_d8.p + kD8Overlap + 2, sizeof (uint64_t));
    uint64_t tail = 0;
    for (int b = 0; b < 64; b++) {
      const long long n = (long long) q.nsteps - 64 + b;               // step of bit b
      bool bit;
      if (n >= 0) { const size_t wi = size_t (n) >> 6; bit = (tw[wi == w_last ? 1 : 0] >> (n & 63)) & 1; }
      else bit = (n + 64 >= 0) ? ((trig_tail_in >> (n + 64)) & 1) : false;   // still a step of the shard before
      if (bit) tail |= 1ull << b;
    }
    o.trig_tail = tail;
    uint32_t namb; memcpy (&namb, e->pin_d8.p + kD8Overlap + 6, sizeof namb);
    o.amb_steps = namb; e->d8_amb_steps = namb;
    o.acc = 0; o.acc_valid = 0;
    e->d8_have_run = true;
  }
  if (timing) fprintf (stderr, "  %.3f ms: run_8000: ordered pass %.3f ms in %u ranges, %zu messages\n", since_entry (), host_ms, nranges, e->messages.size ());
  e->stats.valid_preamble = ps.valid_preamble; e->stats.demodulated = ps.demodulated;
  e->stats.CRC_good = ps.CRC_good; e->stats.CRC_bad = ps.CRC_bad; e->stats.CRC_single_bit_fix = ps.CRC_single_bit_fix;
  e->stats.CRC_two_bits_fix = ps.CRC_two_bits_fix; e->stats.CRC_fixed = ps.CRC_single_bit_fix + ps.CRC_two_bits_fix;
  e->stats.demod_rejected_unknown = ps.rejected_unknown;
  return MODES_OK;
}

int modes_engine_run_device (modes_engine *e, const void *d_iq, size_t nbytes, uint64_t first_sample,
                             const uint16_t *d_halo, void *stream_v)
{
  if (!e || (!d_iq && nbytes)) return fail (MODES_E_ARG, "null argument");
  if (e->mode == 8000) return run_8000 (e, d_iq, nbytes, first_sample, d_halo, static_cast<hipStream_t> (stream_v));
  int rc = run_device_inner (e, d_iq, nbytes, first_sample, d_halo, stream_v);
  if (e->dc_power_unqueued) { const int prc = queue_dc_power (e, static_cast<hipStream_t> (stream_v)); if (rc == MODES_OK) rc = prc; }   // (a path that queued no scan of its own)
  if (e->dc_power_pending) {               // the block powers of a filter_dc run: formed beside the scan, done when the caller's stream is
    e->dc_power_pending = false;
    HIP_TRY (hipStreamWaitEvent (static_cast<hipStream_t> (stream_v), e->dc_power_done, 0));
  }
  if (rc == MODES_OK && e->mode == 2400 && e->cfg.mode_ac)
    rc = run_mode_ac_2400 (e, d_iq, nbytes, first_sample, d_halo, static_cast<hipStream_t> (stream_v));
  return rc;
}

// The scan's records do not depend on the ICAO cache: a shard whose entry cache changed (the hand-over rounds of a
// sharded decode) only repeats the ordered pass and the emit.
int modes_engine_rerun_accept (modes_engine *e, void *stream_v)
{
  if (!e) return fail (MODES_E_ARG, "null engine");
  if (!e->have_scan || e->cfg.mode_ac)
    return fail (MODES_E_ARG, "modes_engine_rerun_accept: no scan to repeat the accept pass on (needs a successful 2.0 / 2.4 MS/s run "
                              "with the ordered pass on the device and without mode_ac)");
  HIP_TRY (hipSetDevice (e->cfg.device));
  hipStream_t stream = static_cast<hipStream_t> (stream_v);
  const uint64_t retries = e->stats.scan_retries, cand = e->stats.candidates;
  e->messages.clear ();
  e->in_pinned = false; e->pin_n = 0;
  memset (&e->stats, 0, sizeof e->stats);
  e->stats.scan_retries = retries;
  (void) cand;
  return run_accept_on_device (e, e->last_sp, e->last_first_sample, stream, false);
}

int modes_engine_can_rerun_accept (const modes_engine *e) { return e && e->have_scan && !e->cfg.mode_ac; }

int modes_engine_d8_chain_map (modes_engine *e, const void *d_iq, size_t nbytes, uint64_t first_sample, const uint16_t *d_halo,
                               void *stream_v, int32_t *map)
{
  if (!e || !map || (!d_iq && nbytes)) return fail (MODES_E_ARG, "null argument");
  if (e->mode != 8000) return fail (MODES_E_ARG, "modes_engine_d8_chain_map: not an 8 MS/s engine");
  if (nbytes / bytes_per_sample (e->cfg.input_format) < 8u * kD8Win)
    return fail (MODES_E_ARG, "modes_engine_d8_chain_map: the shard is shorter than a match window");
  return run_8000 (e, d_iq, nbytes, first_sample, d_halo, static_cast<hipStream_t> (stream_v), map);
}

int modes_engine_d8_last_ambiguous (const modes_engine *e, uint64_t *step_plus_1)
{
  if (!e || !step_plus_1) return fail (MODES_E_ARG, "null argument");
  if (e->mode != 8000) return fail (MODES_E_ARG, "modes_engine_d8_last_ambiguous: not an 8 MS/s engine");
  *step_plus_1 = e->d8_limit;
  return MODES_OK;
}

int modes_engine_d8_reuse_planes (modes_engine *e)
{
  if (!e) return fail (MODES_E_ARG, "null engine");
  if (e->mode != 8000) return fail (MODES_E_ARG, "modes_engine_d8_reuse_planes: not an 8 MS/s engine");
  e->d8_reuse = true;
  return MODES_OK;
}

int modes_engine_set_d8_state (modes_engine *e, const modes_d8_state *in)
{
  if (!e) return fail (MODES_E_ARG, "null engine");
  if (e->mode != 8000) return fail (MODES_E_ARG, "modes_engine_set_d8_state: not an 8 MS/s engine");
  if (!in) { e->d8_in_set = false; return MODES_OK; }
  if (in->pending >= 0 && in->pending != 1) return fail (MODES_E_ARG, "modes_d8_state.pending must be < 0, or 1 for none");
  e->d8_in = *in;
  e->d8_in_set = true;
  return MODES_OK;
}

int modes_engine_get_d8_state (modes_engine *e, modes_d8_state *out, int want_acc, void *stream_v)
{
  if (!e || !out) return fail (MODES_E_ARG, "null argument");
  if (e->mode != 8000 || !e->d8_have_run) return fail (MODES_E_ARG, "modes_engine_get_d8_state: no 8 MS/s run to read the state of");
  hipStream_t stream = static_cast<hipStream_t> (stream_v);
  if (want_acc && !e->d8_out_acc_valid && !e->d8_passthrough) {
    // d8m_phase_av_acc behind the last step: the recurrence over every step of the shard (ref: src/demod-8000.c:275-288)
    HIP_TRY (hipSetDevice (e->cfg.device));
    D8Params q = e->d8_q;
    int rc;
    if ((rc = e->d8_p0.reserve (size_t (q.nsteps) + 64)) != MODES_OK) return rc;
    if ((rc = e->d8_mx.reserve (size_t (q.nsteps) + 64)) != MODES_OK) return rc;
    q.p0 = e->d8_p0.p; q.mx = e->d8_mx.p; q.write_arrays = 1;
    // d8_trig_kernel counts ambiguous steps into amb_count and rewrites the trigger words without the decisions the
    // running average added: both are restored by the acc kernel walking the whole shard (it ORs the same bits back)
    // While the planes are being rewritten they are nobody's to reuse (modes_engine_d8_reuse_planes): a failure on the
    // way leaves them marked invalid, and the next run forms them again.
    const bool planes_were_valid = e->d8_planes_valid;
    e->d8_planes_valid = false;
    if (e->d8_limit < q.nsteps) {
      // the relaunch counts the ambiguous steps again: start the count from zero, so a run that reuses the planes
      // reports them once (modes_d8_state.amb_steps)
      HIP_TRY (hipMemsetAsync (q.amb_count, 0, sizeof (uint32_t), stream));
      HIP_TRY (launch_d8_trig (q, 0, stream));
    }
    int32_t *d_acc = reinterpret_cast<int32_t *> (e->d8_small.p + 7);
    HIP_TRY (launch_d8_acc (q, q.nsteps, d_acc, stream));
    int32_t acc = 0;
    HIP_TRY (hipMemcpyAsync (&acc, d_acc, sizeof acc, hipMemcpyDeviceToHost, stream));
    HIP_TRY (hipStreamSynchronize (stream));
    e->d8_planes_valid = planes_were_valid;
    e->d8_out.acc = acc; e->d8_out.acc_valid = 1;
    e->d8_out_acc_valid = true;
  }
  *out = e->d8_out;
  return MODES_OK;
}

int modes_engine_get_counters (const modes_engine *e, uint64_t *scan_launches, uint64_t *accept_passes)
{
  if (!e) return fail (MODES_E_ARG, "null engine");
  if (scan_launches) *scan_launches = e->scan_launches;
  if (accept_passes) *accept_passes = e->accept_passes;
  return MODES_OK;
}

// ref: src/demod-2400.c:498-505,524-530 (only demod_2400 keeps these)
int modes_engine_get_power_stats (modes_engine *e, modes_power_stats *out)
{
  if (!e || !out) return fail (MODES_E_ARG, "null argument");
  memset (out, 0, sizeof *out);
  if (e->mode != 2400) return MODES_OK;
  const size_t n = modes_engine_num_messages (e);
  const modes_message *m = modes_engine_messages (e);
  const uint32_t B = e->block_samples, blk0 = uint32_t (e->last_first_sample / B);
  std::vector<double> blk_signal;                 // sum of the signal powers of a block's messages, in message order
  for (size_t i = 0; i < n; i++) {
    if (m[i].msg_type == 32) continue;            // Mode A/C replies come from demod_2400_AC
    const int slen = int (m[i].msg_bits) * 12 / 5;
    const double signal_power = m[i].sig_level * slen;
    out->signal_power_sum += signal_power;
    out->signal_power_count += uint64_t (slen);
    if (m[i].sig_level > out->peak_signal_power) out->peak_signal_power = m[i].sig_level;
    if (m[i].sig_level > 0.50119) out->strong_signal_count++;
    const size_t b = m[i].block >= blk0 ? m[i].block - blk0 : 0;
    if (blk_signal.size () <= b) blk_signal.resize (b + 1, 0.0);
    blk_signal[b] += signal_power;
  }
  if (e->cfg.compute_power && e->nblocks_power) {     // Modes.measure_noise: mean_power of every block was formed with the run
    std::vector<double> pw (e->nblocks_power);
    HIP_TRY (hipSetDevice (e->cfg.device));
    if (e->dc_power_done) HIP_TRY (hipEventSynchronize (e->dc_power_done));
    HIP_TRY (hipMemcpy (pw.data (), e->block_power.p, pw.size () * sizeof (double), hipMemcpyDeviceToHost));
    for (size_t b = 0; b < pw.size (); b++) {
      out->noise_power_sum += pw[b] - (b < blk_signal.size () ? blk_signal[b] : 0.0);
      out->noise_power_count += uint64_t (B) + e->ov;                       // mag->total_length: the buffer's capacity
    }
    out->noise_valid = 1;
  }
  return MODES_OK;
}

int modes_engine_run_host (modes_engine *e, const void *iq, size_t nbytes, uint64_t first_sample)
{
  if (!e || (!iq && nbytes)) return fail (MODES_E_ARG, "null argument");
  HIP_TRY (hipSetDevice (e->cfg.device));
  int rc = e->staging.reserve (nbytes + 64);
  if (rc != MODES_OK) return rc;
  HIP_TRY (hipMemcpy (e->staging.p, iq, nbytes, hipMemcpyHostToDevice));
  return modes_engine_run_device (e, e->staging.p, nbytes, first_sample, nullptr, nullptr);
}

int modes_engine_time_scan (modes_engine *e, const void *d_iq, size_t nbytes, int reps, void *stream_v, float *ms_out)
{
  if (!e || !d_iq || reps <= 0 || !ms_out) return fail (MODES_E_ARG, "bad argument");
  hipStream_t stream = static_cast<hipStream_t> (stream_v);
  HIP_TRY (hipSetDevice (e->cfg.device));
  ScanParams sp;
  int rc = fill_scan_params (e, d_iq, nbytes, nullptr, &sp);
  if (rc != MODES_OK) return rc;
  if (const char *ds = getenv ("MODES_DEBUG_STOP")) sp.debug_stop = atoi (ds);
  if (e->cfg.filter_dc) {                     // the scan over the magnitudes the last run's DC chain left (same recording)
    const uint64_t n = nbytes / bytes_per_sample (e->cfg.input_format);
    if (e->dc_mag.cap < size_t (n) + e->ov + 1024) return fail (MODES_E_ARG, "time_scan with filter_dc: decode the recording first");
    sp.iq = reinterpret_cast<const uint8_t *> (e->dc_mag.p + 8); sp.conv = kConvMagU16; sp.navail = n + e->ov; sp.halo = nullptr;
  }
  hipEvent_t a, b;
  HIP_TRY (hipEventCreate (&a));
  HIP_TRY (hipEventCreate (&b));
  HIP_TRY (hipEventRecord (a, stream));
  if (!e->cfg.host_replay && !e->cfg.phase_enhance) {
    if ((rc = reserve_accept_buffers (e, sp)) != MODES_OK) return rc;
    HIP_TRY (hipMemsetAsync (e->rb.p + kRbInexactWord, 0, 2 * sizeof (uint32_t), stream));
  }
  const ScanParams sv = scan_variant (e, sp);                      // the instance a run would launch
  for (int i = 0; i < reps; i++) HIP_TRY (e->mode == 2000 ? launch_scan2000 (sp, stream) : launch_scan2400 (sv, stream));
  HIP_TRY (hipEventRecord (b, stream));
  HIP_TRY (hipEventSynchronize (b));
  float ms = 0;
  HIP_TRY (hipEventElapsedTime (&ms, a, b));
  (void) hipEventDestroy (a);
  (void) hipEventDestroy (b);
  *ms_out = ms / float (reps);
  return MODES_OK;
}

size_t modes_engine_num_messages (const modes_engine *e) { return !e ? 0 : e->in_pinned ? e->pin_n : e->messages.size (); }
const modes_message *modes_engine_messages (const modes_engine *e)
{
  if (!e) return nullptr;
  if (e->in_pinned) return e->pin_n ? e->pin_msgs : nullptr;
  return e->messages.empty () ? nullptr : e->messages.data ();
}

int modes_engine_format_raw (modes_engine *e, int upper_case, const char **text, size_t *len)
{
  if (!e || !text || !len) return fail (MODES_E_ARG, "null argument");
  *text = ""; *len = 0;
  const size_t n = modes_engine_num_messages (e);
  if (!n) return MODES_OK;
  if (n > 0xFFFFFFFFull) return fail (MODES_E_ARG, "too many messages");
  HIP_TRY (hipSetDevice (e->cfg.device));
  int rc;
  if ((rc = e->d_messages.reserve (n)) != MODES_OK) return rc;
  if (!e->in_pinned)                       // the host-ordered paths hold their messages in host memory only
    HIP_TRY (hipMemcpy (e->d_messages.p, e->messages.data (), n * sizeof (modes_message), hipMemcpyHostToDevice));
  const uint32_t nchunks = uint32_t ((n + 1023) / 1024);
  if ((rc = e->raw_len.reserve (n)) != MODES_OK) return rc;
  if ((rc = e->raw_off.reserve (n)) != MODES_OK) return rc;
  if ((rc = e->raw_cs.reserve (nchunks + 1)) != MODES_OK) return rc;
  if ((rc = e->raw_cv.reserve (nchunks + 1)) != MODES_OK) return rc;
  if ((rc = e->raw_co.reserve (nchunks + 1)) != MODES_OK) return rc;
  uint64_t *d_totals = reinterpret_cast<uint64_t *> (e->d_stats) + 24;
  HIP_TRY (launch_raw_format (e->d_messages.p, uint32_t (n), e->raw_len.p, e->raw_cs.p, e->raw_cv.p, e->raw_co.p, d_totals,
                              e->raw_off.p, upper_case, nullptr, 0, nullptr));
  unsigned long long total = 0;
  HIP_TRY (hipMemcpy (&total, d_totals, sizeof total, hipMemcpyDeviceToHost));
  if ((rc = e->raw_text.reserve (size_t (total) + 1)) != MODES_OK) return rc;
  if (total + 1 > e->pin_text_cap) {
    if (e->pin_text) (void) hipHostFree (e->pin_text);
    e->pin_text = nullptr; e->pin_text_cap = 0;
    const size_t want = size_t (total) + size_t (total) / 4 + 4096;
    HIP_TRY (hipHostMalloc (reinterpret_cast<void **> (&e->pin_text), want, hipHostMallocDefault));
    e->pin_text_cap = want;
  }
  HIP_TRY (launch_raw_format (e->d_messages.p, uint32_t (n), e->raw_len.p, e->raw_cs.p, e->raw_cv.p, e->raw_co.p, d_totals,
                              e->raw_off.p, upper_case, e->raw_text.p, 1, nullptr));
  HIP_TRY (hipMemcpy (e->pin_text, e->raw_text.p, size_t (total), hipMemcpyDeviceToHost));
  e->pin_text[total] = 0;
  *text = e->pin_text; *len = size_t (total);
  return MODES_OK;
}

int modes_engine_decode_fields (modes_engine *e, const modes_fields **out, size_t *n_out, void *stream_v)
{
  if (!e || !out || !n_out) return fail (MODES_E_ARG, "null argument");
  *out = nullptr; *n_out = 0;
  const size_t n = modes_engine_num_messages (e);
  if (!n) return MODES_OK;
  if (n > 0xFFFFFFFFull) return fail (MODES_E_ARG, "too many messages");
  HIP_TRY (hipSetDevice (e->cfg.device));
  hipStream_t stream = static_cast<hipStream_t> (stream_v);
  int rc;
  if ((rc = e->d_messages.reserve (n)) != MODES_OK) return rc;
  if ((rc = e->d_fields.reserve (n)) != MODES_OK) return rc;
  if ((rc = e->pin_fields.reserve (n)) != MODES_OK) return rc;
  if (!e->in_pinned)                       // the host-ordered paths hold their messages in host memory only
    HIP_TRY (hipMemcpyAsync (e->d_messages.p, e->messages.data (), n * sizeof (modes_message), hipMemcpyHostToDevice, stream));
  HIP_TRY (launch_fields (e->d_messages.p, uint32_t (n), e->d_fields.p, stream));
  HIP_TRY (hipMemcpyAsync (e->pin_fields.p, e->d_fields.p, n * sizeof (modes_fields), hipMemcpyDeviceToHost, stream));
  HIP_TRY (hipStreamSynchronize (stream));
  *out = e->pin_fields.p; *n_out = n;
  return MODES_OK;
}

// the same kernel on a caller's list (messages that did not come from a run: RAW / SBS input, replays)
int modes_decode_fields_host (const modes_message *msgs, size_t n, modes_fields *out)
{
  if ((!msgs || !out) && n) return fail (MODES_E_ARG, "null argument");
  if (!n) return MODES_OK;
  if (n > 0xFFFFFFFFull) return fail (MODES_E_ARG, "too many messages");
  modes_message *d_in = nullptr;
  modes_fields *d_out = nullptr;
  hipError_t e1 = hipMalloc (reinterpret_cast<void **> (&d_in), n * sizeof (modes_message));
  hipError_t e2 = e1 == hipSuccess ? hipMalloc (reinterpret_cast<void **> (&d_out), n * sizeof (modes_fields)) : e1;
  int rc = MODES_OK;
  if (e2 != hipSuccess) rc = fail (MODES_E_NOMEM, "hipMalloc: %s", hipGetErrorString (e2));
  else {
    hipError_t e = hipMemcpy (d_in, msgs, n * sizeof (modes_message), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_fields (d_in, uint32_t (n), d_out, nullptr);
    if (e == hipSuccess) e = hipMemcpy (out, d_out, n * sizeof (modes_fields), hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail (MODES_E_HIP, "modes_decode_fields_host: %s", hipGetErrorString (e));
  }
  if (d_in) (void) hipFree (d_in);
  if (d_out) (void) hipFree (d_out);
  return rc;
}

int modes_engine_get_stats (const modes_engine *e, modes_stats *out)
{
  if (!e || !out) return fail (MODES_E_ARG, "null argument");
  *out = e->stats;
  return MODES_OK;
}

int modes_engine_kernel_ms (const modes_engine *e, float *main_ms, float *second_ms)
{
  if (!e) return fail (MODES_E_ARG, "null engine");
  if (main_ms) *main_ms = e->kernel_ms[0];
  if (second_ms) *second_ms = e->kernel_ms[1];
  return MODES_OK;
}

size_t modes_engine_block_power (modes_engine *e, double *out, size_t max)
{
  if (!e) return 0;
  const size_t n = e->nblocks_power;
  if (out && n) {
    if (hipSetDevice (e->cfg.device) != hipSuccess) return 0;
    if (e->dc_power_done && hipEventSynchronize (e->dc_power_done) != hipSuccess) return 0;
    if (hipMemcpy (out, e->block_power.p, (n < max ? n : max) * sizeof (double), hipMemcpyDeviceToHost) != hipSuccess) return 0;
  }
  return n;
}

const uint16_t *modes_engine_tail_halo (const modes_engine *e) { return e ? e->tail.p + e->tail_half * ((e->ov + 7u) & ~7u) : nullptr; }

int modes_engine_set_dc_state (modes_engine *e, float z1_I, float z1_Q)
{
  if (!e) return fail (MODES_E_ARG, "null engine");
  e->dc_z[0] = z1_I; e->dc_z[1] = z1_Q;
  return MODES_OK;
}

int modes_engine_get_dc_state (const modes_engine *e, float *z1_I, float *z1_Q)
{
  if (!e || !z1_I || !z1_Q) return fail (MODES_E_ARG, "null argument");
  *z1_I = e->dc_z[0]; *z1_Q = e->dc_z[1];
  return MODES_OK;
}

int modes_engine_set_dc_lead (modes_engine *e, uint64_t lead_samples, uint64_t warm_samples)
{
  if (!e) return fail (MODES_E_ARG, "null engine");
  e->dc_lead = lead_samples;
  e->dc_warm = warm_samples;
  return MODES_OK;
}

int modes_engine_get_dc_info (const modes_engine *e, modes_dc_info *out)
{
  if (!e || !out) return fail (MODES_E_ARG, "null argument");
  *out = e->dc_info;
  return MODES_OK;
}

const uint16_t *modes_engine_dc_magnitudes (const modes_engine *e) { return e && e->dc_mag.p ? e->dc_mag.p + 8 : nullptr; }

int modes_engine_read_dc_magnitudes (modes_engine *e, uint16_t *out, size_t first, size_t count)
{
  if (!e || !out || !e->dc_mag.p) return fail (MODES_E_ARG, "no filter_dc run to read from");
  if (first + count + 8 > e->dc_mag.cap) return fail (MODES_E_ARG, "range past the magnitude buffer");
  HIP_TRY (hipSetDevice (e->cfg.device));
  HIP_TRY (hipMemcpy (out, e->dc_mag.p + 8 + first, count * sizeof (uint16_t), hipMemcpyDeviceToHost));
  return MODES_OK;
}

int modes_engine_halo_at (modes_engine *e, const void *d_iq, size_t nbytes, uint64_t at_sample, uint16_t *d_out, void *stream_v)
{
  if (!e || !d_iq || !d_out) return fail (MODES_E_ARG, "null argument");
  HIP_TRY (hipSetDevice (e->cfg.device));
  ScanParams sp;
  memset (&sp, 0, sizeof sp);
  sp.iq = static_cast<const uint8_t *> (d_iq);
  sp.nsamples = nbytes / bytes_per_sample (e->cfg.input_format);
  sp.ov = e->ov; sp.block_samples = e->block_samples; sp.conv = e->conv; sp.lut256 = e->tabs.lut256;
  HIP_TRY (launch_tail (sp, at_sample, d_out, static_cast<hipStream_t> (stream_v)));
  return MODES_OK;
}

// Stand-alone converter on device buffers.  The tables and the scratch of this call are kept per device; calls are
// serialised by a mutex (the scratch is shared), each call waits for its results.
namespace {
struct ConvDev {
  bool ready = false;
  uint16_t *d_lut = nullptr;
  unsigned long long *d_power = nullptr;
  float *d_small = nullptr;                   // [0..1] z_final, [2] power
  float *d_f[3] = { nullptr, nullptr, nullptr };
  size_t f_cap = 0;
};
constexpr int kMaxConvDevices = 64;
std::mutex g_conv_mu;
ConvDev g_conv[kMaxConvDevices];
}  // namespace

int modes_convert_device (int conv, const void *d_iq, uint16_t *d_mag, size_t nsamples, convert_state *state,
                          double *out_power, void *stream_v)
{
  hipStream_t stream = static_cast<hipStream_t> (stream_v);
  if (conv < 0 || conv > 5) return fail (MODES_E_ARG, "converter id %d", conv);
  if ((!d_iq || !d_mag) && nsamples) return fail (MODES_E_ARG, "null argument");
  int dev = 0;
  HIP_TRY (hipGetDevice (&dev));
  if (dev < 0 || dev >= kMaxConvDevices) return fail (MODES_E_ARG, "device %d", dev);
  std::lock_guard<std::mutex> lk (g_conv_mu);
  ConvDev &c = g_conv[dev];
  if (!c.ready) {
    uint16_t lut[256];
    build_uc8_byte_lut (lut);
    if (!c.d_lut) HIP_TRY (hipMalloc (reinterpret_cast<void **> (&c.d_lut), sizeof lut));
    HIP_TRY (hipMemcpy (c.d_lut, lut, sizeof lut, hipMemcpyHostToDevice));
    if (!c.d_power) HIP_TRY (hipMalloc (reinterpret_cast<void **> (&c.d_power), sizeof (unsigned long long)));
    if (!c.d_small) HIP_TRY (hipMalloc (reinterpret_cast<void **> (&c.d_small), 4 * sizeof (float)));
    CrcTables t;
    float uc8f[256];
    build_uc8_float_lut (uc8f);
    HIP_TRY (upload_constants (t.single_bit (), uc8f));       // __constant__ tables of the device that is current
    c.ready = true;
  }
  if (out_power) *out_power = 0.0;
  if (nsamples == 0) return MODES_OK;
  const bool is_float = conv == kConvUc8Generic || conv == kConvSc16Generic || conv == kConvSc16Q11Generic;
  if (!is_float) {                            // ref: src/convert.c:13-110,247-292 (stateless; state is not touched)
    HIP_TRY (hipMemsetAsync (c.d_power, 0, sizeof (unsigned long long), stream));
    HIP_TRY (launch_convert (conv, static_cast<const uint8_t *> (d_iq), d_mag, nsamples, c.d_lut, c.d_power, stream));
    unsigned long long pw = 0;
    HIP_TRY (hipMemcpyAsync (&pw, c.d_power, sizeof pw, hipMemcpyDeviceToHost, stream));
    HIP_TRY (hipStreamSynchronize (stream));
    if (out_power) *out_power = conv == kConvUc8LutPower ? double (pw) / 65535.0 / 65535.0 : 0.0;
    return MODES_OK;
  }
  // float converters (ref: src/convert.c:112-245): DC block state in and out through `state` (NULL = DC block off)
  const float A = state ? state->DC_A : 0.0f, B = state ? state->DC_B : 1.0f;
  const bool dc = !(A == 0.0f && B == 1.0f);
  if (!dc && state && (state->z1_I != 0.0f || state->z1_Q != 0.0f))
    return fail (MODES_E_ARG, "DC block off with a non-zero filter state: convert_init() never produces that");
  if (nsamples > c.f_cap) {
    for (float *&p : c.d_f) { if (p) (void) hipFree (p); p = nullptr; }
    c.f_cap = 0;
    for (float *&p : c.d_f) HIP_TRY (hipMalloc (reinterpret_cast<void **> (&p), (nsamples + 64) * sizeof (float)));
    c.f_cap = nsamples;
  }
  HIP_TRY (launch_convert_float (conv, static_cast<const uint8_t *> (d_iq), d_mag, nsamples, A, B, state ? state->z1_I : 0.0f,
                                 state ? state->z1_Q : 0.0f, c.d_f[0], c.d_f[1], c.d_f[2], c.d_small,
                                 out_power ? c.d_small + 2 : nullptr, stream));
  float small[3] = { 0, 0, 0 };
  HIP_TRY (hipMemcpyAsync (small, c.d_small, sizeof small, hipMemcpyDeviceToHost, stream));
  HIP_TRY (hipStreamSynchronize (stream));
  if (dc && state) { state->z1_I = small[0]; state->z1_Q = small[1]; }
  if (out_power) *out_power = double (small[2]);
  return MODES_OK;
}

}  // extern "C"
