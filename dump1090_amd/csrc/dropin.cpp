// Part 1 + Part 2 of include/modes_hip.h: the reference's own entry points for the path
// (convert.h, demod.h, crc.h), served by the HIP kernels one mag_buf / one converter call at a time.
//
// These calls take host buffers, as the reference's callers hand them over (ref: src/infile.c:168,183,
// src/dump1090.c:1281,1397), so each call pays one H2D and one D2H copy; the bulk engine (engine.cpp)
// is the throughput path.  No CPU fallback: a HIP failure is reported through modes_last_error() and on
// stderr, and the call produces nothing.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/modes_hip.h"
#include "accept.h"
#include "common.h"
#include "kernels.h"
#include "tables.h"

using namespace modes;

namespace modes { int set_error (int code, const char *fmt, ...); }

namespace {

struct DropIn {
  std::mutex mu;
  bool bound = false;
  modes_hip_host host;
  bool ready = false;
  int device = 0;
  // tables
  CrcTables crc;
  int crc_fix_bits = -1;          // what crc_init() was called with; -1 = not yet
  ErrEntry *d_short = nullptr, *d_long = nullptr;
  uint32_t *d_bloom_short = nullptr, *d_bloom_long = nullptr;
  int short_n = 0, long_n = 0, dev_fix_bits = -2;
  uint16_t *d_lut = nullptr;
  uint16_t *d_log10 = nullptr;
  uint32_t *d_crc_table = nullptr;
  uint32_t *d_noise = nullptr;
  // scratch
  uint8_t *d_in = nullptr;   size_t in_cap = 0;
  uint16_t *d_mag = nullptr; size_t mag_cap = 0;
  float *d_f[3] = { nullptr, nullptr, nullptr }; size_t f_cap = 0;
  float *d_small = nullptr;                                   // [0..1] z_final, [2] power
  unsigned long long *d_power = nullptr;
  Record *d_rec = nullptr;   size_t rec_cap = 0;
  uint32_t *d_counts = nullptr, *d_vp = nullptr; size_t strip_cap = 0;
  uint64_t *d_bitmap = nullptr;
  // built-in accept path (no callbacks bound)
  IcaoCache cache;
  std::vector<modes_message> queue;
  uint32_t block_counter = 0;
  bool d8_allocated = false;
  uint64_t scan_retries = 0;      // scans repeated with more candidate slots per strip
  // demod_8000: what the reference keeps in file-static variables between calls (ref: src/demod-8000.c:80-89)
  struct D8 {
    int32_t *dprev = nullptr, *dnext = nullptr;     // the last 1928 differences (d8m_dbuf's overlap)
    int32_t acc = 0;                                // d8m_phase_av_acc
    long long pending = 1;                          // step (< 0) at which the window still open was opened; 1 = none
    unsigned long long trig_tail = 0;               // trigger decisions of the last 64 steps (re-scanned after a window)
    // scratch
    uint64_t *trig = nullptr, *amb = nullptr, *off = nullptr; size_t words = 0;
    uint32_t *p0 = nullptr, *mx = nullptr; size_t steps = 0;
    uint8_t *table = nullptr, *entry = nullptr; int64_t *win_start = nullptr; uint32_t *win_count = nullptr, *chunk_rec = nullptr;
    D8Window *windows = nullptr, *win_packed = nullptr; Record *records = nullptr, *rec_packed = nullptr; size_t chunks = 0;
    long long *small = nullptr;
  } d8;
};

DropIn g;

#define HIP_OK(expr)                                                                                     \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) {                                                                              \
      set_error (MODES_E_HIP, "%s: %s", #expr, hipGetErrorString (e_));                                  \
      fprintf (stderr, "modes_hip: %s\n", modes_last_error ());                                          \
      return false;                                                                                      \
    }                                                                                                    \
  } while (0)

template <typename T>
bool grow (T *&p, size_t &cap, size_t n)
{
  if (n <= cap) return true;
  if (p) (void) hipFree (p);
  p = nullptr; cap = 0;
  HIP_OK (hipMalloc (reinterpret_cast<void **> (&p), n * sizeof (T)));
  cap = n;
  return true;
}

bool ensure_ready ()
{
  if (g.ready) return true;
  int ndev = 0;
  if (hipGetDeviceCount (&ndev) != hipSuccess || ndev <= 0) {
    set_error (MODES_E_NODEVICE, "no HIP device");
    fprintf (stderr, "modes_hip: no HIP device: the drop-in converters/demodulators cannot run\n");
    return false;
  }
  g.device = g.bound ? g.host.device : 0;
  HIP_OK (hipSetDevice (g.device));
  float uc8f[256];
  build_uc8_float_lut (uc8f);
  HIP_OK (upload_constants (g.crc.single_bit (), uc8f));
  uint16_t lut[256];
  build_uc8_byte_lut (lut);
  HIP_OK (hipMalloc (reinterpret_cast<void **> (&g.d_lut), sizeof lut));
  HIP_OK (hipMemcpy (g.d_lut, lut, sizeof lut, hipMemcpyHostToDevice));
  {
    std::vector<uint16_t> l10 (65536);
    build_log10_lut (l10.data ());
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&g.d_log10), l10.size () * sizeof (uint16_t)));
    HIP_OK (hipMemcpy (g.d_log10, l10.data (), l10.size () * sizeof (uint16_t), hipMemcpyHostToDevice));
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&g.d_crc_table), 256 * sizeof (uint32_t)));
    HIP_OK (hipMemcpy (g.d_crc_table, g.crc.byte_table (), 256 * sizeof (uint32_t), hipMemcpyHostToDevice));
  }
  HIP_OK (hipMalloc (reinterpret_cast<void **> (&g.d_small), 4 * sizeof (float)));
  HIP_OK (hipMalloc (reinterpret_cast<void **> (&g.d_power), sizeof (unsigned long long)));
  g.ready = true;
  return true;
}

int effective_fix_bits ()
{
  if (g.crc_fix_bits >= 0) return g.crc_fix_bits;
  return g.bound ? g.host.fix_bits : 1;          // ref default: error_correct_1 on, _2 off (src/dump1090.c:595-596)
}

bool ensure_tables ()
{
  const int fb = effective_fix_bits ();
  if (g.crc.fix_bits () != fb) g.crc.build (fb);
  if (g.dev_fix_bits == fb) return true;
  for (void *p : { (void *) g.d_short, (void *) g.d_long, (void *) g.d_bloom_short, (void *) g.d_bloom_long })
    if (p) (void) hipFree (p);
  g.d_short = g.d_long = nullptr; g.d_bloom_short = g.d_bloom_long = nullptr;
  for (int bits : { kShortBits, kLongBits }) {
    std::vector<ErrEntry> d = g.crc.device_table (bits);
    ErrEntry *p = nullptr;
    if (!d.empty ()) {
      HIP_OK (hipMalloc (reinterpret_cast<void **> (&p), d.size () * sizeof (ErrEntry)));
      HIP_OK (hipMemcpy (p, d.data (), d.size () * sizeof (ErrEntry), hipMemcpyHostToDevice));
    }
    std::vector<uint32_t> bloom (2048, 0);
    for (const ErrEntry &x : d) { const uint32_t h = (x.syndrome ^ (x.syndrome >> 12)) & 0xFFFF; bloom[h >> 5] |= 1u << (h & 31); }
    uint32_t *pb = nullptr;
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&pb), bloom.size () * sizeof (uint32_t)));
    HIP_OK (hipMemcpy (pb, bloom.data (), bloom.size () * sizeof (uint32_t), hipMemcpyHostToDevice));
    if (bits == kShortBits) { g.d_short = p; g.short_n = int (d.size ()); g.d_bloom_short = pb; }
    else                    { g.d_long = p;  g.long_n = int (d.size ());  g.d_bloom_long = pb; }
  }
  g.dev_fix_bits = fb;
  return true;
}

// ---- converters ------------------------------------------------------------------------

const char *const kFuncName[6] = { "convert_uc8_nodc_nopower", "convert_uc8_nodc_power", "convert_uc8_generic",
                                   "convert_sc16_nodc_nopower", "convert_sc16_generic", "convert_sc16q11_generic" };
const char *const kDescription[6] = { "UC8, integer/table path", "UC8, integer/table path, with power measurement",
                                      "UC8, float path", "SC16, integer path", "SC16, float path", "SC16Q11, float path" };

bool run_converter (int conv, const void *iq, uint16_t *mag, unsigned n, convert_state *st, double *out_power)
{
  std::lock_guard<std::mutex> lk (g.mu);
  if (!ensure_ready ()) return false;
  HIP_OK (hipSetDevice (g.device));
  if (n == 0) { if (out_power) *out_power = 0.0; return true; }
  const bool lut = conv == kConvUc8Lut || conv == kConvUc8LutPower;
  const size_t in_bytes = lut ? n : size_t (n) * (conv == kConvUc8Generic ? 2 : 4);   // as written the LUT path reads n bytes
  if (!grow (g.d_in, g.in_cap, in_bytes + 64) || !grow (g.d_mag, g.mag_cap, size_t (n) + 64)) return false;
  HIP_OK (hipMemcpyAsync (g.d_in, iq, in_bytes, hipMemcpyHostToDevice, nullptr));
  if (lut || conv == kConvSc16Int) {
    HIP_OK (hipMemsetAsync (g.d_power, 0, sizeof (unsigned long long), nullptr));
    HIP_OK (launch_convert (conv, g.d_in, g.d_mag, n, g.d_lut, g.d_power, nullptr));
    unsigned long long pw = 0;
    HIP_OK (hipMemcpyAsync (&pw, g.d_power, sizeof pw, hipMemcpyDeviceToHost, nullptr));
    HIP_OK (hipMemcpyAsync (mag, g.d_mag, size_t (n) * 2, hipMemcpyDeviceToHost, nullptr));
    HIP_OK (hipStreamSynchronize (nullptr));
    if (out_power) *out_power = conv == kConvUc8LutPower ? double (pw) / 65535.0 / 65535.0 : 0.0;
    return true;
  }
  // float path (ref: src/convert.c:112-245)
  if (n > g.f_cap) {
    for (float *&p : g.d_f) { if (p) (void) hipFree (p); p = nullptr; }
    g.f_cap = 0;
    for (float *&p : g.d_f) HIP_OK (hipMalloc (reinterpret_cast<void **> (&p), (size_t (n) + 64) * sizeof (float)));
    g.f_cap = n;
  }
  const bool dc = !(st->DC_A == 0.0f && st->DC_B == 1.0f);
  HIP_OK (launch_convert_float (conv, g.d_in, g.d_mag, n, st->DC_A, st->DC_B, st->z1_I, st->z1_Q, g.d_f[0], g.d_f[1],
                                g.d_f[2], g.d_small, out_power ? g.d_small + 2 : nullptr, nullptr));
  float small[3] = { 0, 0, 0 };
  HIP_OK (hipMemcpyAsync (small, g.d_small, sizeof small, hipMemcpyDeviceToHost, nullptr));
  HIP_OK (hipMemcpyAsync (mag, g.d_mag, size_t (n) * 2, hipMemcpyDeviceToHost, nullptr));
  HIP_OK (hipStreamSynchronize (nullptr));
  if (dc) { st->z1_I = small[0]; st->z1_Q = small[1]; }
  else {
    // z1 = f*0 + z1*1 keeps its value except that the sign of a zero follows the IEEE sum (ref: src/convert.c:137-138)
    st->z1_I = st->z1_I; st->z1_Q = st->z1_Q;
  }
  if (out_power) *out_power = double (small[2]);
  return true;
}

template <int CONV>
void converter_entry (const void *iq, uint16_t *mag, unsigned n, convert_state *st, double *out_power)
{
  if (!run_converter (CONV, iq, mag, n, st, out_power) && out_power) *out_power = 0.0;
}

const convert_func kEntry[6] = { converter_entry<0>, converter_entry<1>, converter_entry<2>,
                                 converter_entry<3>, converter_entry<4>, converter_entry<5> };

// ---- demod_2400 on one mag_buf ------------------------------------------------------------

struct HostMm {
  std::vector<uint8_t> buf;
  template <typename T> T &at (size_t off) { return *reinterpret_cast<T *> (buf.data () + off); }
};

// mode: 2400 / 2000 = the Mode S scan of that demodulator, 24001 = demod_2400_AC
bool scan_mag_buf (const mag_buf *mag, std::vector<Record> &recs, uint64_t *vp_total, int mode = 2400, uint32_t noise_level = 0)
{
  if (!ensure_ready () || !ensure_tables ()) return false;
  HIP_OK (hipSetDevice (g.device));
  const uint32_t mlen = mag->valid_length - mag->overlap;
  const uint32_t nstrips = (mlen + kStrip - 1) / kStrip;
  const bool pe = g.bound && g.host.phase_enhance;
  uint32_t cap = mode == 2000 ? uint32_t (pe ? kCap2000PE : kCap2000) : pe ? 128 : 32;
  if (const char *ev = getenv ("MODES_STRIP_SLOTS")) if (atoi (ev) > 0) cap = uint32_t (atoi (ev));   // test hook: start small, grow on demand
  const uint32_t cap_max = mode == 2000 ? 3u * kStrip : uint32_t (kStrip);     // one slot per position (and retry kind)
  recs.clear ();
  *vp_total = 0;
  if (nstrips == 0) return true;
  if (!grow (g.d_mag, g.mag_cap, size_t (mag->valid_length) + 64)) return false;
  if (nstrips > g.strip_cap) {
    for (void *p : { (void *) g.d_counts, (void *) g.d_vp, (void *) g.d_bitmap }) if (p) (void) hipFree (p);
    g.d_counts = g.d_vp = nullptr; g.d_bitmap = nullptr; g.strip_cap = 0;
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&g.d_counts), nstrips * sizeof (uint32_t)));
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&g.d_vp), nstrips * sizeof (uint32_t)));
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&g.d_bitmap), size_t (nstrips) * kWords * sizeof (uint64_t)));
    g.strip_cap = nstrips;
  }
  HIP_OK (hipMemcpyAsync (g.d_mag, mag->data, size_t (mag->valid_length) * 2, hipMemcpyHostToDevice, nullptr));
  if (mode == 24001) {
    if (!g.d_noise) HIP_OK (hipMalloc (reinterpret_cast<void **> (&g.d_noise), sizeof (uint32_t)));
    HIP_OK (hipMemcpyAsync (g.d_noise, &noise_level, sizeof noise_level, hipMemcpyHostToDevice, nullptr));
  }
  std::vector<uint32_t> counts (nstrips), vp (nstrips);
  // A strip that holds more candidates than its slots is reported, never cut short: scan again with four times the
  // slots until everything fits (as the bulk engine does), so a dense or saturated block is not lost.
  for (;;) {
    if (!grow (g.d_rec, g.rec_cap, size_t (nstrips) * cap)) return false;
    ScanParams sp;
    memset (&sp, 0, sizeof sp);
    sp.iq = reinterpret_cast<const uint8_t *> (g.d_mag);
    sp.nsamples = mlen;
    sp.navail = mag->valid_length;
    sp.ov = mag->overlap;
    sp.block_samples = mlen ? mlen : 1;
    sp.conv = kConvMagU16;
    sp.phase_enhance = g.bound ? g.host.phase_enhance : 0;
    sp.short_tab = g.d_short; sp.short_n = g.short_n; sp.long_tab = g.d_long; sp.long_n = g.long_n;
    sp.bloom_short = g.d_bloom_short; sp.bloom_long = g.d_bloom_long;
    sp.lut256 = g.d_lut;
    sp.records = g.d_rec; sp.counts = g.d_counts; sp.vp_counts = g.d_vp; sp.cap = cap; sp.nstrips = nstrips;
    sp.write_records = 1;
    sp.vp_bitmap = g.d_bitmap;
    sp.log10_lut = g.d_log10; sp.crc_table = g.d_crc_table;
    sp.mode_ac = g.bound ? g.host.mode_AC : 0;
    if (mode == 24001) { sp.noise_level = g.d_noise; HIP_OK (launch_scanac2400 (sp, nullptr)); }
    else HIP_OK (mode == 2000 ? launch_scan2000 (sp, nullptr) : launch_scan2400 (sp, nullptr));
    HIP_OK (hipMemcpyAsync (counts.data (), g.d_counts, nstrips * sizeof (uint32_t), hipMemcpyDeviceToHost, nullptr));
    HIP_OK (hipMemcpyAsync (vp.data (), g.d_vp, nstrips * sizeof (uint32_t), hipMemcpyDeviceToHost, nullptr));
    HIP_OK (hipStreamSynchronize (nullptr));
    bool overflow = false;
    for (uint32_t s = 0; s < nstrips; s++) overflow = overflow || counts[s] == 0xFFFFFFFFu;
    if (!overflow) break;
    if (cap >= cap_max) {
      set_error (MODES_E_OVERFLOW, "a strip overflowed %u candidate slots", cap);
      fprintf (stderr, "modes_hip: %s\n", modes_last_error ());
      return false;
    }
    cap = cap * 4 < cap_max ? cap * 4 : cap_max;
    g.scan_retries++;
  }
  std::vector<Record> strip (cap);
  for (uint32_t s = 0; s < nstrips; s++) {
    *vp_total += vp[s];
    if (!counts[s]) continue;
    HIP_OK (hipMemcpy (strip.data (), g.d_rec + size_t (s) * cap, counts[s] * sizeof (Record), hipMemcpyDeviceToHost));
    for (uint32_t k = 0; k < counts[s]; k++) {
      Record r = strip[k];
      recs.push_back (r);
      recs.back ().pad[0] = uint8_t (s & 0xFF);          // strip index travels in the padding
      recs.back ().pad[1] = uint8_t (s >> 8);
    }
  }
  return true;
}

}  // namespace

extern "C" {

int modes_hip_bind (const modes_hip_host *host)
{
  std::lock_guard<std::mutex> lk (g.mu);
  if (!host) {      // back to the built-in rules, with the state a fresh process would have
    g.bound = false; memset (&g.host, 0, sizeof g.host);
    g.cache.clear (); g.queue.clear (); g.block_counter = 0;
    return MODES_OK;
  }
  if (host->mm_size && (host->off_sig_level + sizeof (double) > host->mm_size || host->off_timestamp_msg + 8 > host->mm_size))
    return set_error (MODES_E_ARG, "modes_hip_bind: offsets outside modeS_message");
  g.host = *host;
  g.bound = true;
  return MODES_OK;
}

convert_func convert_init (convert_format format, double sample_rate, bool filter_dc, bool compute_power,
                           convert_state **state_p)
{
  // ref: src/convert.c:294-328 -- first table row matching the format and the requested capabilities
  static const struct { int fmt; bool dc, pw; } rows[6] = {
    { INPUT_UC8, false, false }, { INPUT_UC8, false, true }, { INPUT_UC8, true, true },
    { INPUT_SC16, false, false }, { INPUT_SC16, true, true }, { INPUT_SC16Q11, true, true } };
  int i;
  for (i = 0; i < 6; i++) {
    if (rows[i].fmt != int (format)) continue;
    if (filter_dc && !rows[i].dc) continue;
    if (compute_power && !rows[i].pw) continue;
    break;
  }
  if (i == 6) {
    fprintf (stderr, "No suitable converter for format=%d (%s), power=%d, DC=%d\n", int (format),
             convert_format_name (format), int (compute_power), int (filter_dc));
    return nullptr;
  }
  convert_state *st = static_cast<convert_state *> (malloc (sizeof *st));
  if (!st) { fprintf (stderr, "Cannot allocate converter state\n"); return nullptr; }
  st->func_name = kFuncName[i];
  st->description = kDescription[i];
  st->z1_I = st->z1_Q = 0;
  if (filter_dc) {                    // ref: src/convert.c:349-355 -- DC block at 1 Hz
    st->DC_B = float (exp (-2.0 * M_PI * 1.0 / sample_rate));
    st->DC_A = float (1.0 - st->DC_B);
  } else {
    st->DC_B = 1.0f;
    st->DC_A = 0.0f;
  }
  *state_p = st;
  return kEntry[i];
}

void convert_cleanup (convert_state **state_p)
{
  if (*state_p) free (*state_p);
  *state_p = nullptr;
}

const char *convert_format_name (convert_format f)
{
  return f == INPUT_ILLEGAL ? "INPUT_ILLEGAL" : f == INPUT_UC8 ? "INPUT_UC8" : f == INPUT_SC16 ? "INPUT_SC16" :
         f == INPUT_SC16Q11 ? "INPUT_SC16Q11" : "?";
}

// ---- crc.h ---------------------------------------------------------------------------------
void crc_init (int fix_bits)
{
  std::lock_guard<std::mutex> lk (g.mu);
  g.crc_fix_bits = fix_bits;
  g.crc.build (fix_bits);
}

void crc_exit (void)
{
  std::lock_guard<std::mutex> lk (g.mu);
  g.crc_fix_bits = -1;
}

uint32_t crc_checksum (const uint8_t *msg, int bitlen) { return g.crc.checksum (msg, bitlen); }

errorinfo *crc_checksum_diagnose (uint32_t syndrome, int bitlen)
{
  static_assert (sizeof (errorinfo) == sizeof (HostErr), "errorinfo layout");
  if (g.crc.fix_bits () < 0) g.crc.build (effective_fix_bits ());
  return reinterpret_cast<errorinfo *> (const_cast<HostErr *> (g.crc.diagnose (syndrome, bitlen)));
}

void crc_checksum_fix (uint8_t *msg, const errorinfo *info)
{
  for (int i = 0; i < info->errors; i++) msg[info->bit[i] >> 3] ^= uint8_t (1 << (7 - (info->bit[i] & 7)));
}

// ---- demod.h -------------------------------------------------------------------------------
void demod_2400 (const mag_buf *mag)
{
  std::lock_guard<std::mutex> lk (g.mu);
  std::vector<Record> recs;
  uint64_t vp = 0;
  const uint32_t block = g.block_counter++;
  if (!mag || !mag->data || mag->valid_length < mag->overlap) return;
  if (!scan_mag_buf (mag, recs, &vp)) return;

  const bool have_cb = g.bound && g.host.modeS_message_score && g.host.decode_mode_S_message && g.host.mm_size;
  const uint16_t *m = mag->data;
  HostMm mm;
  if (have_cb) mm.buf.assign (g.host.mm_size, 0);           // memset (&mm, 0, sizeof mm) at :197
  uint64_t skip_until = 0;

  for (size_t i = 0; i < recs.size ();) {
    size_t e = i + 1;
    const uint64_t j = (uint64_t (recs[i].pad[0]) | (uint64_t (recs[i].pad[1]) << 8)) * kStrip + recs[i].off;
    while (e < recs.size () && recs[e].off == recs[i].off && recs[e].pad[0] == recs[i].pad[0] && recs[e].pad[1] == recs[i].pad[1]) e++;
    const size_t first = i;
    i = e;
    if (j < skip_until) continue;

    int best = -2; size_t besti = first;
    for (size_t k = first; k < e; k++) {
      const int nbits = (recs[k].df == 0 || recs[k].df == 4 || recs[k].df == 5 || recs[k].df == 11) ? 56 : 112;   // i*8 at :440
      const int s = have_cb ? g.host.modeS_message_score (recs[k].msg, nbits) : score_record (recs[k], g.cache);
      if (s > best) { best = s; besti = k; }
    }
    if (best < 0) continue;
    const Record &r = recs[besti];
    const int msglen = g.bound && g.host.modeS_message_len_by_type ? g.host.modeS_message_len_by_type (r.df) : len_by_type (r.df);
    const uint64_t ts = mag->sample_timestamp + j * 5 + r.phase;                                 // ref: :467
    const int64_t ms = g.bound && g.host.receiveclock_ms_elapsed ? g.host.receiveclock_ms_elapsed (mag->sample_timestamp, ts)
                                                                  : int64_t ((ts - mag->sample_timestamp) / 12000);
    const int slen = msglen * 12 / 5;
    uint64_t scaled = 0;
    for (int k = 0; k < slen; k++) { const uint32_t v = m[j + 19 + k]; scaled += uint64_t (v * v); }   // ref: :490-494
    const double signal_power = double (scaled) / 65535.0 / 65535.0;

    if (have_cb) {
      mm.at<uint64_t> (g.host.off_timestamp_msg) = ts;
      mm.at<uint64_t> (g.host.off_sys_timestamp_msg) = mag->sys_timestamp + uint64_t (ms);
      mm.at<int> (g.host.off_score) = best;
      mm.at<uint32_t> (g.host.off_AC_flags) = 0;
      mm.at<int> (g.host.off_error_bits) = 0;
      if (g.host.decode_mode_S_message (mm.buf.data (), r.msg) < 0) continue;
      mm.at<double> (g.host.off_sig_level) = signal_power / slen;
      skip_until = j + uint64_t (slen) + 1;                                                      // ref: :514
      if (g.host.modeS_user_message) g.host.modeS_user_message (mm.buf.data ());
    } else {
      modes_message out;
      memset (&out, 0, sizeof out);
      if (accept_record (r, g.cache, &out) < 0) continue;
      out.timestamp = ts;
      out.score = int16_t (best);
      out.phase = r.phase;
      out.block = block;
      out.j = uint32_t (j);
      out.sig_level = signal_power / slen;
      skip_until = j + uint64_t (slen) + 1;
      g.queue.push_back (out);
    }
  }
}

// ref: src/demod-2000.c:612-1027.  The kernel slices every valid preamble (and its retries); the loop's order,
// the callbacks and the skip-ahead happen here.
void demod_2000 (const mag_buf *mag)
{
  std::lock_guard<std::mutex> lk (g.mu);
  std::vector<Record> recs;
  uint64_t vp = 0;
  const uint32_t block = g.block_counter++;
  if (!mag || !mag->data || mag->valid_length < mag->overlap) return;
  if (!scan_mag_buf (mag, recs, &vp, 2000)) return;
  const uint32_t mlen = mag->valid_length - mag->overlap;
  std::vector<uint64_t> pos (recs.size ());
  for (size_t i = 0; i < recs.size (); i++)
    pos[i] = (uint64_t (recs[i].pad[0]) | (uint64_t (recs[i].pad[1]) << 8)) * kStrip + recs[i].off;

  const bool have_cb = g.bound && g.host.decode_mode_S_message && g.host.mm_size;
  HostMm mm;
  if (have_cb) mm.buf.assign (g.host.mm_size, 0);
  auto elapsed = [&] (uint64_t ts) {
    return g.bound && g.host.receiveclock_ms_elapsed ? g.host.receiveclock_ms_elapsed (mag->sample_timestamp, ts)
                                                      : int64_t ((ts - mag->sample_timestamp) / 12000);
  };
  Pass2000Hooks hooks;
  Pass2000Stats ps;
  hooks.decode = [&] (const Record &r, uint64_t j, uint8_t, modes_message *out) {
    if (have_cb) {
      const uint64_t ts = mag->sample_timestamp + j * 6;                                       // ref: :960
      mm.at<int> (g.host.off_error_bits) = 0;                                                  // ref: :670
      mm.at<uint64_t> (g.host.off_timestamp_msg) = ts;
      mm.at<uint64_t> (g.host.off_sys_timestamp_msg) = mag->sys_timestamp + uint64_t (elapsed (ts));
      const int rc = g.host.decode_mode_S_message (mm.buf.data (), r.msg);
      out->error_bits = uint8_t (mm.at<int> (g.host.off_error_bits));
      return rc;
    }
    return (r.phase & kInfoStaticReject) ? -2 : accept_record (r, g.cache, out);
  };
  hooks.emit = [&] (const modes_message &m, const Record &r) {
    if (have_cb) {
      if (r.cls == kClsModeA) {                                                                 // ref: :677-701
        mm.at<uint64_t> (g.host.off_timestamp_msg) = mag->sample_timestamp + m.timestamp;
        mm.at<uint64_t> (g.host.off_sys_timestamp_msg) = mag->sys_timestamp + uint64_t (elapsed (mag->sample_timestamp + m.timestamp));
        mm.at<double> (g.host.off_sig_level) = m.sig_level;
        if (g.host.decode_mode_A_message) g.host.decode_mode_A_message (mm.buf.data (), (int (r.msg[0]) << 8) | r.msg[1]);
      }
      if (g.host.modeS_user_message) g.host.modeS_user_message (mm.buf.data ());
    } else {
      modes_message out = m;
      out.timestamp += mag->sample_timestamp;
      out.block = block;
      g.queue.push_back (out);
    }
  };
  // block geometry of one call: a single block of mlen positions starting at sample counter 0
  accept_pass_2000 (recs.data (), pos.data (), recs.size (), mlen ? mlen : 1, mlen, 0, 2000000,
                    g.bound && g.host.phase_enhance, hooks, ps);
}
// ref: src/demod-2400AC.c:25-228
void demod_2400_AC (const mag_buf *mag)
{
  std::lock_guard<std::mutex> lk (g.mu);
  if (!mag || !mag->data || mag->valid_length < mag->overlap) return;
  // noise_level as the reference forms it from the block's mean_power / mean_level (ref: :34-35)
  const double noise_stddev = sqrt (mag->mean_power - mag->mean_level * mag->mean_level);
  const double nv = (mag->mean_power + noise_stddev) * 65535 + 0.5;
  const uint32_t noise_level = (nv < 9.2e18 && nv > -9.2e18) ? uint32_t (uint64_t ((long long) nv)) : 0u;
  std::vector<Record> recs;
  uint64_t vp = 0;
  if (!scan_mag_buf (mag, recs, &vp, 24001, noise_level)) return;
  const uint32_t mlen = mag->valid_length - mag->overlap;
  std::vector<uint64_t> pos (recs.size ());
  for (size_t i = 0; i < recs.size (); i++)
    pos[i] = (uint64_t (recs[i].pad[0]) | (uint64_t (recs[i].pad[1]) << 8)) * kStrip + recs[i].off;
  std::vector<modes_message> out;
  accept_pass_2400ac (recs.data (), pos.data (), recs.size (), mlen ? mlen : 1, 0, 2400000, out);
  const bool have_cb = g.bound && g.host.mm_size && g.host.modeS_user_message;
  HostMm mm;
  if (have_cb) mm.buf.assign (g.host.mm_size, 0);                                              // memset at :46
  const uint32_t block = g.block_counter ? g.block_counter - 1 : 0;                            // same block as the demod_2400 call before
  for (modes_message &m : out) {
    const uint64_t ts = mag->sample_timestamp + m.timestamp;
    if (have_cb) {
      const int64_t ms = g.host.receiveclock_ms_elapsed ? g.host.receiveclock_ms_elapsed (mag->sample_timestamp, ts)
                                                        : int64_t ((ts - mag->sample_timestamp) / 12000);
      mm.at<uint64_t> (g.host.off_timestamp_msg) = ts;
      mm.at<uint64_t> (g.host.off_sys_timestamp_msg) = mag->sys_timestamp + uint64_t (ms);
      if (g.host.decode_mode_A_message) g.host.decode_mode_A_message (mm.buf.data (), (int (m.msg[0]) << 8) | m.msg[1]);
      g.host.modeS_user_message (mm.buf.data ());
    } else {
      m.timestamp = ts;
      m.block = block;
      g.queue.push_back (m);
    }
  }
}
// ref: src/demod-8000.c:106-123
int demod_8000_alloc (void)
{
  std::lock_guard<std::mutex> lk (g.mu);
  if (!ensure_ready ()) return 0;
  auto &s = g.d8;
  if (!s.dprev) {
    if (hipMalloc (reinterpret_cast<void **> (&s.dprev), kD8Overlap * sizeof (int32_t)) != hipSuccess) return 0;
    if (hipMalloc (reinterpret_cast<void **> (&s.dnext), kD8Overlap * sizeof (int32_t)) != hipSuccess) return 0;
    if (hipMalloc (reinterpret_cast<void **> (&s.small), 16 * sizeof (long long)) != hipSuccess) return 0;
  }
  if (hipMemset (s.dprev, 0, kD8Overlap * sizeof (int32_t)) != hipSuccess) return 0;        // calloc
  s.acc = 0; s.pending = 1; s.trig_tail = 0;
  g.d8_allocated = true;
  return 1;
}

void demod_8000_free (void)
{
  std::lock_guard<std::mutex> lk (g.mu);
  g.d8_allocated = false;
}

static bool d8_grow (size_t nsteps, uint32_t nchunks)
{
  auto &s = g.d8;
  const size_t words = nsteps / 64 + 64;
  if (words > s.words) {
    for (void *p : { (void *) s.trig, (void *) s.amb }) if (p) (void) hipFree (p);
    s.trig = s.amb = nullptr; s.words = 0;
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&s.trig), words * sizeof (uint64_t)));
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&s.amb), words * sizeof (uint64_t)));
    s.words = words;
  }
  if (nsteps > s.steps) {
    for (void *p : { (void *) s.p0, (void *) s.mx }) if (p) (void) hipFree (p);
    s.p0 = s.mx = nullptr; s.steps = 0;
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&s.p0), (nsteps + 64) * sizeof (uint32_t)));
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&s.mx), (nsteps + 64) * sizeof (uint32_t)));
    s.steps = nsteps;
  }
  if (nchunks > s.chunks) {
    for (void *p : { (void *) s.table, (void *) s.entry, (void *) s.win_start, (void *) s.win_count, (void *) s.chunk_rec,
                     (void *) s.windows, (void *) s.win_packed, (void *) s.records, (void *) s.rec_packed, (void *) s.off })
      if (p) (void) hipFree (p);
    s.chunks = 0;
    const size_t nw = size_t (nchunks) * kD8WinCap;
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&s.table), D8_TABLE_BYTES (nchunks)));
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&s.entry), nchunks + 1));
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&s.win_start), nw * sizeof (int64_t)));
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&s.win_count), nchunks * sizeof (uint32_t)));
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&s.chunk_rec), 2 * size_t (nchunks) * sizeof (uint32_t)));
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&s.windows), nw * sizeof (D8Window)));
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&s.win_packed), nw * sizeof (D8Window)));
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&s.records), nw * kD8Cand * sizeof (Record)));
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&s.rec_packed), nw * kD8Cand * sizeof (Record)));
    HIP_OK (hipMalloc (reinterpret_cast<void **> (&s.off), (2 * size_t (nchunks) + 2) * sizeof (uint64_t)));
    s.chunks = nchunks;
  }
  return true;
}

static bool demod_8000_locked (const mag_buf *mag, uint32_t block)
{
  auto &s = g.d8;
  if (!g.d8_allocated || !s.dprev) {
    set_error (MODES_E_ARG, "demod_8000 called without demod_8000_alloc()");
    fprintf (stderr, "modes_hip: %s\n", modes_last_error ());
    return false;
  }
  if (!ensure_ready () || !ensure_tables ()) return false;
  HIP_OK (hipSetDevice (g.device));
  const uint64_t T = uint64_t (mag->valid_length - mag->overlap) & ~7ull;                   // ref: :203
  const uint64_t nsteps = T / 8;
  if (!nsteps) return true;
  const uint32_t nchunks = uint32_t ((nsteps + kD8ChunkSteps - 1) / kD8ChunkSteps);
  if (!grow (g.d_mag, g.mag_cap, size_t (mag->valid_length) + 64) || !d8_grow (nsteps, nchunks)) return false;
  HIP_OK (hipMemcpyAsync (g.d_mag, mag->data, size_t (mag->valid_length) * 2, hipMemcpyHostToDevice, nullptr));

  D8Params q;
  memset (&q, 0, sizeof q);
  q.sp.iq = reinterpret_cast<const uint8_t *> (g.d_mag); q.sp.nsamples = T; q.sp.navail = mag->valid_length;
  q.sp.ov = mag->overlap; q.sp.block_samples = uint32_t (T); q.sp.conv = kConvMagU16;
  q.sp.short_tab = g.d_short; q.sp.short_n = g.short_n; q.sp.long_tab = g.d_long; q.sp.long_n = g.long_n;
  q.sp.bloom_short = g.d_bloom_short; q.sp.bloom_long = g.d_bloom_long; q.sp.lut256 = g.d_lut; q.sp.crc_table = g.d_crc_table;
  q.T = T; q.nsteps = nsteps; q.bshift = 63; q.dprev = s.dprev; q.acc0 = s.acc; q.pending = s.pending; q.nchunks = nchunks;
  q.trig = s.trig; q.amb = s.amb; q.p0 = s.p0; q.mx = s.mx; q.write_arrays = 1;
  q.table = s.table; q.entry = s.entry; q.win_start = s.win_start; q.win_count = s.win_count; q.windows = s.windows; q.records = s.records;
  long long *d_pre = s.small, *d_pending = s.small + 4;
  q.amb_count = reinterpret_cast<uint32_t *> (s.small + 5);
  int32_t *d_acc = reinterpret_cast<int32_t *> (s.small + 6);
  const long long init[8] = { 0, 0, 0, 0, 1, 0, 0, 0 };
  HIP_OK (hipMemcpyAsync (d_pre, init, sizeof init, hipMemcpyHostToDevice, nullptr));
  const size_t words = size_t (nsteps / 64 + 64);
  HIP_OK (hipMemsetAsync (q.trig, 0, words * sizeof (uint64_t), nullptr));
  HIP_OK (hipMemsetAsync (q.amb, 0, words * sizeof (uint64_t), nullptr));
  HIP_OK (launch_d8_trig (q, 0, nullptr));
  HIP_OK (launch_d8_acc (q, nsteps, d_acc, nullptr));           // the running average is carried to the next call: always formed
  HIP_OK (launch_d8_chain_table (q, nullptr));
  HIP_OK (launch_d8_chain (q, s.trig_tail, 0ll, d_pre, d_pending, nullptr));
  HIP_OK (launch_d8_windows (q, q.nchunks, nullptr));
  HIP_OK (launch_d8_tail (q, s.dnext, nullptr));
  HIP_OK (launch_d8_count (q, q.nchunks, s.chunk_rec, nullptr));
  std::vector<uint32_t> wc (nchunks), rcnt (2 * size_t (nchunks));
  std::vector<uint64_t> trig (words);
  long long small[8];
  HIP_OK (hipMemcpyAsync (wc.data (), s.win_count, nchunks * sizeof (uint32_t), hipMemcpyDeviceToHost, nullptr));
  HIP_OK (hipMemcpyAsync (rcnt.data (), s.chunk_rec, 2 * size_t (nchunks) * sizeof (uint32_t), hipMemcpyDeviceToHost, nullptr));
  HIP_OK (hipMemcpyAsync (trig.data (), s.trig, words * sizeof (uint64_t), hipMemcpyDeviceToHost, nullptr));
  HIP_OK (hipMemcpyAsync (small, s.small, sizeof small, hipMemcpyDeviceToHost, nullptr));
  HIP_OK (hipStreamSynchronize (nullptr));

  // state for the next call
  std::swap (s.dprev, s.dnext);
  memcpy (&s.acc, &small[6], sizeof (int32_t));
  s.pending = small[4];
  {
    unsigned long long tail = 0;
    for (int b = 0; b < 64; b++) {
      const long long n = (long long) nsteps - 64 + b;           // step of bit b
      bool bit;
      if (n >= 0) bit = (trig[size_t (n) >> 6] >> (n & 63)) & 1;
      else bit = (n + 64 >= 0) ? ((s.trig_tail >> (n + 64)) & 1) : false;   // still a step of the call before
      if (bit) tail |= 1ull << b;
    }
    s.trig_tail = tail;
  }

  std::vector<uint64_t> off (2 * size_t (nchunks) + 2);
  uint64_t nwin = 0, nrec = 0;
  for (uint32_t c = 0; c < nchunks; c++) {
    if (wc[c] > uint32_t (kD8WinCap)) { set_error (MODES_E_OVERFLOW, "demod_8000: window list overflow"); return false; }
    off[c] = nwin; off[nchunks + 1 + c] = nrec; nwin += rcnt[nchunks + c]; nrec += rcnt[c];
  }
  if (!nwin) return true;
  HIP_OK (hipMemcpyAsync (s.off, off.data (), off.size () * sizeof (uint64_t), hipMemcpyHostToDevice, nullptr));
  HIP_OK (launch_d8_gather (q, q.nchunks, s.off, s.off + nchunks + 1, s.win_packed, s.rec_packed, nullptr));
  std::vector<D8Window> wins (nwin);
  std::vector<Record> recs (nrec + 1);
  HIP_OK (hipMemcpyAsync (wins.data (), s.win_packed, nwin * sizeof (D8Window), hipMemcpyDeviceToHost, nullptr));
  if (nrec) HIP_OK (hipMemcpyAsync (recs.data (), s.rec_packed, nrec * sizeof (Record), hipMemcpyDeviceToHost, nullptr));
  HIP_OK (hipStreamSynchronize (nullptr));

  const bool have_cb = g.bound && g.host.modeS_message_score && g.host.decode_mode_S_message && g.host.mm_size;
  HostMm mm;
  if (have_cb) mm.buf.assign (g.host.mm_size, 0);
  const uint16_t *m = mag->data;
  Pass8000Hooks hooks;
  Pass8000Stats ps;
  hooks.score = [&] (const Record &r, int valid_bits) {
    return have_cb ? g.host.modeS_message_score (r.msg, valid_bits) : score_record (r, g.cache);
  };
  hooks.decode = [&] (const Record &r, uint64_t ts_rel, modes_message *out) {
    if (!have_cb) return accept_record (r, g.cache, out);
    const uint64_t ts = mag->sample_timestamp + ts_rel;                                        // ref: :421-422
    const int64_t ms = g.host.receiveclock_ms_elapsed ? g.host.receiveclock_ms_elapsed (mag->sample_timestamp, ts)
                                                      : int64_t ((ts - mag->sample_timestamp) / 12000);
    mm.at<uint64_t> (g.host.off_timestamp_msg) = ts;
    mm.at<uint64_t> (g.host.off_sys_timestamp_msg) = mag->sys_timestamp + uint64_t (ms);
    const int rc = g.host.decode_mode_S_message (mm.buf.data (), r.msg);
    out->error_bits = uint8_t (mm.at<int> (g.host.off_error_bits));
    out->addr = mm.at<uint32_t> (g.host.off_addr);
    return rc;
  };
  hooks.emit = [&] (modes_message &out, const Record &, uint64_t, int m_start, int slen) {
    double sig = 0.0;
    if (m_start >= 0) {                                                                          // ref: :469-484
      uint64_t scaled = 0;
      for (int k = 0; k < slen; k++) { const uint32_t v = m[m_start + k]; scaled += uint64_t (v * v); }
      sig = (double (scaled) / 65535.0 / 65535.0) / slen;
    }
    if (have_cb) {
      if (m_start >= 0) mm.at<double> (g.host.off_sig_level) = sig;
      if (out.addr && g.host.modeS_user_message) g.host.modeS_user_message (mm.buf.data ());   // ref: :487
    } else if (out.addr) {
      out.timestamp += mag->sample_timestamp;
      out.sig_level = sig;
      out.block = block;
      g.queue.push_back (out);
    }
  };
  accept_pass_8000 (wins.data (), nwin, recs.data (), uint32_t (T), 63, T, mag->overlap, 0, 8000000, hooks, ps);
  return true;
}

// ref: src/demod-8000.c:143-523
void demod_8000 (const mag_buf *mag)
{
  std::lock_guard<std::mutex> lk (g.mu);
  const uint32_t block = g.block_counter++;
  if (!mag || !mag->data || mag->valid_length < mag->overlap) return;
  (void) demod_8000_locked (mag, block);
}

size_t modes_hip_take_messages (modes_message *out, size_t max)
{
  std::lock_guard<std::mutex> lk (g.mu);
  const size_t n = g.queue.size () < max ? g.queue.size () : max;
  if (out && n) memcpy (out, g.queue.data (), n * sizeof (modes_message));
  g.queue.erase (g.queue.begin (), g.queue.begin () + n);
  return n;
}

}  // extern "C"
