// The ordered accept pass on the host: what the reference does, in order, with every candidate
// that reached modeS_message_score() -- ICAO cache tests and updates, score comparison,
// decode_mode_S_message()'s CRC/repair/accept rules, skip-ahead -- applied to the records the
// scan kernel left.  Works on pre-digested records (syndrome + table hit already known).
#pragma once
#include <stdint.h>
#include <cstring>
#include <functional>
#include <vector>
#include "common.h"
#include "../../include/modes_hip.h"

namespace modes {

// ref: src/dump1090.c:1438-1476.  One address per slot; the reference's TTL compares wall-clock
// seconds (60 s), which a file run never reaches at these speeds: not modelled.
class IcaoCache {
 public:
  IcaoCache () { clear (); }
  void clear ();
  static uint32_t hash (uint32_t a);
  void add (uint32_t a) { slot_[hash (a)] = a; }
  bool test (uint32_t a) const { uint32_t s = slot_[hash (a)]; return s && s == a; }
  // the same with the slot found once
  bool test_at (uint32_t slot, uint32_t a) const { uint32_t s = slot_[slot]; return s && s == a; }
  void add_at (uint32_t slot, uint32_t a) { slot_[slot] = a; }
  uint32_t *raw () { return slot_; }
  const uint32_t *raw () const { return slot_; }
 private:
  uint32_t slot_[kIcaoSlots];
};

struct AcceptStats {
  uint64_t demodulated = 0, out_of_phase = 0, CRC_good = 0, CRC_bad = 0, CRC_fixed = 0,
           CRC_single_bit_fix = 0, CRC_two_bits_fix = 0, rejected_unknown = 0,
           skipped_preambles = 0;   // valid preambles the reference never saw: inside a skipped-over message
};

// ref: src/dump1090.c:1427-1432
inline int len_by_type (int df) { return (df == 16 || df == 17 || df == 19 || df == 20 || df == 21) ? kLongBits : kShortBits; }

// score_record: score of one record against the cache (ref: src/dump1090.c:3279-3365).
// accept_record: accept rules + repair (ref: src/dump1090.c:2372-2578, 2586-2587, 2676-2687); fills msg/addr/... of `out`;
// returns 0, -1 (unknown ICAO), -2 (CRC); adds to the cache exactly where the reference does.
// Defined here: the ordered walks call them once or twice per candidate.
inline uint32_t IcaoCache::hash (uint32_t a)
{
  a = ((a >> 16) ^ a) * 0x45D9F3Bu;
  a = ((a >> 16) ^ a) * 0x45D9F3Bu;
  a = (a >> 16) ^ a;
  return a & (kIcaoSlots - 1);
}

inline uint32_t aa_field (const uint8_t *msg) { return (uint32_t (msg[1]) << 16) | (uint32_t (msg[2]) << 8) | msg[3]; }

// AA as it would read after the table's repair (ref: src/dump1090.c:3225-3241): only flips that land in bits 8..31
inline uint32_t aa_after_repair (const Record &r)
{
  uint32_t a = aa_field (r.msg);
  const int8_t b[2] = { r.bit0, r.bit1 };
  for (int i = 0; i < r.nerr; i++)
    if (b[i] >= 8 && b[i] <= 31) a ^= 1u << (31 - b[i]);
  return a;
}

inline int score_record (const Record &r, const IcaoCache &cache)
{
  switch (r.cls) {
    case kClsAP:  return cache.test (r.crc) ? 1000 : -1;
    case kClsAP2: return cache.test (r.crc) ? 1000 : -2;
    case kClsDF11: {
      const bool known = cache.test (aa_after_repair (r));
      if ((r.crc & 0x7F) == 0) return (known ? 1600 : 750) / (r.nerr + 1);
      return known ? 1000 / (r.nerr + 1) : -1;
    }
    default:
      return (cache.test (aa_after_repair (r)) ? 1800 : 1400) / (r.nerr + 1);
  }
}

// Which extended squitters carry a non-ICAO address (the only output of decode_extended_squitter
// that reaches mm->addr; ref: src/dump1090.c:1918-1963, 1990, 2097, 2130, 2220, 2239).
inline uint32_t non_icao_flag (const uint8_t *msg, int df)
{
  bool look_for_imf = false;
  if (df == 18) {
    const int cf = msg[0] & 7;
    if (cf == 1 || cf == 5) return kNonIcao;
    if (cf == 3) return (msg[4] & 0x80) ? kNonIcao : 0;
    if (cf == 4 || cf == 7) return kNonIcao;
    look_for_imf = (cf == 2 || cf == 6);
  }
  if (!look_for_imf) return 0;
  const int type = msg[4] >> 3;
  bool imf = false;
  if (type == 19) imf = msg[5] & 0x80;
  else if (type >= 5 && type <= 8) imf = msg[6] & 0x08;
  else if (type == 0 || (type >= 9 && type <= 18) || (type >= 20 && type <= 22)) imf = msg[4] & 0x01;
  else if (type == 28) imf = (msg[4] & 7) == 1 && (msg[10] & 0x01);
  else if (type == 31) imf = msg[10] & 0x01;
  return imf ? kNonIcao : 0;
}

inline void flip (uint8_t *msg, int bit) { msg[bit >> 3] ^= uint8_t (0x80u >> (bit & 7)); }

inline int accept_record (const Record &r, IcaoCache &cache, modes_message *out)
{
  uint8_t *msg = out->msg;
  memcpy (msg, r.msg, kLongBytes);
  out->msg_type = r.df;
  out->msg_bits = uint8_t (len_by_type (r.df));
  out->crc = r.crc;
  out->error_bits = 0;
  out->addr = 0;
  out->sig_level = 0.0;

  bool clean_announcer = false;
  switch (r.cls) {
    case kClsAP:
    case kClsAP2:
      if (!cache.test (r.crc)) return -1;
      out->addr = r.crc;
      if (r.df >= 24) out->msg_type = 24;
      return 0;
    case kClsDF11:
      if (r.crc & 0xFFFF80u) {
        out->error_bits = uint8_t (r.nerr);
        if (r.nerr >= 1) flip (msg, r.bit0);
        if (!cache.test (aa_field (msg))) return -1;
      }
      out->addr = aa_field (msg);
      clean_announcer = out->error_bits == 0 && (r.crc & 0x7F) == 0;
      break;
    default: {
      if (r.crc != 0) {
        const uint32_t before = aa_field (msg);
        out->error_bits = uint8_t (r.nerr);
        if (r.nerr >= 1) flip (msg, r.bit0);
        if (r.nerr >= 2) flip (msg, r.bit1);
        const uint32_t after = aa_field (msg);
        if (r.nerr == 2 && !cache.test (after)) return -1;
        if (r.nerr == 1 && before != after && !cache.test (after)) return -1;
      }
      out->addr = aa_field (msg) | non_icao_flag (msg, r.df);
      clean_announcer = out->error_bits == 0;
      break;
    }
  }
  if (clean_announcer) cache.add (out->addr);
  return 0;
}

// score_record () and accept_record () for a window's ONLY candidate (demod_8000's usual window), in one go: both ask the
// cache about the same address, so its slot is found and read once.  Returns the score (< 0: nothing is decoded, `out`
// is scratch); *rc = what accept_record () would have returned.  Same rules, same order of cache reads and writes.
inline int score_and_accept (const Record &r, IcaoCache &cache, modes_message *out, int *rc)
{
  uint8_t *msg = out->msg;
  memcpy (msg, r.msg, kLongBytes);
  out->msg_type = r.df;
  out->msg_bits = uint8_t (len_by_type (r.df));
  out->crc = r.crc;
  out->error_bits = 0;
  out->addr = 0;
  out->sig_level = 0.0;
  *rc = 0;
  if (r.cls == kClsAP || r.cls == kClsAP2) {
    if (!cache.test (r.crc)) return r.cls == kClsAP ? -1 : -2;
    out->addr = r.crc;
    if (r.df >= 24) out->msg_type = 24;
    return 1000;
  }
  const uint32_t aa = aa_after_repair (r), slot = IcaoCache::hash (aa);
  const bool known = cache.test_at (slot, aa);
  int score;
  if (r.cls == kClsDF11) {
    if ((r.crc & 0x7F) == 0) score = (known ? 1600 : 750) / (r.nerr + 1);
    else if (!known) return -1;
    else score = 1000 / (r.nerr + 1);
    if (r.crc & 0xFFFF80u) {
      out->error_bits = uint8_t (r.nerr);
      if (r.nerr >= 1) flip (msg, r.bit0);
      const uint32_t a2 = aa_field (msg);
      if (!(a2 == aa ? known : cache.test (a2))) { *rc = -1; return score; }
    }
    out->addr = aa_field (msg);
    if (out->error_bits == 0 && (r.crc & 0x7F) == 0) { if (out->addr == aa) cache.add_at (slot, aa); else cache.add (out->addr); }
    return score;
  }
  score = (known ? 1800 : 1400) / (r.nerr + 1);
  if (r.crc != 0) {
    const uint32_t before = aa_field (msg);
    out->error_bits = uint8_t (r.nerr);
    if (r.nerr >= 1) flip (msg, r.bit0);
    if (r.nerr >= 2) flip (msg, r.bit1);
    const uint32_t after = aa_field (msg);
    if ((r.nerr == 2 || (r.nerr == 1 && before != after)) && !(after == aa ? known : cache.test (after))) { *rc = -1; return score; }
  }
  out->addr = aa_field (msg) | non_icao_flag (msg, r.df);
  if (out->error_bits == 0) { if (out->addr == aa) cache.add_at (slot, aa); else cache.add (out->addr); }
  return score;
}


// The per-block ordered walk of demod_2400 (ref: src/demod-2400.c:200-520) over records sorted by
// stream position.  `pos[i]` is the scan position of records[i] inside the shard.
// sample_rate/first_sample give the block timestamps (ref: src/infile.c:143).
void accept_pass_2400 (const Record *records, const uint64_t *pos, size_t n, uint32_t block_samples,
                       uint64_t first_sample, uint32_t sample_rate, IcaoCache &cache, AcceptStats &st,
                       std::vector<modes_message> &out, std::vector<uint64_t> &out_pos,
                       const uint64_t *vp_bitmap, uint64_t nsamples);

// ---- demod_2000 ----------------------------------------------------------------------------
// The per-block ordered walk of demod_2000 (ref: src/demod-2000.c:652-1026) over the records scan2000 leaves:
// one per valid preamble, plus (phase_enhance) its phase-corrected retries, plus Mode A/C replies (mode_AC).
// `pos[i]` = scan position of the preamble the record belongs to.  The two callbacks are the reference's
// decode_mode_S_message (returns < 0 to reject; fills msg/addr/error_bits/... of `out`) and
// modeS_user_message.
struct Pass2000Stats {
  uint64_t valid_preamble = 0, demodulated = 0, out_of_phase = 0, demod_modeac = 0, stale = 0;
};
struct Pass2000Hooks {
  // rc of decode_mode_S_message for record r; on success `out` carries msg, msg_bits, msg_type, addr, crc, error_bits
  std::function<int (const Record &r, uint64_t j, uint8_t corrected, modes_message *out)> decode;
  std::function<void (const modes_message &m, const Record &r)> emit;
};
void accept_pass_2000 (const Record *records, const uint64_t *pos, size_t n, uint32_t block_samples, uint64_t nsamples,
                       uint64_t first_sample, uint32_t sample_rate, bool phase_enhance, const Pass2000Hooks &hooks,
                       Pass2000Stats &st);

// ---- demod_2400_AC ----
// The loop of demod_2400_AC over the hits the kernel found (ref: src/demod-2400AC.c:58,206-226): a hit is taken
// unless it lies inside the 69 samples skipped behind the previous one of the same block.
void accept_pass_2400ac (const Record *records, const uint64_t *pos, size_t n, uint32_t block_samples,
                         uint64_t first_sample, uint32_t sample_rate, std::vector<modes_message> &out);

// ---- demod_8000 ----
// The ordered part of demod_8000 over the completed match windows (ref: src/demod-8000.c:368-488): score the
// window's candidates against the live ICAO cache, keep the best (first wins ties), decode, emit.
struct Pass8000Stats {
  uint64_t valid_preamble = 0, demodulated = 0, CRC_good = 0, CRC_bad = 0, CRC_single_bit_fix = 0, CRC_two_bits_fix = 0,
           rejected_unknown = 0;
};
struct Pass8000Hooks {
  std::function<int (const Record &r, int valid_bits)> score;            // modeS_message_score
  // decode_mode_S_message with mm.timestamp_msg already set to `timestamp`; fills `out` on success
  std::function<int (const Record &r, uint64_t timestamp, modes_message *out)> decode;
  // modeS_user_message; m_start >= 0: the signal level is the mean square of `slen` magnitudes from there (block-relative)
  std::function<void (modes_message &m, const Record &r, uint64_t blkc, int m_start, int slen)> emit;
};
void accept_pass_8000 (const D8Window *win, size_t nwin, const Record *rec, uint32_t block_samples, int bshift, uint64_t T,
                       uint32_t ov, uint64_t first_sample, uint32_t sample_rate, const Pass8000Hooks &hooks, Pass8000Stats &st);

// The same walk with the three hooks as plain callables (the bulk engine inlines its own rules here; the function above
// wraps the bound host callbacks of the drop-in entry point).
// `slot ()` says where the next message is to be built; `emit` is handed that place back (a caller that collects the
// messages in an array moves on by one there, or leaves the place to the next message).
// `own_rules`: the cache, when score and decode are this library's own score_record () / accept_record () on it: a window
// with a single candidate then goes through score_and_accept ().
template <class Score, class Decode, class Emit, class Slot>
inline void accept_pass_8000_with (const D8Window *win, size_t nwin, const Record *rec, uint32_t B, int bshift, uint64_t T, uint32_t ov,
                                   uint64_t first_sample, uint32_t sample_rate, Score &&score, Decode &&decode, Emit &&emit, Slot &&slot,
                                   Pass8000Stats &st_out, IcaoCache *own_rules = nullptr)
{
  Pass8000Stats st = st_out;                                        // counted in registers, handed back at the end
  uint64_t ts_blk = ~0ull, block_ts = 0;                           // block_ts of the block seen last (it changes every 64 K samples)
  for (size_t wi = 0; wi < nwin; wi++) {
    const D8Window &w = win[wi];
    const Record *r0 = rec + w.pad;
    int best = -1; uint32_t besti = 0;
    int rc_single = 0;
    const bool single = own_rules && w.nrec == 1;
    if (single) {
      modes_message *mm1 = slot ();
      memset (mm1, 0, sizeof *mm1);
      best = score_and_accept (r0[0], *own_rules, mm1, &rc_single);
    } else
      for (uint32_t k = 0; k < w.nrec; k++) {                      // ref: :368-379, `if (message_result > best_result)`
        const int sc = score (r0[k], (r0[k].phase >> 4) ? kLongBits : kShortBits);
        if (sc > best) { best = sc; besti = k; }
      }
    if (best < 0) continue;
    st.valid_preamble++; st.demodulated++;                         // ref: :412, 424
    const Record &r = r0[besti];
    const int pass = r.phase >> 4, shift = r.phase & 15, nbytes = pass ? kLongBytes : kShortBytes;
    const int64_t position_g = (pass ? w.g_long : w.g_short) + int64_t (nbytes + 1) * 64 + 8 * shift;   // ref: :376
    const uint64_t done = uint64_t (8 * (w.s + kD8Win - 1));
    const uint64_t blkc = bshift >= 63 ? 0 : done >> bshift;      // the block whose call completes the window
    const int position = int (position_g - int64_t (blkc * B));
    const int slen = len_by_type (r.df) * 8;
    if (blkc != ts_blk) {
      const uint64_t counter = first_sample + blkc * B;
      block_ts = uint64_t ((double (counter) * 12E6) / double (sample_rate));   // ref: src/infile.c:143
      ts_blk = blkc;
    }
    const uint64_t ts = block_ts + uint64_t (int64_t ((position - kD8LookAhead - slen - int (ov)) * 12 / 8));   // ref: :421-422
    modes_message &mm = *slot ();
    int rc = rc_single;
    if (!single) {
      memset (&mm, 0, sizeof mm);
      rc = decode (r, ts, &mm);
    }
    if (rc < 0) { st.CRC_bad++; if (rc == -1) st.rejected_unknown++; continue; }
    if (mm.error_bits == 1) st.CRC_single_bit_fix++; else if (mm.error_bits == 2) st.CRC_two_bits_fix++; else st.CRC_good++;
    mm.timestamp = ts;
    mm.score = int16_t (best);
    mm.block = uint32_t (first_sample / B + blkc);
    mm.j = uint32_t (position);
    int m_start = position - kD8LookAhead - slen;                  // ref: :469-484
    const uint64_t mlen_c = B < T - blkc * B ? uint64_t (B) : T - blkc * B;
    if (!(m_start >= 0 && uint64_t (m_start + slen) <= mlen_c)) m_start = -1;
    emit (mm, r, blkc, m_start, slen);
  }
  st_out = st;
}

}  // namespace modes
