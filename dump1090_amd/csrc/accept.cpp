#include "accept.h"

#include <algorithm>
#include <cstring>

namespace modes {

void IcaoCache::clear () { memset (slot_, 0, sizeof slot_); }

void accept_pass_2400 (const Record *rec, const uint64_t *pos, size_t n, uint32_t B, uint64_t first_sample,
                       uint32_t sample_rate, IcaoCache &cache, AcceptStats &st,
                       std::vector<modes_message> &out, std::vector<uint64_t> &out_pos,
                       const uint64_t *vp_bitmap, uint64_t nsamples)
{
  uint64_t cur_block = ~0ull, skip_until = 0;
  for (size_t i = 0; i < n;) {
    size_t g = i + 1;
    while (g < n && pos[g] == pos[i]) g++;          // same offset, other phases (phase_enhance)
    const uint64_t blk = pos[i] / B, j = pos[i] % B;
    if (blk != cur_block) { cur_block = blk; skip_until = 0; }   // each demod_2400 call starts at j = 0
    if (j < skip_until) { i = g; continue; }

    int best = -2; size_t besti = i;
    for (size_t k = i; k < g; k++) {
      const int s = score_record (rec[k], cache);
      if (s > best) { best = s; besti = k; st.out_of_phase++; }
    }
    i = g;
    if (best < 0) continue;

    st.demodulated++;
    modes_message mm;
    memset (&mm, 0, sizeof mm);
    const Record &r = rec[besti];
    const int rc = accept_record (r, cache, &mm);
    if (rc < 0) { st.CRC_bad++; if (rc == -1) st.rejected_unknown++; continue; }
    if (mm.error_bits) {
      st.CRC_fixed++;
      if (mm.error_bits == 1) st.CRC_single_bit_fix++; else st.CRC_two_bits_fix++;
    } else st.CRC_good++;

    const uint64_t counter = first_sample + blk * B;
    const uint64_t block_ts = uint64_t ((double (counter) * 12E6) / double (sample_rate));   // ref: src/infile.c:143
    mm.timestamp = block_ts + j * 5 + r.phase;                                                // ref: src/demod-2400.c:467
    mm.score = int16_t (best);
    mm.phase = r.phase;
    mm.block = uint32_t (first_sample / B + blk);
    mm.j = uint32_t (j);
    out.push_back (mm);
    out_pos.push_back (pos[besti]);
    skip_until = j + uint64_t (len_by_type (r.df)) * 12 / 5 + 1;                              // ref: src/demod-2400.c:514
    if (vp_bitmap) {
      // positions (j, skip_until) of this block are never examined by the reference's loop
      uint64_t a = blk * B + j + 1, b = std::min (blk * B + std::min<uint64_t> (skip_until, B), nsamples);
      for (uint64_t q = a; q < b; q++) st.skipped_preambles += (vp_bitmap[q >> 6] >> (q & 63)) & 1;
    }
  }
}

// ref: src/demod-2000.c:652-1026
void accept_pass_2000 (const Record *rec, const uint64_t *pos, size_t n, uint32_t B, uint64_t nsamples,
                       uint64_t first_sample, uint32_t sample_rate, bool phase_enhance, const Pass2000Hooks &hooks,
                       Pass2000Stats &st)
{
  uint64_t cur_block = ~0ull, skip_until = 0;
  auto kind_of = [] (const Record &r) { return r.cls == kClsModeA ? int (kKindModeA) : int (r.phase >> 6); };

  for (size_t i = 0; i < n;) {
    size_t g = i + 1;
    while (g < n && pos[g] == pos[i]) g++;
    const Record *modea = nullptr, *prim = nullptr, *same = nullptr, *after = nullptr;
    for (size_t k = i; k < g; k++) {
      switch (kind_of (rec[k])) {
        case kKindModeA: modea = &rec[k]; break;
        case kKindPrimary: prim = &rec[k]; break;
        case kKindRetrySame: same = &rec[k]; break;
        default: after = &rec[k]; break;
      }
    }
    const uint64_t blk = pos[i] / B, j = pos[i] % B;
    i = g;
    if (blk != cur_block) { cur_block = blk; skip_until = 0; }          // each demod_2000 call starts at j = 0
    if (j < skip_until) continue;
    const uint64_t mlen = std::min<uint64_t> (B, nsamples - blk * B);
    const uint64_t counter = first_sample + blk * B;
    const uint64_t block_ts = uint64_t ((double (counter) * 12E6) / double (sample_rate));   // ref: src/infile.c:143

    // one iteration of the reference's loop in Mode S mode: slice statistics, decode, emit, skip-ahead
    auto run = [&] (const Record &r, uint64_t jj, uint8_t corrected, bool *ok, int *ebits) {
      *ok = false; *ebits = 0;
      st.demodulated += r.phase & kInfoDemodMask;                        // ref: :848, 934
      if (!(r.phase & kInfoDecode)) return;
      if (r.phase & kInfoStale) st.stale++;
      modes_message mm;
      memset (&mm, 0, sizeof mm);
      const int rc = hooks.decode (r, jj, corrected, &mm);
      if (rc < 0) return;
      *ok = true; *ebits = mm.error_bits;
      mm.timestamp = block_ts + jj * 6;                                  // ref: :960
      mm.score = -1;
      mm.phase = corrected;
      mm.block = uint32_t (first_sample / B + blk);
      mm.j = uint32_t (jj);
      mm.sig_level = 0.0;                                                // wiped by decode's memset (ref: :966)
      hooks.emit (mm, r);
      skip_until = jj + 2 * (8 + uint64_t (r.pad[2]));                   // ref: :997 and the loop's j++
    };

    if (modea) {                                                         // ref: :677-701
      modes_message mm;
      memset (&mm, 0, sizeof mm);
      const uint32_t code = (uint32_t (modea->msg[0]) << 8) | modea->msg[1];
      mm.timestamp = block_ts + (j + 1) * 6;
      mm.msg[0] = modea->msg[0]; mm.msg[1] = modea->msg[1];
      mm.msg_bits = 16; mm.msg_type = 32;
      mm.addr = (code & 0x0000FF7Fu) | kNonIcao;
      mm.score = -1;
      mm.block = uint32_t (first_sample / B + blk);
      mm.j = uint32_t (j);
      const int lv = int (modea->crc);
      mm.sig_level = double (int (uint32_t (lv) * uint32_t (lv))) / (65535.0 * 65535.0);
      hooks.emit (mm, *modea);
      st.demod_modeac++;
      skip_until = j + 51;
      continue;
    }
    if (!prim) continue;
    st.valid_preamble++;
    bool ok; int ebits;
    run (*prim, j, 0, &ok, &ebits);
    if (phase_enhance && (!ok || ebits > 0) && (prim->phase & kInfoOutOfPhase)) {   // ref: :1016-1025
      if (!ok) {
        if (j != 0 && same) {
          st.out_of_phase++;
          bool ok2; int e2;
          run (*same, j, 1, &ok2, &e2);
        }
      } else {
        st.out_of_phase++;
        const uint64_t jn = j + 2 * (8 + uint64_t (prim->pad[2])) - 1;
        if (jn < mlen && after) {
          bool ok3; int e3;
          run (*after, jn, 1, &ok3, &e3);
        }
      }
    }
  }
}

void accept_pass_8000 (const D8Window *win, size_t nwin, const Record *rec, uint32_t B, int bshift, uint64_t T, uint32_t ov,
                       uint64_t first_sample, uint32_t sample_rate, const Pass8000Hooks &hooks, Pass8000Stats &st)
{
  modes_message mm;
  accept_pass_8000_with (win, nwin, rec, B, bshift, T, ov, first_sample, sample_rate, hooks.score, hooks.decode, hooks.emit,
                         [&] { return &mm; }, st);
}

void accept_pass_2400ac (const Record *rec, const uint64_t *pos, size_t n, uint32_t B, uint64_t first_sample,
                         uint32_t sample_rate, std::vector<modes_message> &out)
{
  uint64_t cur_block = ~0ull, skip_until = 0;
  for (size_t i = 0; i < n; i++) {
    const uint64_t blk = pos[i] / B, f1 = pos[i] % B;
    if (blk != cur_block) { cur_block = blk; skip_until = 0; }
    if (f1 < skip_until) continue;
    const uint64_t counter = first_sample + blk * B;
    const uint64_t block_ts = uint64_t ((double (counter) * 12E6) / double (sample_rate));   // ref: src/infile.c:143
    modes_message mm;
    memset (&mm, 0, sizeof mm);
    const uint32_t code = (uint32_t (rec[i].msg[0]) << 8) | rec[i].msg[1];
    mm.timestamp = block_ts + rec[i].crc / 5;                      // f2_clock / 5 (ref: src/demod-2400AC.c:211)
    mm.msg[0] = rec[i].msg[0]; mm.msg[1] = rec[i].msg[1];
    mm.msg_bits = 16; mm.msg_type = 32;                            // ref: src/dump1090.c:2722-2766
    mm.addr = (code & 0x0000FF7Fu) | kNonIcao;
    mm.score = -1;
    mm.block = uint32_t (first_sample / B + blk);
    mm.j = uint32_t (f1);
    out.push_back (mm);
    skip_until = f1 + 20 * 87 / 25 + 1;                            // ref: :226 and the loop's increment
  }
}

}  // namespace modes
