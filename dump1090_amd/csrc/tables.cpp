#include "tables.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <unordered_map>

namespace modes {

CrcTables::CrcTables ()
{
  // byte-at-a-time table for the Mode S generator 0xFFF409 (ref: src/crc.c:43-54)
  for (uint32_t v = 0; v < 256; v++) {
    uint32_t r = v << 16;
    for (int k = 0; k < 8; k++) r = (r & 0x800000u) ? (r << 1) ^ 0xFFF409u : r << 1;
    crc_table_[v] = r & 0xFFFFFFu;
  }
  // syndrome of each single flipped bit of a 112-bit frame (ref: src/crc.c:55-61)
  uint8_t frame[kLongBytes];
  for (int b = 0; b < kLongBits; b++) {
    memset (frame, 0, sizeof frame);
    frame[b / 8] = uint8_t (0x80u >> (b % 8));
    single_bit_[b] = checksum (frame, kLongBits);
  }
}

uint32_t CrcTables::checksum (const uint8_t *msg, int bits) const
{
  const int n = bits / 8;
  uint32_t rem = 0;
  for (int i = 0; i < n - 3; i++)
    rem = ((rem << 8) ^ crc_table_[msg[i] ^ ((rem >> 16) & 0xFF)]) & 0xFFFFFFu;
  return rem ^ (uint32_t (msg[n - 3]) << 16) ^ (uint32_t (msg[n - 2]) << 8) ^ msg[n - 1];
}

// The table the reference builds recursively, sorts and prunes (ref: src/crc.c:189-270):
//  * every pattern of 1..max_correct flipped bits among bits [5, bits) of the frame,
//  * minus every syndrome produced by more than one such pattern,
//  * minus (when max_detect > max_correct) every syndrome that a pattern of
//    max_correct+1..max_detect bits also produces.
std::vector<HostErr> CrcTables::make (int bits, int max_correct, int max_detect) const
{
  std::vector<HostErr> t;
  if (max_correct <= 0) return t;
  const uint32_t *sb = single_bit_ + (kLongBits - bits);

  for (int a = 5; a < bits; a++) {
    t.push_back (HostErr{ sb[a], 1, { int8_t (a), -1 } });
    if (max_correct >= 2)
      for (int b = a + 1; b < bits; b++)
        t.push_back (HostErr{ sb[a] ^ sb[b], 2, { int8_t (a), int8_t (b) } });
  }
  std::sort (t.begin (), t.end (), [] (const HostErr &x, const HostErr &y) { return x.syndrome < y.syndrome; });

  std::vector<HostErr> uniq;
  for (size_t i = 0; i < t.size ();) {
    size_t j = i + 1;
    while (j < t.size () && t[j].syndrome == t[i].syndrome) j++;
    if (j == i + 1) uniq.push_back (t[i]);
    i = j;
  }

  if (max_detect > max_correct) {
    std::unordered_map<uint32_t, uint32_t> where;
    where.reserve (uniq.size () * 2);
    for (uint32_t i = 0; i < uniq.size (); i++) where.emplace (uniq[i].syndrome, i);
    std::vector<char> dead (uniq.size (), 0);
    auto hit = [&] (uint32_t s) { auto it = where.find (s); if (it != where.end ()) dead[it->second] = 1; };
    // patterns of max_correct+1 .. max_detect bits; max_detect is at most 4 in the reference
    for (int a = 5; a < bits; a++)
      for (int b = a + 1; b < bits; b++) {
        const uint32_t s2 = sb[a] ^ sb[b];
        if (max_correct < 2 && max_detect >= 2) hit (s2);
        for (int c = b + 1; c < bits && max_detect >= 3; c++) {
          const uint32_t s3 = s2 ^ sb[c];
          if (max_correct < 3) hit (s3);
          for (int d = c + 1; d < bits && max_detect >= 4; d++) hit (s3 ^ sb[d]);
        }
      }
    std::vector<HostErr> kept;
    for (size_t i = 0; i < uniq.size (); i++) if (!dead[i]) kept.push_back (uniq[i]);
    uniq.swap (kept);
  }
  return uniq;
}

void CrcTables::build (int fix_bits)
{
  fix_bits_ = fix_bits;
  if (fix_bits == 0)      { short_.clear (); long_.clear (); }
  else if (fix_bits == 1) { short_ = make (kShortBits, 1, 1); long_ = make (kLongBits, 1, 1); }
  else                    { short_ = make (kShortBits, 2, 4); long_ = make (kLongBits, 2, 4); }
}

const HostErr *CrcTables::diagnose (uint32_t syndrome, int bits) const
{
  static const HostErr none = { 0, 0, { 0, 0 } };
  if (syndrome == 0) return &none;
  const std::vector<HostErr> &t = table (bits);
  auto it = std::lower_bound (t.begin (), t.end (), syndrome,
                              [] (const HostErr &e, uint32_t s) { return e.syndrome < s; });
  return (it != t.end () && it->syndrome == syndrome) ? &*it : nullptr;
}

std::vector<ErrEntry> CrcTables::device_table (int bits) const
{
  std::vector<ErrEntry> d;
  for (const HostErr &e : table (bits))
    d.push_back (ErrEntry{ e.syndrome, int8_t (e.errors), e.bit[0], e.bit[1], 0 });
  return d;
}

void build_uc8_byte_lut (uint16_t out[256])
{
  // Modes.mag_lut[I*129 + Q] = round(360*hypot(I,Q)); a single byte b lands on (b/129, b%129)
  for (int b = 0; b < 256; b++)
    out[b] = uint16_t (std::round (360 * std::hypot (double (b / 129), double (b % 129))));
}

void build_uc8_float_lut (float out[256])
{
  for (int b = 0; b < 256; b++) out[b] = float ((b - 127.5) / 127.5);
}

void build_log10_lut (uint16_t out[65536])
{
  out[0] = 0;
  for (int i = 1; i < 65536; i++) out[i] = uint16_t (std::round (100.0 * std::log10 (double (i))));
}

}  // namespace modes
