// Host-side tables of the path: CRC-24 byte table, single-bit syndromes, the 1-/2-bit error
// tables (ref: src/crc.c) and the two look-up tables of dump1090.c (ref: src/dump1090.c:385-423).
#pragma once
#include <stdint.h>
#include <vector>
#include "common.h"

namespace modes {

struct HostErr { uint32_t syndrome; int errors; int8_t bit[2]; };   // same members as `errorinfo`

class CrcTables {
 public:
  CrcTables ();
  void build (int fix_bits);                       // crc_init (ref: src/crc.c:281-307)
  uint32_t checksum (const uint8_t *msg, int bits) const;   // ref: src/crc.c:67-82
  const HostErr *diagnose (uint32_t syndrome, int bits) const;  // nullptr = uncorrectable
  const std::vector<HostErr> &table (int bits) const { return bits == kShortBits ? short_ : long_; }
  std::vector<ErrEntry> device_table (int bits) const;
  const uint32_t *single_bit () const { return single_bit_; }
  const uint32_t *byte_table () const { return crc_table_; }
  int fix_bits () const { return fix_bits_; }
 private:
  std::vector<HostErr> make (int bits, int max_correct, int max_detect) const;
  uint32_t crc_table_[256];
  uint32_t single_bit_[kLongBits];
  std::vector<HostErr> short_, long_;
  int fix_bits_ = -1;
};

// mag_lut as the converters index it: one byte (ref: src/convert.c:29 with src/dump1090.c:396-400)
void build_uc8_byte_lut (uint16_t out[256]);
// (float)((b - 127.5) / 127.5) for every byte value (ref: src/convert.c:132-133)
void build_uc8_float_lut (float out[256]);
// round(100*log10(i)), [0] = 0 (ref: src/dump1090.c:410-423)
void build_log10_lut (uint16_t out[65536]);

}  // namespace modes
