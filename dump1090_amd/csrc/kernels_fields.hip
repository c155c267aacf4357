// Batched field decode of the accepted messages: what decode_mode_S_message does behind its CRC / accept
// switch -- AA, AC, CA, CF, FS, ID, MB (BDS 2,0), MD, ME (extended squitter), VS (ref: src/dump1090.c:2580-2675,
// 1844-2249, 1494-1623, 2771-2812) -- and decode_mode_A_message (ref: :2722-2766) for Mode A/C replies.
// One lane per message: modes_message in, one fixed-layout modes_fields record out, message order kept.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/modes_hip.h"
#include "common.h"
#include "kernels.h"

namespace modes {

namespace {

// MODES_ACFLAGS_* (ref: src/aircraft.h:43-64)
enum : uint32_t {
  kAltitudeValid = 0x000001, kAog = 0x000002, kAogValid = 0x000004, kLlEvenValid = 0x000008, kLlOddValid = 0x000010,
  kCallsignValid = 0x000020, kAltitudeHaeValid = 0x000080, kHaeDeltaValid = 0x000100, kHeadingValid = 0x000200,
  kSquawkValid = 0x000400, kSpeedValid = 0x000800, kCategoryValid = 0x001000, kFromTisb = 0x004000,
  kVertRateValid = 0x008000, kNsEwSpdValid = 0x040000, kFsValid = 0x080000, kEwSpeedValid = 0x100000, kNsSpeedValid = 0x200000
};
// addrtype_t (ref: src/aircraft.h:86-97)
enum : uint16_t { kAddrAdsbIcao = 0, kAddrAdsbIcaoNt, kAddrAdsrIcao, kAddrTisbIcao, kAddrAdsbOther, kAddrAdsrOther, kAddrTisbTrackfile,
                  kAddrTisbOther, kAddrModeA, kAddrUnknown };
constexpr int kInvalidAltitude = -9999;

// 13-bit identity / altitude field -> Gillham bits in Mode A layout 0 A4 A2 A1 0 B4 B2 B1 0 C4 C2 C1 0 D4 D2 D1
// (ref: src/dump1090.c:1494-1514).  Field bit k goes to Gillham bit kDst[k]; bit 6 (X / M) goes nowhere.
__device__ __forceinline__ int gillham_of_id13 (int f)
{
  constexpr int kDst[13] = { 2, 10, 1, 9, 0, 8, -1, 14, 6, 13, 5, 12, 4 };
  int g = 0;
#pragma unroll
  for (int k = 0; k < 13; k++) if (kDst[k] >= 0) g |= ((f >> k) & 1) << kDst[k];
  return g;
}

// Mode A code -> Mode C altitude in 100 ft steps, or -9999 (ref: src/dump1090.c:2771-2812): C bits are a 3-bit Gray
// code of the 100 ft digit, D2 D4 A1 A2 A4 B1 B2 B4 an 8-bit Gray code of the 500 ft count
__device__ __forceinline__ int mode_c_of_mode_a (uint32_t a)
{
  if ((a & 0xFFFF8889u) || !(a & 0xF0u)) return kInvalidAltitude;
  uint32_t one = ((a >> 4) & 1 ? 7u : 0u) ^ ((a >> 5) & 1 ? 3u : 0u) ^ ((a >> 6) & 1 ? 1u : 0u);
  if ((one & 5u) == 5u) one ^= 2u;
  if (one > 5u) return kInvalidAltitude;
  // Gray -> binary: bit order D2 D4 A1 A2 A4 B1 B2 B4 from the most significant down
  const uint32_t gray = ((a >> 1) & 1) << 7 | ((a >> 2) & 1) << 6 | ((a >> 12) & 1) << 5 | ((a >> 13) & 1) << 4 |
                        ((a >> 14) & 1) << 3 | ((a >> 8) & 1) << 2 | ((a >> 9) & 1) << 1 | ((a >> 10) & 1);
  uint32_t five = gray;
  five ^= five >> 1; five ^= five >> 2; five ^= five >> 4;
  if (five & 1u) one = 6u - one;
  return int (five * 5u) + int (one) - 13;
}

__device__ __forceinline__ int altitude_of_gillham_or_q (int n25, bool q, int gillham_field, uint8_t *unit)
{
  *unit = 1;                                                   // MODES_UNIT_FEET
  if (q) return n25 * 25 - 1000;
  const int n = mode_c_of_mode_a (uint32_t (gillham_of_id13 (gillham_field)));
  return n < -12 ? kInvalidAltitude : 100 * n;
}

// ref: src/dump1090.c:1524-1559
__device__ __forceinline__ int altitude_of_ac13 (int f, uint8_t *unit)
{
  if (f & 0x40) { *unit = 2; return kInvalidAltitude; }       // metres: not decoded by the reference
  return altitude_of_gillham_or_q (((f & 0x1F80) >> 2) | ((f & 0x20) >> 1) | (f & 0x0F), f & 0x10, f, unit);
}
// ref: src/dump1090.c:1565-1592 (M = 0 inserted at bit 6 for the Gillham form)
__device__ __forceinline__ int altitude_of_ac12 (int f, uint8_t *unit)
{
  return altitude_of_gillham_or_q (((f & 0x0FE0) >> 1) | (f & 0x0F), f & 0x10, ((f & 0x0FC0) << 1) | (f & 0x3F), unit);
}

// ref: src/dump1090.c:1597-1623, as a table of segments (first code, speed at it, codes per knot as a shift / factor)
__device__ __forceinline__ int ground_speed_of_movement (int m)
{
  if (m > 123) return 199;
  if (m > 108) return (m - 108) * 5 + 100;
  if (m > 93) return (m - 93) * 2 + 70;
  if (m > 38) return m - 38 + 15;
  if (m > 12) return ((m - 11) >> 1) + 2;
  if (m > 8) return ((m - 6) >> 2) + 1;
  return 0;
}

// ref: src/dump1090.c:1844-1877: eight 6-bit characters; any character outside A-Z, 0-9, space, '@' voids the callsign;
// leading and trailing blanks are trimmed the way str_trim does (ref: src/misc.c:661-699: the last character stays)
__device__ __forceinline__ void callsign (modes_fields &o, const uint8_t *m)
{
  const uint64_t bits = (uint64_t (m[0]) << 40) | (uint64_t (m[1]) << 32) | (uint64_t (m[2]) << 24) | (uint64_t (m[3]) << 16) |
                         (uint64_t (m[4]) << 8) | m[5];
  if (!bits) return;
  char fl[8];
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const int c = int ((bits >> (42 - 6 * k)) & 63);
    char ch;
    if (c == 0) ch = '@';
    else if (c <= 26) ch = char ('A' + c - 1);
    else if (c == 32) ch = ' ';
    else if (c >= 48 && c <= 57) ch = char ('0' + c - 48);
    else { ch = '?'; ok = false; }
    fl[k] = ch;
  }
  if (!ok) return;
  o.AC_flags |= kCallsignValid;
  int a = 0, b = 8;                                            // [a, b): what str_trim leaves
  while (a < 7 && fl[a] == ' ') a++;
  while (b - 1 > a && fl[b - 1] == ' ') b--;
  for (int k = a; k < b; k++) o.flight[k - a] = fl[k];
}

}  // namespace

struct MsgIn {             // mirrors modes_message (include/modes_hip.h), 56 bytes
  uint64_t timestamp; uint8_t msg[14]; uint8_t msg_bits, msg_type; uint32_t addr, crc; int16_t score; uint8_t error_bits, phase;
  uint32_t block; double sig_level; uint32_t j, reserved;
};
static_assert (sizeof (MsgIn) == 56 && sizeof (modes_message) == 56, "modes_message layout");
static_assert (sizeof (modes_fields) == 120, "modes_fields layout");

__global__ __launch_bounds__ (256)
void fields_kernel (const MsgIn *msgs, uint32_t n, modes_fields *out)
{
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const MsgIn in = msgs[i];
  const uint8_t *msg = in.msg;
  modes_fields o;
  memset (&o, 0, sizeof o);
  const int df = in.msg_type;

  if (df == 32) {                                              // ref: :2722-2766
    const int a = (msg[0] << 8) | msg[1];
    o.addrtype = kAddrModeA;
    o.addr = uint32_t (a & 0xFF7F) | kNonIcao;
    o.identity = a & 0x7777;
    o.AC_flags = kSquawkValid;
    o.flight_status = a & 0x80;
    out[i] = o;
    return;
  }
  const uint32_t aa = (uint32_t (msg[1]) << 16) | (uint32_t (msg[2]) << 8) | msg[3];
  const bool es = df == 17 || df == 18;
  o.addr = (df == 11 || es) ? aa : in.crc;                     // ref: :2584-2585, 2440-2443, 2556
  if (df == 11) o.IID = uint8_t (in.crc & 0x7F);               // ref: :2459
  o.reliable = es && in.crc == 0;                              // ref: :2475, 2500-2503, 2540

  const int f13 = ((msg[2] << 8) | msg[3]) & 0x1FFF;
  if ((df == 0 || df == 4 || df == 16 || df == 20) && f13) {   // AC, ref: :2589-2599
    o.altitude = altitude_of_ac13 (f13, &o.unit);
    if (o.altitude != kInvalidAltitude) o.AC_flags |= kAltitudeValid;
  }
  if (df == 11 || df == 17) {                                  // CA, ref: :2603-2610
    o.capa = int8_t (msg[0] & 7);
    if (o.capa == 4) o.AC_flags |= kAogValid | kAog; else if (o.capa == 5) o.AC_flags |= kAogValid;
  }
  if (df == 18) o.cf = int8_t (msg[0] & 7);                    // CF, ref: :2615-2616
  if (df == 4 || df == 5 || df == 20 || df == 21) {            // FS, ref: :2621-2631
    const int fs = msg[0] & 7;
    o.flight_status = fs;
    o.AC_flags |= kFsValid | (fs <= 3 ? kAogValid | ((fs & 1) ? kAog : 0u) : 0u);
  }
  if ((df == 5 || df == 21) && f13) { o.AC_flags |= kSquawkValid; o.identity = gillham_of_id13 (f13); }      // ID, ref: :2634-2643
  if ((df == 20 || df == 21) && msg[4] == 0x20) callsign (o, msg + 5);                                        // MB, ref: :1882-1899
  if (df == 24)                                                // MD, ref: :2652-2653
    for (int k = 0; k < 10; k++) o.MD[k] = msg[1 + k];
  if (df == 0 || df == 16) o.AC_flags |= kAogValid | ((msg[0] & 4) ? kAog : 0u);                              // VS, ref: :2667-2672

  if (es) {                                                    // ME, ref: :1901-2249
    for (int k = 0; k < 7; k++) o.ME[k] = msg[4 + k];
    const int t = msg[4] >> 3, sub = t == 29 ? (msg[4] & 6) >> 1 : msg[4] & 7;
    o.ME_type = int8_t (t); o.ME_subtype = int8_t (sub);
    bool imf = false, body = true;
    if (df == 18) {                                            // ref: :1918-1963
      switch (o.cf) {
        case 0: o.addrtype = kAddrAdsbIcaoNt; break;
        case 1: o.addr |= kNonIcao; o.addrtype = kAddrAdsbOther; break;
        case 2: o.AC_flags |= kFromTisb; o.addrtype = kAddrTisbIcao; imf = true; break;
        case 3: o.AC_flags |= kFromTisb; o.addrtype = kAddrTisbIcao; if (msg[4] & 0x80) o.addr |= kNonIcao; body = false; break;
        case 5: o.AC_flags |= kFromTisb; o.addr |= kNonIcao; o.addrtype = kAddrTisbOther; break;
        case 6: o.addrtype = kAddrAdsrIcao; imf = true; break;
        default: o.addr |= kNonIcao; o.addrtype = kAddrUnknown; body = false; break;
      }
    }
    const int lat = ((msg[6] & 3) << 15) | (msg[7] << 7) | (msg[8] >> 1), lon = ((msg[8] & 1) << 16) | (msg[9] << 8) | msg[10];
    if (!body) {
    } else if (t >= 1 && t <= 4) {                             // identification and category, ref: :1971-1983
      callsign (o, msg + 5);
      o.category = uint32_t (((0x0E - t) << 4) | sub);
      o.AC_flags |= kCategoryValid;
    } else if (t == 19) {                                      // airborne velocity, ref: :1985-2087
      if (imf && (msg[5] & 0x80)) o.addr |= kNonIcao;
      o.AC_flags |= kAogValid;
      if (sub >= 1 && sub <= 4) {
        const int raw = ((msg[8] & 7) << 6) | (msg[9] >> 2);
        if (raw) { const int v = raw - 1; o.vert_rate = ((msg[8] & 8) ? -v : v) * 64; o.AC_flags |= kVertRateValid; }
      }
      if (sub == 1 || sub == 2) {
        const int sh = sub == 2 ? 2 : 0;                       // supersonic: 4 kt units
        const int ew_raw = ((msg[5] & 3) << 8) | msg[6], ns_raw = ((msg[7] & 0x7F) << 3) | (msg[8] >> 5);
        int ew = (ew_raw - 1) << sh, ns = (ns_raw - 1) << sh;
        if (msg[5] & 4) ew = -ew;
        if (msg[7] & 0x80) ns = -ns;
        if (ew_raw) { o.AC_flags |= kEwSpeedValid; o.EW_velocity = ew; o.EW_dir = (msg[5] >> 2) & 1; }
        if (ns_raw) { o.AC_flags |= kNsSpeedValid; o.NS_velocity = ns; o.NS_dir = msg[7] >> 7; }
        if (ew_raw && ns_raw) {
          o.AC_flags |= kSpeedValid | kHeadingValid | kNsEwSpdValid;
          o.velocity = sqrt (double (ns * ns + ew * ew) + 0.5);
          if (o.velocity > 0.0001) {                           // SMALL_VAL, ref: src/geo.h:37
            double h = atan2 (double (ew), double (ns)) * 180.0 / 3.14159265358979323846;
            if (h < 0.0) h += 360.0;
            o.heading = h;
          }
        }
      } else if (sub == 3 || sub == 4) {
        const int raw = ((msg[7] & 0x7F) << 3) | (msg[8] >> 5);
        if (raw) { o.AC_flags |= kSpeedValid; o.velocity = double ((raw - 1) << (sub == 4 ? 2 : 0)); }
        if (msg[5] & 4) { o.AC_flags |= kHeadingValid; o.heading = double (((((msg[5] & 3) << 8) | msg[6]) * 45) >> 7); }
      }
      if (msg[10]) { o.AC_flags |= kHaeDeltaValid; o.HAE_delta = ((msg[10] & 0x80) ? -25 : 25) * ((msg[10] & 0x7F) - 1); }
    } else if (t >= 5 && t <= 8) {                             // surface position, ref: :2090-2118
      if (imf && (msg[6] & 8)) o.addr |= kNonIcao;
      o.AC_flags |= kAogValid | kAog | ((msg[6] & 4) ? kLlOddValid : kLlEvenValid);
      o.raw_latitude = lat; o.raw_longitude = lon;
      const int mv = ((msg[4] << 4) | (msg[5] >> 4)) & 0x7F;
      if (mv && mv < 125) { o.AC_flags |= kSpeedValid; o.velocity = double (ground_speed_of_movement (mv)); }
      if (msg[5] & 8) { o.AC_flags |= kHeadingValid; o.heading = double (((((msg[5] << 4) | (msg[6] >> 4)) & 0x7F) * 45) >> 4); }
      o.nuc_p = uint8_t (14 - t);
    } else if (t == 0 || (t >= 9 && t <= 18) || (t >= 20 && t <= 22)) {      // airborne position, ref: :2120-2186
      const int ac12 = ((msg[5] << 4) | (msg[6] >> 4)) & 0x0FFF;
      if (imf && (msg[4] & 1)) o.addr |= kNonIcao;
      o.AC_flags |= kAogValid;
      if (t != 0) {
        o.raw_latitude = lat; o.raw_longitude = lon;
        const bool filtered = ac12 == 0 && lon == 0 && (lat & 0x0FFF) == 0 && t == 15;      // Modes.stat.cpr_filtered
        if (!filtered) o.AC_flags |= (msg[6] & 4) ? kLlOddValid : kLlEvenValid;
      }
      if (ac12) {
        const int alt = altitude_of_ac12 (ac12, &o.unit);
        if (t >= 20) { o.altitude_HAE = alt; if (alt != kInvalidAltitude) o.AC_flags |= kAltitudeHaeValid; }
        else         { o.altitude = alt;     if (alt != kInvalidAltitude) o.AC_flags |= kAltitudeValid; }
      }
      o.nuc_p = uint8_t ((t == 0 || t == 18 || t == 22) ? 0 : t < 18 ? 18 - t : 29 - t);
    } else if (t == 23) {                                      // test message, ref: :2188-2202
      const int f = (((msg[5] << 8) | msg[6]) & 0xFFF1) >> 3;
      if (sub == 7 && f) { o.AC_flags |= kSquawkValid; o.identity = gillham_of_id13 (f); }
    } else if (t == 28) {                                      // aircraft status, ref: :2207-2223
      if (sub == 1) {
        const int f = ((msg[5] << 8) | msg[6]) & 0x1FFF;
        if (f) { o.AC_flags |= kSquawkValid; o.identity = gillham_of_id13 (f); }
        if (imf && (msg[10] & 1)) o.addr |= kNonIcao;
      }
    } else if (t == 31) {                                      // operational status, ref: :2235-2241
      if (imf && (msg[10] & 1)) o.addr |= kNonIcao;
    } else if (t != 24 && t != 29 && t != 30)                  // ref: :2244-2247
      o.reliable = 0;
  }
  out[i] = o;
}

hipError_t launch_fields (const void *msgs, uint32_t n, modes_fields *out, hipStream_t stream)
{
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL (fields_kernel, dim3 ((n + 255) / 256), dim3 (256), 0, stream, static_cast<const MsgIn *> (msgs), n, out);
  return hipGetLastError ();
}

}  // namespace modes
