/*
 * modes_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, sequential, single thread) of the raw-IQ hot path of
 * gvanem/Dump1090:  convert.c -> demod-2000/2400/2400AC/8000.c -> crc.c, plus the
 * score / accept / ICAO-cache part of dump1090.c that the demodulators call back
 * into, plus the `bin_read` + `fifo_enqueue` block/overlap geometry.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or
 * load this.  The product (dump1090_amd/) never does.
 *
 * Pinning status: see the header comment of modes_oracle.c.
 *
 * All "ref:" citations are relative to /root/reference/.
 */
#ifndef MODES_ORACLE_H
#define MODES_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ref: src/convert.h:21-26 */
enum { ORC_INPUT_ILLEGAL = 0, ORC_INPUT_UC8 = 1, ORC_INPUT_SC16 = 2, ORC_INPUT_SC16Q11 = 3 };

/* converter ids, in the order of converters_table[] (ref: src/convert.c:301-308) */
enum {
  ORC_CONV_NONE = -1,
  ORC_CONV_UC8_NODC_NOPOWER = 0,
  ORC_CONV_UC8_NODC_POWER   = 1,
  ORC_CONV_UC8_GENERIC      = 2,
  ORC_CONV_SC16_NODC_NOPOWER = 3,
  ORC_CONV_SC16_GENERIC     = 4,
  ORC_CONV_SC16Q11_GENERIC  = 5
};

/* ref: src/convert.h:12-19 (name strings dropped) */
typedef struct orc_convert_state {
  float DC_A, DC_B, z1_I, z1_Q;
} orc_convert_state;

/* ref: src/fifo.h:50-63 */
typedef struct orc_mag_buf {
  uint16_t *data;
  uint32_t  total_length;
  uint32_t  valid_length;
  uint32_t  overlap;
  uint64_t  sample_timestamp;
  uint64_t  sys_timestamp;
  uint32_t  flags;
  double    mean_level;
  double    mean_power;
  uint32_t  dropped;
} orc_mag_buf;

/* One message as handed to modeS_user_message() (ref: src/misc.h:721-806, subset). */
typedef struct orc_msg {
  uint64_t timestamp;      /* mm.timestamp_msg, 12 MHz clock */
  uint64_t sys_timestamp;  /* mm.sys_timestamp_msg with block sys_timestamp = 0 */
  uint8_t  msg[14];        /* mm.msg; only msg_bits/8 bytes are defined */
  uint8_t  pad[2];
  int32_t  msg_bits;
  int32_t  msg_type;       /* DF (24 for DF24..31, 32 for Mode A/C) */
  int32_t  error_bits;
  int32_t  score;          /* score that selected it (demod_2400/8000); -1 where n/a */
  int32_t  phase;          /* demod_2400: bestphase; else 0 */
  uint32_t addr;           /* mm.addr, incl. MODES_NON_ICAO_ADDRESS bit */
  uint32_t crc;            /* mm.CRC (syndrome before repair) */
  int32_t  block;          /* index of the mag_buf (call number) */
  int32_t  j;              /* demod_2000/2400/2400AC: loop index; demod_8000: position */
  double   sig_level;
} orc_msg;

typedef struct orc_stats {
  uint64_t valid_preamble, demodulated, out_of_phase, demod_modeac;
  uint64_t CRC_good, CRC_bad, CRC_fixed, CRC_single_bit_fix, CRC_two_bits_fix;
  uint64_t demod_rejected_unknown;
  uint64_t signal_power_count, strong_signal_count, noise_power_count;
  uint64_t d8m_drift_resets;
  uint64_t score_calls;
  double   signal_power_sum, peak_signal_power, noise_power_sum;
} orc_stats;

typedef struct orc_ctx orc_ctx;

/* fix_bits = error_correct_1 + error_correct_2 as passed to crc_init (ref: src/dump1090.c:964) */
orc_ctx *orc_create (uint32_t sample_rate, int fix_bits, int phase_enhance, int mode_ac, int measure_noise);
void     orc_destroy (orc_ctx *c);
void     orc_reset (orc_ctx *c);                 /* clears ICAO cache, stats, messages, demod_8000 state */

/* ---- convert.c ---- */
int  orc_convert_select (int format, int filter_dc, int compute_power);  /* table walk of convert_init; -1 = none */
void orc_convert_state_init (orc_convert_state *st, double sample_rate, int filter_dc);
void orc_convert (int conv_id, const void *iq, uint16_t *mag, unsigned nsamples, orc_convert_state *st, double *out_power);
const uint16_t *orc_mag_lut (void);              /* 129*129 entries */
const uint16_t *orc_log10_lut (void);            /* 65536 entries   */

/* ---- crc.c ---- */
uint32_t orc_crc_checksum (const uint8_t *msg, int bits);
uint32_t orc_crc_table_entry (int i);            /* crc_table[i]           */
uint32_t orc_single_bit_syndrome (int i);        /* single_bit_syndrome[i] */
int      orc_error_table_size (const orc_ctx *c, int bits);
/* entry k of the 56- or 112-bit table: returns syndrome, fills errors/bit0/bit1 */
uint32_t orc_error_table_entry (const orc_ctx *c, int bits, int k, int *errors, int *bit0, int *bit1);
/* crc_checksum_diagnose: returns errors (0..2) or -1 for NULL; fills bit positions */
int      orc_crc_diagnose (const orc_ctx *c, uint32_t syndrome, int bits, int *bit0, int *bit1);

/* ---- dump1090.c score / accept ---- */
int  orc_message_score (orc_ctx *c, const uint8_t *msg, int valid_bits);
/* returns decode rc (0, -1, -2, -3); on success fills *out (timestamps untouched) */
int  orc_decode_message (orc_ctx *c, const uint8_t *msg, orc_msg *out);
void orc_icao_add (orc_ctx *c, uint32_t addr);
int  orc_icao_test (const orc_ctx *c, uint32_t addr);

/* ---- demodulators (each appends to the context's message list) ---- */
void orc_demod_2000 (orc_ctx *c, const orc_mag_buf *mag);
void orc_demod_2400 (orc_ctx *c, const orc_mag_buf *mag);
void orc_demod_2400_AC (orc_ctx *c, const orc_mag_buf *mag);
void orc_demod_8000 (orc_ctx *c, const orc_mag_buf *mag);

/* demod_2000 building blocks exposed for the tools/frames.js golden vectors */
void orc_apply_phase_correction (uint16_t *payload /* payload[-1 .. 16+224-1] accessible */);
/* Runs the 112-bit Manchester slicer of demod_2000 on `payload` (first data sample);
 * returns msg_len after the DF consistency check (0, 56 or 112); msg[14] is in/out (stale bytes kept). */
int  orc_slice_2000 (const uint16_t *preamble, const uint16_t *payload, uint8_t *msg, int *errors_out);

/* ---- bin_read + fifo geometry (ref: src/infile.c:116-197, src/fifo.c:195-211) ----
 * which_demod: 2000, 2400, 8000; also_ac != 0 runs demod_2400_AC after demod_2400 per block.
 * block_crc32 (may be NULL): receives one CRC-32 (zlib polynomial) of each block's freshly
 * converted magnitudes data[overlap .. valid_length).
 * Returns number of blocks processed. */
int orc_run_stream (orc_ctx *c, const void *iq, size_t nbytes, int format,
                    int filter_dc, int compute_power, int which_demod, int also_ac,
                    uint32_t *block_crc32, int max_blocks, double *block_power);

/* sharded runs (test support): where the next orc_run_stream starts, ICAO cache hand-over, the overlap it leaves */
void     orc_stream_start (orc_ctx *c, uint64_t counter, const uint16_t *tail, uint32_t n);
void     orc_icao_export (const orc_ctx *c, uint32_t *slots1024);
void     orc_icao_import (orc_ctx *c, const uint32_t *slots1024);
uint32_t orc_last_tail (uint16_t *out);

/* ---- field decode of an accepted message (ref: src/dump1090.c:2580-2675 and helpers; see modes_oracle.c) ----
 * msg: the message bytes after CRC repair; msg_type as in orc_msg (24 for DF24..31, 32 for Mode A/C); crc = mm.CRC. */
typedef struct orc_fields {
  uint32_t AC_flags;        /* MODES_ACFLAGS_* (ref: src/aircraft.h:43-64) */
  uint32_t addr;            /* mm.addr, MODES_NON_ICAO_ADDRESS included */
  uint32_t category;
  uint16_t addrtype;        /* addrtype_t (ref: src/aircraft.h:86-97) */
  uint8_t  reliable, unit;  /* unit: 0 unset, 1 feet, 2 meters */
  int8_t   capa; uint8_t IID; int8_t ME_type, ME_subtype, cf, flight_status; uint8_t nuc_p, EW_dir, NS_dir, pad0[3];
  int32_t  flight_status_raw;   /* mm.flight_status as the int it is (Mode A/C stores 0 / 128 there) */
  int32_t  identity, altitude, altitude_HAE, HAE_delta, raw_latitude, raw_longitude, EW_velocity, NS_velocity, vert_rate;
  double   heading, velocity;
  char     flight[9];
  uint8_t  ME[7], MD[10], pad1[6];
} orc_fields;
void orc_decode_fields (const uint8_t *msg, int msg_type, uint32_t crc, int error_bits, orc_fields *out);
size_t orc_sizeof_fields (void);

/* ---- --strip (ref: src/dump1090.c:3423-3448): returns the bytes written to `out` (room for nbytes) ---- */
size_t   orc_strip (const uint8_t *in, size_t nbytes, int level, uint64_t *run_io, uint8_t *out);

size_t          orc_num_messages (const orc_ctx *c);
const orc_msg  *orc_messages (const orc_ctx *c);
void            orc_get_stats (const orc_ctx *c, orc_stats *out);
size_t          orc_sizeof_msg (void);

#ifdef __cplusplus
}
#endif
#endif
