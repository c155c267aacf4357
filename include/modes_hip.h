/*
 * modes_hip.h -- C ABI of libmodes_hip.so, the MI355X (gfx950) engine for the raw-IQ hot
 * path of gvanem/Dump1090:  IQ -> uint16 magnitude -> Mode S / Mode A/C demodulation ->
 * CRC-24 syndrome + 1/2-bit repair.
 *
 * Part 1 are the reference's own entry points for this path, same names, same signatures,
 * same struct layouts, so the reference can link this library instead of convert.c,
 * demod-*.c and crc.c.  Part 2 is the one extra call a drop-in host
 * makes to hand over the callbacks the reference's demodulators reach into.  Part 3 is the
 * bulk API: whole recordings resident in HBM, block geometry of `bin_read` reproduced on
 * the device, one ordered message list out.
 *
 * Plain C, plain pointers and sizes.  "ref:" = file:line under the reference's root.
 */
#ifndef MODES_HIP_H
#define MODES_HIP_H

#include <stdint.h>
#include <stddef.h>
#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden */
#endif

/* ===================================================================================== */
/* Part 1 -- the reference's interface for this path                                      */
/* ===================================================================================== */

/* ref: src/convert.h:12-19 */
typedef struct convert_state {
  float       DC_A;
  float       DC_B;
  float       z1_I;
  float       z1_Q;
  const char *func_name;
  const char *description;
} convert_state;

/* ref: src/convert.h:21-26 */
typedef enum convert_format {
  INPUT_ILLEGAL = 0,
  INPUT_UC8,
  INPUT_SC16,
  INPUT_SC16Q11
} convert_format;

/* ref: src/convert.h:28-32 */
typedef void (*convert_func) (const void *iq_input, uint16_t *mag_output, unsigned nsamples,
                              convert_state *state, double *out_power);

/* ref: src/convert.h:34-38, src/convert.c:310-365.  Returns NULL when no converter in the
 * table supports (format, filter_dc, compute_power). */
convert_func convert_init (convert_format format, double sample_rate, bool filter_dc,
                           bool compute_power, convert_state **state_p);
/* ref: src/convert.h:40, src/convert.c:375-381 */
const char  *convert_format_name (convert_format f);
/* ref: src/convert.h:42, src/convert.c:367-373 */
void         convert_cleanup (convert_state **state_p);

/* ref: src/fifo.h:15-18 */
typedef enum mag_buf_flags { MAGBUF_ZERO = 0, MAGBUF_DISCONTINUOUS = 1 } mag_buf_flags;

/* ref: src/fifo.h:50-63 (u_int == unsigned int) */
typedef struct mag_buf {
  uint16_t       *data;
  unsigned        total_length;
  unsigned        valid_length;
  unsigned        overlap;
  uint64_t        sample_timestamp;
  uint64_t        sys_timestamp;
  mag_buf_flags   flags;
  double          mean_level;
  double          mean_power;
  unsigned        dropped;
  struct mag_buf *next;
} mag_buf;

/* ref: src/demod.h:7-12.  Synchronous: every callback of a block has fired on return. */
void demod_2000 (const mag_buf *mag);
void demod_2400 (const mag_buf *mag);
void demod_2400_AC (const mag_buf *mag);
void demod_8000 (const mag_buf *mag);
int  demod_8000_alloc (void);      /* non-zero = ok */
void demod_8000_free (void);

/* ref: src/crc.h:11-23 */
#define MODES_MAX_BITERRORS 2
typedef struct errorinfo {
  uint32_t syndrome;
  int      errors;
  int8_t   bit [MODES_MAX_BITERRORS];
} errorinfo;

void       crc_init (int fix_bits);
void       crc_exit (void);
uint32_t   crc_checksum (const uint8_t *msg, int bitlen);
errorinfo *crc_checksum_diagnose (uint32_t syndrome, int bitlen);
void       crc_checksum_fix (uint8_t *msg, const errorinfo *info);

/* ===================================================================================== */
/* Part 2 -- binding a drop-in host                                                       */
/* ===================================================================================== */

/* The reference's demodulators call back into dump1090.c (ref: src/misc.h:866-871,900) and
 * read a few `Modes` fields (ref: src/misc.h:496-642).  A host hands them over once.
 * `modeS_message` is opaque to the library: it is told the size and the offsets of the
 * members the demodulators themselves touch (ref: src/misc.h:721-806).
 * Any callback left NULL is served by the library's own implementation of the same rule
 * (score / accept / ICAO cache of ref: src/dump1090.c:1438-1476,2372-2720,3279-3365), and
 * accepted messages are then queued for modes_hip_take_messages(). */
typedef struct modes_hip_host {
  size_t   mm_size;                 /* sizeof(modeS_message)                    */
  size_t   off_timestamp_msg;       /* offsetof(modeS_message, timestamp_msg)   */
  size_t   off_sys_timestamp_msg;
  size_t   off_sig_level;           /* double                                   */
  size_t   off_score;               /* int                                      */
  size_t   off_error_bits;          /* int                                      */
  size_t   off_AC_flags;            /* uint32_t                                 */
  size_t   off_addr;                /* uint32_t                                 */
  int      (*modeS_message_score) (const uint8_t *msg, int valid_bits);
  int      (*decode_mode_S_message) (void *mm, const uint8_t *msg);
  void     (*decode_mode_A_message) (void *mm, int mode_A);
  void     (*modeS_user_message) (void *mm);
  int      (*modeS_message_len_by_type) (int type);
  int64_t  (*receiveclock_ms_elapsed) (uint64_t t1, uint64_t t2);
  int      phase_enhance;           /* Modes.phase_enhance  */
  int      mode_AC;                 /* Modes.mode_AC        */
  int      measure_noise;           /* Modes.measure_noise  */
  int      fix_bits;                /* error_correct_1 + error_correct_2, used when crc_init was not called */
  int      device;                  /* HIP device ordinal   */
} modes_hip_host;

/* Returns 0, or a negative modes_status.  Passing NULL restores the built-in defaults and clears the built-in
 * state (ICAO cache, queued messages, block counter). */
int modes_hip_bind (const modes_hip_host *host);

/* ===================================================================================== */
/* Part 3 -- bulk engine                                                                  */
/* ===================================================================================== */

typedef enum modes_status {
  MODES_OK = 0,
  MODES_E_ARG = -1,          /* bad argument / unsupported combination      */
  MODES_E_NODEVICE = -2,     /* no usable HIP device                        */
  MODES_E_HIP = -3,          /* a HIP call failed; see modes_last_error()   */
  MODES_E_NOMEM = -4,
  MODES_E_OVERFLOW = -5      /* a fixed-capacity device buffer overflowed   */
} modes_status;

const char *modes_last_error (void);

/* One accepted message, in the order modeS_user_message() would have seen it. */
typedef struct modes_message {
  uint64_t timestamp;        /* mm.timestamp_msg, 12 MHz                           */
  uint8_t  msg[14];          /* after CRC repair; msg_bits/8 bytes are meaningful   */
  uint8_t  msg_bits;         /* 56 / 112 (16 for Mode A/C)                          */
  uint8_t  msg_type;         /* DF, 24 for DF24..31, 32 for Mode A/C                */
  uint32_t addr;             /* mm.addr incl. the non-ICAO flag bit 24              */
  uint32_t crc;              /* syndrome before repair (mm.CRC)                     */
  int16_t  score;            /* score that selected the message, -1 where n/a       */
  uint8_t  error_bits;
  uint8_t  phase;            /* demod_2400: phase 4..8; demod_2000: 1 = phase-corrected retry */
  uint32_t block;            /* index of the 256 KiB block (bin_read call)          */
  double   sig_level;        /* mm.sig_level                                        */
  uint32_t j;                /* sample offset inside the block's mag_buf            */
  uint32_t reserved;
} modes_message;             /* 56 bytes */

typedef struct modes_stats {
  uint64_t valid_preamble, demodulated, out_of_phase, demod_modeac;
  uint64_t CRC_good, CRC_bad, CRC_fixed, CRC_single_bit_fix, CRC_two_bits_fix;
  uint64_t demod_rejected_unknown;
  uint64_t candidates;       /* records that left the scan kernel               */
  uint64_t replay_rounds;    /* fixed-point rounds of the device replay; 8 MS/s: ranges walked before a run met the one before (0: walked all) */
  uint64_t scan_retries;     /* scans repeated with more candidate slots per strip */
} modes_stats;

typedef struct modes_engine_config {
  uint32_t sample_rate;      /* 2000000 / 2400000 / 8000000 (ref: src/dump1090.c:702-732) */
  int32_t  input_format;     /* convert_format                                              */
  int32_t  filter_dc;        /* as convert_init()                                           */
  int32_t  compute_power;    /* as convert_init() (Modes.measure_noise)                     */
  int32_t  fix_bits;         /* error_correct_1 + error_correct_2 (ref: src/dump1090.c:964) */
  int32_t  phase_enhance;
  int32_t  mode_ac;          /* run demod_2400_AC after demod_2400 / detect_mode_A in demod_2000 */
  int32_t  device;           /* HIP device ordinal                                          */
  int32_t  host_replay;      /* 1 = do the ordered accept pass on the host (debug/compare)  */
  int32_t  reserved[7];
} modes_engine_config;

typedef struct modes_engine modes_engine;

int  modes_engine_create (const modes_engine_config *cfg, modes_engine **out);
void modes_engine_destroy (modes_engine *e);

/* Seed / read the 1024-slot ICAO cache (ref: src/dump1090.c:1438-1476): one address per slot,
 * 0 = empty.  Used to chain shards of one recording across GPUs. */
int  modes_engine_set_icao (modes_engine *e, const uint32_t *slots1024);
int  modes_engine_get_icao (modes_engine *e, uint32_t *slots1024);

/* Decode `nbytes` of IQ that are ALREADY in device memory (d_iq), exactly as
 * `dump1090 --infile` would (ref: src/infile.c:116-197): 256 KiB blocks, overlap carried
 * from block to block, the stream's first block preceded by zeros.
 *   first_sample : Modes.sample_counter at d_iq[0] (multiple of the block length)
 *   d_halo       : device pointer to the `overlap` magnitudes that precede d_iq[0] (the tail a
 *                  previous shard produced), or NULL for zeros
 *   stream       : hipStream_t to run on (NULL = the default stream)
 * Results stay in the engine until the next run. */
int  modes_engine_run_device (modes_engine *e, const void *d_iq, size_t nbytes,
                              uint64_t first_sample, const uint16_t *d_halo, void *stream);
/* Same, input in host memory (copied with hipMemcpyAsync first). */
int  modes_engine_run_host (modes_engine *e, const void *iq, size_t nbytes, uint64_t first_sample);

/* Only the scan (convert + detect + slice + CRC) kernel of the last configuration, `reps`
 * times on `stream`, bracketed by HIP events on that stream.  ms_out receives the average
 * kernel duration.  For roofline measurements. */
int  modes_engine_time_scan (modes_engine *e, const void *d_iq, size_t nbytes, int reps, void *stream, float *ms_out);

/* Duration of the dominant kernel of the last run, from HIP events recorded on the launch stream around it:
 * main_ms = the scan kernel (2.0 / 2.4 MS/s) or the window kernel (8 MS/s); second_ms = the 8 MS/s trigger kernel, else 0. */
int  modes_engine_kernel_ms (const modes_engine *e, float *main_ms, float *second_ms);

/* The last run's messages as RAW text, one "*<hex>;\n" per message in order, formatted on the device:
 * upper_case = 0: what raw_out_send() sends to RAW-out clients (ref: src/raw-sbs.c:86-97);
 * upper_case = 1: what modeS_user_message() prints with --raw (ref: src/dump1090.c:3086-3097).
 * *text points into host memory owned by the engine (valid until the next call), NUL-terminated. */
int  modes_engine_format_raw (modes_engine *e, int upper_case, const char **text, size_t *len);

/* Sharded decodes hand the ICAO cache from shard to shard and repeat until no entry cache changes.  What the scan
 * kernel found does not depend on the cache: after modes_engine_set_icao() this repeats only the ordered accept pass
 * and the emit of the LAST run (2.0 / 2.4 MS/s, ordered pass on the device, no mode_ac; the run's d_iq must still be
 * valid: the signal level is read from it).  Results replace those of the last run. */
int  modes_engine_rerun_accept (modes_engine *e, void *stream);
/* 1 when modes_engine_rerun_accept() would be accepted for the last run, 0 when the shard has to be decoded again with
 * modes_engine_run_device() (mode_ac, phase_enhance and host_replay engines, 8 MS/s, or no successful run yet) */
int  modes_engine_can_rerun_accept (const modes_engine *e);
/* scan kernel launches and ordered passes since the engine was created */
int  modes_engine_get_counters (const modes_engine *e, uint64_t *scan_launches, uint64_t *accept_passes);

/* Modes.stat.signal_power_* / peak_signal_power / strong_signal_count / noise_power_* as demod_2400 keeps them
 * (ref: src/demod-2400.c:498-505,524-530), for the last run.  The noise figures need mag_buf.mean_power of every block:
 * they are formed when compute_power is set (Modes.measure_noise) and the converter is not a DC-filter one
 * (noise_valid = 0 otherwise).  Floats agree with the reference within 1e-12 relative (sums of doubles in message order). */
typedef struct modes_power_stats {
  double   signal_power_sum;
  uint64_t signal_power_count;
  double   peak_signal_power;
  uint64_t strong_signal_count;
  double   noise_power_sum;
  uint64_t noise_power_count;
  int32_t  noise_valid, reserved;
} modes_power_stats;
int  modes_engine_get_power_stats (modes_engine *e, modes_power_stats *out);

/* ---- demod_8000: what the reference keeps in file-static variables between calls (ref: src/demod-8000.c:80-89,
 * 186-220,509-522), for shards of one 8 MS/s recording.  A run that does not start the stream (first_sample != 0) must
 * be given the state the shard before it left; modes_engine_get_d8_state() reads what the last run leaves behind.
 *   acc         d8m_phase_av_acc.  Forming it means walking every step of the shard in order, so it is only done on
 *               request (want_acc); a shard only reads it if it holds steps whose trigger depends on the running
 *               average (amb_steps != 0 after its run: near-silent input).
 *   pending     step (< 0, relative to the next shard's first step) at which a match window still open was opened; 1 = none
 *   trig_tail   trigger decisions of the last 64 steps (the search resumes 55 steps before a window's end)
 *   dprev       the last 1928 four-sample differences (d8m_dbuf's overlap)                                        */
typedef struct modes_d8_state {
  int32_t  acc;
  int32_t  acc_valid;        /* out: acc was formed; in: 0 = unknown (fails with MODES_E_ARG if the shard needs it) */
  int64_t  pending;
  uint64_t trig_tail;
  uint64_t amb_steps;        /* out only: steps of the last run whose trigger depended on the running average */
  int32_t  dprev[1928];
} modes_d8_state;
int  modes_engine_set_d8_state (modes_engine *e, const modes_d8_state *in);      /* applies to the next run only */
/* Sharded 8 MS/s decode without a round per shard.  Where the window chain enters a shard depends on every shard before
 * it (where the input is never silent, windows follow each other at 113 steps and never re-synchronise), but only
 * through `pending`; the differences and the trigger history in front of a shard are functions of the samples in front
 * of it.  modes_engine_d8_chain_map() forms the shard's trigger planes (with the dprev / trig_tail / acc of the state
 * set for the run, which stays set) and reports, for each of the MODES_D8_CHAIN_STATES values `pending` can enter with
 * -- index 0: none, index i: -i -- the `pending` the shard would hand on.  The ranks exchange their maps once and each
 * follows them from the stream's start to its own entry (dump1090_amd/shard.py).  The planes are kept for the run that
 * follows on the same samples; modes_engine_d8_reuse_planes() makes the same promise for a repeated run (same d_iq,
 * nbytes, first_sample, d_halo and differences carried in: only then is anything reused).
 * No counterpart in the reference, which is one sequential stream (ref: src/demod-8000.c:80-89,493-522). */
#define MODES_D8_CHAIN_STATES 168
int  modes_engine_d8_chain_map (modes_engine *e, const void *d_iq, size_t nbytes, uint64_t first_sample, const uint16_t *d_halo,
                                void *stream, int32_t map[MODES_D8_CHAIN_STATES]);
int  modes_engine_d8_reuse_planes (modes_engine *e);
/* 1 + the last step of the last 8 MS/s run whose trigger depended on the running average (0: none did); steps = samples / 8. */
int  modes_engine_d8_last_ambiguous (const modes_engine *e, uint64_t *step_plus_1);
int  modes_engine_get_d8_state (modes_engine *e, modes_d8_state *out, int want_acc, void *stream);

/* Field decode of the last run's messages, batched on the device: what decode_mode_S_message() does behind its CRC /
 * accept switch (ref: src/dump1090.c:2580-2675 with decode_extended_squitter :1901-2249 and the AC / ID / movement /
 * callsign helpers :1494-1623,1844-1899,2771-2812) and decode_mode_A_message() (:2722-2766) for Mode A/C replies.
 * One record per message, in message order; members carry the names and meanings of modeS_message
 * (ref: src/misc.h:721-806).  Members the reference never writes on this path (aircraft_type, odd_flag, UTC_flag,
 * vert_rate_source/_sign, DR_status, UM_status, BDS, MB, MV, position, source) are not part of the record. */
typedef struct modes_fields {
  uint32_t AC_flags;        /* MODES_ACFLAGS_* (ref: src/aircraft.h:43-64)                       */
  uint32_t addr;            /* mm.addr, MODES_NON_ICAO_ADDRESS (bit 24) included                 */
  uint32_t category;
  uint16_t addrtype;        /* addrtype_t (ref: src/aircraft.h:86-97)                            */
  uint8_t  reliable, unit;  /* unit: 0 unset, MODES_UNIT_FEET = 1, MODES_UNIT_METERS = 2         */
  int8_t   capa; uint8_t IID; int8_t ME_type, ME_subtype, cf, reserved0; uint8_t nuc_p, EW_dir, NS_dir, reserved1[3];
  int32_t  flight_status;
  int32_t  identity, altitude, altitude_HAE, HAE_delta, raw_latitude, raw_longitude, EW_velocity, NS_velocity, vert_rate;
  double   heading, velocity;
  char     flight[9];       /* NUL-terminated */
  uint8_t  ME[7], MD[10], reserved2[6];
} modes_fields;             /* 120 bytes */
/* *out points into host memory owned by the engine (valid until the next call on it). */
int  modes_engine_decode_fields (modes_engine *e, const modes_fields **out, size_t *n, void *stream);
/* The same on a caller's list in host memory (messages that did not come from a run: RAW / SBS input, replays); only
 * msg, msg_type, crc and error_bits of every modes_message are read.  Runs on the current device. */
int  modes_decode_fields_host (const modes_message *msgs, size_t n, modes_fields *out);

size_t               modes_engine_num_messages (const modes_engine *e);
const modes_message *modes_engine_messages (const modes_engine *e);      /* host memory */
int                  modes_engine_get_stats (const modes_engine *e, modes_stats *out);
/* mag_buf.mean_power of every block of the last run, as the selected converter reports it (ref: src/infile.c:168);
 * filled when mode_ac is set at 2.4 MS/s.  Copies up to `max` values, returns the number of blocks. */
size_t               modes_engine_block_power (modes_engine *e, double *out, size_t max);
/* the last block's trailing `overlap` magnitudes (device memory), to pass as d_halo of the next shard */
const uint16_t      *modes_engine_tail_halo (const modes_engine *e);

/* ---- float converters with the DC block on (filter_dc = 1; ref: src/convert.c:112-245) ----
 * The filter state z1_I / z1_Q of convert_state runs through the whole stream.  An engine keeps it between runs the
 * way convert_state does between convert_func calls (a new engine starts at 0, 0); set it to decode another stream.
 * d_iq must be 16-byte aligned. */
int  modes_engine_set_dc_state (modes_engine *e, float z1_I, float z1_Q);
int  modes_engine_get_dc_state (const modes_engine *e, float *z1_I, float *z1_Q);
/* For shards decoded in parallel: tells the NEXT run that d_iq[0] is preceded in device memory by `lead_samples`
 * (a multiple of 256) valid samples of the same stream and that the filter state in front of them is unknown.  The
 * filter is then settled on the lead-in: the run fails with MODES_E_ARG if it is too short to do so (about
 * 35 / DC_A samples are needed: 13.4 M at 2.4 MS/s).  A lead-in that covers the overlap also replaces d_halo.
 * warm_samples != 0 overrides the warm-up length chunks inside a run use (a performance knob: results do not
 * depend on it); 0 = default. */
int  modes_engine_set_dc_lead (modes_engine *e, uint64_t lead_samples, uint64_t warm_samples);
typedef struct modes_dc_info {
  uint32_t chunks;           /* independent chains of the last run                            */
  uint32_t rerun_chunks;     /* chunks re-run as a whole, in order, because an earlier chunk's end state changed under them */
  uint64_t chunk_samples, warm_samples;
  float    chain_ms;         /* duration of the chain kernel (HIP events on the launch stream) */
  uint32_t prefix_reruns;    /* chunks that were not entered with their predecessor's end state: their first samples were stepped again, up to where that chain met the first pass's */
  uint64_t guess_rounds;     /* 1024-sample rounds of all chains of the first pass                */
  uint64_t plain_rounds;     /* ragged ends stepped one by one                                     */
  uint64_t chain_steps;      /* filter steps of all chains of the first pass (warm-up included)  */
} modes_dc_info;
int  modes_engine_get_dc_info (const modes_engine *e, modes_dc_info *out);
/* The magnitudes of the last filter_dc run in device memory: `overlap` entries in front of the shard's first sample,
 * then one per sample (what convert_func wrote into the mag_bufs, back to back). */
const uint16_t *modes_engine_dc_magnitudes (const modes_engine *e);
/* ... `count` of them starting at entry `first`, copied to host memory */
int  modes_engine_read_dc_magnitudes (modes_engine *e, uint16_t *out, size_t first, size_t count);

/* The `overlap` magnitudes that precede sample `at_sample` of the IQ in d_iq (zeros before sample 0), written to
 * d_out (device): the d_halo of a shard that starts at `at_sample`, for callers that decode shards in parallel and so
 * cannot wait for the previous shard's tail.  at_sample must be a multiple of the block length. */
int  modes_engine_halo_at (modes_engine *e, const void *d_iq, size_t nbytes, uint64_t at_sample, uint16_t *d_out, void *stream);

/* ---- live receivers: rx_callback() + mag_buf FIFO + main_data_loop() around the device path ----
 * (ref: src/dump1090.c:1181-1282,1367-1400, src/fifo.c:195-211).  A reader thread hands over IQ blocks with
 * modes_live_rx(); a worker owned by the object converts and demodulates them on the device in arrival order, carrying
 * the overlap from block to block; modes_live_take() drains the decoded messages.  Blocks are converted before they are
 * queued for the demodulator (the reference enqueues first, src/dump1090.c:1273-1282) and staged in pinned host buffers.
 * Block timestamps come from a sample counter that only counts accepted blocks; when every buffer is busy the block is
 * dropped and the next accepted block is discontinuous: its overlap is zeros (ref: src/dump1090.c:1187-1207,
 * src/fifo.c:197-203).  In the messages `block` counts the accepted blocks and `reserved` carries
 * mm.sys_timestamp_msg in ms since modes_live_open(). */
typedef struct modes_live modes_live;
typedef struct modes_live_stats {
  uint64_t blocks, samples;              /* accepted                                   */
  uint64_t dropped_blocks, dropped_samples, discontinuities;
  uint64_t blocks_decoded, messages;
} modes_live_stats;
/* max_block_bytes: the largest block modes_live_rx() will see (0 = 256 KiB, MODES_ASYNC_BUF_SIZE, also the maximum);
 * nbuffers: pinned buffers in the ring (2..64). */
int    modes_live_open (const modes_engine_config *cfg, uint32_t max_block_bytes, uint32_t nbuffers, modes_live **out);
void   modes_live_close (modes_live *l);
int    modes_live_rx (modes_live *l, const void *iq, uint32_t len);    /* 1 = queued, 0 = dropped, < 0 = error; never waits for the device */
size_t modes_live_take (modes_live *l, modes_message *out, size_t max, int wait_ms);
int    modes_live_flush (modes_live *l);                               /* until every queued block is decoded */
int    modes_live_hold (modes_live *l, int hold);                      /* 1: the worker leaves the queue alone (a stalled consumer); 0: resume */
int    modes_live_get_stats (modes_live *l, modes_live_stats *out);

/* --strip as a streaming pass (ref: src/dump1090.c:3423-3448, strip_mode_loop; README "What is --strip mode?"):
 * uc8 IQ pairs whose I and Q both lie within `level` of 127 are dropped from the 33rd pair of such a run on, all other
 * pairs are copied in order.  d_iq (16-byte aligned) and d_out (room for nbytes) are device memory; *out_bytes receives
 * the bytes written.  *run_io (may be NULL = 0) is the reference's counter of quiet pairs in a row, carried from one
 * piece of a stream to the next (it comes back saturated at 32, which behaves like any longer run). */
int  modes_strip_device (const void *d_iq, size_t nbytes, int level, uint64_t *run_io, void *d_out, size_t *out_bytes,
                         void *stream);

/* Stand-alone converter on device buffers: what convert_func computes, for n samples. */
int  modes_convert_device (int converter_id /* row of converters_table, ref: src/convert.c:301-308 */,
                           const void *d_iq, uint16_t *d_mag, size_t nsamples,
                           convert_state *state, double *out_power, void *stream);

/* The same for callers that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  NO COUNTERPART IN THE REFERENCE.  It runs the kernels of modes_convert_device(), but calls no hipMalloc /
 * hipFree / allocating routine and never waits for the stream: the scratch those kernels need (table, power
 * accumulator, the float converters' z1 sequences and squared magnitudes) is a fixed array in device memory, one per
 * GPU, held until the process ends (about 3 MiB), and the results modes_convert_device() returns through host memory
 * are written to *d_result in DEVICE memory by the last kernel of the call.
 *   - nsamples > MODES_NOALLOC_MAX_SAMPLES is refused with MODES_E_ARG.
 *   - DC_A, DC_B, z1_I, z1_Q: the fields of convert_state (DC_A = 0, DC_B = 1: DC block off); ignored by the integer
 *     converters.
 *   - runs on the CURRENT device.  The first call on a device uploads the tables (blocking) and must not be made
 *     under capture.
 *   - every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two threads) race. */
#define MODES_NOALLOC_MAX_SAMPLES 262144
typedef struct modes_convert_result {
  double out_power;          /* what convert_func stores to *out_power       */
  float  z1_I, z1_Q;         /* convert_state.z1_I / z1_Q after the call     */
} modes_convert_result;
int  modes_convert_device_noalloc (int converter_id, const void *d_iq, uint16_t *d_mag, size_t nsamples,
                                   float DC_A, float DC_B, float z1_I, float z1_Q,
                                   modes_convert_result *d_result /* device memory, may be NULL */, void *stream);

/* messages queued by the drop-in demod_*() when no modeS_user_message callback is bound */
size_t modes_hip_take_messages (modes_message *out, size_t max);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
