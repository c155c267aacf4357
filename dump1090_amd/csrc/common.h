// Shared definitions between the host side of libmodes_hip.so and its gfx950 kernels.
#pragma once
#include <stdint.h>

namespace modes {

constexpr int kShortBytes = 7, kLongBytes = 14, kShortBits = 56, kLongBits = 112;
constexpr uint32_t kAsyncBufSize = 256 * 1024;       // ref: src/misc.h:649 (bytes per bin_read block)
constexpr uint32_t kIcaoSlots = 1024;                // ref: src/misc.h:664
constexpr uint32_t kNonIcao = 1u << 24;              // ref: src/misc.h:699

// Converter rows of converters_table[] (ref: src/convert.c:301-308)
enum Conv : int {
  kConvUc8Lut = 0, kConvUc8LutPower = 1, kConvUc8Generic = 2,
  kConvSc16Int = 3, kConvSc16Generic = 4, kConvSc16Q11Generic = 5,
  kConvMagU16 = 6,    // not a converter: the input already is the uint16 magnitude buffer of a mag_buf
  // scan2400_kernel only: kConvUc8Lut with the LDS tile holding, per sample, the rank of its magnitude among the byte
  // table's values (one byte instead of two: half the LDS, more waves per SIMD); magnitudes are looked up where needed
  kConvUc8LutR8 = 7,
  // scan2400_kernel only: kConvUc8Lut with the LDS tile holding the RAW input bytes, brought in by LDS-DMA
  // (global_load_lds_dwordx4: no register staging, every byte of a workgroup's span in flight at once).  Stage B looks
  // the ranks of its own 33 bytes up (byte -> rank, what only orders samples), later stages the magnitudes (byte ->
  // table value) of the few positions they look at.
  kConvUc8LutRaw = 8
};

// ---- scan kernel geometry (2.4 MS/s) -------------------------------------------------
// One wavefront owns one strip of kStrip consecutive scan positions of the stream.
constexpr int kStrip = 2048;
constexpr int kWords = kStrip / 64;   // 64-position words per strip, one per lane
constexpr int kQ1Cap = 1024;          // stage-B candidates per strip (50 % of the positions)
constexpr int kQ2Cap = 256;           // stage-C survivors per strip
constexpr int kLook2400 = 320;     // samples a position may look ahead (<= 290 needed; ref: src/demod-2400.c:339-412,488-494)
constexpr int kTilePad = 16;       // slack so global loads can start 16-sample aligned

// ---- scan kernel geometry (2.0 MS/s) -------------------------------------------------
constexpr int kLook2000 = 256;        // samples a position may look ahead (16 + 224 payload, Mode A ~75; ref: src/demod-2000.c:757,768)
constexpr int kBack2000 = 16;         // tile slack in front: pre[-1] of the phase correction + 16-sample load alignment
constexpr int kTile2000 = kStrip + kLook2000 + kBack2000 + 16;
constexpr int kPlaneWords2000 = (kStrip + kLook2000) / 64;   // 36 words of GT/LT bits
constexpr int kLook2000PE = 512;      // phase_enhance: the retry behind an accepted message starts up to 239 samples later
constexpr int kCap2000 = 64;          // records per strip: every valid preamble leaves one
constexpr int kCap2000PE = 192;       // ... up to three with phase_enhance
constexpr int kQ2Cap2000 = 256;

// Record.phase carries these for the 2.0 MS/s path
constexpr uint8_t kInfoDemodMask = 3;     // Modes.stat.demodulated increments of the slice (0..2)
constexpr uint8_t kInfoDecode = 4;        // reached decode_mode_S_message (msg_len, SNR, errors <= 3; ref: src/demod-2000.c:954)
constexpr uint8_t kInfoStaticReject = 8;  // ... which fails whatever the ICAO cache holds (rc -2 / -3)
constexpr uint8_t kInfoStale = 16;        // message bytes the slicer never wrote would be read
constexpr uint8_t kInfoOutOfPhase = 32;   // detect_out_of_phase() != 0 at this preamble
// bits 6..7: which iteration of demod_2000's loop the record stands for
constexpr uint8_t kKindPrimary = 0;       // the slice at a valid preamble
constexpr uint8_t kKindRetrySame = 1;     // phase-corrected retry at the same position (first try not accepted)
constexpr uint8_t kKindRetryAfter = 2;    // phase-corrected slice behind an accepted message that needed a repair
constexpr uint8_t kKindModeA = 3;         // Mode A/C reply (Modes.mode_AC)

constexpr int kLookAC = 80;           // demod_2400_AC reads up to f1 + 69 (ref: src/demod-2400AC.c:145-173)

// ---- 8 MS/s path (demod_8000; ref: src/demod-8000.c:14-62) ----------------------------
constexpr int kD8LookAhead = 544, kD8LookBack = 1384, kD8Overlap = 1928, kD8Win = 168, kD8Net = 113;
constexpr int kD8Ambig = 4576;           // 2*max > 3*av is settled without av above this: av <= 50000000 >> 14 = 3051
constexpr int kD8SaneMax = 50000000;
constexpr int kD8StripSteps = 256;       // bit steps (8 samples each) per wave of the trigger kernel
#ifndef D8_CHUNK_STEPS
#define D8_CHUNK_STEPS 65536
#endif
constexpr int kD8ChunkSteps = D8_CHUNK_STEPS;     // bit steps per chunk of the window-chain kernels
#define D8_TABLE_BYTES_PER_CHUNK (128 + 2 * 16 * 128)
#define D8_TABLE_BYTES(nchunks) (size_t (nchunks) * D8_TABLE_BYTES_PER_CHUNK + (size_t (nchunks) / 64 + 1) * 128)   /* + the tables of groups of 64 chunks */
constexpr int kD8WinCap = kD8ChunkSteps / kD8Net + 6;   // openings are >= kD8Net steps apart (580 in a chunk); the first chunk also lists up to three carried in from the steps before it
constexpr int kD8Cand = 32;              // candidate messages per window: 16 bit shifts x (short, long)

// Candidate classes (what the ordered accept pass needs to know about a record)
enum Cls : uint8_t {
  kClsAP = 0,      // DF0/4/5/16/24: address = CRC, score 1000 / -1      (ref: src/dump1090.c:3302-3307)
  kClsAP2 = 1,     // DF20/21:       address = CRC, score 1000 / -2      (ref: src/dump1090.c:3355-3359)
  kClsDF11 = 2,    // table hit on syndrome & 0xFFFF80, <= 1 error       (ref: src/dump1090.c:3309-3338)
  kClsES = 3,      // DF17/18 with a table hit                           (ref: src/dump1090.c:3340-3353)
  kClsModeA = 4    // Mode A/C reply: always emitted                     (ref: src/demod-2000.c:677-701)
};

// One candidate that survived every state-independent rejection.  32 bytes.
struct alignas(16) Record {
  uint16_t off;        // position inside the strip
  uint8_t  phase;      // try_phase 4..8 (demod_2400)
  uint8_t  df;         // msg[0] >> 3
  uint8_t  cls;        // Cls
  int8_t   nerr;       // errorinfo.errors (0..2)
  int8_t   bit0, bit1; // errorinfo.bit[]
  uint32_t crc;        // crc_checksum over modeS_message_len_by_type(df) bits
  uint8_t  msg[14];
  uint8_t  flags;      // filled by the accept pass: bit0 accepted
  uint8_t  pad[5];
};
static_assert(sizeof(Record) == 32, "Record must be 32 bytes");

// Error table entry on the device: sorted by syndrome.
struct ErrEntry { uint32_t syndrome; int8_t nerr, bit0, bit1, pad; };
static_assert(sizeof(ErrEntry) == 8, "ErrEntry must be 8 bytes");

struct ScanParams {
  const uint8_t  *iq;          // device, whole shard
  uint64_t        nsamples;    // samples in the shard == scan positions
  uint64_t        navail;      // kConvMagU16 only: valid magnitudes in `iq` (mag_buf.valid_length)
  const uint16_t *halo;        // `ov` magnitudes that precede sample 0, or nullptr (zeros)
  uint32_t        ov;          // Modes.trailing_samples (ref: src/dump1090.c:734)
  uint32_t        block_samples;
  uint32_t        block_strips; // block_samples / kStrip when the block is whole strips, else 0 (one block: only strip 0 starts one)
  int32_t         conv;        // Conv
  int32_t         phase_enhance;
  const ErrEntry *short_tab;  int32_t short_n;
  const ErrEntry *long_tab;   int32_t long_n;
  const uint32_t *bloom_short, *bloom_long;   // 65536-bit filters over the table syndromes
  const uint16_t *lut256;      // as-written uc8 table: mag_lut[byte] (ref: src/convert.c:29)
  Record         *records;     // [nstrips][cap]
  uint32_t       *counts;      // [nstrips]
  uint32_t        cap;
  uint32_t        nstrips;
  uint32_t       *vp_counts;   // [nstrips] positions that reached scoring (Modes.stat.valid_preamble before skip-ahead)
  int32_t         write_records;   // 0 = timing variant that keeps everything but the final store
  int32_t         debug_stop;      // profiling aid: 1..3 = leave after stage A/B/C (results are then meaningless)
  uint64_t       *vp_bitmap;       // [nstrips][64]: bit = position reached scoring (Modes.stat.valid_preamble), or nullptr
  // 2.0 MS/s path
  const uint16_t *log10_lut;       // 65536 entries (ref: src/dump1090.c:410-423)
  const uint32_t *crc_table;       // 256 entries (ref: src/crc.c:43-54)
  int32_t         mode_ac;
  // demod_2400_AC
  const uint32_t *noise_level;     // [blocks]: (unsigned)((mean_power + sqrt(mean_power - mean_level^2)) * 65535 + 0.5)
  // demod_2000 / demod_2400 scan kernels: the launch covers strips [strip_begin, strip_end) of the shard (strip_end 0 = nstrips)
  uint32_t        strip_begin, strip_end;
  // kConvUc8LutR8: rank256[byte] = rank of lut256[byte] among the table's distinct values, rankmag[rank] = that value
  // (0xFFFF behind the last rank); *inexact is set when a halo magnitude is none of the table's values
  // kConvUc8LutRaw: rank256 and lut256; the tile holds input bytes, a halo magnitude is mapped back to a byte with that value
  const uint8_t  *rank256;
  const uint16_t *rankmag;
  uint32_t       *inexact;
};
constexpr uint32_t kOv2400 = 326;                               // (8 + 112 + 16) us at 2.4 MS/s (ref: src/dump1090.c:734)
constexpr int kRawShift2400 = int ((16 - kOv2400 % 16) % 16);   // sample of V[16 k] sits this many bytes into a 16-byte chunk of input

}  // namespace modes

namespace modes {

// ---- ordered accept pass on the device ------------------------------------------------------
constexpr uint32_t kDeltaWritten = 0x80000000u;   // marks an ICAO slot a block wrote (addresses use 25 bits)

enum BlkStat : int { kStDemodulated = 0, kStOutOfPhase, kStGood, kStBad, kStFix1, kStFix2, kStUnknown, kStSkippedVp, kStAccepted, kStVp, kStModeAC, kBlkStats = 12 };

struct AcceptParams {
  Record         *records;       // [nstrips][cap]; flags/score are written back
  const uint32_t *counts;        // [nstrips]
  uint32_t        cap, nstrips, strips_per_block, nblocks;
  uint64_t        nsamples;
  uint32_t        block_samples;
  const uint32_t *snap_in;       // [nblocks][1024] ICAO cache at the start of each block
  uint32_t       *delta;         // [nblocks][1024] kDeltaWritten | address for the slots the block wrote
  uint32_t       *acc_counts;    // [nstrips] accepted messages per strip
  uint32_t       *blk_stats;     // [nblocks][kBlkStats]
  const uint64_t *vp_bitmap;     // may be null
  uint32_t       *changed;       // set to 1 when a delta entry differs from the previous round
  const uint32_t *dirty;         // [nblocks] 1 = the block's entry cache changed since the round before (null: run every block)
  int32_t         mode;          // 2400 or 2000: which demodulator's loop is replayed
  uint32_t        blk_begin, blk_end;   // the launch covers these blocks of the shard (blk_end 0 = nblocks)
  // First round of a pass: every block starts from `cache_in` (snap_in holds nothing yet) and nothing of the round
  // before is compared.  blk_wrote[block] = the block wrote a cache slot (only then its delta row is valid).
  const uint32_t *cache_in;
  uint32_t       *blk_wrote;
  int32_t         first_round;
};

// One completed match window of demod_8000 (ref: src/demod-8000.c:331-420)
struct D8Window {
  int64_t  s;            // bit step at which the window opened
  int64_t  g_short;      // first sample (look-ahead domain) of the 72-bit search string
  int64_t  g_long;       // ... of the 128-bit search string
  uint32_t nrec;         // candidates that are not -2 whatever the ICAO cache holds
  uint32_t pad;
};

struct D8Params {
  ScanParams      sp;            // input geometry (iq, nsamples, conv, ov, block_samples, halo, lut256, tables)
  uint64_t        T;             // samples processed: sum over blocks of (valid - overlap) & ~7
  uint64_t        nsteps;        // T / 8
  int32_t         bshift;        // block_samples == 1 << bshift (63: one block)
  const int32_t  *dprev;         // [1928] differences that precede the stream (a previous call's tail), or null = zeros
  uint64_t       *trig;          // [nsteps/64 + 2] bit n: the search would start a window at step n
  uint64_t       *amb;           // same shape: 0 < max <= kD8Ambig, needs the running average
  uint32_t       *amb_count;     // [1] number of ambiguous steps
  uint32_t       *p0;            // [nsteps] phase[0] per step   (written only when write_arrays)
  uint32_t       *mx;            // [nsteps] max phase per step  (written only when write_arrays)
  int32_t         write_arrays;
  int32_t         acc0;          // d8m_phase_av_acc at the start of the stream
  int64_t         pending;       // step (< 0) at which a window carried in from a previous call opened, or 1 = none
  uint8_t        *table;         // [nchunks][128] chain exit for every entry offset, then [nchunks][2][16][128]: the same and the
                                 // number of openings for the sixteenths of every chunk (D8_TABLE_BYTES_PER_CHUNK per chunk in all), then
                                 // [nchunks / 64 + 1][128]: the tables of groups of 64 chunks -- D8_TABLE_BYTES (nchunks)
  uint8_t        *entry;         // [nchunks + 1]
  int64_t        *win_start;     // [nchunks][kD8WinCap]
  uint32_t       *win_count;     // [nchunks]
  D8Window       *windows;       // [nchunks][kD8WinCap]
  Record         *records;       // [nchunks][kD8WinCap][kD8Cand]
  uint32_t        nchunks;
  uint32_t        chunk0;        // first chunk of the launch: window / count / gather kernels run on chunk ranges
  uint32_t       *announce;      // [2^24 / 32] addresses that can enter the ICAO cache during the run, or null = no filter
  // The magnitude plane: d8_trig_kernel, which converts every sample once, leaves V[x] here (x = scan position), and the
  // window kernel reads its tiles from it instead of converting the samples a second time.  Null = no plane.
  uint16_t       *vmag;          // [vcap], 16-byte aligned
  uint64_t        vcap;          // entries allocated, a multiple of 8
};

// ---- float converters with the DC block on, whole recording (kernels_dc.hip) ----------------
// One chain of the filter per chunk: it starts at sample `start` (<= the chunk's first sample, may reach back
// into the lead-in in front of the shard) from the bracket [lo, hi] per channel; lo == hi when the state is known.
#ifndef DC_BATCH
#define DC_BATCH 4096
#endif
constexpr int kDcBatch = DC_BATCH;     // samples a workgroup stages per round: sixteen per thread
constexpr int kDcCkpt = 4 * kDcBatch; // samples between two checkpoints of a chunk's chains (DcParams::ckpt)
constexpr int kDcSeg = 65536;          // samples per segment of the real-arithmetic estimate (dc_linear_kernel)
struct DcChunk {
  long long start;                     // first sample the chain steps (<= the chunk's first sample)
  float zIlo, zIhi, zQlo, zQhi;        // bracket in front of `start`
  uint32_t chunk;                      // which chunk of the shard
  uint32_t slot;                       // where the results of this chain go (z_start, z_end, unmerged, closed_at, z_close)
  long long stop;                      // sample to stop in front of (a multiple of kDcBatch inside the chunk); 0: the chunk's end;
                                       // -1: the chunk's end, or the first checkpoint where the chain has met the first pass's (DcParams::ckpt)
};
static_assert(sizeof(DcChunk) == 40, "DcChunk must be 40 bytes");

struct DcParams {
  const uint8_t  *iq;          // sample 0 of the shard (16-byte aligned); `start` may be negative
  uint64_t        n;           // samples in the shard
  uint64_t        chunk_len;   // samples per chunk, a multiple of kDcBatch
  float           A, B;        // convert_state.DC_A, DC_B (ref: src/convert.c:349-353)
  const DcChunk  *desc;        // one per launched wave
  uint16_t       *mag;         // [n] magnitudes of the shard's samples (4-byte aligned)
  float          *mag_sq;      // [n] the clamped squared magnitudes `power += mag_sq` adds up (ref: src/convert.c:142-146), or null (16-byte aligned)
  float          *z_start;     // [slots][4] I-lo, I-hi, Q-lo, Q-hi in front of the chunk's first sample
  float          *z_end;       // [slots][2] z1_I, z1_Q behind the last sample stepped
  uint32_t       *unmerged;    // [slots] 1 = the bracket was still open at the chunk's first sample
  long long      *closed_at;   // [slots] first batch of the chunk in front of which both brackets were closed (-1: never)
  float          *z_close;     // [slots][2] z1_I, z1_Q in front of that batch
  float          *ckpt;        // [chunks][nck][2] z1_I, z1_Q of the first pass in front of every kDcCkpt-th sample of a chunk
  uint32_t        nck;
  unsigned long long *ticks;   // [nchunks][16] profiling aid (wave 0: lifetime in clock ticks, guess rounds, plain rounds, ticks in the chain, at barriers); may be null
  unsigned long long *rounds;  // [2] guess rounds (256 samples tried), plain rounds (64 samples stepped one by one); may be null
};

struct EmitParams {
  const Record   *records; const uint32_t *counts; const uint32_t *acc_counts; const uint64_t *offsets;     // messages of all earlier strips
  uint32_t        cap, nstrips;
  uint32_t        strip_begin, strip_end;   // the launch covers these strips (the message list is emitted in parts)
  // stream geometry for timestamps and the signal level
  const uint8_t  *iq; uint64_t nsamples, navail; const uint16_t *halo; uint32_t ov, block_samples; int32_t conv;
  const uint16_t *lut256;
  uint64_t        first_sample; uint32_t sample_rate;
  void           *messages;      // modes_message[total]
  int32_t         mode;          // 2400 or 2000
  uint32_t        block_shift;   // log2 (block_samples): set by launch_emit2400, which refuses other block sizes
};

}  // namespace modes
