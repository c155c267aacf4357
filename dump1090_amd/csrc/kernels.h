// Launchers implemented in kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include "common.h"

struct modes_fields;

namespace modes {
hipError_t upload_constants (const uint32_t *single_bit, const float *uc8_float);
hipError_t launch_scan2400 (const ScanParams &p, hipStream_t stream);
hipError_t launch_tail (const ScanParams &p, uint64_t at, uint16_t *out, hipStream_t stream);
// kernels_2000.hip
hipError_t upload_constants_2000 (const uint32_t *single_bit, const float *uc8_float);
hipError_t launch_scan2000 (const ScanParams &p, hipStream_t stream);
// kernels_ac.hip
hipError_t upload_constants_ac (const uint32_t *single_bit, const float *uc8_float);
hipError_t launch_scanac2400 (const ScanParams &p, hipStream_t stream);
hipError_t launch_noise_level (const double *mean_power, uint32_t nblocks, uint32_t *noise_level, hipStream_t stream);
hipError_t launch_block_power (const ScanParams &p, uint32_t nblocks, double *mean_power, hipStream_t stream);
// kernels_8000.hip
hipError_t upload_constants_8000 (const uint32_t *single_bit, const float *uc8_float);
hipError_t launch_d8_trig (const D8Params &q, uint64_t step_limit, hipStream_t stream, uint32_t c_begin = 0, uint32_t c_count = 0);   // chunks [c_begin, + c_count), 0: to the end
hipError_t launch_d8_acc (const D8Params &q, uint64_t limit, int32_t *acc_out, hipStream_t stream);
hipError_t launch_d8_last_amb (const uint64_t *amb, uint64_t w_begin, uint64_t w_end, unsigned long long *limit, uint32_t *flag_host, hipStream_t stream);
hipError_t launch_d8_chain_table (const D8Params &q, hipStream_t stream, uint32_t c_begin = 0, uint32_t c_count = 0);
hipError_t launch_d8_chain_exit (const D8Params &q, int32_t *exit_of, hipStream_t stream);
hipError_t launch_d8_chain (const D8Params &q, unsigned long long trig_tail, long long search0, long long *pre,
                            long long *pending_out, hipStream_t stream, uint32_t c_begin = 0, uint32_t c_count = 0);
hipError_t launch_d8_windows (const D8Params &q, uint32_t nch, hipStream_t stream, int which = 3);   // chunks q.chunk0 .. + nch; 1: inside a block, 2: at a block edge
hipError_t launch_d8_tail (const D8Params &q, int32_t *dnext, hipStream_t stream);
hipError_t launch_d8_seed (uint32_t *announce, const uint32_t *addr, uint32_t n, hipStream_t stream);
hipError_t launch_d8_count (const D8Params &q, uint32_t nch, uint32_t *chunk_rec, hipStream_t stream);
hipError_t launch_d8_gather (const D8Params &q, uint32_t nch, const uint64_t *win_off, const uint64_t *rec_off, D8Window *win_out,
                             Record *rec_out, hipStream_t stream, const uint64_t *range_tot = nullptr);
hipError_t launch_d8_offsets (const uint32_t *chunk_rec, const uint32_t *win_count, uint32_t nchunks, uint32_t c0, uint32_t n,
                              uint64_t *win_off, uint64_t *rec_off, uint64_t cap_w, uint64_t cap_r, uint64_t *tot, uint64_t *tot_host,
                              hipStream_t stream);
hipError_t launch_convert (int conv, const uint8_t *iq, uint16_t *mag, size_t n, const uint16_t *lut256,
                           unsigned long long *power_u64, hipStream_t stream);
hipError_t launch_rssi2400 (int conv, const uint8_t *iq, uint64_t nsamples, const uint16_t *halo, uint32_t ov,
                            uint32_t block_samples, const uint16_t *lut256, const uint64_t *pos, const uint32_t *len,
                            uint64_t *sum_out, uint32_t nmsg, hipStream_t stream);
hipError_t launch_convert_float (int conv, const uint8_t *iq, uint16_t *mag, size_t n, float A, float B, float zI0, float zQ0,
                                 float *zI, float *zQ, float *mag_sq, float *z_final, float *power_out, hipStream_t stream);
hipError_t launch_raw_format (const void *msgs, uint32_t n, uint32_t *len, uint64_t *chunk_sum, uint64_t *chunk_vp,
                              uint64_t *chunk_ovf, uint64_t *totals, uint64_t *offsets, int upper, char *text, int pass, hipStream_t stream);
// kernels_fields.hip
hipError_t launch_fields (const void *msgs, uint32_t n, struct ::modes_fields *out, hipStream_t stream);
// kernels_dc.hip
hipError_t launch_dc_chain (int conv, const DcParams &p, uint32_t ndesc, hipStream_t stream);
hipError_t launch_dc_block_power (const float *sq, uint64_t n, uint32_t block_samples, uint32_t nblocks, double *mean_power, hipStream_t stream);
hipError_t launch_dc_linear (int conv, const uint8_t *iq, uint64_t n, double A, const double *wtab /* [1025]: B^(1023 - j), B^1024 */, double *out, hipStream_t stream);
hipError_t launch_accept2400 (const AcceptParams &p, hipStream_t stream);
hipError_t launch_snapshots (const uint32_t *delta, const uint32_t *blk_wrote, uint32_t nblocks, const uint32_t *cache_in, uint32_t *chunk_last,
                             uint32_t *chunk_entry, uint32_t *snap_in, uint32_t *cache_out, uint32_t *dirty, const uint32_t *changed, uint32_t *marked,
                             uint32_t *marked_host, int first_round, hipStream_t stream);      // marked_host (may be null): `marked` again, in pinned host memory
hipError_t launch_emit2400 (const EmitParams &p, hipStream_t stream);
hipError_t launch_strip_offsets (const uint32_t *acc_counts, uint32_t nstrips, const uint64_t *chunk_sum, uint64_t *offsets, hipStream_t stream);
hipError_t launch_fill_u32 (uint32_t *p, size_t n, uint32_t v, hipStream_t stream);
hipError_t launch_copy_u32 (uint32_t *dst, const uint32_t *src, size_t n, hipStream_t stream);
hipError_t launch_blk_stats_sum (const uint32_t *blk_stats, uint32_t nblocks, unsigned long long *out, hipStream_t stream);
hipError_t launch_count_scan (const uint32_t *counts, const uint32_t *vp_counts, uint32_t nstrips, uint64_t *chunk_sum,
                              uint64_t *chunk_vp, uint64_t *chunk_ovf, uint64_t *totals, hipStream_t stream);
hipError_t launch_gather (const uint32_t *counts, uint32_t nstrips, const uint64_t *chunk_sum, const Record *records,
                          uint32_t cap, Record *packed, uint64_t *packed_pos, uint64_t *offsets, hipStream_t stream);
}
