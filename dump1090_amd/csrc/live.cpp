// Live-path shim: what rx_callback() + the mag_buf FIFO + main_data_loop() do around the hot path
// (ref: src/dump1090.c:1181-1282,1367-1400, src/fifo.c:195-211), for receivers that hand over IQ blocks from a reader
// thread (RTL-SDR, SDRplay, rtl_tcp ...).
//
//   reader thread:  modes_live_rx()   copies the block into the next free pinned slot of a small ring and returns
//   worker thread:  per slot, in order: H2D copy, converter + demodulator on the device (the bulk engine on one block,
//                   overlap carried from block to block as fifo_enqueue() does), messages appended to the output queue
//   main thread:    modes_live_take() drains the queue
//
// Differences to the reference, on purpose: the block is converted BEFORE it is queued for the demodulator (the
// reference enqueues first, ref: src/dump1090.c:1273-1282); the slots are pinned host memory, so the
// H2D copy of one block runs while the reader fills the next.  Kept: the block timestamps from a sample counter that
// only counts accepted blocks, dropped-block accounting when every slot is busy, MAGBUF_DISCONTINUOUS zeroing the
// overlap of the block behind a drop (ref: src/dump1090.c:1187-1207, src/fifo.c:197-203).
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/modes_hip.h"
#include "common.h"

namespace modes { int set_error (int code, const char *fmt, ...); }
using namespace modes;

struct modes_live {
  modes_engine_config cfg;
  modes_engine *eng = nullptr;
  uint32_t bps = 2, max_bytes = 0, ov = 0;
  struct Slot { uint8_t *pin = nullptr; uint32_t len = 0; bool discontinuous = false; uint32_t dropped = 0; uint64_t counter = 0; int64_t sys_ms = 0; };
  std::vector<Slot> slots;
  uint8_t *d_block = nullptr;                 // one block of IQ on the device
  uint16_t *d_halo = nullptr;                 // the overlap in front of it (what fifo.c keeps in overlap_buffer)
  hipStream_t stream = nullptr;
  // ring state (mu)
  std::mutex mu;
  std::condition_variable cv_work, cv_out, cv_idle;
  std::deque<int> queued;                     // slots waiting for the worker, in arrival order
  std::vector<int> free_slots;
  std::deque<modes_message> out;
  bool stop = false, hold = false, busy = false, failed = false;
  std::string error;
  // reader-side state (only touched under mu)
  uint64_t sample_counter = 0;                // Modes.sample_counter
  uint32_t dropped = 0;                       // samples dropped since the last accepted block
  modes_live_stats st;
  // worker-side state
  bool have_tail = false;
  modes_d8_state d8;
  bool have_d8 = false;
  uint32_t block_index = 0;
  std::chrono::steady_clock::time_point t0;
  std::thread worker;
};

namespace {

int64_t now_ms (const modes_live *l)
{
  return std::chrono::duration_cast<std::chrono::milliseconds> (std::chrono::steady_clock::now () - l->t0).count ();
}

// one block through the engine; false = the engine failed (modes_last_error holds why)
bool decode_block (modes_live *l, const modes_live::Slot &s, std::vector<modes_message> &found)
{
  if (hipSetDevice (l->cfg.device) != hipSuccess) return false;
  if (hipMemcpyAsync (l->d_block, s.pin, s.len, hipMemcpyHostToDevice, l->stream) != hipSuccess) return false;
  const bool use_halo = l->have_tail && !s.discontinuous;          // ref: src/fifo.c:197-207
  if (l->cfg.sample_rate == 8000000 && l->have_d8 && modes_engine_set_d8_state (l->eng, &l->d8) != MODES_OK) return false;
  if (modes_engine_run_device (l->eng, l->d_block, s.len, s.counter, use_halo ? l->d_halo : nullptr, l->stream) != MODES_OK) return false;
  // the tail of this block is the overlap of the next (ref: src/fifo.c:209-211)
  if (hipMemcpyAsync (l->d_halo, modes_engine_tail_halo (l->eng), l->ov * sizeof (uint16_t), hipMemcpyDeviceToDevice, l->stream) != hipSuccess) return false;
  if (hipStreamSynchronize (l->stream) != hipSuccess) return false;
  l->have_tail = true;
  if (l->cfg.sample_rate == 8000000) {
    if (modes_engine_get_d8_state (l->eng, &l->d8, 1, l->stream) != MODES_OK) return false;
    l->have_d8 = true;
  }
  const size_t n = modes_engine_num_messages (l->eng);
  const modes_message *m = modes_engine_messages (l->eng);
  const uint64_t block_ts = uint64_t ((double (s.counter) * 12E6) / double (l->cfg.sample_rate));
  for (size_t i = 0; i < n; i++) {
    modes_message x = m[i];
    x.block = l->block_index;
    // mm.sys_timestamp_msg = block's system time + receiveclock_ms_elapsed (ref: src/demod-2400.c:470-472, src/misc.c:1259-1262)
    // (signed throughout, like the reference's int64_t receiveclock_ms_elapsed: demod_8000 can stamp a message in front of
    // its block; a block taken in within its own duration of modes_live_open starts at 0, not before it)
    const int64_t at_ms = s.sys_ms + (int64_t (x.timestamp) - int64_t (block_ts)) / 12000;
    x.reserved = uint32_t (at_ms < 0 ? 0 : at_ms);
    found.push_back (x);
  }
  l->block_index++;
  return true;
}

void worker_main (modes_live *l)
{
  std::vector<modes_message> found;
  for (;;) {
    int slot;
    {
      std::unique_lock<std::mutex> lk (l->mu);
      l->cv_work.wait (lk, [&] { return l->stop || (!l->hold && !l->queued.empty ()); });
      if (l->stop && (l->queued.empty () || l->hold)) break;
      slot = l->queued.front ();
      l->queued.pop_front ();
      l->busy = true;
    }
    found.clear ();
    const bool ok = l->failed ? false : decode_block (l, l->slots[slot], found);
    {
      std::lock_guard<std::mutex> lk (l->mu);
      if (!ok && !l->failed) { l->failed = true; l->error = modes_last_error (); }
      for (const modes_message &m : found) l->out.push_back (m);
      l->st.messages += found.size ();
      l->st.blocks_decoded++;
      l->free_slots.push_back (slot);
      l->busy = false;
    }
    l->cv_out.notify_all ();
    l->cv_idle.notify_all ();
  }
}

}  // namespace

extern "C" {

__attribute__ ((visibility ("default")))
int modes_live_open (const modes_engine_config *cfg, uint32_t max_block_bytes, uint32_t nbuffers, modes_live **out)
{
  if (!cfg || !out) return set_error (MODES_E_ARG, "null argument");
  *out = nullptr;
  if (max_block_bytes == 0) max_block_bytes = kAsyncBufSize;       // ref: src/misc.h:649, what the drivers hand over
  if (max_block_bytes > kAsyncBufSize) return set_error (MODES_E_ARG, "a live block is at most %u bytes (MODES_ASYNC_BUF_SIZE)", kAsyncBufSize);
  if (nbuffers < 2) nbuffers = 2;
  if (nbuffers > 64) nbuffers = 64;
  modes_live *l = new modes_live ();
  l->cfg = *cfg;
  l->bps = cfg->input_format == INPUT_UC8 ? 2 : 4;
  l->max_bytes = max_block_bytes / l->bps * l->bps;
  l->ov = uint32_t ((8 + 112 + 16) * 1E-6 * cfg->sample_rate);     // ref: src/dump1090.c:734
  memset (&l->st, 0, sizeof l->st);
  l->t0 = std::chrono::steady_clock::now ();
  int rc = modes_engine_create (cfg, &l->eng);
  if (rc != MODES_OK) { delete l; return rc; }
  bool ok = hipSetDevice (cfg->device) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags (&l->stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipMalloc (reinterpret_cast<void **> (&l->d_block), size_t (l->max_bytes) + 64) == hipSuccess;
  ok = ok && hipMalloc (reinterpret_cast<void **> (&l->d_halo), (l->ov + 8) * sizeof (uint16_t)) == hipSuccess;
  l->slots.resize (nbuffers);
  for (uint32_t i = 0; ok && i < nbuffers; i++) {
    ok = hipHostMalloc (reinterpret_cast<void **> (&l->slots[i].pin), l->max_bytes, hipHostMallocDefault) == hipSuccess;
    l->free_slots.push_back (int (i));
  }
  if (!ok) { modes_live_close (l); return set_error (MODES_E_NOMEM, "modes_live_open: device / pinned buffers"); }
  l->worker = std::thread (worker_main, l);
  *out = l;
  return MODES_OK;
}

__attribute__ ((visibility ("default")))
void modes_live_close (modes_live *l)
{
  if (!l) return;
  {
    std::lock_guard<std::mutex> lk (l->mu);
    l->stop = true; l->hold = false;
  }
  l->cv_work.notify_all ();
  l->cv_idle.notify_all ();                       // a flush that waits on another thread gives up
  if (l->worker.joinable ()) l->worker.join ();
  for (auto &s : l->slots) if (s.pin) (void) hipHostFree (s.pin);
  if (l->d_block) (void) hipFree (l->d_block);
  if (l->d_halo) (void) hipFree (l->d_halo);
  if (l->stream) (void) hipStreamDestroy (l->stream);
  if (l->eng) modes_engine_destroy (l->eng);
  delete l;
}

// ref: src/dump1090.c:1181-1282 (rx_callback + rx_callback_to_fifo).  Returns 1 = queued, 0 = dropped (every buffer
// busy), < 0 = error.  Never blocks on the device.
__attribute__ ((visibility ("default")))
int modes_live_rx (modes_live *l, const void *iq, uint32_t len)
{
  if (!l || (!iq && len)) return set_error (MODES_E_ARG, "null argument");
  const uint32_t samples_read = len / l->bps;
  if (samples_read == 0) return 1;
  int slot;
  uint32_t to_convert = samples_read;
  {
    std::lock_guard<std::mutex> lk (l->mu);
    if (l->failed) return set_error (MODES_E_HIP, "modes_live: the decoder failed: %s", l->error.c_str ());
    if (l->free_slots.empty ()) {                                  // FIFO is full: drop this block
      l->dropped += samples_read;
      l->st.dropped_blocks++; l->st.dropped_samples += samples_read;
      return 0;
    }
    slot = l->free_slots.back ();
    l->free_slots.pop_back ();
    modes_live::Slot &s = l->slots[slot];
    s.discontinuous = l->dropped != 0;
    s.dropped = l->dropped;
    if (s.discontinuous) l->st.discontinuities++;
    l->dropped = 0;
    s.counter = l->sample_counter;
    l->sample_counter += samples_read;
    s.sys_ms = now_ms (l) - int64_t ((1E3 * samples_read) / l->cfg.sample_rate);
    if (s.sys_ms < 0) s.sys_ms = 0;
    if (to_convert * l->bps > l->max_bytes) {                      // "how did that happen?": the excess counts as dropped
      to_convert = l->max_bytes / l->bps;
      l->dropped = samples_read - to_convert;
      l->st.dropped_samples += l->dropped;
    }
    s.len = to_convert * l->bps;
    l->st.blocks++; l->st.samples += to_convert;
  }
  memcpy (l->slots[slot].pin, iq, size_t (to_convert) * l->bps);    // outside the lock: the slot is ours
  {
    std::lock_guard<std::mutex> lk (l->mu);
    l->queued.push_back (slot);
  }
  l->cv_work.notify_one ();
  return 1;
}

// ref: src/dump1090.c:1367-1400 (main_data_loop's dequeue side): decoded messages in order, up to `max`; waits up to
// wait_ms for the first one (0 = do not wait)
__attribute__ ((visibility ("default")))
size_t modes_live_take (modes_live *l, modes_message *out, size_t max, int wait_ms)
{
  if (!l || !out || !max) return 0;
  std::unique_lock<std::mutex> lk (l->mu);
  if (l->out.empty () && wait_ms > 0)
    l->cv_out.wait_for (lk, std::chrono::milliseconds (wait_ms), [&] { return !l->out.empty () || l->failed; });
  const size_t n = l->out.size () < max ? l->out.size () : max;
  for (size_t i = 0; i < n; i++) out[i] = l->out[i];
  l->out.erase (l->out.begin (), l->out.begin () + n);
  return n;
}

// waits until every queued block has been decoded
__attribute__ ((visibility ("default")))
int modes_live_flush (modes_live *l)
{
  if (!l) return set_error (MODES_E_ARG, "null argument");
  std::unique_lock<std::mutex> lk (l->mu);
  // on hold the worker leaves the queue alone: wait for the block in progress only, and say that blocks are still queued
  l->cv_idle.wait (lk, [&] { return l->failed || l->stop || (!l->busy && (l->hold || l->queued.empty ())); });
  if (l->failed) return set_error (MODES_E_HIP, "modes_live: the decoder failed: %s", l->error.c_str ());
  if (!l->queued.empty ()) return set_error (MODES_E_ARG, "modes_live_flush: %zu blocks are still queued (the worker is on hold or stopping)", l->queued.size ());
  return MODES_OK;
}

// hold != 0: the worker finishes its block and then leaves the queue alone (a stalled consumer, for tests of the
// drop accounting); 0 resumes
__attribute__ ((visibility ("default")))
int modes_live_hold (modes_live *l, int hold)
{
  if (!l) return set_error (MODES_E_ARG, "null argument");
  {
    std::unique_lock<std::mutex> lk (l->mu);
    l->hold = hold != 0;
    if (l->hold) l->cv_idle.wait (lk, [&] { return !l->busy; });
  }
  l->cv_work.notify_all ();
  return MODES_OK;
}

__attribute__ ((visibility ("default")))
int modes_live_get_stats (modes_live *l, modes_live_stats *out)
{
  if (!l || !out) return set_error (MODES_E_ARG, "null argument");
  std::lock_guard<std::mutex> lk (l->mu);
  *out = l->st;
  return MODES_OK;
}

}  // extern "C"
