"""dump1090_amd -- host-side Python mirror of libmodes_hip.so (include/modes_hip.h).

The library is the product: hand-written gfx950 kernels behind a C ABI that keeps the
reference's convert.h / demod.h entry points.  This module only loads it (ctypes) and gives
tests and bench.py the same calls with numpy / torch buffers.  There is no CPU fallback: if
the shared library is missing the import fails.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MODES_LIB") or os.path.join(_HERE, "libmodes_hip.so")   # MODES_LIB: development builds (tools/build_variant.sh)

INPUT_ILLEGAL, INPUT_UC8, INPUT_SC16, INPUT_SC16Q11 = 0, 1, 2, 3
MODES_OK = 0

CONVERTER_NAMES = ["convert_uc8_nodc_nopower", "convert_uc8_nodc_power", "convert_uc8_generic",
                   "convert_sc16_nodc_nopower", "convert_sc16_generic", "convert_sc16q11_generic"]

MESSAGE_DTYPE = np.dtype([("timestamp", "<u8"), ("msg", "u1", (14,)), ("msg_bits", "u1"), ("msg_type", "u1"),
                          ("addr", "<u4"), ("crc", "<u4"), ("score", "<i2"), ("error_bits", "u1"), ("phase", "u1"),
                          ("block", "<u4"), ("sig_level", "<f8"), ("j", "<u4"), ("reserved", "<u4")])
assert MESSAGE_DTYPE.itemsize == 56


class EngineConfig(C.Structure):
    _fields_ = [("sample_rate", C.c_uint32), ("input_format", C.c_int32), ("filter_dc", C.c_int32),
                ("compute_power", C.c_int32), ("fix_bits", C.c_int32), ("phase_enhance", C.c_int32),
                ("mode_ac", C.c_int32), ("device", C.c_int32), ("host_replay", C.c_int32),
                ("reserved", C.c_int32 * 7)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("valid_preamble", "demodulated", "out_of_phase", "demod_modeac", "CRC_good", "CRC_bad", "CRC_fixed",
                 "CRC_single_bit_fix", "CRC_two_bits_fix", "demod_rejected_unknown", "candidates", "replay_rounds", "scan_retries")]

    def asdict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class DcInfo(C.Structure):
    _fields_ = [("chunks", C.c_uint32), ("rerun_chunks", C.c_uint32), ("chunk_samples", C.c_uint64),
                ("warm_samples", C.c_uint64), ("chain_ms", C.c_float), ("prefix_reruns", C.c_uint32),
                ("guess_rounds", C.c_uint64), ("plain_rounds", C.c_uint64), ("chain_steps", C.c_uint64)]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved"}


D8_CHAIN_STATES = 168


class D8State(C.Structure):
    """modes_d8_state: what demod_8000 carries from one shard of a recording to the next."""
    _fields_ = [("acc", C.c_int32), ("acc_valid", C.c_int32), ("pending", C.c_int64), ("trig_tail", C.c_uint64),
                ("amb_steps", C.c_uint64), ("dprev", C.c_int32 * 1928)]

    def key(self):
        """What must be equal for two hand-overs to be the same (amb_steps describes the run, not the state)."""
        return (self.acc if self.acc_valid else None, self.pending, self.trig_tail, bytes(self.dprev))


class PowerStats(C.Structure):
    _fields_ = [("signal_power_sum", C.c_double), ("signal_power_count", C.c_uint64), ("peak_signal_power", C.c_double),
                ("strong_signal_count", C.c_uint64), ("noise_power_sum", C.c_double), ("noise_power_count", C.c_uint64),
                ("noise_valid", C.c_int32), ("reserved", C.c_int32)]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved"}


class ConvertState(C.Structure):
    _fields_ = [("DC_A", C.c_float), ("DC_B", C.c_float), ("z1_I", C.c_float), ("z1_Q", C.c_float),
                ("func_name", C.c_char_p), ("description", C.c_char_p)]


class MagBuf(C.Structure):
    pass


MagBuf._fields_ = [("data", C.POINTER(C.c_uint16)), ("total_length", C.c_uint), ("valid_length", C.c_uint),
                   ("overlap", C.c_uint), ("sample_timestamp", C.c_uint64), ("sys_timestamp", C.c_uint64),
                   ("flags", C.c_int), ("mean_level", C.c_double), ("mean_power", C.c_double),
                   ("dropped", C.c_uint), ("next", C.POINTER(MagBuf))]

CONVERT_FUNC = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_uint, C.POINTER(ConvertState), C.POINTER(C.c_double))


class ErrorInfo(C.Structure):
    _fields_ = [("syndrome", C.c_uint32), ("errors", C.c_int), ("bit", C.c_int8 * 2)]


SCORE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
DECODE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
DECODE_A_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int)
USER_CB = C.CFUNCTYPE(None, C.c_void_p)
LEN_CB = C.CFUNCTYPE(C.c_int, C.c_int)
CLOCK_CB = C.CFUNCTYPE(C.c_int64, C.c_uint64, C.c_uint64)


class HostBinding(C.Structure):
    """modes_hip_host (include/modes_hip.h, Part 2)."""
    _fields_ = [("mm_size", C.c_size_t), ("off_timestamp_msg", C.c_size_t), ("off_sys_timestamp_msg", C.c_size_t),
                ("off_sig_level", C.c_size_t), ("off_score", C.c_size_t), ("off_error_bits", C.c_size_t),
                ("off_AC_flags", C.c_size_t), ("off_addr", C.c_size_t),
                ("modeS_message_score", SCORE_CB), ("decode_mode_S_message", DECODE_CB),
                ("decode_mode_A_message", DECODE_A_CB), ("modeS_user_message", USER_CB),
                ("modeS_message_len_by_type", LEN_CB), ("receiveclock_ms_elapsed", CLOCK_CB),
                ("phase_enhance", C.c_int), ("mode_AC", C.c_int), ("measure_noise", C.c_int), ("fix_bits", C.c_int),
                ("device", C.c_int)]


class ModesError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libmodes_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ModesError("%s is missing: run `make -C dump1090_amd/csrc` (or __graft_entry__.build())" % LIB_PATH)
        # One HIP runtime per process: if PyTorch is around, let it load its bundled libamdhip64 first so
        # that this library binds to the same one (device pointers are then shared with torch tensors).
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        L.modes_last_error.restype = C.c_char_p
        L.modes_engine_create.argtypes = [C.POINTER(EngineConfig), C.POINTER(C.c_void_p)]
        L.modes_engine_destroy.argtypes = [C.c_void_p]
        L.modes_engine_set_icao.argtypes = [C.c_void_p, C.c_void_p]
        L.modes_engine_get_icao.argtypes = [C.c_void_p, C.c_void_p]
        L.modes_engine_run_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p, C.c_void_p]
        L.modes_engine_run_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64]
        L.modes_engine_time_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p,
                                             C.POINTER(C.c_float)]
        L.modes_engine_rerun_accept.argtypes = [C.c_void_p, C.c_void_p]
        L.modes_engine_can_rerun_accept.argtypes = [C.c_void_p]
        L.modes_engine_get_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.modes_engine_get_power_stats.argtypes = [C.c_void_p, C.POINTER(PowerStats)]
        L.modes_engine_set_d8_state.argtypes = [C.c_void_p, C.POINTER(D8State)]
        L.modes_engine_get_d8_state.argtypes = [C.c_void_p, C.POINTER(D8State), C.c_int, C.c_void_p]
        L.modes_engine_d8_chain_map.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
        L.modes_engine_d8_reuse_planes.argtypes = [C.c_void_p]
        L.modes_engine_d8_last_ambiguous.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.modes_engine_decode_fields.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_void_p]
        L.modes_decode_fields_host.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.modes_live_open.argtypes = [C.POINTER(EngineConfig), C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        L.modes_live_close.argtypes = [C.c_void_p]
        L.modes_live_close.restype = None
        L.modes_live_rx.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.modes_live_take.restype = C.c_size_t
        L.modes_live_take.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.modes_live_flush.argtypes = [C.c_void_p]
        L.modes_live_hold.argtypes = [C.c_void_p, C.c_int]
        L.modes_live_get_stats.argtypes = [C.c_void_p, C.POINTER(LiveStats)]
        L.modes_engine_num_messages.restype = C.c_size_t
        L.modes_engine_num_messages.argtypes = [C.c_void_p]
        L.modes_engine_messages.restype = C.c_void_p
        L.modes_engine_messages.argtypes = [C.c_void_p]
        L.modes_engine_format_raw.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t)]
        L.modes_engine_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.modes_engine_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.modes_engine_block_power.restype = C.c_size_t
        L.modes_engine_block_power.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.modes_engine_halo_at.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p, C.c_void_p]
        L.modes_engine_tail_halo.restype = C.c_void_p
        L.modes_engine_tail_halo.argtypes = [C.c_void_p]
        L.modes_engine_set_dc_state.argtypes = [C.c_void_p, C.c_float, C.c_float]
        L.modes_engine_get_dc_state.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.modes_engine_set_dc_lead.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        L.modes_engine_get_dc_info.argtypes = [C.c_void_p, C.POINTER(DcInfo)]
        L.modes_engine_dc_magnitudes.restype = C.c_void_p
        L.modes_engine_dc_magnitudes.argtypes = [C.c_void_p]
        L.modes_engine_read_dc_magnitudes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
        L.modes_convert_device.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(ConvertState),
                                           C.POINTER(C.c_double), C.c_void_p]
        L.modes_convert_device_noalloc.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float,
                                                   C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.modes_strip_device.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint64), C.c_void_p,
                                         C.POINTER(C.c_size_t), C.c_void_p]
        L.modes_hip_bind.argtypes = [C.c_void_p]
        L.convert_init.restype = C.c_void_p
        L.convert_init.argtypes = [C.c_int, C.c_double, C.c_bool, C.c_bool, C.POINTER(C.POINTER(ConvertState))]
        L.convert_cleanup.argtypes = [C.POINTER(C.POINTER(ConvertState))]
        L.convert_format_name.restype = C.c_char_p
        L.convert_format_name.argtypes = [C.c_int]
        for n in ("demod_2000", "demod_2400", "demod_2400_AC", "demod_8000"):
            getattr(L, n).argtypes = [C.POINTER(MagBuf)]
            getattr(L, n).restype = None
        L.crc_init.argtypes = [C.c_int]
        L.crc_checksum.restype = C.c_uint32
        L.crc_checksum.argtypes = [C.c_void_p, C.c_int]
        L.crc_checksum_diagnose.restype = C.POINTER(ErrorInfo)
        L.crc_checksum_diagnose.argtypes = [C.c_uint32, C.c_int]
        L.crc_checksum_fix.argtypes = [C.c_void_p, C.POINTER(ErrorInfo)]
        L.modes_hip_take_messages.restype = C.c_size_t
        L.modes_hip_take_messages.argtypes = [C.c_void_p, C.c_size_t]
        _lib = L
    return _lib


def _check(rc):
    if rc != MODES_OK:
        raise ModesError("modes_hip error %d: %s" % (rc, lib().modes_last_error().decode()))


class Engine:
    """Bulk engine (Part 3 of include/modes_hip.h)."""

    def __init__(self, sample_rate=2400000, input_format=INPUT_UC8, filter_dc=False, compute_power=False, fix_bits=1,
                 phase_enhance=False, mode_ac=False, device=0, host_replay=False):
        self.L = lib()
        cfg = EngineConfig(sample_rate=sample_rate, input_format=input_format, filter_dc=int(filter_dc),
                           compute_power=int(compute_power), fix_bits=fix_bits, phase_enhance=int(phase_enhance),
                           mode_ac=int(mode_ac), device=device, host_replay=int(host_replay))
        h = C.c_void_p()
        _check(self.L.modes_engine_create(C.byref(cfg), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.modes_engine_destroy(self.h)
            self.h = None

    __del__ = close

    def set_icao(self, slots=None):
        if slots is None:
            _check(self.L.modes_engine_set_icao(self.h, None))
        else:
            a = np.ascontiguousarray(slots, dtype=np.uint32)
            assert a.size == 1024
            _check(self.L.modes_engine_set_icao(self.h, a.ctypes.data_as(C.c_void_p)))

    def get_icao(self):
        a = np.zeros(1024, dtype=np.uint32)
        _check(self.L.modes_engine_get_icao(self.h, a.ctypes.data_as(C.c_void_p)))
        return a

    def run_device(self, data_ptr, nbytes, first_sample=0, halo_ptr=None, stream=None):
        """data_ptr: device address (e.g. torch_tensor.data_ptr())."""
        _check(self.L.modes_engine_run_device(self.h, C.c_void_p(data_ptr), nbytes, first_sample,
                                              C.c_void_p(halo_ptr) if halo_ptr else None,
                                              C.c_void_p(stream) if stream else None))

    def decode_fields(self, stream=None):
        """Field decode of the last run's messages (FIELDS_DTYPE array, message order)."""
        p, n = C.c_void_p(), C.c_size_t()
        _check(self.L.modes_engine_decode_fields(self.h, C.byref(p), C.byref(n), C.c_void_p(stream) if stream else None))
        if not n.value:
            return np.zeros(0, dtype=FIELDS_DTYPE)
        buf = (C.c_char * (n.value * FIELDS_DTYPE.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=FIELDS_DTYPE).copy()

    def rerun_accept(self, stream=None):
        """Repeat only the ordered accept pass + emit of the last run (after set_icao)."""
        _check(self.L.modes_engine_rerun_accept(self.h, C.c_void_p(stream) if stream else None))

    def can_rerun_accept(self):
        """False for engines whose ordered pass is not on the device (mode_ac, phase_enhance, host_replay) or with no run yet."""
        return bool(self.L.modes_engine_can_rerun_accept(self.h))

    def counters(self):
        a, b = C.c_uint64(), C.c_uint64()
        _check(self.L.modes_engine_get_counters(self.h, C.byref(a), C.byref(b)))
        return {"scan_launches": int(a.value), "accept_passes": int(b.value)}

    def power_stats(self):
        s = PowerStats()
        _check(self.L.modes_engine_get_power_stats(self.h, C.byref(s)))
        return s.asdict()

    def set_d8_state(self, state):
        _check(self.L.modes_engine_set_d8_state(self.h, C.byref(state) if state is not None else None))

    def get_d8_state(self, want_acc=False, stream=None):
        s = D8State()
        _check(self.L.modes_engine_get_d8_state(self.h, C.byref(s), int(want_acc), C.c_void_p(stream) if stream else None))
        return s

    def d8_chain_map(self, data_ptr, nbytes, first_sample=0, halo_ptr=None, stream=None):
        """int32[168]: the `pending` the shard hands on for every `pending` it can be entered with (index 0: none,
        index i: -i).  Keeps the state set with set_d8_state() and the trigger planes for the run that follows."""
        m = (C.c_int32 * D8_CHAIN_STATES)()
        _check(self.L.modes_engine_d8_chain_map(self.h, C.c_void_p(data_ptr), nbytes, first_sample,
                                                C.c_void_p(halo_ptr) if halo_ptr else None,
                                                C.c_void_p(stream) if stream else None, m))
        return np.frombuffer(m, dtype=np.int32).copy()

    def d8_last_ambiguous(self):
        v = C.c_uint64()
        _check(self.L.modes_engine_d8_last_ambiguous(self.h, C.byref(v)))
        return v.value

    def d8_reuse_planes(self):
        _check(self.L.modes_engine_d8_reuse_planes(self.h))

    def run_host(self, iq, first_sample=0):
        iq = np.ascontiguousarray(iq)
        _check(self.L.modes_engine_run_host(self.h, iq.ctypes.data_as(C.c_void_p), iq.nbytes, first_sample))

    def time_scan(self, data_ptr, nbytes, reps=10, stream=None):
        ms = C.c_float()
        _check(self.L.modes_engine_time_scan(self.h, C.c_void_p(data_ptr), nbytes, reps,
                                             C.c_void_p(stream) if stream else None, C.byref(ms)))
        return ms.value

    def halo_at(self, data_ptr, nbytes, at_sample, out_ptr, stream=None):
        _check(self.L.modes_engine_halo_at(self.h, C.c_void_p(data_ptr), nbytes, at_sample, C.c_void_p(out_ptr),
                                           C.c_void_p(stream) if stream else None))

    def tail_halo(self):
        return self.L.modes_engine_tail_halo(self.h)

    def num_messages(self):
        return int(self.L.modes_engine_num_messages(self.h))

    def messages(self):
        n = self.L.modes_engine_num_messages(self.h)
        if n == 0:
            return np.zeros(0, dtype=MESSAGE_DTYPE)
        buf = (C.c_char * (n * MESSAGE_DTYPE.itemsize)).from_address(self.L.modes_engine_messages(self.h))
        return np.frombuffer(buf, dtype=MESSAGE_DTYPE).copy()

    def format_raw(self, upper_case=False):
        """RAW text of the last run's messages (bytes): '*<hex>;\\n' per message."""
        t, n = C.c_char_p(), C.c_size_t()
        _check(self.L.modes_engine_format_raw(self.h, int(upper_case), C.byref(t), C.byref(n)))
        return C.string_at(t, n.value) if n.value else b""

    def kernel_ms(self):
        """(main, second) kernel durations of the last run in ms (HIP events on the launch stream)."""
        a, b = C.c_float(), C.c_float()
        _check(self.L.modes_engine_kernel_ms(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def block_power(self):
        """mean_power of every block of the last run (only computed with mode_ac at 2.4 MS/s)."""
        n = self.L.modes_engine_block_power(self.h, None, 0)
        a = np.zeros(n, dtype=np.float64)
        if n:
            self.L.modes_engine_block_power(self.h, a.ctypes.data_as(C.c_void_p), n)
        return a

    # ---- float converters with the DC block on ----
    def set_dc_state(self, z1_I=0.0, z1_Q=0.0):
        _check(self.L.modes_engine_set_dc_state(self.h, z1_I, z1_Q))

    def get_dc_state(self):
        a, b = C.c_float(), C.c_float()
        _check(self.L.modes_engine_get_dc_state(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_dc_lead(self, lead_samples, warm_samples=0):
        _check(self.L.modes_engine_set_dc_lead(self.h, lead_samples, warm_samples))

    def dc_info(self):
        i = DcInfo()
        _check(self.L.modes_engine_get_dc_info(self.h, C.byref(i)))
        return i.asdict()

    def dc_magnitudes(self, first, count):
        """Magnitudes of the last filter_dc run: entry 0 = first overlap sample in front of the shard."""
        a = np.zeros(count, dtype=np.uint16)
        _check(self.L.modes_engine_read_dc_magnitudes(self.h, a.ctypes.data_as(C.c_void_p), first, count))
        return a

    def stats(self):
        s = Stats()
        _check(self.L.modes_engine_get_stats(self.h, C.byref(s)))
        return s.asdict()


class LiveStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("blocks", "samples", "dropped_blocks", "dropped_samples", "discontinuities",
                                          "blocks_decoded", "messages")]

    def asdict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Live:
    """modes_live_*: rx_callback() + FIFO + main_data_loop() around the device path (include/modes_hip.h)."""

    def __init__(self, sample_rate=2400000, input_format=INPUT_UC8, max_block_bytes=0, nbuffers=4, **kw):
        self.L = lib()
        cfg = EngineConfig(sample_rate=sample_rate, input_format=input_format, filter_dc=int(kw.get("filter_dc", False)),
                           compute_power=int(kw.get("compute_power", False)), fix_bits=kw.get("fix_bits", 1),
                           phase_enhance=int(kw.get("phase_enhance", False)), mode_ac=int(kw.get("mode_ac", False)),
                           device=kw.get("device", 0), host_replay=0)
        h = C.c_void_p()
        _check(self.L.modes_live_open(C.byref(cfg), max_block_bytes, nbuffers, C.byref(h)))
        self.h = h

    def rx(self, block):
        b = np.ascontiguousarray(block)
        rc = self.L.modes_live_rx(self.h, b.ctypes.data_as(C.c_void_p), b.nbytes)
        if rc < 0:
            _check(rc)
        return rc

    def take(self, max_n=1 << 16, wait_ms=0):
        buf = np.zeros(max_n, dtype=MESSAGE_DTYPE)
        n = self.L.modes_live_take(self.h, buf.ctypes.data_as(C.c_void_p), max_n, wait_ms)
        return buf[:n].copy()

    def flush(self):
        _check(self.L.modes_live_flush(self.h))

    def hold(self, on):
        _check(self.L.modes_live_hold(self.h, int(on)))

    def stats(self):
        s = LiveStats()
        _check(self.L.modes_live_get_stats(self.h, C.byref(s)))
        return s.asdict()

    def close(self):
        if getattr(self, "h", None):
            self.L.modes_live_close(self.h)
            self.h = None

    __del__ = close


def convert_device(conv_id, d_iq_ptr, d_mag_ptr, nsamples, state=None, want_power=True, stream=None):
    p = C.c_double(-1.0)
    _check(lib().modes_convert_device(conv_id, C.c_void_p(d_iq_ptr), C.c_void_p(d_mag_ptr), nsamples,
                                      C.byref(state) if state is not None else None,
                                      C.byref(p) if want_power else None, C.c_void_p(stream) if stream else None))
    return p.value


def strip_device(d_iq_ptr, nbytes, level, d_out_ptr, run=0, stream=None):
    """modes_strip_device(): returns (bytes written, run of quiet pairs carried out)."""
    r, n = C.c_uint64(run), C.c_size_t(0)
    _check(lib().modes_strip_device(C.c_void_p(d_iq_ptr), nbytes, int(level), C.byref(r), C.c_void_p(d_out_ptr), C.byref(n),
                                    C.c_void_p(stream) if stream else None))
    return int(n.value), int(r.value)


FIELDS_DTYPE = np.dtype([("AC_flags", "<u4"), ("addr", "<u4"), ("category", "<u4"), ("addrtype", "<u2"), ("reliable", "u1"),
                         ("unit", "u1"), ("capa", "i1"), ("IID", "u1"), ("ME_type", "i1"), ("ME_subtype", "i1"), ("cf", "i1"),
                         ("reserved0", "i1"), ("nuc_p", "u1"), ("EW_dir", "u1"), ("NS_dir", "u1"), ("reserved1", "u1", (3,)),
                         ("flight_status", "<i4"), ("identity", "<i4"), ("altitude", "<i4"), ("altitude_HAE", "<i4"),
                         ("HAE_delta", "<i4"), ("raw_latitude", "<i4"), ("raw_longitude", "<i4"), ("EW_velocity", "<i4"),
                         ("NS_velocity", "<i4"), ("vert_rate", "<i4"), ("pad", "<u4"), ("heading", "<f8"), ("velocity", "<f8"),
                         ("flight", "S9"), ("ME", "u1", (7,)), ("MD", "u1", (10,)), ("reserved2", "u1", (6,))])
assert FIELDS_DTYPE.itemsize == 120


def decode_fields_host(msgs):
    """modes_decode_fields_host(): field decode of a MESSAGE_DTYPE array on the device."""
    m = np.ascontiguousarray(msgs, dtype=MESSAGE_DTYPE)
    out = np.zeros(len(m), dtype=FIELDS_DTYPE)
    _check(lib().modes_decode_fields_host(m.ctypes.data_as(C.c_void_p), len(m), out.ctypes.data_as(C.c_void_p)))
    return out


NOALLOC_MAX_SAMPLES = 262144
CONVERT_RESULT_DTYPE = np.dtype([("out_power", "<f8"), ("z1_I", "<f4"), ("z1_Q", "<f4")])


def convert_device_noalloc(conv_id, d_iq_ptr, d_mag_ptr, nsamples, DC_A=0.0, DC_B=1.0, z1_I=0.0, z1_Q=0.0,
                           d_result_ptr=None, stream=None):
    """modes_convert_device_noalloc(): allocates nothing, waits for nothing; results land in device memory."""
    _check(lib().modes_convert_device_noalloc(conv_id, C.c_void_p(d_iq_ptr), C.c_void_p(d_mag_ptr), nsamples, DC_A, DC_B,
                                              z1_I, z1_Q, C.c_void_p(d_result_ptr) if d_result_ptr else None,
                                              C.c_void_p(stream) if stream else None))


def msg_hex(rec):
    return bytes(rec["msg"][: int(rec["msg_bits"]) // 8]).hex()


# ---- the reference's own entry points (Part 1), host buffers ------------------------------------

class Converter:
    """convert_init() / convert_func / convert_cleanup() (ref: src/convert.h:28-42)."""

    def __init__(self, fmt, sample_rate, filter_dc, compute_power):
        self.L = lib()
        self.state = C.POINTER(ConvertState)()
        fp = self.L.convert_init(fmt, float(sample_rate), bool(filter_dc), bool(compute_power), C.byref(self.state))
        if not fp:
            raise ModesError("convert_init returned NULL")
        self.func = CONVERT_FUNC(fp)
        self.name = self.state.contents.func_name.decode()

    def __call__(self, iq, nsamples, want_power=True):
        iq = np.ascontiguousarray(iq)
        mag = np.zeros(nsamples, dtype=np.uint16)
        p = C.c_double(-1.0)
        self.func(iq.ctypes.data_as(C.c_void_p), mag.ctypes.data_as(C.c_void_p), nsamples, self.state,
                  C.byref(p) if want_power else None)
        return mag, p.value

    def close(self):
        if self.state:
            self.L.convert_cleanup(C.byref(self.state))

    __del__ = close


def make_mag_buf(data, overlap, sample_timestamp=0, mean_power=0.0, total_length=None):
    data = np.ascontiguousarray(data, dtype=np.uint16)
    mb = MagBuf()
    mb.data = data.ctypes.data_as(C.POINTER(C.c_uint16))
    mb.total_length = total_length if total_length is not None else len(data)
    mb.valid_length = len(data)
    mb.overlap = overlap
    mb.sample_timestamp = sample_timestamp
    mb.mean_power = mean_power
    mb._keep = data
    return mb


def take_messages(max_n=1 << 20):
    buf = np.zeros(max_n, dtype=MESSAGE_DTYPE)
    n = lib().modes_hip_take_messages(buf.ctypes.data_as(C.c_void_p), max_n)
    return buf[:n].copy()
