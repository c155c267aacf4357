"""ctypes binding of the CPU oracle (oracle/libmodes_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (dump1090_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmodes_oracle.so")

INPUT_UC8, INPUT_SC16, INPUT_SC16Q11 = 1, 2, 3
CONV_NAMES = ["convert_uc8_nodc_nopower", "convert_uc8_nodc_power", "convert_uc8_generic",
              "convert_sc16_nodc_nopower", "convert_sc16_generic", "convert_sc16q11_generic"]


class ConvertState(C.Structure):
    _fields_ = [("DC_A", C.c_float), ("DC_B", C.c_float), ("z1_I", C.c_float), ("z1_Q", C.c_float)]


class MagBuf(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint16)), ("total_length", C.c_uint32), ("valid_length", C.c_uint32),
                ("overlap", C.c_uint32), ("sample_timestamp", C.c_uint64), ("sys_timestamp", C.c_uint64),
                ("flags", C.c_uint32), ("mean_level", C.c_double), ("mean_power", C.c_double),
                ("dropped", C.c_uint32)]


MSG_DTYPE = np.dtype([("timestamp", "<u8"), ("sys_timestamp", "<u8"), ("msg", "u1", (14,)), ("pad", "u1", (2,)),
                      ("msg_bits", "<i4"), ("msg_type", "<i4"), ("error_bits", "<i4"), ("score", "<i4"),
                      ("phase", "<i4"), ("addr", "<u4"), ("crc", "<u4"), ("block", "<i4"), ("j", "<i4"),
                      ("pad2", "<i4"), ("sig_level", "<f8")], align=False)


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("valid_preamble", "demodulated", "out_of_phase", "demod_modeac", "CRC_good", "CRC_bad",
                 "CRC_fixed", "CRC_single_bit_fix", "CRC_two_bits_fix", "demod_rejected_unknown",
                 "signal_power_count", "strong_signal_count", "noise_power_count", "d8m_drift_resets",
                 "score_calls")] + [(n, C.c_double) for n in
                                    ("signal_power_sum", "peak_signal_power", "noise_power_sum")]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def build(force=False):
    """Compile the oracle with gcc (seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "modes_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_reset.argtypes = [C.c_void_p]
        L.orc_convert_select.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_convert_state_init.argtypes = [C.POINTER(ConvertState), C.c_double, C.c_int]
        L.orc_convert.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint, C.POINTER(ConvertState),
                                  C.POINTER(C.c_double)]
        L.orc_mag_lut.restype = C.POINTER(C.c_uint16)
        L.orc_log10_lut.restype = C.POINTER(C.c_uint16)
        L.orc_crc_checksum.restype = C.c_uint32
        L.orc_crc_checksum.argtypes = [C.c_void_p, C.c_int]
        L.orc_crc_table_entry.restype = C.c_uint32
        L.orc_single_bit_syndrome.restype = C.c_uint32
        L.orc_error_table_size.argtypes = [C.c_void_p, C.c_int]
        L.orc_error_table_entry.restype = C.c_uint32
        L.orc_error_table_entry.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 3
        L.orc_crc_diagnose.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_message_score.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_decode_message.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_icao_add.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_icao_test.argtypes = [C.c_void_p, C.c_uint32]
        for n in ("orc_demod_2000", "orc_demod_2400", "orc_demod_2400_AC", "orc_demod_8000"):
            getattr(L, n).argtypes = [C.c_void_p, C.POINTER(MagBuf)]
        L.orc_apply_phase_correction.argtypes = [C.c_void_p]
        L.orc_slice_2000.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        L.orc_run_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_stream_start.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]
        L.orc_icao_export.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_icao_import.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_last_tail.argtypes = [C.c_void_p]
        L.orc_last_tail.restype = C.c_uint32
        L.orc_strip.restype = C.c_size_t
        L.orc_strip.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_uint64), C.c_void_p]
        L.orc_num_messages.restype = C.c_size_t
        L.orc_num_messages.argtypes = [C.c_void_p]
        L.orc_messages.restype = C.c_void_p
        L.orc_messages.argtypes = [C.c_void_p]
        L.orc_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.orc_sizeof_msg.restype = C.c_size_t
        assert L.orc_sizeof_msg() == MSG_DTYPE.itemsize, (L.orc_sizeof_msg(), MSG_DTYPE.itemsize)
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """One reference 'process': CRC tables, ICAO cache, statistics, demod_8000 state."""

    def __init__(self, sample_rate=2400000, fix_bits=1, phase_enhance=False, mode_ac=False, measure_noise=False):
        self.L = lib()
        self.sample_rate = sample_rate
        self.h = self.L.orc_create(sample_rate, fix_bits, int(phase_enhance), int(mode_ac), int(measure_noise))

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    __del__ = close

    def reset(self):
        self.L.orc_reset(self.h)

    # ---- crc ----
    def table_size(self, bits):
        return self.L.orc_error_table_size(self.h, bits)

    def table(self, bits):
        n = self.table_size(bits)
        out = []
        e, b0, b1 = C.c_int(), C.c_int(), C.c_int()
        for k in range(n):
            s = self.L.orc_error_table_entry(self.h, bits, k, C.byref(e), C.byref(b0), C.byref(b1))
            out.append((s, e.value, b0.value, b1.value))
        return out

    def diagnose(self, syndrome, bits):
        b0, b1 = C.c_int(), C.c_int()
        r = self.L.orc_crc_diagnose(self.h, syndrome, bits, C.byref(b0), C.byref(b1))
        return r, b0.value, b1.value

    def score(self, msg, valid_bits):
        m = np.frombuffer(bytes(msg).ljust(14, b"\0"), dtype=np.uint8).copy()
        return self.L.orc_message_score(self.h, _ptr(m), valid_bits)

    def decode(self, msg):
        m = np.frombuffer(bytes(msg).ljust(14, b"\0"), dtype=np.uint8).copy()
        out = np.zeros(1, dtype=MSG_DTYPE)
        rc = self.L.orc_decode_message(self.h, _ptr(m), _ptr(out))
        return rc, out[0]

    def icao_add(self, addr):
        self.L.orc_icao_add(self.h, addr)

    def icao_test(self, addr):
        return bool(self.L.orc_icao_test(self.h, addr))

    # ---- demod on an explicit mag_buf ----
    def demod(self, which, data, overlap, sample_timestamp=0, mean_power=0.0, total_length=None):
        data = np.ascontiguousarray(data, dtype=np.uint16)
        mb = MagBuf()
        mb.data = data.ctypes.data_as(C.POINTER(C.c_uint16))
        mb.total_length = total_length if total_length is not None else len(data)
        mb.valid_length = len(data)
        mb.overlap = overlap
        mb.sample_timestamp = sample_timestamp
        mb.mean_power = mean_power
        fn = {2000: self.L.orc_demod_2000, 2400: self.L.orc_demod_2400, "2400AC": self.L.orc_demod_2400_AC,
              8000: self.L.orc_demod_8000}[which]
        fn(self.h, C.byref(mb))

    # ---- whole stream ----
    def run_stream(self, iq, fmt, demod, filter_dc=False, compute_power=False, also_ac=False, max_blocks=0,
                   want_crc=False):
        iq = np.ascontiguousarray(iq)
        nbytes = iq.nbytes
        nb_max = nbytes // (256 * 1024) + 2
        crcs = np.zeros(nb_max, dtype=np.uint32)
        powers = np.zeros(nb_max, dtype=np.float64)
        n = self.L.orc_run_stream(self.h, _ptr(iq), nbytes, fmt, int(filter_dc), int(compute_power), demod,
                                  int(also_ac), _ptr(crcs) if want_crc else None, max_blocks, _ptr(powers))
        if n < 0:
            raise ValueError("no suitable converter")
        self.last_block_crc32 = crcs[:n]
        self.last_block_power = powers[:n]
        return n

    # ---- sharded runs (test support) ----
    def stream_start(self, counter, tail=None):
        if tail is None:
            self.L.orc_stream_start(self.h, counter, None, 0)
        else:
            t = np.ascontiguousarray(tail, dtype=np.uint16)
            self.L.orc_stream_start(self.h, counter, _ptr(t), len(t))

    def icao_export(self):
        a = np.zeros(1024, dtype=np.uint32)
        self.L.orc_icao_export(self.h, _ptr(a))
        return a

    def icao_import(self, slots):
        a = np.ascontiguousarray(slots, dtype=np.uint32)
        self.L.orc_icao_import(self.h, _ptr(a))

    def last_tail(self):
        a = np.zeros(4096, dtype=np.uint16)
        n = self.L.orc_last_tail(_ptr(a))
        return a[:n].copy()

    def messages(self):
        n = self.L.orc_num_messages(self.h)
        if n == 0:
            return np.zeros(0, dtype=MSG_DTYPE)
        buf = (C.c_char * (n * MSG_DTYPE.itemsize)).from_address(self.L.orc_messages(self.h))
        return np.frombuffer(buf, dtype=MSG_DTYPE).copy()

    def stats(self):
        s = Stats()
        self.L.orc_get_stats(self.h, C.byref(s))
        return s.asdict()


def convert_select(fmt, filter_dc, compute_power):
    return lib().orc_convert_select(fmt, int(filter_dc), int(compute_power))


def convert(conv_id, iq, nsamples, sample_rate=2400000.0, filter_dc=False, state=None):
    """Run one converter call; returns (mag u16[nsamples], out_power, state)."""
    L = lib()
    iq = np.ascontiguousarray(iq)
    if state is None:
        state = ConvertState()
        L.orc_convert_state_init(C.byref(state), float(sample_rate), int(filter_dc))
    mag = np.zeros(nsamples, dtype=np.uint16)
    p = C.c_double(-1.0)
    L.orc_convert(conv_id, _ptr(iq), _ptr(mag), nsamples, C.byref(state), C.byref(p))
    return mag, p.value, state


FIELDS_DTYPE = np.dtype([("AC_flags", "<u4"), ("addr", "<u4"), ("category", "<u4"), ("addrtype", "<u2"), ("reliable", "u1"),
                         ("unit", "u1"), ("capa", "i1"), ("IID", "u1"), ("ME_type", "i1"), ("ME_subtype", "i1"), ("cf", "i1"),
                         ("flight_status8", "i1"), ("nuc_p", "u1"), ("EW_dir", "u1"), ("NS_dir", "u1"), ("pad0", "u1", (3,)),
                         ("flight_status", "<i4"), ("identity", "<i4"), ("altitude", "<i4"), ("altitude_HAE", "<i4"),
                         ("HAE_delta", "<i4"), ("raw_latitude", "<i4"), ("raw_longitude", "<i4"), ("EW_velocity", "<i4"),
                         ("NS_velocity", "<i4"), ("vert_rate", "<i4"), ("pad", "<u4"), ("heading", "<f8"), ("velocity", "<f8"),
                         ("flight", "S9"), ("ME", "u1", (7,)), ("MD", "u1", (10,)), ("pad1", "u1", (6,))])


def decode_fields(msgs):
    """orc_decode_fields over an array with fields msg, msg_type, crc, error_bits (engine or oracle message dtype)."""
    L = lib()
    L.orc_sizeof_fields.restype = C.c_size_t
    assert L.orc_sizeof_fields() == FIELDS_DTYPE.itemsize
    L.orc_decode_fields.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_int, C.c_void_p]
    out = np.zeros(len(msgs), dtype=FIELDS_DTYPE)
    for i, m in enumerate(msgs):
        b = np.ascontiguousarray(m["msg"], dtype=np.uint8)
        L.orc_decode_fields(_ptr(b), int(m["msg_type"]), int(m["crc"]), int(m["error_bits"]), C.c_void_p(out.ctypes.data + i * out.itemsize))
    return out


def strip(iq, level, run=0):
    """strip_mode_loop (ref: src/dump1090.c:3423-3448) over a uint8 buffer; returns (kept bytes, counter carried out)."""
    a = np.ascontiguousarray(iq, dtype=np.uint8)
    out = np.zeros(a.size + 2, dtype=np.uint8)
    r = C.c_uint64(run)
    n = lib().orc_strip(_ptr(a), a.size, int(level), C.byref(r), _ptr(out))
    return out[:n].copy(), int(r.value)


def crc_checksum(msg, bits):
    m = np.frombuffer(bytes(msg), dtype=np.uint8).copy()
    return lib().orc_crc_checksum(_ptr(m), bits)


def msg_hex(rec):
    return bytes(rec["msg"][: rec["msg_bits"] // 8]).hex()
