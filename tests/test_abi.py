"""CPU-side checks of the C ABI: the library loads and exports every function include/modes_hip.h declares;
host-side logic that needs no device (converter selection, CRC entry points)."""
import ctypes as C
import os
import re

import numpy as np

import dump1090_amd as da
from oracle import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "modes_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)                        # preprocessor lines
    src = re.sub(r"typedef[^;{]*\{.*?\}[^;]*;", "", src, flags=re.S)      # struct/enum bodies (function pointers inside)
    src = re.sub(r"typedef[^;]*;", "", src)
    return sorted(set(re.findall(r"\b([a-zA-Z_][a-zA-Z0-9_]*)\s*\([^;{]*\)\s*;", src)))


def test_library_exports_every_declared_symbol():
    names = declared_functions()
    for must in ("convert_init", "convert_cleanup", "convert_format_name", "demod_2000", "demod_2400", "demod_2400_AC",
                 "demod_8000", "demod_8000_alloc", "demod_8000_free", "crc_init", "crc_checksum",
                 "crc_checksum_diagnose", "crc_checksum_fix", "modes_hip_bind", "modes_engine_create",
                 "modes_engine_run_device", "modes_engine_messages"):
        assert must in names, must
    L = C.CDLL(da.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "libmodes_hip.so does not export %s" % n


def test_convert_init_selection_matches_reference_table():
    """ref: src/convert.c:301-328; state values ref: src/convert.c:346-362"""
    L = da.lib()
    for fmt in (da.INPUT_UC8, da.INPUT_SC16, da.INPUT_SC16Q11):
        for dc in (False, True):
            for pw in (False, True):
                c = da.Converter(fmt, 2400000.0, dc, pw)
                assert c.name == orc.CONV_NAMES[orc.convert_select(fmt, dc, pw)]
                st = c.state.contents
                ref = orc.ConvertState()
                orc.lib().orc_convert_state_init(C.byref(ref), 2400000.0, int(dc))
                assert (st.DC_A, st.DC_B, st.z1_I, st.z1_Q) == (ref.DC_A, ref.DC_B, 0.0, 0.0)
                c.close()
                assert not c.state
    st = C.POINTER(da.ConvertState)()
    assert not L.convert_init(da.INPUT_ILLEGAL, 2.4e6, False, False, C.byref(st))
    assert L.convert_format_name(da.INPUT_SC16Q11) == b"INPUT_SC16Q11" and L.convert_format_name(9) == b"?"


def test_crc_entry_points_match_oracle():
    """ref: src/crc.h:19-23"""
    L = da.lib()
    rng = np.random.default_rng(0)
    for fix_bits in (1, 2):
        L.crc_init(fix_bits)
        o = orc.Oracle(fix_bits=fix_bits)
        for bits in (56, 112):
            for s, n, b0, b1 in o.table(bits)[::37]:
                ei = L.crc_checksum_diagnose(s, bits)
                assert ei and (ei.contents.errors, ei.contents.bit[0], ei.contents.bit[1]) == (n, b0, b1)
            for s in rng.integers(1, 1 << 24, size=300):
                ei = L.crc_checksum_diagnose(int(s), bits)
                assert bool(ei) == (o.diagnose(int(s), bits)[0] >= 0)
        for _ in range(200):
            m = rng.integers(0, 256, size=14, dtype=np.uint8)
            for bits in (56, 112):
                assert L.crc_checksum(m.ctypes.data_as(C.c_void_p), bits) == orc.crc_checksum(bytes(m[: bits // 8]), bits)
        m = np.frombuffer(bytes.fromhex("8D4840D6202CC371C32CE0576098"), dtype=np.uint8).copy()
        assert L.crc_checksum(m.ctypes.data_as(C.c_void_p), 112) == 0
        m[5] ^= 0x10
        ei = L.crc_checksum_diagnose(L.crc_checksum(m.ctypes.data_as(C.c_void_p), 112), 112)
        assert ei and ei.contents.errors == 1 and ei.contents.bit[0] == 43
        L.crc_checksum_fix(m.ctypes.data_as(C.c_void_p), ei)
        assert bytes(m).hex().upper() == "8D4840D6202CC371C32CE0576098"
    assert L.crc_checksum_diagnose(0, 112).contents.errors == 0
    # the frames the reference itself prints (tests/golden/reference_vectors.json: README.md, tools/SBS_client.py)
    import json, os
    from conftest import GOLDEN
    vec = json.load(open(os.path.join(GOLDEN, "reference_vectors.json")))
    for f in vec["frames_112"]:
        m = np.frombuffer(bytes.fromhex(f["hex"]), dtype=np.uint8).copy()
        assert L.crc_checksum(m.ctypes.data_as(C.c_void_p), 112) == 0
    for f in vec["frames_56"]:
        m = np.frombuffer(bytes.fromhex(f["hex"]), dtype=np.uint8).copy()
        assert "%06x" % L.crc_checksum(m.ctypes.data_as(C.c_void_p), 56) == f["syndrome"]
    L.crc_exit()
