"""Synthetic Mode S captures: complex baseband, OOK pulses box-integrated over
each sample period, random sub-sample phase, carrier phase and amplitude, AWGN, quantised to
uc8 or sc16.  Used by the tests and by bench.py; no reference code involved.
"""
import numpy as np

_POLY = 0xFFF409


def crc24(data: bytes) -> int:
    """Mode S parity over `data` (the message without its last three bytes)."""
    rem = 0
    for b in data:
        rem ^= b << 16
        for _ in range(8):
            rem = ((rem << 1) ^ _POLY) if rem & 0x800000 else (rem << 1)
        rem &= 0xFFFFFF
    return rem


def make_df17(icao: int, me: bytes, ca: int = 5) -> bytes:
    body = bytes([(17 << 3) | ca, (icao >> 16) & 0xFF, (icao >> 8) & 0xFF, icao & 0xFF]) + bytes(me[:7]).ljust(7, b"\0")
    p = crc24(body)
    return body + bytes([(p >> 16) & 0xFF, (p >> 8) & 0xFF, p & 0xFF])


def make_df11(icao: int, ca: int = 5, iid: int = 0) -> bytes:
    body = bytes([(11 << 3) | ca, (icao >> 16) & 0xFF, (icao >> 8) & 0xFF, icao & 0xFF])
    p = crc24(body) ^ iid
    return body + bytes([(p >> 16) & 0xFF, (p >> 8) & 0xFF, p & 0xFF])


def make_ap(df: int, icao: int, payload: bytes) -> bytes:
    """Address/Parity formats (DF0/4/5 short, DF16/20/21 long): parity = CRC xor address."""
    n = 11 if df in (16, 20, 21) else 4
    body = bytes([(df << 3) | (payload[0] & 7)]) + bytes(payload[1:n]).ljust(n - 1, b"\0")
    p = crc24(body) ^ icao
    return body + bytes([(p >> 16) & 0xFF, (p >> 8) & 0xFF, p & 0xFF])


def chips_of(msg: bytes) -> np.ndarray:
    """0.5 us chips: 8 us preamble (pulses at 0, 1, 3.5, 4.5 us) then Manchester bits (1 = on,off)."""
    pre = np.zeros(16, dtype=np.float64)
    pre[[0, 2, 7, 9]] = 1.0
    bits = np.unpackbits(np.frombuffer(msg, dtype=np.uint8))
    data = np.empty(2 * len(bits), dtype=np.float64)
    data[0::2] = bits
    data[1::2] = 1 - bits
    return np.concatenate([pre, data])


def envelope(msg: bytes, sample_rate: float, frac: float) -> np.ndarray:
    """Per-sample mean of the on/off envelope; the frame starts `frac` samples (0..1) into sample 0."""
    ch = chips_of(msg)
    cps = sample_rate * 0.5e-6                      # samples per chip
    edges = frac + cps * np.arange(len(ch) + 1)     # chip boundaries in sample units
    cum = np.concatenate([[0.0], np.cumsum(ch) * cps])
    n = int(np.ceil(edges[-1])) + 1
    t = np.arange(n + 1, dtype=np.float64)
    integ = np.interp(t, edges, cum, left=0.0, right=cum[-1])
    return np.diff(integ)


def envelope_mode_a(code: int, sample_rate: float, frac: float, spi: bool = False) -> np.ndarray:
    """Mode A/C reply: 0.45 us pulses on a 1.45 us pitch, F1 .. F2 (+ SPI three slots later).
    `code` uses the reference's bit layout 00:A4:A2:A1:00:B4:B2:B1:00:C4:C2:C1:00:D4:D2:D1 with SPI = 0x0080."""
    slot_bits = [None, 0x0010, 0x1000, 0x0020, 0x2000, 0x0040, 0x4000, 0, 0x0100, 0x0001, 0x0200, 0x0002, 0x0400,
                 0x0004, None]                       # F1 C1 A1 C2 A2 C4 A4 X B1 D1 B2 D2 B4 D4 F2
    on = [True if b is None else bool(code & b) for b in slot_bits]
    on += [False, False, spi]
    rate_us = sample_rate * 1e-6
    n = int(np.ceil(frac + (len(on) * 1.45 + 1) * rate_us)) + 2
    t = np.arange(n + 1, dtype=np.float64)
    cum = np.zeros(n + 1)
    for k, o in enumerate(on):
        if o:
            a, b = frac + k * 1.45 * rate_us, frac + (k * 1.45 + 0.45) * rate_us
            cum += np.clip(t, a, b) - a
    return np.diff(cum)


def quantise(x: np.ndarray, fmt: str) -> np.ndarray:
    """x: complex array, full scale 1.0.  Returns interleaved I,Q in uc8 (uint8) or sc16 (int16)."""
    iq = np.empty(2 * len(x), dtype=np.float64)
    iq[0::2] = x.real
    iq[1::2] = x.imag
    if fmt == "uc8":
        return np.clip(np.round(iq * 127.5 + 127.5), 0, 255).astype(np.uint8)
    if fmt == "sc16":
        return np.clip(np.round(iq * 32767.0), -32768, 32767).astype(np.int16)
    if fmt == "sc16q11":
        return np.clip(np.round(iq * 2047.0), -2048, 2047).astype(np.int16)
    raise ValueError(fmt)


def capture(nsamples: int, sample_rate: float = 2.4e6, fmt: str = "uc8", seed: int = 1, **kw):
    """Returns (iq array, list of injected frames as dicts {start, msg (as sent), clean (before bit flips), amp})."""
    x, frames = capture_complex(nsamples, sample_rate, seed, **kw)
    return quantise(x, fmt), frames


# ---- byte streams for the uc8 converters AS WRITTEN (ref: src/convert.c:19-40) ----------------------
# convert_uc8_nodc_nopower / _power read ONE byte per output sample: out[i] = mag_lut[byte i] for the first `nsamples`
# bytes of each 256 KiB block, with mag_lut[b] = 360 * b for b <= 128.  A byte stream whose image under that table is
# a Mode S frame train is therefore: byte = round(envelope magnitude / 360) in the first half of every block.
LUT_BLOCK_SAMPLES = 131072           # uc8: 256 KiB / 2 bytes per sample
LUT_STEP = 360                       # mag_lut[b] = 360 * b, b in 0..128


def lut_image_bytes(mag_bytes: np.ndarray, seed: int = 0) -> np.ndarray:
    """mag_bytes[i] (0..128) = the byte the as-written converter must read for output sample i.  Returns the uc8
    "IQ" buffer (2 bytes per sample) that puts them where the converter looks: the first half of each block; the
    second half, which the converter never reads, is filled with random bytes."""
    n = len(mag_bytes)
    B = LUT_BLOCK_SAMPLES
    rng = np.random.default_rng(seed)
    out = rng.integers(0, 256, size=2 * n, dtype=np.uint8)
    for b0 in range(0, n, B):
        cnt = min(B, n - b0)
        out[2 * b0: 2 * b0 + cnt] = mag_bytes[b0: b0 + cnt]
    return out


def capture_lut_image(nsamples: int, sample_rate: float = 2.4e6, seed: int = 1, **kw):
    """A uc8 buffer whose image under the as-written byte table is the envelope of capture_complex()'s signal
    (full scale 1.0 -> byte 128 -> magnitude 46080).  Returns (uint8[2 * nsamples], frames)."""
    x, frames = capture_complex(nsamples, sample_rate, seed, **kw)
    mb = np.clip(np.round(np.abs(x) * 128.0), 0, 128).astype(np.uint8)
    return lut_image_bytes(mb, seed), frames


def capture_complex(nsamples: int, sample_rate: float = 2.4e6, seed: int = 1, frames_per_sec: float = 1000.0,
                    n_icao: int = 16, noise_sigma: float = 0.01, overlap_fraction: float = 0.1, bitflip1: float = 0.0,
                    bitflip2: float = 0.0, dc_offset: complex = 0.0, ap_fraction: float = 0.0, min_amp_db: float = -32.0,
                    max_amp_db: float = -4.0):
    """Complex baseband (full scale 1.0) and the list of injected frames."""
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(nsamples) + 1j * rng.standard_normal(nsamples)) * noise_sigma + dc_offset
    icaos = rng.integers(0x400000, 0x4FFFFF, size=n_icao)
    nframes = int(frames_per_sec * nsamples / sample_rate)
    frames = []
    gap = int(150e-6 * sample_rate)
    pos = rng.integers(0, gap)
    mean_step = nsamples / max(nframes, 1)
    for _ in range(nframes):
        kind = rng.random()
        icao = int(icaos[rng.integers(0, n_icao)])
        if kind < ap_fraction:
            df = int(rng.choice([0, 4, 5, 16, 20, 21]))
            clean = make_ap(df, icao, bytes(rng.integers(0, 256, size=11, dtype=np.uint8)))
        elif kind < ap_fraction + (1 - ap_fraction) * 0.6:
            clean = make_df17(icao, bytes(rng.integers(0, 256, size=7, dtype=np.uint8)))
        else:
            clean = make_df11(icao)
        msg = bytearray(clean)
        r = rng.random()
        nflip = 2 if r < bitflip2 else 1 if r < bitflip2 + bitflip1 else 0
        for b in rng.choice(np.arange(5, 8 * len(msg)), size=nflip, replace=False):
            msg[b >> 3] ^= 0x80 >> (b & 7)
        amp = 10 ** (rng.uniform(min_amp_db, max_amp_db) / 20.0)
        env = envelope(bytes(msg), sample_rate, rng.random())
        if pos + len(env) >= nsamples:
            break
        ph = np.exp(2j * np.pi * rng.random())
        x[pos:pos + len(env)] += amp * ph * env
        frames.append({"start": int(pos), "msg": bytes(msg), "clean": bytes(clean), "amp": float(amp), "nflip": nflip})
        if rng.random() < overlap_fraction:                       # an overlapping straggler
            step = int(rng.integers(20, len(env)))
        else:
            step = len(env) + gap + int(rng.exponential(max(mean_step - len(env) - gap, 1.0)))
        pos += step
    return x, frames
