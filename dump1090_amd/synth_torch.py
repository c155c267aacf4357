"""Large synthetic captures generated on the GPU with torch (bench.py): same signal model as
synth.py (box-integrated OOK pulses, random sub-sample phase / carrier phase / amplitude, AWGN),
frames drawn from a bank of pre-rendered envelopes so that multi-GiB recordings take seconds."""
import numpy as np
import torch

from . import synth


def _bank(rng, sample_rate, n_bank, n_icao, df17_fraction):
    icaos = rng.integers(0x400000, 0x4FFFFF, size=n_icao)
    envs, msgs = [], []
    for _ in range(n_bank):
        icao = int(icaos[rng.integers(0, n_icao)])
        if rng.random() < df17_fraction:
            msg = synth.make_df17(icao, bytes(rng.integers(0, 256, size=7, dtype=np.uint8)))
        else:
            msg = synth.make_df11(icao)
        envs.append(synth.envelope(msg, sample_rate, rng.random()))
        msgs.append(msg)
    L = max(len(e) for e in envs)
    bank = np.zeros((n_bank, L), dtype=np.float32)
    for i, e in enumerate(envs):
        bank[i, :len(e)] = e
    return bank, msgs


def capture_gpu(nsamples, sample_rate=2.4e6, fmt="uc8", seed=1, frames_per_sec=1000.0, device="cuda:0",
                noise_sigma=0.01, n_icao=64, n_bank=4096, overlap_fraction=0.1, df17_fraction=0.6,
                min_amp_db=-32.0, max_amp_db=-4.0, dc_offset=(0.0, 0.0), chunk=1 << 24, sample_range=None):
    """Returns (iq tensor on `device`: uint8[2n] or int16[2n], dict with frame positions / bank indices).
    fmt "uc8lut": a uc8 buffer whose image under the reference's as-written byte table is the signal's envelope
    (synth.capture_lut_image): byte = round(|x| * 128) in the first half of every 256 KiB block, random bytes in the
    half the converter never reads.
    sample_range = (a, b), multiples of `chunk`: only samples [a, b) of the recording are generated (every chunk has
    its own noise seed, so any rank can form its part of one long recording); the tensor then holds 2 (b - a) values."""
    rng = np.random.default_rng(seed)
    bank, msgs = _bank(rng, sample_rate, n_bank, n_icao, df17_fraction)
    L = bank.shape[1]
    nframes = int(frames_per_sec * nsamples / sample_rate)
    gap = int(150e-6 * sample_rate)
    mean_step = nsamples / max(nframes, 1)
    steps = L + gap + rng.exponential(max(mean_step - L - gap, 1.0), size=nframes).astype(np.int64)
    pos = np.cumsum(steps) - steps[0] + rng.integers(0, gap)
    pos = pos[pos + L < nsamples]
    n_over = int(len(pos) * overlap_fraction)
    if n_over:
        base = rng.choice(pos, size=n_over, replace=False)
        extra = base + rng.integers(20, L, size=n_over)
        pos = np.sort(np.concatenate([pos, extra[extra + L < nsamples]]))
    nf = len(pos)
    which = rng.integers(0, n_bank, size=nf)
    amp = 10 ** (rng.uniform(min_amp_db, max_amp_db, size=nf) / 20.0)
    phi = rng.uniform(0, 2 * np.pi, size=nf)

    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    bank_t = torch.from_numpy(bank).to(dev)
    r0, r1 = (0, nsamples) if sample_range is None else (max(0, sample_range[0]), min(nsamples, sample_range[1]))
    assert r0 % chunk == 0 and (r1 % chunk == 0 or r1 == nsamples), "sample_range must lie on the chunk grid"
    whole = out = torch.empty(2 * (r1 - r0), dtype=torch.uint8 if fmt in ("uc8", "uc8lut") else torch.int16, device=dev)
    if fmt == "uc8lut":
        B = synth.LUT_BLOCK_SAMPLES
        assert chunk % B == 0
    ar = torch.arange(L, device=dev)
    for c0 in range(r0, r1, chunk):
        c1 = min(c0 + chunk, nsamples)
        n = c1 - c0
        gen.manual_seed(seed * 1000003 + c0 // chunk)            # a chunk's noise does not depend on the chunks before it
        out = whole[2 * (c0 - r0):]                              # this chunk's part: indices below are relative to r0
        if fmt == "uc8lut":
            out[: 2 * n].random_(0, 256, generator=gen)
        x = torch.randn(2, n + L, generator=gen, device=dev, dtype=torch.float32) * noise_sigma
        x[0] += dc_offset[0]
        x[1] += dc_offset[1]
        # frames that start inside [c0 - L, c1): the part falling in this chunk is added
        lo, hi = np.searchsorted(pos, c0 - L, side="left"), np.searchsorted(pos, c1, side="left")
        if hi > lo:
            p = torch.from_numpy(pos[lo:hi] - c0).to(dev)
            w = torch.from_numpy(which[lo:hi]).to(dev)
            a = torch.from_numpy(amp[lo:hi].astype(np.float32)).to(dev)
            ph = torch.from_numpy(phi[lo:hi].astype(np.float32)).to(dev)
            idx = p[:, None] + ar[None, :]
            env = bank_t[w] * a[:, None]
            ok = (idx >= 0) & (idx < n)
            idx = idx.clamp(0, n + L - 1)
            x[0].index_put_((idx[ok],), (env * torch.cos(ph)[:, None])[ok], accumulate=True)
            x[1].index_put_((idx[ok],), (env * torch.sin(ph)[:, None])[ok], accumulate=True)
        if fmt == "uc8lut":
            mb = torch.clamp(torch.round(torch.sqrt(x[0, :n] ** 2 + x[1, :n] ** 2) * 128.0), 0, 128).to(torch.uint8)
            nfull = n // B
            if nfull:
                out[: 2 * nfull * B].view(nfull, 2, B)[:, 0, :] = mb[: nfull * B].view(nfull, B)
            if n > nfull * B:
                out[2 * nfull * B: 2 * nfull * B + (n - nfull * B)] = mb[nfull * B:]
            del x, mb
            continue
        iq = x[:, :n].t().contiguous().view(-1)                  # interleave I,Q
        if fmt == "uc8":
            q = torch.clamp(torch.round(iq * 127.5 + 127.5), 0, 255).to(torch.uint8)
        elif fmt == "sc16":
            q = torch.clamp(torch.round(iq * 32767.0), -32768, 32767).to(torch.int16)
        else:
            q = torch.clamp(torch.round(iq * 2047.0), -2048, 2047).to(torch.int16)
        out[: 2 * n] = q
        del x, iq, q
    return whole, {"pos": pos, "which": which, "msgs": msgs, "amp": amp}
