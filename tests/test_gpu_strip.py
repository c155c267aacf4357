"""--strip (ref: src/dump1090.c:3423-3448) on the device against the oracle's sequential loop: byte-exact output,
the run counter carried across pieces of a stream, and the reference's own fixture: testfiles/modes1.bin was made
with `--strip 25` (README.md:482-489), so stripping it again at that level must give it back unchanged."""
import numpy as np
import pytest
import torch

import dump1090_amd as da
from dump1090_amd import synth
from oracle import orc

pytestmark = pytest.mark.gpu


def gpu_strip(iq, level, run=0):
    d = torch.from_numpy(np.ascontiguousarray(iq)).cuda()
    out = torch.zeros(max(iq.size, 2), dtype=torch.uint8, device="cuda")
    n, r = da.strip_device(d.data_ptr(), iq.size, level, out.data_ptr(), run=run)
    return out[:n].cpu().numpy(), r


def test_modes1_is_a_fixed_point_at_level_25(modes1):
    got, _ = gpu_strip(modes1, 25)
    assert np.array_equal(got, modes1)
    less, _ = gpu_strip(modes1, 30)                 # ... and at a higher level it is not
    exp, _ = orc.strip(modes1, 30)
    assert less.size < modes1.size and np.array_equal(less, exp)


@pytest.mark.parametrize("level", [1, 4, 12, 25, 127, 200])
def test_synthetic_matches_oracle(level):
    iq, _ = synth.capture(700_001, 2.4e6, "uc8", seed=50 + level, frames_per_sec=2500, noise_sigma=0.02)
    got, r = gpu_strip(iq, level)
    exp, c = orc.strip(iq, level)
    assert np.array_equal(got, exp)
    assert r == min(c, 32)
    if level in (12, 25):
        assert 0 < exp.size < iq.size


@pytest.mark.parametrize("nbytes", [0, 1, 2, 3, 64, 65, 66, 127, 128, 8191, 8192, 8194, 4096 * 2 * 3 + 10])
def test_ragged_sizes_and_all_quiet(nbytes):
    rng = np.random.default_rng(nbytes)
    iq = rng.integers(120, 135, size=nbytes, dtype=np.uint8)
    for level in (3, 9):
        got, r = gpu_strip(iq, level)
        exp, c = orc.strip(iq, level)
        assert np.array_equal(got, exp) and r == min(c, 32)
    quiet = np.full(nbytes, 127, dtype=np.uint8)
    got, r = gpu_strip(quiet, 5, run=7)
    exp, c = orc.strip(quiet, 5, run=7)
    assert np.array_equal(got, exp) and r == min(c, 32)


def test_pieces_with_carried_run_equal_one_pass():
    rng = np.random.default_rng(3)
    iq = rng.integers(118, 137, size=2 * 300_000, dtype=np.uint8)
    iq[2 * 100_000: 2 * 100_400] = 127                # a long quiet run across a cut
    exp, _ = orc.strip(iq, 8)
    cuts = [0, 2 * 99_990, 2 * 100_010, 2 * 100_016, 2 * 100_200, 2 * 250_000, iq.size]
    run, parts = 0, []
    for a, b in zip(cuts[:-1], cuts[1:]):
        piece = np.ascontiguousarray(iq[a:b])
        got, run = gpu_strip(piece, 8, run=run)
        parts.append(got)
    assert np.array_equal(np.concatenate(parts), exp)
