"""The DEVICE arithmetic coder (k_ac_encode, k_dec_init, k_dec_plane, k_imp_dec_plane of csrc/codec_fused.hip) against the
fixtures produced by the REFERENCE coder compiled in the build container (tests/golden/ac_golden.npz, oracle/gen_golden.py:
extension/ArithmeticCoder.cpp + BitIoStream.cpp): raw tables + symbols go through the C-ABI test hooks
lic360_devcoder_encode / _decode, bytes must be identical and every symbol must come back."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest
import torch

from gen_golden import FIXED, draw_fixed

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ac_golden.npz")


@pytest.fixture(scope="module")
def lic():
    import lic360
    assert torch.cuda.is_available()
    return lic360


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLD)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def dev_encode(lic, tab, ncode, lab, mask):
    n = len(lab)
    cap = (max(4096, 3 * n) + 3) // 4 * 4
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda:0")
    nb = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    er = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    t, l = dev(tab.astype(np.int32)), dev(lab.astype(np.int32))
    m = None if mask is None else dev(mask.astype(np.float32))
    lic._chk(lic._lib.lic360_devcoder_encode(lic._stream(0), lic._p(t), ncode, lic._p(l), lic._p(m), n, lic._p(out), cap, lic._p(nb), lic._p(er)))
    assert int(er.item()) == 0
    return bytes(out[:int(nb.item())].cpu().numpy().tobytes())


def dev_decode(lic, data, tab, ncode, mask, n, chunk):
    cap = (max(4096, len(data) + 8) + 3) // 4 * 4
    buf = torch.zeros(cap, dtype=torch.uint8, device="cuda:0")
    if len(data):
        buf[:len(data)] = dev(np.frombuffer(data, np.uint8).copy())
    nb = torch.tensor([len(data)], dtype=torch.int32, device="cuda:0")
    er = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    out = torch.full((max(n, 1),), -1.0, dtype=torch.float32, device="cuda:0")
    t = dev(tab.astype(np.int32))
    m = None if mask is None else dev(mask.astype(np.float32))
    lic._chk(lic._lib.lic360_devcoder_decode(lic._stream(0), lic._p(t), ncode, lic._p(m), n, chunk, lic._p(buf), cap, lic._p(nb), lic._p(out), lic._p(er)))
    return out[:n].cpu().numpy(), int(er.item())


@pytest.mark.parametrize("name,ncode,masked", [("rand8", 8, True), ("rand49", 49, False), ("skew8", 8, False)])
@pytest.mark.parametrize("chunk", [64, 100, 4096])
def test_reference_fixtures_through_the_device_coder(lic, golden, name, ncode, masked, chunk):
    tab, lab = golden[name + "_tables"], golden[name + "_labels"].astype(np.int32)
    mask = golden[name + "_mask"].astype(np.float32) if masked else None
    want = golden[name + "_bytes"].tobytes()
    if chunk == 64:                                                  # the encoder has no chunk parameter: once is enough
        assert dev_encode(lic, tab, ncode, lab, mask) == want
    got, err = dev_decode(lic, want, tab, ncode, mask, len(lab), chunk)
    assert err == 0
    keep = np.ones(len(lab), bool) if mask is None else mask > 0.5
    assert np.array_equal(got[keep].astype(np.int32), lab[keep])
    assert np.all(got[~keep] == 0)


def test_fixed_small_and_empty(lic, golden):
    lab = golden["fixed_small_labels"].astype(np.int32)
    tab = np.tile(FIXED, (len(lab), 1))
    want = golden["fixed_small_bytes"].tobytes()
    assert dev_encode(lic, tab, 8, lab, None) == want
    got, err = dev_decode(lic, want, tab, 8, None, len(lab), 333)
    assert err == 0 and np.array_equal(got.astype(np.int32), lab)
    assert dev_encode(lic, np.zeros((0, 9), np.int32), 8, np.zeros(0, np.int32), None) == golden["empty_bytes"].tobytes()


def test_config1_full_latent_through_the_device_coder(lic, golden):
    """BASELINE.json configs[0] (393 216 symbols of a 32x64x192 latent on the fixed CDF), here on the device coder:
    SHA-256 of the bytes == the reference coder's, and the decode returns every symbol."""
    n = 393216
    lab = draw_fixed(1234, n)
    assert int(lab.sum()) == int(golden["fixed_full_label_sum"][0])
    tab = np.tile(FIXED, (n, 1))
    data = dev_encode(lic, tab, 8, lab, None)
    assert len(data) == int(golden["fixed_full_nbytes"][0])
    assert hashlib.sha256(data).digest() == golden["fixed_full_sha256"].tobytes()
    got, err = dev_decode(lic, data, tab, 8, None, n, 3072)           # 3072 = the widest plane of a 48x64x128 latent
    assert err == 0 and np.array_equal(got.astype(np.int32), lab)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_full_range_recurrence_and_last_symbol(lic, seed):
    """The corner cases of the scalar-assembly symbol steps (csrc/codec_fused.hip): with T[1] a power of two, every
    symbol 0 coded while low = 0 and high = 2^32 - 1 leaves exactly that state, so range == 2^32 RECURS (the interval starts are then
    T << 16 themselves, not the 32-bit product); symbol 7 ends at T[8] = 65536, which does not fit the 16-bit table words; tables
    with entries one apart give the longest shift runs.  Device bytes == the oracle coder's bytes, and every symbol comes back."""
    import oracle as orc
    rng = np.random.default_rng(900 + seed)
    n = 6000
    tab = np.zeros((n, 9), np.int32)
    kinds = rng.integers(0, 4, n)
    for i in range(n):
        if kinds[i] == 0:   tab[i] = [0, 1 << 15, 40000, 45000, 50000, 55000, 60000, 65000, 65536]     # T[1] = 2^15
        elif kinds[i] == 1: tab[i] = [0, 1 << 14, 1 << 15, 3 << 14, 50000, 50001, 50002, 50003, 65536]  # powers of two + width-1 symbols
        elif kinds[i] == 2: tab[i] = [0, 1, 2, 3, 4, 5, 6, 7, 65536]                                  # all the mass on the last symbol
        else:               tab[i] = np.concatenate([[0], np.sort(rng.choice(np.arange(1, 65536), 7, replace=False)), [65536]])
    lab = np.zeros(n, np.int32)
    lab[n // 3:] = rng.integers(0, 8, n - n // 3)                     # a long run of zeros first (range stays 2^32), then everything
    lab[rng.random(n) < 0.15] = 7
    lab[:64] = 0
    mask = None if seed == 0 else (rng.random(n) > 0.3).astype(np.float32)
    e = orc.Encoder()
    e.encode(tab, 8, lab, mask, n)
    want = e.finish()
    assert dev_encode(lic, tab, 8, lab, mask) == want
    for chunk in (64, 1000):
        got, err = dev_decode(lic, want, tab, 8, mask, n, chunk)
        keep = np.ones(n, bool) if mask is None else mask > 0.5
        assert err == 0 and np.array_equal(got[keep].astype(np.int32), lab[keep]) and np.all(got[~keep] == 0)


def test_truncated_stream_and_oversized_length_are_flagged(lic, golden):
    tab, lab = golden["rand8_tables"], golden["rand8_labels"].astype(np.int32)
    want = golden["rand8_bytes"].tobytes()
    got, err = dev_decode(lic, want[:len(want) // 3], tab, 8, None, len(lab), 256)     # zeros past the end, like the reference's BitInputStream
    assert got.min() >= 0 and got.max() <= 7
    # a device-side length beyond the slot is clamped and reported (error bit 32), never read
    cap = 4096
    buf = torch.zeros(cap, dtype=torch.uint8, device="cuda:0")
    nb = torch.tensor([1 << 30], dtype=torch.int32, device="cuda:0")
    er = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    out = torch.zeros(64, dtype=torch.float32, device="cuda:0")
    t = dev(tab[:64].astype(np.int32))
    lic._chk(lic._lib.lic360_devcoder_decode(lic._stream(0), lic._p(t), 8, None, 64, 64, lic._p(buf), cap, lic._p(nb), lic._p(out), lic._p(er)))
    assert int(er.item()) & 32
