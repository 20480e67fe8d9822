"""Product host Coder (C ABI lic360_coder_*, closed-form renormalisation in csrc/ac_core.h) against the
reference coder's golden bitstreams and against the oracle.  No GPU needed: the Coder API takes host
tables, exactly like the reference's Coder (extension/coder.h:10-63)."""
import os
import re

import numpy as np
import pytest
import torch

import oracle as orc
from gen_golden import FIXED, draw_fixed, rand_tables, draw_from_tables


@pytest.fixture(scope="module")
def lic():
    import lic360
    return lic360


def _roundtrip(lic, tmp_path, tab, ncode, lab, mask, cuts=None):
    f = str(tmp_path / "code.bin")
    c = lic.Coder("tmp", 3.5)
    c.reset_fname(f)
    c.start_encoder()
    n = len(lab)
    cuts = cuts or [0, n]
    tt, ll = torch.from_numpy(np.ascontiguousarray(tab, np.int32)), torch.from_numpy(np.ascontiguousarray(lab, np.int32))
    mm = None if mask is None else torch.from_numpy(np.ascontiguousarray(mask, np.float32))
    for a, b in zip(cuts[:-1], cuts[1:]):
        if mm is None:
            c.encodes(tt[a:b], ncode, ll[a:b], b - a)
        else:
            c.encodes_mask(tt[a:b], ncode, ll[a:b], mm[a:b], b - a)
    c.end_encoder()
    data = open(f, "rb").read()
    c.start_decoder()
    outs = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        o = c.decodes(tt[a:b], ncode, b - a) if mm is None else c.decodes_mask(tt[a:b], ncode, mm[a:b], b - a)
        outs.append(o.numpy())
    return data, np.concatenate(outs) if outs else np.zeros(0, np.float32)


@pytest.mark.parametrize("name,ncode,masked", [("rand8", 8, True), ("rand49", 49, False), ("skew8", 8, False)])
def test_golden(lic, golden, tmp_path, name, ncode, masked):
    tab, lab = golden[name + "_tables"].astype(np.int32), golden[name + "_labels"].astype(np.int32)
    mask = golden[name + "_mask"].astype(np.float32) if masked else None
    data, dec = _roundtrip(lic, tmp_path, tab, ncode, lab, mask, cuts=[0, 3, 500, len(lab)])
    assert data == golden[name + "_bytes"].tobytes()
    if masked:
        assert np.array_equal(dec[mask > 0.5], lab[mask > 0.5]) and np.all(dec[mask < 0.5] == 3.5)
    else:
        assert np.array_equal(dec.astype(np.int32), lab)


def test_config1_full(lic, golden, tmp_path):
    import hashlib
    n = 393216
    lab = draw_fixed(1234, n)
    data, dec = _roundtrip(lic, tmp_path, np.tile(FIXED, (n, 1)), 8, lab, None)
    assert hashlib.sha256(data).digest() == golden["fixed_full_sha256"].tobytes()
    assert np.array_equal(dec.astype(np.int32), lab)


def test_empty_and_errors(lic, golden, tmp_path):
    data, _ = _roundtrip(lic, tmp_path, np.zeros((0, 9), np.int32), 8, np.zeros(0, np.int32), None, cuts=[0, 0])
    assert data == b"\x80"
    c = lic.Coder(str(tmp_path / "e.bin"), 3.5)
    c.start_encoder()
    bad = torch.tensor([[0, 5, 5, 65536]], dtype=torch.int32)          # zero-frequency symbol 1
    with pytest.raises(RuntimeError):
        c.encodes(bad, 3, torch.tensor([1], dtype=torch.int32), 1)


@pytest.mark.parametrize("seed", range(4))
def test_random_vs_oracle(lic, tmp_path, seed):
    rng = np.random.default_rng(100 + seed)
    n, ncode = 5000, [8, 49, 2, 8][seed]
    tab = rand_tables(rng, n, ncode, skew=(seed == 3))
    lab = draw_from_tables(rng, tab)
    mask = (rng.random(n) > 0.5).astype(np.float32)
    data, dec = _roundtrip(lic, tmp_path, tab, ncode, lab, mask)
    e = orc.Encoder()
    e.encode(tab, ncode, lab, mask, n)
    assert data == e.finish()
    assert np.array_equal(dec[mask > 0.5], lab[mask > 0.5])


def test_c_abi_exports_every_declared_symbol(lic):
    """include/lic360_hip.h is the boundary: every declared entry point must be exported by the .so."""
    import ctypes
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for hdr in os.listdir(os.path.join(root, "include")):
        txt = open(os.path.join(root, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(lic360_[a-z0-9_]+)\s*\(", txt))
    assert len(names) > 30
    so = ctypes.CDLL(lic.LIBRARY_PATH)
    missing = [n for n in sorted(names) if not hasattr(so, n)]
    assert not missing, missing
