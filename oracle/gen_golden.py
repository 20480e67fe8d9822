#!/usr/bin/env python3
"""Generate tests/golden/ac_golden.npz with the REFERENCE arithmetic coder.

Runs only where /root/reference is mounted (this container): `make -C oracle ref` compiles
extension/ArithmeticCoder.cpp + extension/BitIoStream.cpp in place into oracle/_ref/, and
this script drives it through oracle/ref_driver.cpp.  The outputs are DATA (tables, symbols,
masks, reference bitstreams / digests) -- no reference source text is stored.

Cases (SURVEY.md §8c pins (1), §8d config 1):
  fixed_small   4096 symbols, config-1 fixed 9-entry CDF
  fixed_full    393 216 symbols (= 32x64x192 latent), config-1 CDF; sha256 + length only
  rand8_mask    4096 symbols, per-symbol random strictly increasing 9-entry tables, 40 % masked
  rand49        2048 symbols, 49-symbol alphabet (importance-map codec), unmasked
  skew8         4096 symbols, near-degenerate tables (frequencies of 1) -> long underflow runs
  empty         0 symbols -> the bare terminator byte
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle as orc  # noqa: E402

FIXED = np.array([0, 1200, 5200, 14000, 32768, 51536, 60336, 64336, 65536], np.int32)


def xorshift32(seed, n):
    """Documented, dependency-free symbol source (so fixed_full needs no stored symbols)."""
    out = np.empty(n, np.uint32)
    x = np.uint32(seed)
    for i in range(n):
        x ^= np.uint32((int(x) << 13) & 0xFFFFFFFF)
        x ^= np.uint32(int(x) >> 17)
        x ^= np.uint32((int(x) << 5) & 0xFFFFFFFF)
        out[i] = x
    return out


def draw_fixed(seed, n):
    u = (xorshift32(seed, n) >> np.uint32(16)).astype(np.int64)     # 16-bit uniform
    return (np.searchsorted(FIXED, u, side="right") - 1).astype(np.int32)


def rand_tables(rng, n, ncode, total=65536, skew=False):
    t = np.zeros((n, ncode + 1), np.int64)
    for i in range(n):
        if skew:
            big = rng.integers(0, ncode)
            f = np.ones(ncode, np.int64)
            f[big] = total - (ncode - 1)
        else:
            cuts = np.sort(rng.choice(np.arange(1, total), size=ncode - 1, replace=False))
            f = np.diff(np.concatenate([[0], cuts, [total]]))
        t[i, 1:] = np.cumsum(f)
    return t.astype(np.int32)


def draw_from_tables(rng, t):
    n, m = t.shape
    u = rng.integers(0, t[:, -1])
    return np.array([np.searchsorted(t[i], u[i], side="right") - 1 for i in range(n)], np.int32)


def main():
    orc.build(ref=True)
    assert orc.have_ref(), "reference coder not built"
    out = {}
    # fixed small / full
    for name, n, seed in (("fixed_small", 4096, 1234), ("fixed_full", 393216, 1234)):
        lab = draw_fixed(seed, n)
        tab = np.tile(FIXED, (n, 1))
        data = orc.ref_encode(tab, 8, lab, None)
        dec = orc.ref_decode(data, tab, 8, None, n)
        assert np.array_equal(dec.astype(np.int32), lab)
        if name == "fixed_small":
            out[name + "_labels"] = lab.astype(np.uint8)
            out[name + "_bytes"] = np.frombuffer(data, np.uint8)
        else:
            out[name + "_sha256"] = np.frombuffer(hashlib.sha256(data).digest(), np.uint8)
            out[name + "_nbytes"] = np.array([len(data)], np.int64)
            out[name + "_head"] = np.frombuffer(data[:64], np.uint8)
            out[name + "_tail"] = np.frombuffer(data[-64:], np.uint8)
            out[name + "_label_sum"] = np.array([int(lab.sum())], np.int64)
    rng = np.random.default_rng(2024)
    # random 8-symbol tables with mask
    n = 4096
    tab = rand_tables(rng, n, 8)
    lab = draw_from_tables(rng, tab)
    mask = (rng.random(n) > 0.4).astype(np.float32)
    data = orc.ref_encode(tab, 8, lab, mask)
    dec = orc.ref_decode(data, tab, 8, mask, n)
    assert np.array_equal(dec[mask > 0.5].astype(np.int32), lab[mask > 0.5]) and np.all(dec[mask < 0.5] == 3.5)
    out.update(rand8_tables=tab.astype(np.uint32).astype(np.int32), rand8_labels=lab.astype(np.uint8),
               rand8_mask=mask.astype(np.uint8), rand8_bytes=np.frombuffer(data, np.uint8))
    # 49-symbol alphabet
    n = 2048
    tab = rand_tables(rng, n, 49)
    lab = draw_from_tables(rng, tab)
    data = orc.ref_encode(tab, 49, lab, None)
    assert np.array_equal(orc.ref_decode(data, tab, 49, None, n).astype(np.int32), lab)
    out.update(rand49_tables=tab, rand49_labels=lab.astype(np.uint8), rand49_bytes=np.frombuffer(data, np.uint8))
    # skewed (underflow stress): mostly the big symbol, 10 % uniformly random rare ones
    n = 4096
    tab = rand_tables(rng, n, 8, skew=True)
    lab = draw_from_tables(rng, tab)
    rare = rng.random(n) < 0.1
    lab[rare] = rng.integers(0, 8, rare.sum()).astype(np.int32)
    data = orc.ref_encode(tab, 8, lab, None)
    assert np.array_equal(orc.ref_decode(data, tab, 8, None, n).astype(np.int32), lab)
    out.update(skew8_tables=tab, skew8_labels=lab.astype(np.uint8), skew8_bytes=np.frombuffer(data, np.uint8))
    # empty stream
    data = orc.ref_encode(np.zeros((0, 9), np.int32), 8, np.zeros(0, np.int32), None)
    out["empty_bytes"] = np.frombuffer(data, np.uint8)
    dst = os.path.join(HERE, "..", "tests", "golden", "ac_golden.npz")
    np.savez_compressed(dst, **out)
    print("wrote", os.path.normpath(dst), os.path.getsize(dst), "bytes; fixed_full nbytes =", int(out["fixed_full_nbytes"][0]))


if __name__ == "__main__":
    main()
