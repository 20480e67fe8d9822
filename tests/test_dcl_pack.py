"""The packing of the dead-cone decode task lists on the CPU (csrc/need.h:dcl_wave_pieces through lic360_dcl_pack_layout, no GPU work): random live row
windows of a chunk of <= 8 samples -- empty ones, full-height ones, one-row ones -- must be covered exactly once by pieces that keep the decode
kernel's lane rules (cconv4v6_dc.inc: (a0 - slo) % 4 == 0, neighbours in different band quads, a0 >= 2 unless the piece starts with row 0, last lane
<= 61 unless it ends with the image's last row, at most three pieces per wave, samples in order)."""
import ctypes as C

import numpy as np
import pytest

import lic360


def pack(h, lo, hi):
    c = len(lo)
    lo_a, hi_a = (C.c_int * c)(*lo), (C.c_int * c)(*hi)
    pieces = (C.c_uint * (6 * c))()
    nw = C.c_int(0)
    assert lic360._lib.lic360_dcl_pack_layout(h, c, lo_a, hi_a, pieces, C.byref(nw)) == 0, lic360._lib.lic360_last_error()
    return [[int(pieces[3 * w + i]) for i in range(3)] for w in range(nw.value)]


def check(h, lo, hi):
    waves = pack(h, lo, hi)
    rows = [np.zeros(h, np.int32) for _ in lo]
    last_k = (-1, -1)
    for wv in waves:
        assert wv[0] >> 21, "a wave holds at least one piece"
        prev_last = None
        for w in wv:
            if not w >> 21:
                continue
            k, slo, shi, a0 = w & 7, (w >> 3) & 63, (w >> 9) & 63, (w >> 15) & 63
            assert lo[k] <= slo <= shi <= hi[k]
            assert (a0 - slo) % 4 == 0 and a0 + shi - slo <= 63
            assert a0 >= 2 or slo == 0
            assert a0 + shi - slo <= 61 or shi == h - 1
            if prev_last is not None:
                assert a0 >= ((prev_last + 4) // 4 + 1) * 4
            prev_last = a0 + shi - slo
            assert (k, slo) > last_k                                       # samples, and rows inside a sample, in order
            last_k = (k, slo)
            rows[k][slo:shi + 1] += 1
    for k in range(len(lo)):
        want = np.zeros(h, np.int32)
        if hi[k] >= lo[k]:
            want[lo[k]:hi[k] + 1] = 1
        assert np.array_equal(rows[k], want), (k, lo, hi)
    live = sum(1 for k in range(len(lo)) if hi[k] >= lo[k])
    return len(waves), live


@pytest.mark.parametrize("h", [64, 50, 20, 7])
def test_random_windows_are_covered_once_by_legal_pieces(h):
    rng = np.random.default_rng(h)
    saved = 0
    for case in range(400):
        c = int(rng.integers(1, 9))
        lo, hi = [], []
        for k in range(c):
            kind = rng.integers(0, 6)
            if kind == 0:
                a, b = 1, 0                                                # no live row
            elif kind == 1:
                a, b = 0, h - 1
            elif kind == 2:
                a = b = int(rng.integers(h))
            else:
                a = int(rng.integers(h)); b = int(rng.integers(a, h))
            lo.append(a); hi.append(b)
        nw, live = check(h, lo, hi)
        assert nw <= max(live, 1) * 2 and (live == 0) == (nw == 0)
        saved += live - nw
    assert saved > 0 or h <= 7                                             # packing does pay on short windows


def test_the_uniform_case_is_round_5s_tape():
    """equal windows for every sample = what lic360_dc4_tape_layout packs (the tape is the special case)"""
    for h, lo, hi, c in ((64, 10, 30, 6), (64, 0, 63, 4), (40, 5, 39, 3), (64, 20, 22, 8)):
        waves = pack(h, [lo] * c, [hi] * c)
        tot = sum(((w >> 9) & 63) - ((w >> 3) & 63) + 1 for wv in waves for w in wv if w >> 21)
        assert tot == c * (hi - lo + 1)
        if hi - lo + 1 <= 18:
            assert len(waves) < c                                           # three short windows share a wave
