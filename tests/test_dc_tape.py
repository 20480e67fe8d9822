"""Host logic of the decode-order kernel's TAPE packing (csrc/cconv4v6_dc.inc: dc6_build_tape, through the host-only entry
lic360_dc4_tape_layout): on every plane of several layer shapes, the pieces the launcher hands to the kernel obey the rules the kernel's
lane / column arithmetic relies on, and cover every row of every sample of a tape exactly once.  No GPU work (the library only has to load).

What the tape replaces: one task (= one wave per group and lane class) per sample of an anti-diagonal plane of the latent nets' masked
convolution in decode order (reference: extension/cconv_dc_cuda.cu:313-398, one launch per plane, lane = image row)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "360-image-compression_amd"))

NB_MAX, NW_MAX, PS = 24, 6, 3


@pytest.fixture(scope="module")
def L():
    import lic360 as lic
    return lic._lib


def layout(L, G, cin, n, nb, h, w, psum, x_mod):
    tape_c, n_blocks = C.c_int(), C.c_int()
    blocks, nwaves = (C.c_int * NB_MAX)(), (C.c_int * NB_MAX)()
    windows = (C.c_uint * (NB_MAX * NW_MAX * 3))()
    assert L.lic360_dc4_tape_layout(G, cin, n, nb, h, w, psum, x_mod, C.byref(tape_c), C.byref(n_blocks), blocks, nwaves, windows) == 0
    return tape_c.value, n_blocks.value, list(blocks), list(nwaves), np.array(windows, dtype=np.uint32).reshape(NB_MAX, NW_MAX, 3)


def rows_of_block(G, h, w, psum, g0):
    """union of the row ranges of the block's (up to) three staggered diagonals"""
    lo, hi = 1 << 30, -1
    for q in range(PS):
        s = psum - g0 - q
        if g0 + q >= G or s < 0 or s >= h + w - 1:
            continue
        lo, hi = min(lo, s - w + 1 if s >= w else 0), max(hi, s if s < h else h - 1)
    return lo, hi


CASES = [(48, 4, 144, 3, 64, 128), (48, 1, 144, 3, 64, 128), (48, 4, 48, 3, 64, 128), (12, 4, 40, 1, 50, 12), (12, 4, 64, 2, 33, 9),
         (9, 4, 72, 1, 64, 30), (15, 4, 32, 2, 30, 30), (6, 4, 144, 3, 18, 7), (48, 4, 96, 2, 64, 20), (20, 4, 80, 2, 47, 61)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "g%d_cin%d_n%d_nb%d_%dx%d" % c)
def test_tape_pieces_obey_the_kernel_rules(L, case):
    G, cin, n, nb, h, w = case
    taped = saved = 0
    for psum in range(h + w + G - 2):
        c, nblk, blocks, nwaves, win = layout(L, G, cin, n, nb, h, w, psum, n)
        if nblk == 0 or c == 0:
            continue
        taped += 1
        assert 2 <= c <= NW_MAX and (n // nb // 8) % c == 0 and n % 8 == 0
        tasks = plain = 0
        for j in range(nblk):
            lo, hi = rows_of_block(G, h, w, psum, blocks[j])
            assert hi >= 0, (psum, j)
            assert 1 <= nwaves[j] <= c, "a tape never needs more tasks than one per sample"
            tasks, plain = tasks + nwaves[j], plain + c
            if nwaves[j] == c:                                           # this block keeps plain tasks (one sample each): no pieces
                assert not win[j].any()
                continue
            cover = np.zeros((c, h), np.int32)
            for t in range(NW_MAX):
                pieces = [int(v) for v in win[j, t] if v]
                assert (t < nwaves[j]) == bool(pieces), (psum, j, t)
                assert [bool(v) for v in win[j, t]] == [True] * len(pieces) + [False] * (3 - len(pieces))
                quads_taken, last_a0 = set(), -1
                for wd in pieces:
                    k, slo, shi, a0 = wd & 7, (wd >> 3) & 63, (wd >> 9) & 63, (wd >> 15) & 63
                    rows = shi - slo + 1
                    assert wd >> 21 == 1 and k < c and lo <= slo <= shi <= hi
                    assert (a0 - slo) % 4 == 0, "band quads stay 16-byte aligned"
                    assert a0 > last_a0
                    last_a0 = a0
                    assert a0 + rows - 1 <= (63 if shi == h - 1 else 61), "source lanes + 1, + 2 of the last stored lane"
                    assert a0 >= (0 if slo == 0 else 2), "source lanes - 1, - 2 of the first stored lane"
                    cols = range(a0, a0 + rows + 4)                       # a stored lane reads columns lane .. lane + 4
                    assert cols[-1] < 68
                    q = {col // 4 for col in cols}
                    assert not (q & quads_taken), "pieces of a wave share no band quad"
                    quads_taken |= q
                    cover[k, slo:shi + 1] += 1
            assert (cover[:, lo:hi + 1] == 1).all() and cover.sum() == c * (hi - lo + 1), (psum, j)
        assert tasks < plain, "a launch tapes only where that saves tasks"
        saved += plain - tasks
    assert taped > 0 and saved > 0


def test_tape_is_off_where_its_conditions_fail(L):
    G, h, w = 48, 64, 128
    for (n, nb, x_mod) in [(3, 3, 3), (51, 3, 51), (24, 3, 24), (144, 3, 40)]:           # latency mode, 8 does not divide, one sample per list, x_mod
        for psum in (5, 40, 100, 200):
            assert layout(L, G, 4, n, nb, h, w, psum, x_mod)[0] == 0
    assert layout(L, G, 4, 144, 3, 128, 256, 50, 144)[0] == 0                              # taller than a wave: row segments, not tapes
    assert layout(L, G, 4, 144, 3, h, w, 40, 144)[0] == 6 and layout(L, G, 4, 144, 3, h, w, 40, 48)[0] == 6
    assert layout(L, G, 4, 96, 3, h, w, 40, 96)[0] == 4 and layout(L, G, 4, 48, 3, h, w, 40, 48)[0] == 2
    assert layout(L, G, 4, 144, 3, h, w, 100, 144)[0] == 0                                # full-length diagonals: nothing to pack
