"""The ctypes binding table (lic360/_abi_table.py) must be exactly what include/lic360_hip.h declares, and every
entry point must carry argtypes (a Python int that does not fit a C `int` raises instead of being truncated)."""
import ctypes
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gen():
    spec = importlib.util.spec_from_file_location("gen_abi", os.path.join(ROOT, "tools", "gen_abi.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_table_matches_header():
    g = _gen()
    from lic360._abi_table import ABI
    assert g.parse(open(g.HEADER).read()) == ABI, "run tools/gen_abi.py after editing include/lic360_hip.h"


def test_every_entry_point_is_typed():
    import lic360
    from lic360._abi_table import ABI
    assert len(ABI) > 60
    for name, (res, args) in ABI.items():
        fn = getattr(lic360._lib, name)
        assert fn.argtypes is not None and len(fn.argtypes) == len(args), name
    # a value that does not fit `int` is refused, not truncated (no GPU call happens: the conversion fails first)
    idx = (ctypes.c_int * 8)()
    with pytest.raises(ctypes.ArgumentError):
        lic360._lib.lic360_code_contex(2 ** 40, 2, idx, idx)


def test_layout_queries_and_buffer_sizes_run_without_a_gpu():
    """the layout functions are plain host arithmetic: buffer sizes include the slack the band fetches of the last plane may touch,
    so callers never add a magic number (round-1 verdict: the 16 KB over-read contract)"""
    import lic360
    L = lic360._lib
    rows, pitch, r0, c0 = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert L.lic360_dc4_layout(64, 128, ctypes.byref(rows), ctypes.byref(pitch), ctypes.byref(r0), ctypes.byref(c0)) == 0
    planes = 3 * 192
    need = L.lic360_conv4_buffer_floats(0, planes, 64, 128)
    assert need > planes * rows.value * pitch.value and need - planes * rows.value * pitch.value <= 1 << 14
    hp, wp = ctypes.c_int(), ctypes.c_int()
    assert L.lic360_ec16_layout(64, 128, ctypes.byref(hp), ctypes.byref(wp)) == 0 and (hp.value, wp.value) == (68, 132)
    assert L.lic360_conv4_buffer_floats(7, planes, 64, 128) == 0 and L.lic360_conv4_buffer_floats(0, 0, 64, 128) == 0
    # importance-net decode layout: any map height (maps taller than a task window are cut into diagonal segments)
    for h, w in ((32, 64), (64, 128), (5, 3)):
        assert L.lic360_dc144_layout(h, w, ctypes.byref(rows), ctypes.byref(pitch)) == 0
        assert rows.value == h + w - 1 + 8 and pitch.value % 4 == 0 and pitch.value >= h + 36


def test_kernel_names_of_the_bench_classes_exist_in_the_library():
    """bench.py attaches committed PMC traffic to a kernel class only when the entry's kernels are the ones lic360_codec_kernel_names lists
    for it; every base name listed there must be a kernel of the loaded library (its mangled name holds the base name)"""
    import lic360
    blob = open(lic360.LIBRARY_PATH, "rb").read()
    classes = dict(p.split("=") for p in lic360._lib.lic360_codec_kernel_names().decode().split(";"))
    assert set(classes) == {"ec_first", "ec_hidden", "ec_last", "dc_first", "dc_hidden", "dc_last", "imp_ec", "imp_dc"}
    for cls, names in classes.items():
        for name in names.split("+"):
            base = name.split("<")[0].strip()
            assert ("%d%s" % (len(base), base)).encode() in blob, (cls, name)       # Itanium mangling: <length><identifier>
