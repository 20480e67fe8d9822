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
