"""Row b of SURVEY.md §8: the drop-in claim, checked where the reference tree is mounted (the build container).
On the GPU box /root/reference does not exist and the check is skipped; the shim-only part always runs."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dropin_directory_provides_only_lic360():
    """`360-image-compression_amd/dropin` alone on the path gives the real `lic360` package and no `lic360_operator`."""
    code = ("import sys; sys.path.insert(0, %r); import lic360, importlib.util; "
            "assert lic360.__file__.endswith('360-image-compression_amd/lic360/__init__.py'), lic360.__file__; "
            "assert hasattr(lic360, 'CconvEcOp') and hasattr(lic360, 'Coder'); "
            "assert importlib.util.find_spec('lic360_operator') is None") % os.path.join(ROOT, "360-image-compression_amd", "dropin")
    env = dict(os.environ, PYTHONPATH="")
    subprocess.check_call([sys.executable, "-c", code], env=env, cwd="/")


def test_every_reference_operator_name_is_exported():
    import lic360_operator as lo
    for n in ("MultiProject ImpMap Dtow QUANT GDN SSIM ModuleSaver Logger ContextShift EntropyGmm ContextReshape DropGrad MaskConv2 "
              "SpherePad SphereTrim SphereCutEdge SphereLatScaleNet CodeContex CconvDc CconvDcBatch CconvEc CconvEcBatch TileExtract "
              "TileExtractBatch TileInput TileAdd EntropyGmmTable EntropyBatchGmmTable Dquant EntropyTable Scale Imp2mask").split():
        assert hasattr(lo, n), n


@pytest.mark.skipif(not os.path.isdir("/root/reference/extension"), reason="reference tree not mounted (build container only)")
def test_reference_python_runs_over_the_shim():
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "check_dropin.py")])
