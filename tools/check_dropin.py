#!/usr/bin/env python3
"""BUILD-CONTAINER-ONLY check of the drop-in boundary -- needs /root/reference, no GPU:

  1. parse the reference's pybind11 bindings (extension/main.cpp:4-178) into {class: (ctor argument types, methods)} and
     check that the `lic360` shim has every class, accepts every constructor arity and has every bound method
     (methods of out-of-scope ops may raise NotImplementedError when CALLED; they must exist);
  2. with ONLY `360-image-compression_amd/dropin` of this repository on sys.path, import the REFERENCE's own
     `lic360_operator` package (it sits on top of the shim) and construct every wrapper class it exports with the argument
     lists the reference's model zoo and demo use;
  3. import the reference's `test/model_zoo.py` and `test/lic360_demo.py` module bodies the same way (cv2 and tkinter are absent
     from this image; bare placeholder modules stand in for `import cv2` and for the stray `from tkinter.messagebox import NO`
     of lic360_operator/Dquant.py:1 -- nothing of either is called);
  4. the same imports against THIS repository's `lic360_operator` (second recipe).

Constructing the codec drivers themselves needs a HIP device (`.to('cuda:0')` in their __init__): the GPU tests cover them
through lic360_codec.py.  Exit code 0 = every check passed.
"""
import importlib
import importlib.util
import inspect
import os
import re
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
DROPIN = os.path.join(ROOT, "360-image-compression_amd", "dropin")
PKG = os.path.join(ROOT, "360-image-compression_amd")


def parse_bindings(text):
    out = {}
    for m in re.finditer(r'py::class_<\s*(\w+)\s*>\s*\(\s*m\s*,\s*"(\w+)"\s*\)(.*?);', text, flags=re.S):
        body = m.group(3)
        ctors = [[a.strip() for a in c.split(",") if a.strip()] for c in re.findall(r"py::init<(.*?)>\(\)", body, flags=re.S)]
        methods = re.findall(r'\.def\(\s*"(\w+)"', body)
        out[m.group(2)] = (ctors, methods)
    return out


def dummy(ctype):
    c = ctype.replace("std::", "").replace(" ", "")
    if c.startswith("vector<float>"):
        return [0.0, 0.5] * 7                            # ProjectsOp: 14 viewport angles
    return {"int": 1, "float": 0.5, "bool": False, "string": "tmp", "double": 0.5}[c]


def check_shim(bindings):
    import lic360
    bad = []
    for name, (ctors, methods) in sorted(bindings.items()):
        cls = getattr(lic360, name, None)
        if cls is None:
            bad.append("%s: class missing" % name)
            continue
        for args in ctors:
            try:
                cls(*[dummy(a) for a in args])
            except NotImplementedError:
                pass                                    # out-of-scope op: the class exists and says so
            except TypeError as e:
                bad.append("%s%s: %s" % (name, tuple(args), e))
        for meth in methods:
            if not callable(getattr(cls, meth, None)):
                bad.append("%s.%s missing" % (name, meth))
    return bad


WRAPPER_ARGS = {   # argument lists as the reference's own callers write them (test/model_zoo.py, test/lic360_demo.py)
    "ImpMap": (200, 1.0, 6, 48, 0.61, 0.61, 5), "Dtow": (2, True), "QUANT": (192, 8), "ContextShift": (False, 4), "EntropyGmm": (3,),
    "ContextReshape": (48,), "SpherePad": (2,), "SphereTrim": (2,), "SphereCutEdge": (2,), "SphereLatScaleNet": (32,),
    "CodeContex": (), "CconvDc": (1, 144, 144, 5, True, True), "CconvDcBatch": (48, 4, 4, 5, 3, True, True),
    "CconvEc": (1, 144, 49, 5, True, False), "CconvEcBatch": (48, 1, 4, 5, 3, False, True), "TileExtract": (48, True),
    "TileExtractBatch": (48, True), "TileInput": (48, -3.5, 1, 3), "TileAdd": (48,), "EntropyGmmTable": (8, 3.5, 3, 65536),
    "EntropyBatchGmmTable": (8, 3.5, 3, 65536), "Dquant": (192, 8), "EntropyTable": (49, 65536), "Scale": (-1, 2 / 47.0), "Imp2mask": (48, 192),
    "DropGrad": (True,), "SSIM": (11,), "MultiProject": (171, 256, 0.5, False, 0),
}
NEEDS_GPU_OR_OUT_OF_SCOPE = {"GDN": "allocates its parameters on cuda at construction (lic360_operator/GDN.py:44-47)",
                              "MaskConv2": "runs MaskConstrainOp on its weight at construction time only on a GPU (lic360_operator/MaskConstrain.py:27-33)",
                             "ModuleSaver": "filesystem utility", "Logger": "filesystem utility"}


def child(which):
    """Runs in a fresh interpreter: sys.path decides whose lic360_operator is on top."""
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))         # placeholder for the import line only
    if importlib.util.find_spec("tkinter") is None:                # the reference's Dquant.py:1 has a stray `from tkinter.messagebox import NO`
        tk, mb = types.ModuleType("tkinter"), types.ModuleType("tkinter.messagebox")
        mb.NO = "no"
        tk.messagebox = mb
        sys.modules["tkinter"], sys.modules["tkinter.messagebox"] = tk, mb
    if which == "reference":
        sys.path[:0] = [DROPIN, REF, os.path.join(REF, "test")]
    else:
        sys.path[:0] = [PKG, os.path.join(REF, "test")]
    import lic360
    import lic360_operator as lo
    origin = os.path.dirname(os.path.abspath(lo.__file__))
    expect = os.path.join(REF, "lic360_operator") if which == "reference" else os.path.join(PKG, "lic360_operator")
    assert origin == expect, (origin, expect)
    bad, need_gpu = [], []
    names = re.findall(r"import\s+(.+)$", open(os.path.join(REF, "lic360_operator", "__init__.py")).read(), flags=re.M)
    names = [n.strip() for line in names for n in line.split(",") if n.strip() != "torch"]
    for n in names:
        cls = getattr(lo, n, None)
        if cls is None:
            bad.append("lic360_operator.%s missing" % n)
            continue
        if n in NEEDS_GPU_OR_OUT_OF_SCOPE:
            continue
        try:
            cls(*WRAPPER_ARGS[n])
        except RuntimeError as e:
            if "No HIP GPUs" not in str(e):                           # wrappers that put parameters on cuda in __init__ (QUANT.py:36-39 ...)
                bad.append("lic360_operator.%s%s: RuntimeError: %s" % (n, WRAPPER_ARGS[n], e))
            else:
                need_gpu.append(n)
        except Exception as e:                                        # noqa: BLE001
            bad.append("lic360_operator.%s%s: %s: %s" % (n, WRAPPER_ARGS[n], type(e).__name__, e))
    for mod in ("model_zoo", "lic360_demo"):
        try:
            m = importlib.import_module(mod)
            if mod == "lic360_demo":
                for drv in ("EntEncoderFast", "EntDecoder", "ImpEntEncoderFast", "ImpEntDecoder", "cast_entropy_parameter", "cast_imp_entropy_parameter"):
                    assert hasattr(m, drv), drv
        except Exception as e:                                        # noqa: BLE001
            bad.append("import %s: %s: %s" % (mod, type(e).__name__, e))
    print("[%s lic360_operator on top] %d names checked, lic360 from %s%s" % (which, len(names), os.path.dirname(lic360.__file__),
          ("; constructed only up to their cuda allocation (no device here): " + ", ".join(need_gpu)) if need_gpu else ""))
    for b in bad:
        print("  FAIL", b)
    sys.exit(1 if bad else 0)


def main():
    if len(sys.argv) > 1:
        child(sys.argv[1])
    if not os.path.isdir(REF):
        print("reference tree not mounted: nothing to check here")
        return 0
    sys.path.insert(0, DROPIN)
    bindings = parse_bindings(open(os.path.join(REF, "extension", "main.cpp")).read())
    assert len(bindings) == 26, len(bindings)
    bad = check_shim(bindings)
    print("[shim vs extension/main.cpp] %d classes, %d constructors, %d methods" %
          (len(bindings), sum(len(c) for c, _ in bindings.values()), sum(len(m) for _, m in bindings.values())))
    for b in bad:
        print("  FAIL", b)
    rc = 1 if bad else 0
    for which in ("reference", "repo"):
        rc |= subprocess.call([sys.executable, os.path.abspath(__file__), which])
    print("drop-in check:", "FAILED" if rc else "ok")
    return rc


if __name__ == "__main__":
    sys.exit(main())
