"""Single-file bitstream container: raw payloads stay byte-identical; corruption is detected."""
import numpy as np
import pytest

import lic360_container as lc


def test_roundtrip_and_reference_file_pair(tmp_path):
    rng = np.random.default_rng(1)
    lat, imp = rng.integers(0, 256, 65751, dtype=np.uint8).tobytes(), rng.integers(0, 256, 913, dtype=np.uint8).tobytes()
    blob = lc.pack(lat, imp, 512, 1024, model_idx=3, ssim=True)
    assert len(blob) == 24 + len(lat) + len(imp) and blob[:4] == b"L360"
    d = lc.unpack(blob)
    assert d == {"latent": lat, "imp": imp, "height": 512, "width": 1024, "model_idx": 3, "ssim": True}
    p = str(tmp_path / "img.l360")
    lc.write_file(p, lat, imp, 1024, 2048)
    assert lc.read_file(p)["height"] == 1024 and lc.read_file(p)["ssim"] is False
    # the reference's two headerless files <-> container, payloads untouched
    code = str(tmp_path / "code")
    lc.to_reference_files(blob, code)
    assert open(code, "rb").read() == lat and open(code + "_imp", "rb").read() == imp
    assert lc.from_reference_files(code, 512, 1024, 3, True) == blob


def test_empty_streams_and_errors():
    blob = lc.pack(b"\x80", b"", 512, 1024)
    assert lc.unpack(blob)["latent"] == b"\x80" and lc.unpack(blob)["imp"] == b""
    with pytest.raises(lc.ContainerError):
        lc.unpack(blob[:10])
    with pytest.raises(lc.ContainerError):
        lc.unpack(b"XXXX" + blob[4:])
    with pytest.raises(lc.ContainerError):
        lc.unpack(blob + b"\x00")
    bad = bytearray(lc.pack(b"abcdef", b"gh", 512, 1024))
    bad[-1] ^= 1
    with pytest.raises(lc.ContainerError):
        lc.unpack(bytes(bad))
    with pytest.raises(lc.ContainerError):
        lc.pack(b"", b"", 70000, 10)
