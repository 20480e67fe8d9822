"""Single-file container for one compressed ERP.

The reference writes two headerless files per image -- `<name>` (latent stream) and `<name>_imp` (importance-map stream,
test/lic360_demo.py:361-365, extension/coder.h:15-26) -- and keeps image size, model index and the --ssim flag implicit.
This container puts both raw payloads, byte for byte as the coder produced them, behind a 24-byte header:

    offset  size  field
    0       4     magic  b"L360"
    4       1     version (1)
    5       1     flags   bit 0: model trained for SSIM (--ssim)
    6       1     model index (0..7, test/lic360_demo.py:11-17)
    7       1     reserved (0)
    8       2     image height  (u16, little endian)
    10      2     image width
    12      4     length of the latent stream in bytes (u32)
    16      4     length of the importance-map stream in bytes
    20      4     CRC-32 (zlib) of the two payloads, latent first
    24      ...   latent stream, then importance-map stream
"""
import struct
import zlib

MAGIC = b"L360"
VERSION = 1
_HDR = struct.Struct("<4sBBBBHHIII")


class ContainerError(ValueError):
    pass


def pack(latent_stream, imp_stream, height, width, model_idx=0, ssim=False):
    latent_stream, imp_stream = bytes(latent_stream), bytes(imp_stream)
    if not (0 < height < 65536 and 0 < width < 65536 and 0 <= model_idx < 256):
        raise ContainerError("height/width must fit 16 bits and model_idx 8 bits")
    crc = zlib.crc32(imp_stream, zlib.crc32(latent_stream)) & 0xFFFFFFFF
    return _HDR.pack(MAGIC, VERSION, 1 if ssim else 0, model_idx, 0, height, width, len(latent_stream), len(imp_stream), crc) + latent_stream + imp_stream


def unpack(blob):
    """-> dict(latent=bytes, imp=bytes, height, width, model_idx, ssim); raises ContainerError on any inconsistency."""
    blob = bytes(blob)
    if len(blob) < _HDR.size:
        raise ContainerError("truncated header (%d bytes)" % len(blob))
    magic, ver, flags, model_idx, _, h, w, n_lat, n_imp, crc = _HDR.unpack_from(blob)
    if magic != MAGIC:
        raise ContainerError("bad magic %r" % magic)
    if ver != VERSION:
        raise ContainerError("unsupported version %d" % ver)
    if len(blob) != _HDR.size + n_lat + n_imp:
        raise ContainerError("payload length mismatch: header says %d + %d, file holds %d" % (n_lat, n_imp, len(blob) - _HDR.size))
    lat = blob[_HDR.size:_HDR.size + n_lat]
    imp = blob[_HDR.size + n_lat:]
    if (zlib.crc32(imp, zlib.crc32(lat)) & 0xFFFFFFFF) != crc:
        raise ContainerError("CRC mismatch")
    return {"latent": lat, "imp": imp, "height": h, "width": w, "model_idx": model_idx, "ssim": bool(flags & 1)}


def write_file(path, latent_stream, imp_stream, height, width, model_idx=0, ssim=False):
    with open(path, "wb") as f:
        f.write(pack(latent_stream, imp_stream, height, width, model_idx, ssim))


def read_file(path):
    with open(path, "rb") as f:
        return unpack(f.read())


def from_reference_files(code_path, height=512, width=1024, model_idx=0, ssim=False):
    """Wrap the reference's file pair `<code_path>` + `<code_path>_imp` (read as they are) into one container blob."""
    with open(code_path, "rb") as f:
        lat = f.read()
    with open(code_path + "_imp", "rb") as f:
        imp = f.read()
    return pack(lat, imp, height, width, model_idx, ssim)


def to_reference_files(blob, code_path):
    """Write the two raw payloads back as the reference's file pair (byte-identical to what its coder would have written)."""
    d = unpack(blob)
    with open(code_path, "wb") as f:
        f.write(d["latent"])
    with open(code_path + "_imp", "wb") as f:
        f.write(d["imp"])
    return d
