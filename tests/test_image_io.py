"""Image files either side of the path: rt_write_png (the reference writes PNG through stb, main.cu:697),
rt_write_ppm, rt_read_ppm (texture input; the reference decodes a JPEG through stb, main.cu:566-570)."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest


def decode_png(raw):
    """minimal reader for 8-bit RGB, non-interlaced PNG; checks every chunk CRC"""
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(raw):
        (n,) = struct.unpack(">I", raw[pos:pos + 4])
        kind, body = raw[pos + 4:pos + 8], raw[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(kind + body), kind
        chunks.append((kind, body))
        pos += 12 + n
    assert [k for k, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    w, h, depth, colour, comp, filt, lace = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, colour, comp, filt, lace) == (8, 2, 0, 0, 0)
    rows = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, 1 + 3 * w)      # checks the Adler-32
    assert not rows[:, 0].any()                                                              # filter type 0 on every row
    return rows[:, 1:].reshape(h, w, 3)


@pytest.mark.parametrize("w,h", [(1, 1), (7, 5), (160, 137), (300, 91)])     # 160x137 and 300x91 span several stored blocks
def test_png_round_trip(binding, tmp_path, w, h):
    lib = binding.lib()
    px = np.random.default_rng(w * h).integers(0, 256, (h, w, 3), dtype=np.uint8)
    path = str(tmp_path / "a.png").encode()
    assert lib.rt_write_png(path, px.ctypes.data_as(C.c_void_p), w, h) == 0
    assert np.array_equal(decode_png(open(path, "rb").read()), px)
    assert lib.rt_write_png(path, px.ctypes.data_as(C.c_void_p), 0, h) < 0
    assert lib.rt_write_png(str(tmp_path / "no_such_dir" / "a.png").encode(), px.ctypes.data_as(C.c_void_p), w, h) < 0


def read_ppm(lib, path):
    ptr, w, h = C.c_void_p(), C.c_int(), C.c_int()
    rc = lib.rt_read_ppm(str(path).encode(), C.byref(ptr), C.byref(w), C.byref(h))
    if rc != 0:
        assert not ptr.value
        return rc, None
    out = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(h.value, w.value, 3)).copy()
    lib.rt_free_pixels(ptr)
    return rc, out


def test_ppm_round_trip_and_rejects(binding, tmp_path):
    lib = binding.lib()
    px = np.random.default_rng(3).integers(0, 256, (9, 13, 3), dtype=np.uint8)
    path = tmp_path / "t.ppm"
    assert lib.rt_write_ppm(str(path).encode(), px.ctypes.data_as(C.c_void_p), 13, 9) == 0
    rc, got = read_ppm(lib, path)
    assert rc == 0 and np.array_equal(got, px)
    # header with comments and mixed whitespace, pixel bytes that look like whitespace
    body = bytes([10, 32, 9]) * (2 * 2)
    path.write_bytes(b"P6 # made by hand\n2\t2\n# another\n255\n" + body)
    rc, got = read_ppm(lib, path)
    assert rc == 0 and got.tobytes() == body
    for bad in (b"P5\n2 2\n255\n" + body, b"P6\n2 2\n65535\n" + body * 2, b"P6\n2 2\n255\n" + body[:-1], b"P6\n-2 2\n255\n", b"P6\n2\n"):
        path.write_bytes(bad)
        assert read_ppm(lib, path)[0] < 0, bad[:12]
    assert read_ppm(lib, tmp_path / "missing.ppm")[0] < 0
    assert b"rt_read_ppm" in lib.rt_last_error()
