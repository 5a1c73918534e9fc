"""rsr_mjx_amd/png.py (height-field asset loader) against PNG files written here with every scanline filter."""
import struct
import zlib

import numpy as np
import pytest

from rsr_mjx_amd.png import read_png, read_png_gray


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def _write_png(path, img, depth, ctype, filters):
    h, w, ch = img.shape
    raw = img.astype(">u2").tobytes() if depth == 16 else img.astype(np.uint8).tobytes()
    bpp = ch * depth // 8
    stride = w * bpp
    lines = bytearray()
    prev = bytes(stride)
    for y in range(h):
        cur = raw[y * stride:(y + 1) * stride]
        ft = filters[y % len(filters)]
        out = bytearray(stride)
        for i in range(stride):
            a = cur[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = (0, a, b, (a + b) >> 1, _paeth(a, b, c))[ft]
            out[i] = (cur[i] - pred) & 255
        lines += bytes([ft]) + out
        prev = cur
    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)
    z = zlib.compress(bytes(lines), 6)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) +
                chunk(b"IDAT", z[: len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b""))


@pytest.mark.parametrize("depth,ctype,ch", [(8, 0, 1), (8, 2, 3), (8, 6, 4), (8, 4, 2), (16, 0, 1), (16, 2, 3)])
def test_png_roundtrip_all_filters(tmp_path, depth, ctype, ch):
    rng = np.random.default_rng(depth + ctype)
    img = rng.integers(0, 2 ** depth, size=(13, 11, ch))
    img[3:6] = img[2:3]                         # runs that make Up / Paeth predictions exact
    p = str(tmp_path / "t.png")
    _write_png(p, img, depth, ctype, filters=[0, 1, 2, 3, 4])
    out = read_png(p)
    assert out.shape == img.shape and out.dtype == (np.uint16 if depth == 16 else np.uint8)
    np.testing.assert_array_equal(out, img)
    g = read_png_gray(p)
    np.testing.assert_array_equal(g, (img[:, :, 0] >> 8) if depth == 16 else img[:, :, 0])


def test_png_rejects_damage(tmp_path):
    p = str(tmp_path / "t.png")
    _write_png(p, np.zeros((4, 4, 1), dtype=np.int64), 8, 0, [0])
    data = bytearray(open(p, "rb").read())
    data[40] ^= 0xFF
    open(p, "wb").write(bytes(data))
    with pytest.raises(ValueError):
        read_png(p)
    open(p, "wb").write(b"not a png at all")
    with pytest.raises(ValueError):
        read_png(p)
