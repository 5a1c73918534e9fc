"""Minimal PNG decoder for height-field assets (`<hfield file="*.png">`): 8/16-bit greyscale, RGB, RGBA and
their alpha variants, non-interlaced.  zlib inflate is the standard library's; the scanline filters (PNG
specification section 9: None, Sub, Up, Average, Paeth) are undone here.  No image library exists on the target.

`read_png_gray` returns what MuJoCo's compiler feeds its height field with: the image decoded to 8-bit grey
the way lodepng does for LCT_GREY output (grey = red channel of colour images; 16-bit samples keep their high byte).
"""
from __future__ import annotations

import struct
import zlib

import numpy as np

_CHANNELS = {0: 1, 2: 3, 4: 2, 6: 4}     # colour type -> samples per pixel (palette type 3 not needed)


def _unfilter(raw: np.ndarray, height: int, stride: int, bpp: int) -> np.ndarray:
    out = np.zeros((height, stride), dtype=np.uint8)
    prev = np.zeros(stride, dtype=np.int32)
    pos = 0
    for y in range(height):
        ft = int(raw[pos])
        line = raw[pos + 1: pos + 1 + stride].astype(np.int32)
        pos += 1 + stride
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        elif ft == 1:
            cur = line.copy()
            for i in range(bpp, stride):
                cur[i] = (cur[i] + cur[i - bpp]) & 255
        elif ft == 3:
            cur = line.copy()
            for i in range(stride):
                left = cur[i - bpp] if i >= bpp else 0
                cur[i] = (cur[i] + ((left + prev[i]) >> 1)) & 255
        elif ft == 4:
            cur = line.copy()
            for i in range(stride):
                a = int(cur[i - bpp]) if i >= bpp else 0
                b = int(prev[i])
                c = int(prev[i - bpp]) if i >= bpp else 0
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (cur[i] + pred) & 255
        else:
            raise ValueError(f"PNG: unknown filter type {ft}")
        out[y] = cur
        prev = cur
    return out


def read_png(path: str) -> np.ndarray:
    """Decodes to an array [height, width, channels] of uint8 or uint16."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos + 8 <= len(data):
        (length,), ctype = struct.unpack(">I", data[pos:pos + 4]), data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + length]
        if zlib.crc32(ctype + body) & 0xFFFFFFFF != struct.unpack(">I", data[pos + 8 + length:pos + 12 + length])[0]:
            raise ValueError(f"{path}: CRC mismatch in chunk {ctype!r}")
        if ctype == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif ctype == b"IDAT":
            idat.append(body)
        elif ctype == b"IEND":
            break
        pos += 12 + length
    if hdr is None or not idat:
        raise ValueError(f"{path}: missing IHDR/IDAT")
    width, height, depth, ctype, _, _, interlace = hdr
    if interlace != 0 or ctype not in _CHANNELS or depth not in (8, 16):
        raise NotImplementedError(f"{path}: PNG variant (depth {depth}, colour type {ctype}, interlace {interlace}) not supported")
    ch = _CHANNELS[ctype]
    bpp = ch * depth // 8
    stride = width * bpp
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8)
    if raw.size != height * (stride + 1):
        raise ValueError(f"{path}: decompressed size mismatch")
    px = _unfilter(raw, height, stride, bpp)
    if depth == 16:
        px = (px[:, 0::2].astype(np.uint16) << 8) | px[:, 1::2].astype(np.uint16)
    return px.reshape(height, width, ch)


def read_png_gray(path: str) -> np.ndarray:
    """uint8 [height, width]: grey value as lodepng's RGB(A)->grey conversion gives it (the red channel)."""
    img = read_png(path)
    g = img[:, :, 0]
    return (g >> 8).astype(np.uint8) if g.dtype == np.uint16 else g
