"""Minimal stdlib (zlib + struct) PNG reader/writer for the test fixtures, so the tests need no
imaging library on the GPU box.  Handles what the reference's fixtures use: non-interlaced,
8-bit gray / gray+alpha / RGB / RGBA and 1-8 bit palette images.  Returns uint8 (h, w, c) with
c as `image::open(..).as_flat_samples_u8()` would see it (palette expands to RGB or RGBA)."""
import struct
import zlib

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    if pa <= pb and pa <= pc:
        return a
    if pb <= pc:
        return b
    return c


def read_png(path):
    with open(path, "rb") as f:
        data = f.read()
    assert data[:8] == _SIG, "not a PNG"
    pos = 8
    idat = []
    plte = trns = None
    ihdr = None
    while pos < len(data):
        (length,) = struct.unpack(">I", data[pos:pos + 4])
        ctype = data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + length]
        pos += 12 + length
        if ctype == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif ctype == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif ctype == b"tRNS":
            trns = np.frombuffer(body, np.uint8)
        elif ctype == b"IDAT":
            idat.append(body)
        elif ctype == b"IEND":
            break
    w, h, depth, color, _comp, _filt, interlace = ihdr
    assert interlace == 0, "interlaced PNG not supported"
    chans = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color]
    assert depth == 8 or (color == 3 and depth in (1, 2, 4)), "unsupported bit depth"
    raw = zlib.decompress(b"".join(idat))
    bpp = max(1, chans * depth // 8)
    stride = (w * chans * depth + 7) // 8
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    p = 0
    for y in range(h):
        ft = raw[p]
        line = np.frombuffer(raw, np.uint8, stride, p + 1).astype(np.int32)
        p += 1 + stride
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
                a = cur[i - bpp] if i >= bpp else 0
                c = prev[i - bpp] if i >= bpp else 0
                cur[i] = (cur[i] + _paeth(int(a), int(prev[i]), int(c))) & 255
        else:
            raise ValueError("bad PNG filter %d" % ft)
        out[y] = cur
        prev = cur
    if color == 3:
        if depth != 8:
            bits = np.unpackbits(out, axis=1)[:, :w * depth].reshape(h, w, depth)
            idx = np.zeros((h, w), np.int32)
            for b in range(depth):
                idx = (idx << 1) | bits[:, :, b]
        else:
            idx = out[:, :w].astype(np.int32)
        rgb = plte[idx]
        if trns is not None:
            alpha = np.full(256, 255, np.uint8)
            alpha[:len(trns)] = trns
            return np.concatenate([rgb, alpha[idx][:, :, None]], axis=2)
        return rgb
    return out.reshape(h, w, chans)


def write_png(path, px):
    """px: uint8 (h, w, 4) -> RGBA8 PNG (filter 0), the format Write nodes / tests save."""
    px = np.ascontiguousarray(px, np.uint8)
    h, w, c = px.shape
    color = {1: 0, 2: 4, 3: 2, 4: 6}[c]

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)

    raw = b"".join(b"\x00" + px[y].tobytes() for y in range(h))
    with open(path, "wb") as f:
        f.write(_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
