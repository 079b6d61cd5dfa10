"""PNG variants the Image node has to read the way `image::open(..)` does (src/node/image.rs:10-26, src/shared.rs:16-56):
1 / 2 / 4-bit gray (scaled to 8 bits), a tRNS colour key on gray and RGB files (-> alpha channel), Adam7 interlacing for
every colour type, all five row filters.  The files are written here by a 40-line encoder (zlib + struct), so the expected
samples are known exactly; a valid file must never come back as the 1x1 magenta "unreadable" pixel."""
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]


@pytest.fixture(scope="module")
def kc():
    import kanter_core_amd as kc
    kc.init(0)
    return kc


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def _filter_rows(rows, bpp):
    """rows: list of bytes (packed scanlines); cycles through filter types 0..4."""
    out = bytearray()
    prev = bytes(len(rows[0])) if rows else b""
    for y, row in enumerate(rows):
        ft = y % 5
        enc = bytearray(len(row))
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = [0, a, b, (a + b) >> 1, _paeth(a, b, c)][ft]
            enc[i] = (v - pred) & 255
        out.append(ft)
        out += enc
        prev = row
    return bytes(out)


def _pack(samples, depth):
    """samples: (h, w, chans) uint8 at `depth` bits each -> list of packed scanlines"""
    h, w, c = samples.shape
    rows = []
    for y in range(h):
        flat = samples[y].reshape(-1)
        if depth == 8:
            rows.append(bytes(flat))
        else:
            bits = np.zeros(((len(flat) * depth + 7) // 8) * 8, np.uint8)
            for k in range(depth):
                bits[np.arange(len(flat)) * depth + k] = (flat >> (depth - 1 - k)) & 1
            rows.append(bytes(np.packbits(bits)))
    return rows


def write_png(path, samples, color, depth, interlace=False, plte=None, trns=None):
    h, w, chans = samples.shape
    bpp = max(1, chans * depth // 8)
    raw = b""
    if interlace:
        for (x0, y0, dx, dy) in ADAM7:
            sub = samples[y0::dy, x0::dx]
            if sub.size:
                raw += _filter_rows(_pack(sub, depth), bpp)
    else:
        raw = _filter_rows(_pack(samples, depth), bpp)

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)

    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color, 0, 0, 1 if interlace else 0))
    if plte is not None:
        data += chunk(b"PLTE", bytes(np.asarray(plte, np.uint8).reshape(-1)))
    if trns is not None:
        data += chunk(b"tRNS", bytes(trns))
    comp = zlib.compress(raw, 6)
    data += chunk(b"IDAT", comp[:len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:]) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(data)


def planes_of(kc, path):
    img = kc.SlotImage.read_png(path)
    assert img.is_rgba()
    return img.planes()


def expect(u8_channels, h, w):
    """what deconstruct_image makes of `n` interleaved u8 channels: channel c -> plane c, missing R/G/B = 0, A = 1"""
    out = []
    for c in range(4):
        if c < len(u8_channels):
            out.append(u8_channels[c].astype(np.float32) / np.float32(255.0))
        else:
            out.append(np.full((h, w), 1.0 if c == 3 else 0.0, np.float32))
    return out


def check(got, want, what):
    for c in range(4):
        assert got[c].shape == want[c].shape and np.array_equal(got[c], want[c]), "%s plane %d" % (what, c)


@pytest.mark.parametrize("interlace", [False, True])
@pytest.mark.parametrize("depth", [1, 2, 4, 8])
def test_gray_every_depth(kc, tmp_path, depth, interlace):
    rng = np.random.default_rng(depth * 10 + interlace)
    h, w = 13, 21  # not multiples of 8: the late Adam7 passes have ragged edges
    g = rng.integers(0, 1 << depth, (h, w, 1), dtype=np.uint8)
    p = tmp_path / "g.png"
    write_png(p, g, 0, depth, interlace)
    scaled = (g[..., 0].astype(np.uint16) * (255 // ((1 << depth) - 1))).astype(np.uint8)
    check(planes_of(kc, p), expect([scaled], h, w), "gray %d-bit" % depth)
    # with a colour key: gray + alpha, exactly the pixels that equal the key are transparent
    key = int(g[3, 5, 0])
    write_png(p, g, 0, depth, interlace, trns=struct.pack(">H", key))
    alpha = np.where(g[..., 0] == key, 0, 255).astype(np.uint8)
    check(planes_of(kc, p), expect([scaled, alpha], h, w), "gray %d-bit + tRNS" % depth)


@pytest.mark.parametrize("interlace", [False, True])
def test_rgb_rgba_gray_alpha_and_colour_key(kc, tmp_path, interlace):
    rng = np.random.default_rng(7 + interlace)
    h, w = 19, 9
    for color, chans in ((2, 3), (6, 4), (4, 2)):
        px = rng.integers(0, 256, (h, w, chans), dtype=np.uint8)
        p = tmp_path / ("c%d.png" % color)
        write_png(p, px, color, 8, interlace)
        check(planes_of(kc, p), expect([px[..., c] for c in range(chans)], h, w), "colour type %d" % color)
    rgb = rng.integers(0, 4, (h, w, 3), dtype=np.uint8) * 85  # few distinct colours, so the key occurs several times
    key = rgb[2, 2]
    p = tmp_path / "key.png"
    write_png(p, rgb, 2, 8, interlace, trns=struct.pack(">HHH", *[int(v) for v in key]))
    alpha = np.where((rgb == key).all(axis=2), 0, 255).astype(np.uint8)
    assert (alpha == 0).sum() >= 1
    check(planes_of(kc, p), expect([rgb[..., 0], rgb[..., 1], rgb[..., 2], alpha], h, w), "RGB + tRNS")


@pytest.mark.parametrize("interlace", [False, True])
@pytest.mark.parametrize("depth", [1, 4, 8])
def test_palette(kc, tmp_path, depth, interlace):
    rng = np.random.default_rng(depth + 100 * interlace)
    h, w = 10, 17
    n = 1 << depth
    plte = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    idx = rng.integers(0, n, (h, w, 1), dtype=np.uint8)
    p = tmp_path / "p.png"
    write_png(p, idx, 3, depth, interlace, plte=plte)
    rgb = plte[idx[..., 0]]
    check(planes_of(kc, p), expect([rgb[..., 0], rgb[..., 1], rgb[..., 2]], h, w), "palette %d-bit" % depth)
    ta = rng.integers(0, 256, n // 2 + 1, dtype=np.uint8)  # shorter than the palette: the rest is opaque
    write_png(p, idx, 3, depth, interlace, plte=plte, trns=bytes(ta))
    alpha = np.where(idx[..., 0] < len(ta), ta[np.minimum(idx[..., 0], len(ta) - 1)], 255).astype(np.uint8)
    check(planes_of(kc, p), expect([rgb[..., 0], rgb[..., 1], rgb[..., 2], alpha], h, w), "palette %d-bit + tRNS" % depth)


def test_sixteen_bit_and_truncated_files_are_errors_not_wrong_pixels(kc, tmp_path):
    p = tmp_path / "bad.png"
    with open(p, "wb") as f:  # a 16-bit gray header: the reference panics on these (as_flat_samples_u8().unwrap())
        body = struct.pack(">IIBBBBB", 4, 4, 16, 0, 0, 0, 0)
        f.write(b"\x89PNG\r\n\x1a\n" + struct.pack(">I", 13) + b"IHDR" + body + struct.pack(">I", zlib.crc32(b"IHDR" + body)))
    with pytest.raises(kc.TexProError):
        kc.SlotImage.read_png(p)
    g = np.zeros((8, 8, 1), np.uint8)
    write_png(p, g, 0, 8, True)
    data = open(p, "rb").read()
    open(p, "wb").write(data[:len(data) - 30])
    with pytest.raises(kc.TexProError):
        kc.SlotImage.read_png(p)
    # through an Image node an unreadable file is the reference's 1x1 magenta pixel (src/node/image.rs:13-18)
    tp = kc.TextureProcessor.new()
    lg = tp.new_live_graph()
    n = lg.add_node(kc.Node.new(kc.NodeType.Image(str(p))))
    assert lg.await_clean(n).slot_data_size(n, 0) == (1, 1)
