"""The u8 host boundary as a pipeline (csrc/u8pipe.cpp, kc_u8_pipe_*): deconstruct_image (src/shared.rs:16-56) and
SlotImage::to_u8 / to_u8_srgb (src/slot_image.rs:141-207) on pinned buffers with copy streams of their own.  What must hold:
what an upload yields is the oracle's deconstruct_u8, what a download leaves on the host is the oracle's to_u8 -- byte for byte,
for every channel count, through slot reuse and with uploads, evaluations and downloads of different images in flight at once."""
import numpy as np
import pytest

from util import assert_planes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kc():
    import kanter_core_amd as kc
    kc.init(0)
    return kc


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as orc
    return orc


@pytest.mark.parametrize("channels", [1, 2, 3, 4])
@pytest.mark.parametrize("shape", [(17, 33), (64, 256), (5, 1)])
def test_upload_is_deconstruct_image_and_download_is_to_u8(kc, orc, channels, shape):
    h, w = shape
    rng = np.random.default_rng(channels * 1000 + h)
    pipe = kc.U8Pipe(w, h, channels, depth=2)
    px = [rng.integers(0, 256, size=(h, w, channels), dtype=np.uint8) for _ in range(5)]
    for i, p in enumerate(px):  # five images through two slots
        s = i % 2
        pipe.in_buffer(s)[...] = p
        img = pipe.upload(s)
        assert img.is_rgba()
        assert_planes(img.planes(), orc.deconstruct_u8(p), what="upload %d" % i)
        for srgb in (False, True):
            pipe.download(s, img, srgb=srgb)
            got = pipe.wait_download(s).copy()
            want = orc.to_u8(orc.Image(orc.deconstruct_u8(p)), srgb=srgb)
            assert (got == want).all(), "download %d srgb=%s" % (i, srgb)
    pipe.close()


def test_pipelined_loop_equals_the_oracle(kc, orc):
    """upload(k + 1) is issued while image k's kernels and image k - 1's download are in flight; slots are reused every 3 images."""
    h, w, depth, n_img = 96, 200, 3, 10
    rng = np.random.default_rng(7)
    imgs = [rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8) for _ in range(n_img)]
    b = [rng.random((h, w), dtype=np.float32) * 2 - 0.5 for _ in range(4)]  # values outside [0, 1] too: to_u8 clamps
    ib = kc.SlotImage.from_planes(b)
    pipe = kc.U8Pipe(w, h, 4, depth)

    def graph(x):
        return kc.mix_process(kc.mix_process(kc.mix_process(x, ib, kc.MixType.Add), ib, kc.MixType.Multiply), x, kc.MixType.Subtract)

    def want(k):
        a = orc.deconstruct_u8(imgs[k])
        r = [orc.mix_plane("Subtract", orc.mix_plane("Multiply", orc.mix_plane("Add", a[c], b[c]), b[c]), a[c]) for c in range(3)]
        return orc.to_u8(orc.Image(r + [np.ones((h, w), np.float32)]))

    results = {}
    pipe.in_buffer(0)[...] = imgs[0]
    cur = pipe.upload(0)
    for k in range(n_img):
        s = k % depth
        res = graph(cur)
        if k >= depth:
            results[k - depth] = pipe.wait_download(s).copy()
        pipe.download(s, res)
        if k + 1 < n_img:
            pipe.in_buffer((k + 1) % depth)[...] = imgs[k + 1]
            cur = pipe.upload((k + 1) % depth)
    for k in range(max(0, n_img - depth), n_img):
        results[k] = pipe.wait_download(k % depth).copy()
    for k in range(n_img):
        assert (results[k] == want(k)).all(), "image %d" % k
    pipe.close()


def test_pipe_argument_errors(kc):
    with pytest.raises(kc.TexProError):
        kc.U8Pipe(0, 4)
    with pytest.raises(kc.TexProError):
        kc.U8Pipe(4, 4, channels=5)
    pipe = kc.U8Pipe(8, 4, 4, depth=1)
    with pytest.raises(kc.TexProError):
        pipe.upload(1)
    other = kc.SlotImage.from_value((4, 4), 0.5, True)
    with pytest.raises(kc.TexProError):
        pipe.download(0, other)  # another size than the pipe's
    assert (pipe.wait_download(0) is not None)  # nothing pending: returns at once
    pipe.close()
