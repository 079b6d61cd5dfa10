"""Host time of kc_image_from_u8 at the reference's own sizes: python profiles/from_u8_latency.py  (KC_UPLOAD_RING=0 for the old path)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import kanter_core_amd as kc
kc.init(0)
for S in (64, 256, 1024):
    px = (np.arange(S * S * 4, dtype=np.uint32) * 2654435761 >> 24).astype(np.uint8).reshape(S, S, 4)
    imgs = [kc.SlotImage.from_u8(px) for _ in range(20)]
    kc.sync()
    n = 1000 if S <= 256 else 200
    t0 = time.perf_counter()
    for _ in range(n):
        img = kc.SlotImage.from_u8(px)
    t1 = time.perf_counter()
    kc.sync()
    t2 = time.perf_counter()
    back = img.to_u8()
    assert (np.asarray(back).reshape(S, S, 4) == px).all()
    print("%4d x %4d RGBA8: %.1f us per call (%.1f us incl. the final sync), round trip exact" % (S, S, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
kc.shutdown()
