"""Shared helpers for the parity tests: synthetic planes, ulp distance,
bitwise comparison that treats NaN == NaN."""
import numpy as np

SEED_A, SEED_B = 0x5EED0001, 0x5EED0002


def splitmix_plane(seed, channel, h, w):
    """u in [0, 1): (splitmix64(seed, idx) >> 40) * 2^-24 with idx = (c*H + y)*W + x."""
    idx = (np.uint64(channel) * np.uint64(h) * np.uint64(w)) + np.arange(h * w, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = np.uint64(seed) + (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).reshape(h, w)


def splitmix_rows(seed, channel, h, w, y0, y1):
    """Rows [y0, y1) of splitmix_plane(seed, channel, h, w) without generating the rest."""
    idx = (np.uint64(channel) * np.uint64(h) * np.uint64(w)) + np.uint64(y0) * np.uint64(w) + np.arange((y1 - y0) * w, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = np.uint64(seed) + (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return ((z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).reshape(y1 - y0, w)


def synthetic_rgba(seed, h, w):
    return [splitmix_plane(seed, c, h, w) for c in range(4)]


EDGE_VALUES = np.array([0.0, -0.0, 1.0, 2.0, -1.0, np.inf, -np.inf, np.nan, 5.877e-39, 0.1, 0.5, 254.5 / 255,
                        1e-30, 3.0, -2.5, 1e30], np.float32)


def with_edge_cases(plane, shift=0):
    """Overwrites the head of the plane with IEEE edge cases (all pairs appear when two planes
    use different shifts)."""
    p = plane.copy().reshape(-1)
    n = len(EDGE_VALUES)
    reps = min(len(p) // n, n)
    for r in range(reps):
        p[r * n:(r + 1) * n] = np.roll(EDGE_VALUES, shift * r)
    return p.reshape(plane.shape)


def ordered_bits(a):
    """Maps f32 to integers that are monotone in the float order (for ulp distances)."""
    i = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    return np.where(i < 0, np.int64(-2147483648) - i, i)


def max_ulp(a, b):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    both_nan = np.isnan(a) & np.isnan(b)
    one_nan = np.isnan(a) ^ np.isnan(b)
    if one_nan.any():
        return np.inf
    d = np.abs(ordered_bits(a) - ordered_bits(b))
    d[both_nan] = 0
    return int(d.max()) if d.size else 0


def bit_equal(a, b):
    """Bit-exact, except that any NaN equals any NaN (payloads are not part of the contract)."""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    if a.shape != b.shape:
        return False
    nan = np.isnan(a) & np.isnan(b)
    return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | nan))


def assert_planes(got, want, ulp=0, what=""):
    assert len(got) == len(want), what
    for c, (g, w) in enumerate(zip(got, want)):
        if ulp == 0:
            assert bit_equal(g, w), "%s plane %d: %d mismatches, max ulp %s" % (
                what, c, int((~((g.view(np.uint32) == w.view(np.uint32)) | (np.isnan(g) & np.isnan(w)))).sum()), max_ulp(g, w))
        else:
            assert max_ulp(g, w) <= ulp, "%s plane %d: max ulp %s > %d" % (what, c, max_ulp(g, w), ulp)
