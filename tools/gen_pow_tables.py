#!/usr/bin/env python3
"""Generates the tables and polynomial coefficients of kanter_core_amd/csrc/pow_positive.inc (mpmath, 100 digits).

    python tools/gen_pow_tables.py > /tmp/pow_tables.txt

log2 a:  a = z * 2^k with z in [OFF, 2 OFF), OFF = 0x3f330000 (0.699..), the 32 equal subintervals of z IN BIT SPACE
         indexed by i; c_i = the subinterval's midpoint (1.0 for the one that contains 1.0, so that values of a
         next to 1 keep full RELATIVE accuracy: log c = 0 and r = z - 1 exactly); invc_i = double(1 / c_i),
         logc_i = double(-log2(invc_i)) -- the log of what the kernel really multiplies by.
         log2 a = k + logc_i + log2(1 + r), r = z * invc_i - 1, |r| < 2^-5.9;
         log2(1 + r) = r * (L1 + L2 r + ... + L7 r^6), Taylor: truncation < 2^-45 relative.
2^y:     y * 32 = kd + 32 r', kd integer, |r'| <= 1/64;  2^y = 2^(kd >> 5) * T[kd & 31] * 2^r',
         T[j] = double(2^(j / 32)), 2^r' = 1 + E1 r' + ... + E5 r'^5, Taylor: truncation < 2^-47.
"""
import struct

from mpmath import mp, mpf, log, floor

mp.dps = 100
OFF = 0x3F330000
BITS = 5
N = 1 << BITS


def f32(bits):
    return mpf(struct.unpack("<f", struct.pack("<I", bits))[0])


def to_double(x):
    return float(x)  # mpmath rounds to nearest


def hexd(x):
    return float(x).hex()


def main():
    step = 1 << (23 - BITS)
    one = 0x3F800000
    rows = []
    max_r = 0
    for i in range(N):
        lo, hi = OFF + i * step, OFF + (i + 1) * step
        if lo <= one < hi:
            c = mpf(1)
        else:
            c = (f32(lo) + f32(hi)) / 2
        invc = to_double(1 / c)
        logc = to_double(-log(mpf(invc), 2))
        if c == 1:
            assert invc == 1.0 and logc == 0.0
        for z in (f32(lo), f32(hi - 1)):
            max_r = max(max_r, abs(z * mpf(invc) - 1))
        rows.append((invc, logc))
    print("// max |r| = %s = 2^%s" % (mp.nstr(max_r, 6), mp.nstr(log(max_r, 2), 5)))
    print("__constant__ double kPowLogTab[%d][2] = {  // { 1 / c_i, log2 c_i }" % N)
    for invc, logc in rows:
        print("    { %s, %s }," % (hexd(invc), hexd(logc)))
    print("};")
    ln2 = log(mpf(2))
    L = [to_double((-1) ** (k + 1) / (k * ln2)) for k in range(1, 8)]
    print("// log2(1 + r) = r * (L[0] + L[1] r + ... + L[6] r^6); listed highest power first for Horner")
    print("__constant__ double kPowLog[7] = { %s };" % ", ".join(hexd(x) for x in reversed(L)))
    T = [to_double(mpf(2) ** (mpf(j) / N)) for j in range(N)]
    print("__constant__ double kPowExpTab[%d] = {  // 2^(j / 32)" % N)
    for j in range(0, N, 4):
        print("    " + " ".join("%s," % hexd(x) for x in T[j:j + 4]))
    print("};")
    fact = 1
    E = []
    for k in range(1, 6):
        fact *= k
        E.append(to_double(ln2 ** k / fact))
    print("// 2^r = 1 + E[0] r + ... + E[4] r^5; listed highest power first")
    print("__constant__ double kPowExp[5] = { %s };" % ", ".join(hexd(x) for x in reversed(E)))
    # truncation bounds
    r = max_r
    print("// log truncation (relative): 2^%s" % mp.nstr(log(r ** 7 / 8, 2), 5))
    x = ln2 / 64
    print("// exp truncation: 2^%s" % mp.nstr(log(x ** 6 / 720, 2), 5))


if __name__ == "__main__":
    main()
