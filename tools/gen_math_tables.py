#!/usr/bin/env python3
"""Generate / cross-check the constants used by curvis_amd/csrc/cv_math.h.

* 2/pi bit table for the Payne-Hanek large-argument reduction (1280 bits, with
  one leading zero word so that windows starting at bit positions <= 0 need no
  special case).
* Cody-Waite pi/2 split (33-bit heads + rounded tails), derived from pi itself.
* decimal <-> hex consistency of the polynomial coefficients (published fdlibm
  minimax coefficients; they are re-validated numerically against mpmath in
  tests/test_cv_math.py).
"""
import struct
from mpmath import mp, mpf, pi, floor, ldexp

mp.prec = 4000

def d2h(x):
    return struct.unpack('<Q', struct.pack('<d', float(x)))[0]

def h2d(h):
    return struct.unpack('<d', struct.pack('<Q', h))[0]

def trunc_bits(x, nbits):
    """first nbits significant bits of positive mpf x (truncation)"""
    from mpmath import frexp
    m, e = frexp(x)          # x = m * 2^e, 0.5 <= m < 1
    mi = int(floor(ldexp(m, nbits)))
    return ldexp(mpf(mi), e - nbits)

def main():
    two_over_pi = 2 / pi
    nwords = 20
    bits = int(floor(ldexp(two_over_pi, 64 * (nwords - 1))))
    words = [0] + [(bits >> (64 * (nwords - 2 - i))) & 0xFFFFFFFFFFFFFFFF for i in range(nwords - 1)]
    print("/* 2/pi, 64 zero bits then %d fractional bits, big-endian 64-bit words */" % (64 * (nwords - 1)))
    for i in range(0, nwords, 2):
        print("    0x%016XULL, 0x%016XULL," % (words[i], words[i + 1]))

    p = pi / 2
    names = []
    rem = p
    for i in (1, 2, 3):
        head = trunc_bits(rem, 33)
        tail = rem - head
        print("pio2_%d  = %.20e  0x%016X" % (i, float(head), d2h(head)))
        print("pio2_%dt = %.20e  0x%016X" % (i, float(tail), d2h(tail)))
        rem = tail
    print("invpio2 = %.20e 0x%016X" % (float(2 / pi), d2h(2 / pi)))
    print("pio2_hi = %.20e 0x%016X" % (float(p), d2h(p)))
    lo = p - mpf(float(p))
    print("pio2_lo = %.20e 0x%016X" % (float(lo), d2h(lo)))
    pil = pi - mpf(float(pi))
    print("pi      = %.20e 0x%016X" % (float(pi), d2h(pi)))
    print("pi_lo   = %.20e 0x%016X" % (float(pil), d2h(pil)))
    from mpmath import log, atan
    l2 = log(2)
    hi = trunc_bits(l2, 32)
    print("ln2_hi = %.20e 0x%016X" % (float(hi), d2h(hi)))
    print("ln2_lo = %.20e 0x%016X" % (float(l2 - hi), d2h(l2 - hi)))
    for i, a in enumerate((atan(mpf(0.5)), atan(mpf(1)), atan(mpf(1.5)), pi / 2)):
        h = mpf(float(a))
        print("atanhi[%d] = %.20e 0x%016X   atanlo = %.20e 0x%016X" % (i, float(h), d2h(h), float(a - h), d2h(a - h)))

if __name__ == "__main__":
    main()
