#!/usr/bin/env python3
"""Coefficients of the float64 exp kernel polynomial in atx_common.hpp (atx_exp): exp(r) = 1 + r (1 + r q(r)), |r| <= ln2 / 2, with q the
degree-9 Chebyshev fit of g(r) = (e^r - 1 - r) / r^2 on an interval a hair wider than [-ln2/2, ln2/2] (k = rint(x log2e) is computed with a
rounded log2e).  300-bit arithmetic (mpmath); prints the coefficients as C hexadecimal literals and the relative error of the exact and of
the double-rounded polynomial in units of 2^-53."""
import mpmath as mp

mp.mp.prec = 300
h = mp.log(2) / 2 * mp.mpf("1.0005")


def g(r):
    return (mp.exp(r) - 1 - r) / (r * r) if r != 0 else mp.mpf(1) / 2


coef = mp.chebyfit(g, [-h, h], 10)  # highest degree first


def worst(c):
    w = mp.mpf(0)
    for i in range(4001):
        r = -h + 2 * h * i / 4000
        w = max(w, abs((1 + r + r * r * mp.polyval(c, r)) / mp.exp(r) - 1))
    return w / mp.mpf(2) ** -53


print("relative error of 1 + r + r^2 q(r): exact coefficients %s x 2^-53, double coefficients %s x 2^-53"
      % (mp.nstr(worst(coef), 4), mp.nstr(worst([mp.mpf(float(c)) for c in coef]), 4)))
for c in coef:
    print("    %s   // %.17e" % (float(c).hex(), float(c)))
