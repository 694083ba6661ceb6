#!/usr/bin/env python3
"""tools/idct_bound.py -- rigorous error constant of K4's fast IDCT (libkpeg_amd/csrc/idct_colour.hip.h).

K4 decides per sample whether its fast f32 value may be rounded directly or must be re-evaluated
in the reference's own order.  That decision needs a bound on

        | fast value  -  reference's float result |          (both in sample units, i.e. ic)

The reference result `ic` (src/MCU.cpp:184-198) differs from the ideal V = 0.25 * sum fc*cos*cos
(real arithmetic on the reference's own float factors fc) by the roundings of its float
accumulator:   |ic - V| <= U * nnz_ac * A            (one rounding per non-zero AC term, each
at most half an ulp of a partial sum whose magnitude is <= 4A; the DC term, when first, is exact)
with A = sum |in| = sum |0.25 fc| over the block and U = 2^-24 (1 + 2^-20).

This script bounds the other half, |fast - V| <= KAPPA * U * A, by a forward error analysis of the
exact operation sequence of row_idct8() + the DPP column pass:
  * every f32 add/mul/fma result carries a relative rounding error <= 2^-24,
  * every non-trivial f32 constant differs from its real value by a relative 2^-24,
  * the AC inputs differ from 0.25*fc by three relative roundings (scale table, product, fc),
  * a node whose exact value is sum_i L[i] in_i has magnitude <= sum_i |L[i]| |in_i|, so its rounding
    adds the weight vector |L| to the error functional  |err| <= U * sum_i w[i] |in_i|.
Weights are propagated with absolute values of the gains (triangle inequality), so the result
KAPPA = max_i w_out[i]  (over all 64 outputs) is a bound, not an estimate:
        |fast - V| <= U * sum_i w[i]|in_i| <= KAPPA * U * A.  It also checks that the operation sequence computes exactly
sum_uv in[u][v] cos((2x+1)u pi/16) cos((2y+1)v pi/16) in real arithmetic.

    python tools/idct_bound.py           # prints KAPPA; tests/test_bound.py asserts the kernel's
                                         # KPEG_KAPPA is >= this value
"""
import math

import numpy as np

C = [math.cos(k * math.pi / 16) for k in range(8)]


class Node:
    """value = L . inputs (exact real arithmetic); |computed - value| <= U * sum_i w[i] |in_i|."""

    def __init__(self, L, w):
        self.L = L
        self.w = w


def inp(i, eps):
    L = np.zeros(64)
    L[i] = 1.0
    w = np.zeros(64)
    w[i] = eps
    return Node(L, w)


def add(a, b, sign=1.0):
    L = a.L + sign * b.L
    return Node(L, a.w + b.w + np.abs(L))  # + rounding of the result


def mulc(a, c):
    L = a.L * c
    exact_const = abs(c) == 1.0
    # a product by an exactly representable +-1 is exact; otherwise rounding + constant representation
    return Node(L, abs(c) * a.w + (0.0 if exact_const else 2.0) * np.abs(L))


def fmac(a, c, b):
    """fma(a, c, b) = a*c + b with one rounding; constant c has a relative representation error."""
    prod = a.L * c
    L = prod + b.L
    exact_const = abs(c) == 1.0
    return Node(L, abs(c) * a.w + b.w + np.abs(L) + (0.0 if exact_const else 1.0) * np.abs(prod))


def row_idct8(a):
    c1, c2, c3, c4, c5, c6, c7 = C[1], C[2], C[3], C[4], C[5], C[6], C[7]
    t0 = fmac(a[4], c4, a[0])
    t1 = fmac(a[4], -c4, a[0])
    p = fmac(a[6], c6, mulc(a[2], c2))
    q = fmac(a[6], -c2, mulc(a[2], c6))
    e0, e3, e1, e2 = add(t0, p), add(t0, p, -1), add(t1, q), add(t1, q, -1)
    o0 = fmac(a[7], c7, fmac(a[5], c5, fmac(a[3], c3, mulc(a[1], c1))))
    o1 = fmac(a[7], -c5, fmac(a[5], -c1, fmac(a[3], -c7, mulc(a[1], c3))))
    o2 = fmac(a[7], c3, fmac(a[5], c7, fmac(a[3], -c1, mulc(a[1], c5))))
    o3 = fmac(a[7], -c1, fmac(a[5], c3, fmac(a[3], -c5, mulc(a[1], c7))))
    out = [None] * 8
    out[0], out[7] = add(e0, o0), add(e0, o0, -1)
    out[1], out[6] = add(e1, o1), add(e1, o1, -1)
    out[2], out[5] = add(e2, o2), add(e2, o2, -1)
    out[3], out[4] = add(e3, o3), add(e3, o3, -1)
    return out


def cos_k(k):
    return math.cos((k % 32) * math.pi / 16)


def kappa():
    # inputs: column 0 exact (reference's own chain), other columns three relative roundings
    a = [[inp(u * 8 + v, 0.0 if v == 0 else 3.0) for v in range(8)] for u in range(8)]
    g = [row_idct8(a[u]) for u in range(8)]  # g[u][y]
    worst = 0.0
    max_dev = 0.0
    for y in range(8):
        for x in range(4):
            # even lane x: E_x = sum_k g[2k][y] * cos((2x+1)(2k)pi/16); k = 0 constant is exactly 1
            E = mulc(g[0][y], cos_k((2 * x + 1) * 0))
            for k in range(1, 4):
                E = fmac(g[2 * k][y], cos_k((2 * x + 1) * 2 * k), E)
            # odd lane: N = -O_x
            N = mulc(g[1][y], -cos_k((2 * x + 1) * 1))
            for k in range(1, 4):
                N = fmac(g[2 * k + 1][y], -cos_k((2 * x + 1) * (2 * k + 1)), N)
            out_lo = fmac(N, -1.0, E)   # row x     : E + O
            out_hi = fmac(E, 1.0, N)    # row 7 - x : E - O
            for xx, node in ((x, out_lo), (7 - x, out_hi)):
                ideal = np.array([math.cos((2 * xx + 1) * u * math.pi / 16) * math.cos((2 * y + 1) * v * math.pi / 16)
                                  for u in range(8) for v in range(8)])
                max_dev = max(max_dev, float(np.max(np.abs(node.L - ideal))))
                worst = max(worst, float(np.max(node.w)))
    assert max_dev < 1e-12, "operation sequence does not compute the IDCT kernel (dev %g)" % max_dev
    return worst * 1.001  # second-order terms: (1 + 2^-24)^30 - 1 << 0.1 %


if __name__ == "__main__":
    k = kappa()
    print("KAPPA (fast-path error constant, relative to U*A) = %.3f" % k)
    print("kernel constant KPEG_KAPPA must be >= %.3f" % k)
