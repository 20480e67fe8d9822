"""Independent restatement of the CDF-table arithmetic, written from the reference kernels and NOT from
csrc/lic360_exact_math.h (which the HIP kernels and the C oracle share):

  gmm_table      extension/entropy_gmm_table_cuda.cu:29-48 (softmax), :51-57 (sigma floor), :138-159 (entries), :85-107 (fix-up)
  entropy_table  extension/entropy_table_cuda.cu:24-50 (softmax CDF + clamp), :53-76 (fix-up with the running bias in the compare)

Scalar Python over numpy float32 scalars; every float32 operation is spelled out, the `0.5` literals of the reference are
doubles (C++ promotion rules), `ps = ps + w*f` is ONE fused multiply-add (nvcc's default contraction) evaluated exactly
with rationals and rounded once.  exp / erf are parameters: the tests pass the oracle's elementwise values, so that what
is compared is everything AROUND the transcendental functions.
"""
from fractions import Fraction

import numpy as np

F = np.float32


def round_f32(q):
    """Fraction -> nearest float32, ties to even (one rounding)."""
    if q == 0:
        return F(0.0)
    c = F(float(q))                                   # may be double-rounded: repair against the exact neighbours
    best = None
    for cand in (np.nextafter(c, F(-np.inf)), c, np.nextafter(c, F(np.inf))):
        if not np.isfinite(cand):
            continue
        err = abs(Fraction(float(cand)) - q)
        even = (int(np.array(cand, F).view(np.uint32)) & 1) == 0
        key = (err, 0 if even else 1)
        if best is None or key < best[0]:
            best = (key, cand)
    return F(best[1])


def fmaf(a, b, c):
    return round_f32(Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c)))


def fixup(T, n, bias_in_compare):
    """entropy_gmm_table_check_kernel (False) / entropy_table_forward_kernel (True); T: list of n+1 float32, in place."""
    bias, mval, midx = F(0), F(0), 0
    for i in range(n):
        nxt = F(T[i + 1] + bias) if bias_in_compare else T[i + 1]
        if nxt <= T[i]:
            bias = F(bias + F(1))
        T[i + 1] = F(T[i + 1] + bias)
        d = F(T[i + 1] - T[i])
        if d > mval:
            mval, midx = d, i
    if bias > 0:
        for i in range(midx, n):
            T[i + 1] = F(T[i + 1] - bias)
    return T


def gmm_table_row(w, d, m, exp, erf, nstep=8, bias=3.5, total=65536.0, beta=1e-6):
    """One symbol: logits w[ng], raw sigma d[ng], mu m[ng] -> (softmaxed w, floored sigma, T[nstep+1])."""
    ng = len(w)
    w = [F(v) for v in w]
    mval = w[0]
    for i in range(1, ng):
        if mval < w[i]:
            mval = w[i]
    tmp, psum = [], F(0)
    for i in range(ng):
        t = F(exp(F(w[i] - mval)))
        tmp.append(t)
        psum = F(psum + t)
    w = [F(t / psum) for t in tmp]
    beta, bias, total = F(beta), F(bias), F(total)
    d = [beta if F(v) < 0 else F(F(v) + beta) for v in d]
    s2 = F(1.0 / np.sqrt(2.0))                        # scalar_t s2 = 1. / sqrt(2.0)
    T = [F(0)] * (nstep + 1)
    T[nstep] = F(int(total))
    for pt in range(1, nstep):
        v = F(float(F(F(pt - 1) - bias)) + 0.5)       # float - float, then + 0.5 (double), stored to a float
        ps = F(0)
        for i in range(ng):
            arg = F(F(s2 * F(v - F(m[i]))) / d[i])
            f = F(0.5 + 0.5 * float(erf(arg)))        # double arithmetic around the float erf
            ps = fmaf(w[i], f, ps)
        T[pt] = F(int(float(F(total * ps)) + 0.5))    # static_cast<int>(total*ps + 0.5): float product, double add, truncation
    return w, d, fixup(T, nstep, False)


def entropy_table_row(logits, exp, total=65536.0):
    w = len(logits)
    lg = [F(v) for v in logits]
    mval = lg[0]
    for i in range(1, w):
        if mval < lg[i]:
            mval = lg[i]
    tmp, psum = [], F(0)
    for i in range(w):
        t = F(exp(F(lg[i] - mval)))
        tmp.append(t)
        psum = F(psum + t)
    total = F(total)
    T = [F(0)] * (w + 1)
    dp = F(total / psum)
    for i in range(w - 1):
        ts = F(T[i] + F(int(float(F(tmp[i] * dp)) + 0.5)))
        T[i + 1] = ts if ts < total else total
    T[w] = total
    return fixup(T, w, True)
