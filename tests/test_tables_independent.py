"""Oracle CDF tables vs an INDEPENDENT restatement (tests/independent_tables.py, written from the reference kernels without
csrc/lic360_exact_math.h) on inputs that trigger the monotonic fix-up walks, the total clamp and the integer rounding.
The GPU tests compare the HIP kernels with the oracle; this closes the loop on the arithmetic both of those share."""
import numpy as np

import oracle as orc
import independent_tables as it

exp1 = lambda x: orc.expf(np.float32(x))[0]
erf1 = lambda x: orc.erff(np.float32(x))[0]


def _gmm_inputs(rng, n):
    w = rng.normal(0, 2, (n, 3)).astype(np.float32)
    d = np.abs(rng.normal(1.0, 0.8, (n, 3))).astype(np.float32)
    m = rng.normal(0, 2.5, (n, 3)).astype(np.float32)
    k = n // 6
    d[:k] = rng.uniform(-1, 1e-4, (k, 3))                 # degenerate sigma: floored to beta or ~1e-4 -> step CDFs, equal entries
    m[k:2 * k] = rng.choice([-40.0, 40.0], (k, 1))        # mixture far outside the alphabet: entries all 0 or all 65536
    d[k:2 * k] = rng.uniform(0.05, 0.5, (k, 3))
    w[2 * k:3 * k] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, k)] * 60   # one-hot mixture weights
    d[2 * k:3 * k, :] = rng.uniform(1e-3, 0.2, (k, 3))
    m[3 * k:4 * k] = np.round(m[3 * k:4 * k]) + 0.5       # means exactly on bin edges with small sigma
    d[3 * k:4 * k] = rng.uniform(1e-6, 1e-2, (k, 3))
    return w, d, m


def test_gmm_table_vs_independent_restatement():
    rng = np.random.default_rng(11)
    n = 1200
    w, d, m = _gmm_inputs(rng, n)
    wo, do = w.copy(), d.copy()
    ref = orc.gmm_table(wo, do, m.copy(), n)               # mutates wo/do in place like the reference
    fixed = 0
    for i in range(n):
        wi, di, T = it.gmm_table_row(w[i], d[i], m[i], exp1, erf1)
        assert np.array_equal(np.array(wi, np.float32), wo[i]), i
        assert np.array_equal(np.array(di, np.float32), do[i]), i
        T = np.array(T, np.float32)
        assert np.array_equal(T, ref[i]), (i, T, ref[i])
        assert T[0] == 0 and T[8] == 65536 and np.all(np.diff(T) >= 1), (i, T)
        # did this row need the fix-up?  (raw entries, recomputed without it)
        raw = T.copy()
        fixed += int(not np.array_equal(np.array(it.fixup([np.float32(v) for v in _raw_gmm(w[i], d[i], m[i])], 8, False), np.float32),
                                        np.array(_raw_gmm(w[i], d[i], m[i]), np.float32)))
    assert fixed > n // 5, "the inputs were meant to trigger the fix-up walk (%d rows did)" % fixed


def _raw_gmm(w, d, m):
    """entries before the fix-up (the independent restatement with the walk disabled)."""
    saved = it.fixup
    try:
        it.fixup = lambda T, n, b: T
        return it.gmm_table_row(w, d, m, exp1, erf1)[2]
    finally:
        it.fixup = saved


def test_entropy_table_vs_independent_restatement():
    rng = np.random.default_rng(12)
    n, nsym = 400, 49
    lg = rng.normal(0, 3, (n, nsym)).astype(np.float32)
    k = n // 4
    lg[:k] = -30.0
    lg[np.arange(k), rng.integers(0, nsym, k)] = 30.0     # one-hot: 48 zero-width bins, clamp at total, long fix-up walks
    lg[k:2 * k] = 0.0                                     # uniform: 49 x 1337.47 -> rounding of every increment
    lg[2 * k:3 * k, :40] -= 25.0                          # mass in the last bins: leading zero-width bins
    ref = orc.entropy_table(lg.reshape(-1).copy(), n, nsym)
    walked = 0
    for i in range(n):
        T = np.array(it.entropy_table_row(lg[i], exp1), np.float32)
        assert np.array_equal(T, ref[i]), (i, T, ref[i])
        assert T[0] == 0 and T[nsym] == 65536
        saved = it.fixup
        it.fixup = lambda T_, n_, b: T_
        raw = np.array(it.entropy_table_row(lg[i], exp1), np.float32)
        it.fixup = saved
        walked += int(not np.array_equal(raw, T))
    assert walked > n // 3, walked


def test_shared_exp_erf_close_to_float64():
    """the transcendental values themselves: within 2 ulp (exp) / 1.2e-7 absolute (erf) of float64"""
    from scipy.special import erf
    x = np.linspace(-20, 20, 20001).astype(np.float32)
    e = orc.expf(x).astype(np.float64)
    assert np.max(np.abs(e - np.exp(x.astype(np.float64))) / np.exp(x.astype(np.float64))) < 2.5e-7
    x = np.linspace(-6, 6, 24001).astype(np.float32)
    assert np.max(np.abs(orc.erff(x).astype(np.float64) - erf(x.astype(np.float64)))) < 1.2e-7
