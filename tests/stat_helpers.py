"""Small-sample statistics for the Table I reproduction (tests/test_gpu_table1.py): is an error count observed here
compatible with a count the paper reports, given BOTH sides' sampling error and the paper's rounding?"""
from scipy.stats import binom


def two_sample_pvalue(x, n, y, m):
    """Two-sided exact conditional test that x errors in n trials and y errors in m trials share one (small) rate:
    given x + y = t, x ~ Binomial(t, n / (n + m)) (Poisson approximation of two binomials, rates << 1)."""
    t = x + y
    if t == 0:
        return 1.0
    q = n / float(n + m)
    lo = binom.cdf(x, t, q)
    hi = binom.sf(x - 1, t, q)
    return min(1.0, 2.0 * min(lo, hi))


def consistent_with_reported(x, n, rate_lo, rate_hi, m, alpha=1e-3):
    """x errors in n trials here; the paper reports a rate in [rate_lo, rate_hi] (its rounding) measured on m trials.
    True if some error count of the paper inside that interval passes the two-sample test at level alpha."""
    y_lo, y_hi = int(round(rate_lo * m)), int(round(rate_hi * m))
    best = 0.0
    step = max(1, (y_hi - y_lo) // 200)
    for y in list(range(y_lo, y_hi + 1, step)) + [y_hi]:
        best = max(best, two_sample_pvalue(x, n, y, m))
    return best >= alpha, best


def consistent_with_rate(x, n, rate, alpha=1e-3):
    """x errors in n trials against an exactly known rate (binomial, two-sided)."""
    lo = binom.cdf(x, n, rate)
    hi = binom.sf(x - 1, n, rate)
    p = min(1.0, 2.0 * min(lo, hi))
    return p >= alpha, p
