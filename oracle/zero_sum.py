"""CPU restatement of csrc/zero_sum.hip (TEST INFRASTRUCTURE ONLY - imported by tests/, never by the product).

Zero-sum rounding is this build's own preparation step for fp16 weights, not an algorithm of the reference: what the
reference pins is the NETWORK OUTPUT (`net(x)`, /root/reference/sykepic/compute/probability.py:189), and that parity is
checked at network level (tests/test_gpu_calibrated.py against the goldens and the fp32 oracle).  This file restates
the kernel's float32 arithmetic operation by operation (the kernel is compiled with FMA contraction off), including
the lane-strided partial sums and the butterfly that produce the initial weighted sum, so the GPU result can be
compared bit for bit.
"""

import numpy as np

F = np.float32


def _neighbours(v):
    """nearest fp16 value q of every v and the fp16 value on the far side of v (alt); ok: alt exists and is finite."""
    hq = np.clip(v, F(-65504.0), F(65504.0)).astype(np.float16)
    q = hq.astype(F)
    up = q < v
    alt16 = np.where(up, np.nextafter(hq, np.float16(np.inf)), np.nextafter(hq, np.float16(-np.inf)))
    ok = (q != v) & np.isfinite(alt16)
    alt = np.where(ok, alt16.astype(F), q)
    return q, alt, ok


def zero_sum_round(w, mu=None, mu_period=None):
    """w [rows, row_len] float32; mu [mu_period] float32 or None (ones).  Returns float32 [rows, row_len]."""
    w = np.ascontiguousarray(w, dtype=F)
    rows, n = w.shape
    if mu is None:
        m_all = np.ones(n, F)
    else:
        mu = np.asarray(mu, F)
        m_all = mu[np.arange(n) % int(mu_period or mu.size)]
    max_iter = n if n < 64 else max(n // 4, 64)
    out = np.empty_like(w)
    lanes = np.arange(64)
    for r in range(rows):
        v = w[r]
        q, alt, ok = _neighbours(v)
        d, da = (q - v).astype(F), (alt - v).astype(F)
        step = np.where(ok, (m_all * (alt - q).astype(F)).astype(F), F(0))
        cost = ((da * da).astype(F) - (d * d).astype(F)).astype(F)
        res = q.copy()
        # lane l adds m*d of elements l, l + 64, ... in that order; then the butterfly over the 64 lanes
        md = (m_all * d).astype(F)
        part = np.zeros(64, F)
        for k0 in range(0, n, 64):
            seg = md[k0:k0 + 64]
            part[:seg.size] = (part[:seg.size] + seg).astype(F)
        for o in (32, 16, 8, 4, 2, 1):
            part = (part + part[lanes ^ o]).astype(F)
        S = F(part[0])
        for _ in range(max_iter):
            aS = F(abs(S))
            elig = ((step * S).astype(F) < 0) & (np.abs(step) < F(2) * aS)
            if not elig.any():
                break
            gain = (aS - np.abs((S + step).astype(F))).astype(F)
            sc = np.where(elig, (cost / np.maximum(gain, F(1e-37))).astype(F), F(np.inf))
            best = sc.min()
            if not best < F(3.0e38):
                break
            k = int(np.flatnonzero(sc == best)[0])      # smallest k among the equal scores
            S = F(S + step[k])
            res[k] = alt[k]
            step[k] = F(0)
        out[r] = res
    return out


def weighted_sum(w, q, mu=None, mu_period=None):
    """sum_k mu_k (q_k - w_k) per row in float64: what the rounding drives to zero."""
    n = w.shape[1]
    m = np.ones(n) if mu is None else np.asarray(mu, np.float64)[np.arange(n) % int(mu_period or len(mu))]
    return ((q.astype(np.float64) - w.astype(np.float64)) * m).sum(1)
