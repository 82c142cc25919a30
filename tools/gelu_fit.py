"""Minimax fits behind common.cuh's transcendental-free GELU (gelu2_fwd / gelu2_both), checked over EVERY finite bf16 input.

    GELU(x)  = relu(x) - t Q(t),  t = min(|x|, L),  Q ~ Phi(-t)                     (degree 9)
    GELU'(x) = 0.5 + clamp(x, -L, L) R(t),          t R(t) ~ 0.5 - Phi(-t) + t phi(t) (degree 8)

Lawson-weighted Chebyshev least squares on [0, L] (weight t: absolute error of t Q(t)), converted to the power basis
and evaluated in float32 Horner form like the kernel. Prints the coefficients (lowest power first) and the errors.
"""
import numpy as np
from scipy.special import erfc
from numpy.polynomial import chebyshev as Ch, polynomial as P

L = 4.5


def Phi(x): return 0.5 * erfc(-x / np.sqrt(2))
def phi(x): return np.exp(-0.5 * x * x) / np.sqrt(2 * np.pi)


def fit(target_over_t, deg, iters=80, N=6000):
    t = 0.5 * L * (1 - np.cos(np.pi * (np.arange(N) + 0.5) / N))
    y = target_over_t(t)
    w = np.ones(N)
    for _ in range(iters):
        c = Ch.chebfit(2 * t / L - 1, y, deg, w=np.sqrt(w) * t)
        e = np.abs(Ch.chebval(2 * t / L - 1, c) - y) * t
        w = w * (e / e.max() + 1e-3)
        w /= w.sum()
    pc, a, b = Ch.cheb2poly(c), 2 / L, -1.0
    res, pw = np.zeros(1), np.ones(1)
    for ck in pc:
        res = P.polyadd(res, ck * pw)
        pw = P.polymul(pw, np.array([b, a]))
    return res, e.max()


def horner32(coef, t):
    q = np.full_like(t, np.float32(coef[-1]))
    for ck in coef[-2::-1]:
        q = (q * t + np.float32(ck)).astype(np.float32)
    return q


def main():
    bits = np.arange(0, 0x7f80, dtype=np.uint32)
    xp = (bits << 16).view(np.float32)
    x = np.concatenate([xp, -xp]).astype(np.float32)
    x64 = x.astype(np.float64)
    t = np.minimum(np.abs(x), np.float32(L))
    q, eq = fit(lambda t: Phi(-t), 9)
    g = (np.maximum(x, np.float32(0)) - t * horner32(q, t)).astype(np.float64)
    err = np.abs(g - x64 * Phi(x64))
    m = np.abs(x) < 64           # beyond: 1 ulp of x
    print("Q:", ", ".join(f"{np.float32(c):.9e}f" for c in q))
    print(f"  fit {eq:.2e}; fp32 over all bf16 |x| < 64: max |GELU error| {err[m].max():.2e} at {x[m][err[m].argmax()]}")
    r, er = fit(lambda t: (0.5 - (Phi(-t) - t * phi(t))) / t, 8)
    d = (np.float32(0.5) + np.clip(x, -np.float32(L), np.float32(L)) * horner32(r, t)).astype(np.float64)
    errd = np.abs(d - (Phi(x64) + x64 * phi(x64)))
    print("R:", ", ".join(f"{np.float32(c):.9e}f" for c in r))
    print(f"  fit {er:.2e}; fp32 over all bf16: max |GELU' error| {errd.max():.2e} at {x[errd.argmax()]}")


if __name__ == "__main__":
    main()
