"""Minimax fit of the transcendental-free erf-GELU that the GEMM epilogues apply to 16-bit outputs (csrc/common.h GeluPoly):

    gelu(x) ~ x * clamp(1/2 + x * Q(x^2), 0, 1),   Q of degree n in x^2, fitted on |x| <= R.

An LP (scipy `linprog`, HiGHS) minimises the maximum ABSOLUTE error of x * Phi(x) over Chebyshev-spaced nodes of [-R, R]; the
result is then evaluated the way the kernel evaluates it — f32 Horner with fused multiply-adds, the [0, 1] clamp, no clamp of x —
over 6 million points of [-30, 30] plus magnitudes up to 1e30, and R is scanned for the smallest global error (the clamp turns
the region past R into a truncation error gelu(-R), so too small an R loses as much as too large a one).  Even degrees only:
the leading coefficient must be positive for x * Q(x^2) to run monotonically to +-infinity past R (exact saturation).

    python tools/fit_gelu.py            # prints the tables and the coefficient lists of degree 6 (bf16) and 8 (f16)"""
import numpy as np
from scipy.optimize import linprog
from scipy.special import erf


def Phi(x):
    return 0.5 * (1 + erf(x / np.sqrt(2)))


def gelu(x):
    return x * Phi(x)


def fit(n, R, npts=4000):
    x = np.cos(np.linspace(0, np.pi, npts)) * R
    x = x[np.abs(x) > 1e-9]
    u = (x / R) ** 2
    A = np.stack([x * x * u ** k for k in range(n + 1)], 1)      # gelu - x/2 = x^2 * Q(x^2)
    b = x * (Phi(x) - 0.5)
    c = np.zeros(n + 2)
    c[-1] = 1
    one = np.ones((len(x), 1))
    res = linprog(c, A_ub=np.block([[A, -one], [-A, -one]]), b_ub=np.concatenate([b, -b]),
                  bounds=[(None, None)] * (n + 1) + [(0, None)], method="highs")
    assert res.status == 0, res.message
    return res.x[:-1] / np.array([R ** (2 * k) for k in range(n + 1)]), res.x[-1]


def kernel_eval(q, x):
    """f32 Horner with FMAs (emulated: the product-sum in f64, rounded once), clamp of Phi, x * Phi — as common.h does."""
    x = x.astype(np.float32)
    q = [np.float32(c) for c in q]
    s = (x * x).astype(np.float32)
    p = (np.float64(q[-1]) * s.astype(np.float64) + np.float64(q[-2])).astype(np.float32)
    for k in range(len(q) - 3, -1, -1):
        p = (p.astype(np.float64) * s.astype(np.float64) + np.float64(q[k])).astype(np.float32)
    phi = np.clip((x.astype(np.float64) * p.astype(np.float64) + 0.5).astype(np.float32), 0, 1).astype(np.float32)
    return (x * phi).astype(np.float32)


def main():
    grid = np.linspace(-30, 30, 6000001).astype(np.float32)
    ref = gelu(grid.astype(np.float64))
    far = np.array([31., 100., 1e3, 1e6, 1e10, 1e18, 3e19, 1e30, -31., -100., -1e3, -1e6, -1e10, -1e18, -3e19, -1e30], np.float32)
    for n, what in ((6, "bf16 outputs"), (8, "f16 outputs")):
        best = None
        for R in np.arange(3.5, 5.01, 0.125):
            q, t = fit(n, R)
            with np.errstate(all="ignore"):
                e = np.abs(kernel_eval(q, grid) - ref).max()
            print(f"degree {n}  R = {R:5.3f}  fit {t:.2e}  global max |error| {e:.3e}")
            if best is None or e < best[0]:
                best = (e, R, q)
        e, R, q = best
        err = np.abs(kernel_eval(q, grid) - ref)
        with np.errstate(all="ignore"):
            yf = kernel_eval(q, far)
        exact = bool(np.all(np.where(far > 0, yf == far, yf == 0)))
        print(f"==> degree {n} ({what}): R = {R}, max |error| {err.max():.3e} at x = {grid[err.argmax()]:.3f}, rms on |x| < 4 "
              f"{np.sqrt((err[np.abs(grid) < 4] ** 2).mean()):.2e}, exact saturation for |x| up to 1e30: {exact}")
        print("    q = {" + ", ".join(f"{c:.9e}f" for c in q) + "}")


if __name__ == "__main__":
    main()
