"""Coefficients of the matmul engine's GELU (csrc/gswm_mmtypes.h: mm_gelu): Phi(g) = 1/2 + g Q(z), z = 2 g^2 / A^2 - 1, g clamped to [-A, A], Q a polynomial of degree m in z.
Fitted for the smallest maximum of |g| |Phi_fit - Phi| over [0, A] (the absolute error of gelu = g Phi) by weighted least squares on Chebyshev-distributed points with Lawson's
re-weighting, then checked in float32 Horner arithmetic against the erf form over [-1.4 A, 1.4 A].  usage: python tools/gelu_poly_fit.py [A] [m]"""
import sys
import numpy as np
from scipy.special import erf

A = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
m = int(sys.argv[2]) if len(sys.argv) > 2 else 12


def phi(g):
    return 0.5 * (1.0 + erf(g / np.sqrt(2.0)))


t = np.cos(np.linspace(0, np.pi, 20001))            # Chebyshev-distributed z
g = A * np.sqrt((t + 1) / 2)
V = np.polynomial.polynomial.polyvander(t, m)        # Q(z) monomials
# residual of gelu: g * (0.5 + g Q - Phi) = g^2 Q - g (Phi - 0.5)
Mx, y = V * (g * g)[:, None], g * (phi(g) - 0.5)
w = np.ones_like(g)
for _ in range(200):
    c = np.linalg.lstsq(Mx * w[:, None], y * w, rcond=None)[0]
    r = np.abs(Mx @ c - y)
    w = w * (0.2 + r / r.max())
    w /= w.max()
print(f"A = {A}, degree {m} in z: max |gelu error| on [0, A] in float64: {r.max():.3e}")
f = np.float32
gg = np.linspace(-1.4 * A, 1.4 * A, 4000001)
g32 = gg.astype(f)
gc = np.clip(g32, f(-A), f(A))
z = gc * gc * f(2 / (A * A)) + f(-1)
q = np.full_like(z, f(c[-1]))
for ck in c[-2::-1]:
    q = q * z + f(ck)
out = (g32 * (gc * q + f(0.5))).astype(np.float64)
e = np.abs(out - gg * phi(gg))
print(f"float32 Horner, g in [-1.4 A, 1.4 A]: max |error| {e.max():.3e} at g = {gg[e.argmax()]:.3f}; max relative error for g >= 0.01: {(e / np.abs(gg * phi(gg)))[gg >= 0.01].max():.3e}")
print("coefficients, highest degree first:")
print(", ".join(f"{float(f(v))!r}f" for v in c[::-1]))
