#!/usr/bin/env python3
"""Coefficients of the packed-FMA GELU in csrc/gemm.hip: h(x) = 0.5*erf(x/sqrt2) ~= xc*P(xc^2)."""
import numpy as np
from scipy.special import erf
zmax, n = 3.0, 9
u = (np.cos(np.pi * (np.arange(4000) + 0.5) / 4000) + 1) / 2 * zmax ** 2
z = np.sqrt(u)
f = np.where(z > 1e-8, erf(z) / np.maximum(z, 1e-30), 2 / np.sqrt(np.pi))
V = np.vander(u, n, increasing=True)
w = z + 1e-3
a, *_ = np.linalg.lstsq(V * w[:, None], f * w, rcond=None)
c = [0.5 / np.sqrt(2) * a[i] / 2 ** i for i in range(n)]
print("X0 =", zmax * np.sqrt(2))
print(", ".join("%.9ef" % v for v in c))
x = np.linspace(-8, 8, 400001).astype(np.float32)
xc = np.clip(x, -np.float32(zmax * np.sqrt(2)), np.float32(zmax * np.sqrt(2)))
acc = np.full_like(x, np.float32(c[-1]))
for v in c[-2::-1]:
    acc = acc * (xc * xc) + np.float32(v)
g = x * (xc * acc) + np.float32(0.5) * x
ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
print("max |gelu err| (fp32 Horner) = %.3e" % np.abs(g - ref).max())
