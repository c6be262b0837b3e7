"""Legendre-Gauss quadrature tables (host side, computed once per config).

Mirror of legendre_gauss_type / initialize_legendre_gauss / calc_legendre_gauss
(radtool/radtool_legendre_gauss.F90:25-170).  The tables are tiny (<= 16
doubles x 6) and enter the HIP kernels as kernel constants (SURVEY.md
section 8 row a12).
"""
import math

import numpy as np

from . import abi

Pi = 3.14159265358979323846  # radtool/radiation_constants.F90:24


def calc_legendre_gauss(nnode, x1, x2):
    """Nodes and weights on [x1,x2] by Newton iteration on the Legendre
    polynomial (radtool_legendre_gauss.F90:119-170).  The nodes come out
    ASCENDING in mu because the linear map uses (1-y) for both end points
    (:166), exactly as the reference."""
    jn = np.arange(1, nnode + 1, dtype=np.float64)
    # the 0.27 of the first guess is a default-real (float32) literal (:143)
    ynode = (np.cos((2.0 * (jn - 1.0) + 1.0) * Pi / (2 * nnode))
             + (float(np.float32(0.27)) / nnode)
             * np.sin(Pi * (-1.0 + 2.0 * jn) / (nnode + 1)))
    ynode0 = np.full(nnode, 2.0)
    eps = np.finfo(np.float64).eps
    lgvm = np.zeros((nnode, nnode + 1))
    lgvm_deriv = np.zeros(nnode)
    while np.max(np.abs(ynode - ynode0)) > eps:
        lgvm[:, 0] = 1.0
        lgvm[:, 1] = ynode
        for k in range(2, nnode + 1):
            lgvm[:, k] = ((2 * k - 1) * ynode * lgvm[:, k - 1]
                          - (k - 1) * lgvm[:, k - 2]) / k
        lgvm_deriv = ((nnode + 1) * (lgvm[:, nnode - 1] - ynode * lgvm[:, nnode])
                      / (1.0 - ynode * ynode))
        ynode0 = ynode
        ynode = ynode0 - lgvm[:, nnode] / lgvm_deriv
    xnode = 0.5 * (x1 * (1.0 - ynode) + x2 * (1.0 - ynode))
    weight = (((nnode + 1) * (nnode + 1) / float(nnode * nnode)) * (x2 - x1)
              / ((1.0 - ynode * ynode) * lgvm_deriv * lgvm_deriv))
    return xnode, weight


class LegendreGauss:
    """legendre_gauss_type (radtool_legendre_gauss.F90:25-48) initialised for
    ``nstream`` streams per hemisphere (:52-100)."""

    def __init__(self, nstream):
        if not 1 <= nstream <= abi.SS_MAX_STREAMS:
            raise ValueError("nstream must be in 1..%d" % abi.SS_MAX_STREAMS)
        self.nstream = int(nstream)
        self.mu, self.weight = calc_legendre_gauss(self.nstream, 0.0, 1.0)
        self.sin_ang = np.sqrt(1.0 - self.mu * self.mu)
        self.tan_ang = self.sin_ang / self.mu
        hweight = self.weight * self.mu
        vweight = self.weight * self.sin_ang
        self.hweight = hweight / _fsum(hweight)
        self.vweight = vweight / _fsum(vweight)
        self.vadjustment = 1.0
        self.vadjustment2 = (Pi / 4.0) / _fsum(self.weight * self.sin_ang)

    def to_c(self):
        c = abi.LegendreGaussC()
        c.nstream = self.nstream
        for name in ("mu", "sin_ang", "tan_ang", "weight", "hweight", "vweight"):
            arr = getattr(c, name)
            src = getattr(self, name)
            for i in range(self.nstream):
                arr[i] = float(src[i])
        c.vadjustment = self.vadjustment
        c.vadjustment2 = self.vadjustment2
        return c


def _fsum(a):
    # Fortran sum() of a short vector: plain left-to-right accumulation
    s = 0.0
    for x in a:
        s += float(x)
    return s
