"""spartacus_surface_amd - MI355X-native SPARTACUS-Surface column solver.

One hot path of ecmwf/spartacus-surface (the per-column multi-stream,
multi-region SW/LW canopy flux solver behind ``radsurf``) as hand-written HIP
kernels for gfx950, reached through the C-ABI of
``include/spartacus_surface_hip.h``.  This package is the Python host-side
mirror of the reference's call surface (same type and member names); the
Fortran host-side shim lives in ``spartacus_surface_amd/fortran``.
"""
from .types import (BoundaryCondsOut, CanopyFlux, CanopyProperties, Config,  # noqa: F401
                    LwSpectralProperties, SwSpectralProperties,
                    ITileFlat, ITileForest, ITileUrban, ITileVegetatedUrban,
                    ITileSimpleUrban, ITileInfiniteStreet)
from .legendre_gauss import LegendreGauss  # noqa: F401

__version__ = "0.1.0"
