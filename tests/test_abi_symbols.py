"""The drop-in boundary without a GPU: the HIP library loads, exports every
function include/spartacus_surface_hip.h declares (and the Python binding lists
exactly those), and - where no device is visible - refuses to compute instead of
falling back to anything."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    with open(os.path.join(ROOT, "include", "spartacus_surface_hip.h")) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ss_[a-z0-9_]+|spartacus_radsurf_hip)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from spartacus_surface_amd import radsurf
    if not os.path.exists(radsurf.library_path()):
        pytest.skip("HIP library not built (run __graft_entry__.build())")
    lib = radsurf.load_library()
    names = _declared()
    assert len(names) >= 24
    for n in names:
        assert getattr(lib, n) is not None, n
    assert sorted(radsurf.EXPORTS) == names


def test_no_device_means_error_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by the gpu tests")
    from spartacus_surface_amd import radsurf
    if not os.path.exists(radsurf.library_path()):
        pytest.skip("HIP library not built")
    lib = radsurf.load_library()
    n = C.c_int32(-1)
    assert lib.ss_device_count(C.byref(n)) == 1 and n.value == 0      # SS_ERR_NO_DEVICE
    import numpy as np
    with pytest.raises(radsurf.SpartacusHipError, match="NO_DEVICE"):
        radsurf.eigen_decomposition_real(np.eye(3)[None])
