"""The register-resident eigen routine (ss_eigen_static.hpp, orders 3..6) must be
bit-identical to the run-time-indexed one (ss_eigen.hpp): same arithmetic,
different addressing.  tests/emul/eigen_check.cpp runs both on 16000 random
Gamma-like matrices (block-diagonal, triangular, diagonal, degenerate and
isolated-eigenvalue cases included) compiled for the host.  The dynamic routine
in turn is checked against the reference's eigen_decomposition_real through the
oracle (tests/test_oracle_vs_reference.py)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_static_eigen_is_bit_identical_to_dynamic(tmp_path):
    exe = str(tmp_path / "eigen_check")
    # -ffp-contract=off: both forms must round every operation the same way
    subprocess.run(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-Wno-unknown-pragmas", "-o", exe,
                    os.path.join(HERE, "emul", "eigen_check.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert r.stdout.count("0 mismatches") == 4, r.stdout


def test_symmetric_route_matches_general_solver(tmp_path):
    """ss_eigen_sym.hpp (Cholesky + symmetric tridiagonal QL on the symmetrised
    Gamma product) against the general solver on 12000 random matrices with the
    hidden symmetry: eigenvalues to 1e-11 of the largest, residuals to 1e-12."""
    exe = str(tmp_path / "eigen_sym_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-Wno-unknown-pragmas", "-o", exe,
                    os.path.join(HERE, "emul", "eigen_sym_check.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert r.stdout.count("0 failures") == 4, r.stdout


def test_row16_primitives_on_lane_packs(tmp_path):
    """The dense-algebra primitives of the row-of-16 kernels (orders 7..16: product, unpivoted LU and
    substitution, symmetric elimination with both Cholesky factors, sums over lanes) on packs of 16
    'lanes' against plain loops (tests/emul/row16_check.cpp; the device runs the same source with
    v_fmac_f64_dpp row_newbcast as the broadcast-multiply-add)."""
    exe = str(tmp_path / "row16_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wno-unknown-pragmas", "-I", os.path.join(HERE, "emul"),
                    '-DSS_ROW16_PACK_HEADER="row16_pack.hpp"', "-o", exe,
                    os.path.join(HERE, "emul", "row16_check.cpp")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    seen = 0
    for line in out.splitlines():
        n, name, val = line.split()
        if name == "cholesky_ok":
            assert val == "1", line
        else:
            assert float(val) < 1e-12, line
        seen += 1
    assert seen == 21
