"""The driver's default pattern through the Fortran boundary: the reference's derived types -> module
radsurf_interface (ISO_C_BINDING shim) -> spartacus_radsurf_hip, called from the harness's OpenMP loop
(driver/spartacus_surface_driver.F90:206-234) with blocks of nblocksize columns, against the compiled
reference run the same way on the host cores.  usage: python tools/shim_blocks_rate.py [ncol] [nblocksize]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("OMP_NUM_THREADS", "16")
os.environ.setdefault("OMP_SCHEDULE", "dynamic")
import bench
from oracle import checker
ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
case = bench.make_case("c3", ncol, 999)
a = (case["config"], case["canopy"], case["sw"], case["lw"])
checker.run_shim(*a, want_profiles=True, nblocksize=nb)                    # plans, kernels loaded
for rep in range(3):
    _, sec = checker.run_shim(*a, want_profiles=True, nblocksize=nb)
    print("GPU through the Fortran shim, %d threads, blocks of %d: %.3e columns/s" % (int(os.environ["OMP_NUM_THREADS"]), nb, ncol / sec), flush=True)
_, sec = checker.run_ref(*a, want_profiles=True, nblocksize=nb, nrepeat=1)
print("reference on the host cores, same loop: %.3e columns/s" % (ncol / sec))
