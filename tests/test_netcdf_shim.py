"""netCDF-free I/O: the classic-format engine
(spartacus_surface_amd/netcdf_shim/cdf_classic.c) and the `module netcdf`
stand-in (netcdf.F90) that lets the reference's untouched easy_netcdf /
read_input / radsurf_save / driver link without netcdf-fortran.

The independent implementation of the format used as the checker here is
scipy.io.netcdf_file."""
import glob
import os
import shutil
import subprocess

import numpy as np
import pytest
from scipy.io import netcdf_file

from spartacus_surface_amd import netcdf_shim as ns

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLANG = "/opt/rocm/lib/llvm/bin/flang"
REF_TEST = "/root/reference/test"


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not shutil.which("gcc") or not os.path.exists(FLANG):
        if not os.path.exists(ns.LIB):
            pytest.skip("no toolchain and no prebuilt libspartacus_netcdf.so")
    else:
        ns.build()
    ns.load()


def _scipy_write(path, version):
    rng = np.random.default_rng(3)
    nc = netcdf_file(path, "w", version=version)
    nc.title = "scipy written"
    nc.createDimension("time", None)
    nc.createDimension("column", 5)
    nc.createDimension("layer", 3)
    data = {}
    for name, typ, dims in (("d", "d", ("column", "layer")), ("f", "f", ("layer",)),
                            ("i", "i", ("column",)), ("h", "h", ("column", "layer")),
                            ("b", "b", ("layer",)), ("rec", "d", ("time", "layer")),
                            ("rec1", "h", ("time",))):
        v = nc.createVariable(name, typ, dims)
        shape = tuple({"time": 4, "column": 5, "layer": 3}[d] for d in dims)
        a = (rng.uniform(-100, 100, size=shape)).astype(np.dtype(typ))
        v[:] = a
        v.units = "W m-2"
        data[name] = a
    nc.variables["d"].scale = 2.5
    nc.close()
    return data


@pytest.mark.parametrize("version", [1, 2])
def test_engine_reads_what_scipy_writes(tmp_path, version):
    path = str(tmp_path / "in.nc")
    data = _scipy_write(path, version)
    with ns.File(path) as f:
        for name, a in data.items():
            got = f.get(name)
            assert got.shape == a.shape
            assert np.array_equal(got, a.astype(np.float64)), name
            assert f.get_att(f.varid(name), "units") == b"W m-2"
        assert f.get_att(-1, "title") == b"scipy written"
        assert f.get_att(f.varid("d"), "scale")[0] == 2.5
        # hyperslabs, incl. across records, with conversion to the caller's type
        assert np.array_equal(f.get("d", start=[1, 1], count=[3, 2]), data["d"][1:4, 1:3])
        assert np.array_equal(f.get("rec", start=[1, 0], count=[2, 3]), data["rec"][1:3])
        assert np.array_equal(f.get("rec1", start=[2], count=[2], dtype=np.int32), data["rec1"][2:4])
        assert np.array_equal(f.get("h", dtype=np.int32), data["h"].astype(np.int32))
        with pytest.raises(ns.CdfError):
            f.get("d", start=[0, 0], count=[6, 3])
        with pytest.raises(ns.CdfError):
            f.varid("absent")


def test_engine_writes_what_scipy_reads(tmp_path):
    path = str(tmp_path / "out.nc")
    f = ns.File(path, "w")
    t, c, l = f.def_dim("time", None), f.def_dim("column", 5), f.def_dim("layer", 3)
    va = f.def_var("a", ns.FLOAT, [c, l])
    f.def_var("s", ns.SHORT, [c])
    f.def_var("r", ns.DOUBLE, [t, l])
    f.def_var("sc", ns.DOUBLE, [])
    f.def_var("r1", ns.INT, [t])
    f.def_var("odd", ns.BYTE, [l])            # 3 bytes: padded to 4 in the file
    f.def_var("after", ns.SHORT, [l])
    f.put_att(va, "units", "W m-2")
    f.put_att(va, "_FillValue", np.float32(-9999.0))
    f.put_att(-1, "title", "engine written")
    f.enddef()
    a = np.arange(15, dtype=np.float64).reshape(5, 3) * 1.1
    f.put("a", a[1:], start=[1, 0])
    f.put("s", np.array([1, 2, 3, 4, 5], dtype=np.int32))
    r = np.arange(12, dtype=np.float64).reshape(4, 3)
    f.put("r", r[2:], start=[2, 0])            # records 2,3 first: 0,1 are filled
    f.put("r", r[:1], start=[0, 0])
    f.put("sc", np.array(3.25))
    f.put("r1", np.array([7, 8], dtype=np.int32), start=[2])
    f.put("odd", np.array([1, -2, 3], dtype=np.int8))
    f.put("after", np.array([10, 20, 30], dtype=np.int16))
    f.close()
    nc = netcdf_file(path, "r", mmap=False)
    assert nc.dimensions == {"time": None, "column": 5, "layer": 3}
    A = nc.variables["a"][:]
    assert A.dtype == np.dtype(">f4")
    assert np.all(A[0] == np.float32(-9999.0)) and np.array_equal(A[1:], a[1:].astype(np.float32))
    assert nc.variables["a"].units == b"W m-2" and nc.title == b"engine written"
    assert np.array_equal(nc.variables["s"][:], [1, 2, 3, 4, 5])
    R = nc.variables["r"][:]
    assert R.shape == (4, 3) and np.array_equal(R[2:], r[2:]) and np.array_equal(R[0], r[0])
    assert np.all(R[1] == 9.9692099683868690e+36)        # default fill of an unwritten record
    assert nc.variables["sc"][()] == 3.25
    assert np.array_equal(nc.variables["r1"][2:], [7, 8]) and nc.variables["r1"][0] == -2147483647
    assert np.array_equal(nc.variables["odd"][:], [1, -2, 3])
    assert np.array_equal(nc.variables["after"][:], [10, 20, 30])


def test_define_mode_rules(tmp_path):
    f = ns.File(str(tmp_path / "x.nc"), "w")
    c = f.def_dim("column", 2)
    f.def_var("v", ns.DOUBLE, [c])
    with pytest.raises(ns.CdfError):
        f.def_dim("column", 3)                  # name in use
    with pytest.raises(ns.CdfError):
        f.def_var("v", ns.DOUBLE, [c])          # name in use
    with pytest.raises(ns.CdfError):
        f.put("v", np.zeros(2))                 # still in define mode
    f.enddef()
    with pytest.raises(ns.CdfError):
        f.def_dim("late", 3)                    # data mode
    f.close()
    with pytest.raises(ns.CdfError):
        ns.File(str(tmp_path / "x.nc"), "w", noclobber=True)
    with open(str(tmp_path / "junk.nc"), "wb") as g:
        g.write(b"\x89HDF\r\n\x1a\n" + b"\0" * 64)
    with pytest.raises(ns.CdfError, match="classic"):
        ns.File(str(tmp_path / "junk.nc"))


@pytest.mark.skipif(not os.path.isdir(REF_TEST), reason="reference test inputs not present")
def test_reads_every_input_file_of_the_reference():
    files = glob.glob(os.path.join(REF_TEST, "**", "*.nc"), recursive=True)
    assert len(files) >= 15
    nvars = 0
    for path in files:
        nc = netcdf_file(path, "r", mmap=False)
        with ns.File(path) as f:
            for name, var in nc.variables.items():
                ref = np.array(var[:])
                if ref.dtype.kind == "S":
                    continue
                got = f.get(name)
                assert np.array_equal(got, ref.astype(np.float64)), (path, name)
                nvars += 1
    assert nvars > 300


@pytest.mark.skipif(not os.path.exists(FLANG), reason="flang not available")
def test_fortran_module_roundtrip(tmp_path):
    """`use netcdf` through every generic easy_netcdf.F90 needs; the program
    checks what it reads back, scipy checks the file."""
    exe = str(tmp_path / "nf90_roundtrip")
    libdir = os.path.join(ROOT, "spartacus_surface_amd", "lib")
    subprocess.run([FLANG, "-cpp", "-O1", "-I" + os.path.join(libdir, "mod"),
                    os.path.join(ROOT, "tests", "netcdf_shim", "nf90_roundtrip.F90"), "-o", exe,
                    "-L" + libdir, "-lspartacus_netcdf", "-Wl,-rpath," + libdir],
                   check=True, cwd=str(tmp_path))
    path = str(tmp_path / "rt.nc")
    r = subprocess.run([exe, path], capture_output=True, text=True)
    assert r.returncode == 0 and "nf90_roundtrip: OK" in r.stdout, r.stdout + r.stderr
    nc = netcdf_file(path, "r", mmap=False)
    a3 = nc.variables["a3"]
    assert a3.dimensions == ("column", "layer", "band") and a3.shape == (4, 3, 2)
    k, j, i = np.meshgrid(np.arange(1, 5), np.arange(1, 4), np.arange(1, 3), indexing="ij")
    assert np.array_equal(a3[:], 100 * k + 10 * j + i + 0.5)
    assert a3.long_name == b"Three-dimensional test array" and a3.units == b"W m-2" and a3.scale == 2.5
    assert nc.variables["a2"][:].dtype == np.dtype(">f4")
    assert np.array_equal(nc.variables["surface_type"][:], [-1, 0, 1, 2])
    assert np.array_equal(nc.variables["ivec"][:], [1000, 2000, 77, 4000])
    assert nc.variables["s0"][()] == 42.125
    assert np.array_equal(nc.variables["rec"][:], [[1, 2, 3], [4, 5, 6]])
    filled = nc.variables["filled"][:]
    assert np.all(filled[0] == -9999.0) and np.all(filled[2:] == -9999.0) and np.all(filled[1] != -9999.0)
    assert nc.title == b"round trip"
