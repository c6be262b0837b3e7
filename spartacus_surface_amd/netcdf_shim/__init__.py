"""netCDF-free I/O layer.

cdf_classic.c   classic-format (CDF-1 / CDF-2) reader + writer, C ABI (cdf_classic.h)
netcdf.F90      `module netcdf` with the nf90_* subset utilities/easy_netcdf.F90 uses
Makefile        -> ../lib/libspartacus_netcdf.so, ../lib/mod/netcdf.mod

This Python file only loads the C engine through ctypes (for tests and for
hosts that want to read the driver's files without netCDF4/scipy).
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "lib", "libspartacus_netcdf.so")

BYTE, CHAR, SHORT, INT, FLOAT, DOUBLE = 1, 2, 3, 4, 5, 6
_NP = {BYTE: np.int8, SHORT: np.int16, INT: np.int32, FLOAT: np.float32, DOUBLE: np.float64}
_I64 = ctypes.c_int64


class CdfError(RuntimeError):
    pass


def build():
    subprocess.run(["make", "-C", HERE], check=True, stdout=subprocess.DEVNULL)


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = ctypes.CDLL(LIB)
        _lib.cdf_strerror.restype = ctypes.c_char_p
    return _lib


def _chk(rc):
    if rc != 0:
        raise CdfError(load().cdf_strerror(rc).decode())


class File:
    """Minimal object view of one open file."""

    def __init__(self, path, mode="r", noclobber=False):
        self.L = load()
        ncid = ctypes.c_int()
        if mode == "r":
            _chk(self.L.cdf_open(path.encode(), ctypes.byref(ncid)))
        else:
            _chk(self.L.cdf_create(path.encode(), int(noclobber), ctypes.byref(ncid)))
        self.ncid = ncid.value

    # -- define mode --
    def def_dim(self, name, length):
        d = ctypes.c_int()
        _chk(self.L.cdf_def_dim(self.ncid, name.encode(), _I64(length or 0), ctypes.byref(d)))
        return d.value

    def def_var(self, name, xtype, dimids):
        v = ctypes.c_int()
        arr = (ctypes.c_int * max(1, len(dimids)))(*dimids)
        _chk(self.L.cdf_def_var(self.ncid, name.encode(), xtype, len(dimids), arr, ctypes.byref(v)))
        return v.value

    def put_att(self, varid, name, value):
        if isinstance(value, (bytes, str)):
            b = value.encode() if isinstance(value, str) else value
            _chk(self.L.cdf_put_att(self.ncid, varid, name.encode(), CHAR, _I64(len(b)), b))
        else:
            a = np.atleast_1d(np.asarray(value))
            xtype = {v: k for k, v in _NP.items()}[a.dtype.type]
            _chk(self.L.cdf_put_att(self.ncid, varid, name.encode(), xtype, _I64(a.size),
                                    a.ctypes.data_as(ctypes.c_void_p)))

    def enddef(self):
        _chk(self.L.cdf_enddef(self.ncid))

    # -- inquiry --
    def varid(self, name):
        v = ctypes.c_int()
        _chk(self.L.cdf_inq_varid(self.ncid, name.encode(), ctypes.byref(v)))
        return v.value

    def inq_var(self, varid):
        name = ctypes.create_string_buffer(257)
        xt, nd, na = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        dimids = (ctypes.c_int * 1024)()
        _chk(self.L.cdf_inq_var(self.ncid, varid, name, ctypes.byref(xt), ctypes.byref(nd), dimids,
                                ctypes.byref(na)))
        shape = []
        for d in range(nd.value):
            n = _I64()
            _chk(self.L.cdf_inq_dim(self.ncid, dimids[d], None, ctypes.byref(n)))
            shape.append(n.value)
        return name.value.decode(), xt.value, tuple(shape)

    def get_att(self, varid, name):
        xt, n = ctypes.c_int(), _I64()
        _chk(self.L.cdf_inq_att(self.ncid, varid, name.encode(), ctypes.byref(xt), ctypes.byref(n)))
        if xt.value == CHAR:
            buf = ctypes.create_string_buffer(n.value + 1)
            _chk(self.L.cdf_get_att(self.ncid, varid, name.encode(), CHAR, buf, n))
            return buf.raw[:n.value]
        out = np.empty(n.value, dtype=_NP[xt.value])
        _chk(self.L.cdf_get_att(self.ncid, varid, name.encode(), xt.value,
                                out.ctypes.data_as(ctypes.c_void_p), n))
        return out

    # -- data --
    def get(self, name, start=None, count=None, dtype=np.float64):
        vid = self.varid(name)
        _, _, shape = self.inq_var(vid)
        nd = len(shape)
        if count is None:
            count = [s - (start[i] if start else 0) for i, s in enumerate(shape)]
        out = np.empty(count, dtype=dtype)
        memtype = {v: k for k, v in _NP.items()}[np.dtype(dtype).type]
        st = (_I64 * max(1, nd))(*(start or [0] * nd))
        ct = (_I64 * max(1, nd))(*count)
        _chk(self.L.cdf_get_vara(self.ncid, vid, st, ct, memtype, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def put(self, name, values, start=None):
        vid = self.varid(name)
        a = np.ascontiguousarray(values)
        _, _, shape = self.inq_var(vid)
        nd = len(shape)
        memtype = {v: k for k, v in _NP.items()}[a.dtype.type]
        lead = nd - a.ndim
        ct = [1] * lead + list(a.shape)
        st = (_I64 * max(1, nd))(*(start or [0] * nd))
        _chk(self.L.cdf_put_vara(self.ncid, vid, st, (_I64 * max(1, nd))(*ct), memtype,
                                 a.ctypes.data_as(ctypes.c_void_p)))

    def close(self):
        if self.ncid is not None:
            rc = self.L.cdf_close(self.ncid)
            self.ncid = None
            _chk(rc)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
