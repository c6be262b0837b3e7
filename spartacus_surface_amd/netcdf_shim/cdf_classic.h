/* cdf_classic.h - a small reader/writer for the netCDF "classic" file format
 * (CDF-1, and CDF-2 / 64-bit offset for files beyond 2 GiB), written from the
 * published format specification.  It is the storage engine under the
 * `module netcdf` stand-in of this directory (netcdf.F90), which provides
 * exactly the nf90_* symbols the reference's utilities/easy_netcdf.F90 uses,
 * so that the reference's untouched driver, read_input and radsurf_save can be
 * linked where netcdf-fortran is not installed.
 *
 * Dimension order in this API is the C / file order (slowest first); the
 * Fortran layer reverses it.  start/count are 0-based.  All functions return 0
 * or a (negative) netCDF-style error code, see cdf_strerror().
 */
#ifndef CDF_CLASSIC_H
#define CDF_CLASSIC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { CDF_BYTE = 1, CDF_CHAR = 2, CDF_SHORT = 3, CDF_INT = 4, CDF_FLOAT = 5, CDF_DOUBLE = 6 };
enum { CDF_GLOBAL = -1, CDF_MAX_VAR_DIMS = 1024, CDF_MAX_NAME = 256 };

enum {
  CDF_NOERR = 0,
  CDF_EBADID = -33,
  CDF_EEXIST = -35,
  CDF_EINVAL = -36,
  CDF_EPERM = -37,
  CDF_ENOTINDEFINE = -38,
  CDF_EINDEFINE = -39,
  CDF_EINVALCOORDS = -40,
  CDF_ENAMEINUSE = -42,
  CDF_ENOTATT = -43,
  CDF_EBADTYPE = -45,
  CDF_EBADDIM = -46,
  CDF_EUNLIMIT = -54,
  CDF_ENOTVAR = -49,
  CDF_ENOTNC = -51,
  CDF_EEDGE = -57,
  CDF_ENOMEM = -61,
  CDF_EIO = -68
};

int cdf_open(const char *path, int *ncid);                    /* read only */
int cdf_create(const char *path, int noclobber, int *ncid);   /* enters define mode */
int cdf_enddef(int ncid);
int cdf_close(int ncid);

int cdf_def_dim(int ncid, const char *name, int64_t len, int *dimid);   /* len 0 = record dimension */
int cdf_def_var(int ncid, const char *name, int xtype, int ndims, const int *dimids, int *varid);
/* values are in the host representation of xtype (CDF_CHAR: n bytes of text) */
int cdf_put_att(int ncid, int varid, const char *name, int xtype, int64_t n, const void *values);

int cdf_inq_dimid(int ncid, const char *name, int *dimid);
int cdf_inq_dim(int ncid, int dimid, char *name, int64_t *len);
int cdf_inq_varid(int ncid, const char *name, int *varid);
int cdf_inq_var(int ncid, int varid, char *name, int *xtype, int *ndims, int *dimids, int *natts);
int cdf_inq_att(int ncid, int varid, const char *name, int *xtype, int64_t *len);
/* memtype CDF_CHAR copies the text (no terminator), otherwise converts numbers */
int cdf_get_att(int ncid, int varid, const char *name, int memtype, void *values, int64_t maxn);

/* Hyperslab I/O; start/count NULL = the whole variable.  memtype is the type
 * of the caller's buffer (CDF_BYTE int8, CDF_SHORT int16, CDF_INT int32,
 * CDF_FLOAT, CDF_DOUBLE; CDF_CHAR raw bytes) - converted to/from the
 * variable's external type. */
int cdf_get_vara(int ncid, int varid, const int64_t *start, const int64_t *count, int memtype, void *out);
int cdf_put_vara(int ncid, int varid, const int64_t *start, const int64_t *count, int memtype, const void *in);

const char *cdf_strerror(int status);

#ifdef __cplusplus
}
#endif
#endif
