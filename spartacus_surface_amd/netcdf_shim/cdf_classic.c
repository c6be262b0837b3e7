/* cdf_classic.c - netCDF classic-format (CDF-1 / CDF-2) reader and writer.
 * See cdf_classic.h.  Layout written from the format specification:
 *
 *   header   = magic("CDF" version) numrecs dim_list gatt_list var_list
 *   dim_list = ABSENT | 0x0A nelems { name length }
 *   att_list = ABSENT | 0x0C nelems { name type nelems values(pad 4) }
 *   var_list = ABSENT | 0x0B nelems { name ndims dimids att_list type vsize begin }
 *   ABSENT   = two zero words;  all integers big-endian 32 bit, except `begin`
 *              which is 64 bit when version = 2.
 *   data     = fixed-size variables, each padded to 4 bytes, in definition
 *              order, followed by the records (one slab of every record
 *              variable per record).
 */
#define _FILE_OFFSET_BITS 64
#define _POSIX_C_SOURCE 200809L   /* fseeko / ftello / off_t */
#include "cdf_classic.h"

#include <sys/types.h>

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_FILES 64

typedef struct {
  char name[CDF_MAX_NAME + 1];
  int64_t len;          /* 0 = record dimension */
} cdf_dim;

typedef struct {
  char name[CDF_MAX_NAME + 1];
  int xtype;
  int64_t n;
  void *data;           /* host representation */
} cdf_att;

typedef struct {
  char name[CDF_MAX_NAME + 1];
  int xtype, ndims;
  int *dimids;
  int natts;
  cdf_att *atts;
  int64_t vsize, begin;
  int is_record;
} cdf_var;

typedef struct {
  FILE *fp;
  int writable, define_mode, version;
  int64_t numrecs;
  int ndims, nvars, ngatts;
  cdf_dim *dims;
  cdf_var *vars;
  cdf_att *gatts;
  int64_t recsize;      /* bytes of one record */
  int64_t rec_begin;    /* offset of the first record */
  int numrecs_dirty;
} cdf_file;

static cdf_file *g_files[MAX_FILES];

static int type_size(int t) {
  switch (t) {
    case CDF_BYTE: case CDF_CHAR: return 1;
    case CDF_SHORT: return 2;
    case CDF_INT: case CDF_FLOAT: return 4;
    case CDF_DOUBLE: return 8;
    default: return 0;
  }
}

static int64_t pad4(int64_t n) { return (n + 3) & ~(int64_t)3; }

static cdf_file *lookup(int ncid) {
  if (ncid < 0 || ncid >= MAX_FILES) return NULL;
  return g_files[ncid];
}

/* ---- big-endian primitives ---- */
static void be_store(unsigned char *p, uint64_t v, int nbytes) {
  for (int i = nbytes - 1; i >= 0; --i) { p[i] = (unsigned char)(v & 0xff); v >>= 8; }
}
static uint64_t be_load(const unsigned char *p, int nbytes) {
  uint64_t v = 0;
  for (int i = 0; i < nbytes; ++i) v = (v << 8) | p[i];
  return v;
}

/* external (big-endian, xtype) -> double, and back */
static double ext_to_double(const unsigned char *p, int xtype) {
  switch (xtype) {
    case CDF_BYTE: return (double)(int8_t)p[0];
    case CDF_CHAR: return (double)p[0];
    case CDF_SHORT: return (double)(int16_t)be_load(p, 2);
    case CDF_INT: return (double)(int32_t)be_load(p, 4);
    case CDF_FLOAT: { uint32_t u = (uint32_t)be_load(p, 4); float f; memcpy(&f, &u, 4); return (double)f; }
    default: { uint64_t u = be_load(p, 8); double d; memcpy(&d, &u, 8); return d; }
  }
}
static void double_to_ext(double v, unsigned char *p, int xtype) {
  switch (xtype) {
    case CDF_BYTE: p[0] = (unsigned char)(int8_t)v; break;
    case CDF_CHAR: p[0] = (unsigned char)v; break;
    case CDF_SHORT: be_store(p, (uint64_t)(uint16_t)(int16_t)v, 2); break;
    case CDF_INT: be_store(p, (uint64_t)(uint32_t)(int32_t)v, 4); break;
    case CDF_FLOAT: { float f = (float)v; uint32_t u; memcpy(&u, &f, 4); be_store(p, u, 4); break; }
    default: { uint64_t u; memcpy(&u, &v, 8); be_store(p, u, 8); break; }
  }
}
/* host memory (memtype) <-> double */
static double mem_to_double(const void *p, int64_t i, int memtype) {
  switch (memtype) {
    case CDF_BYTE: return (double)((const int8_t *)p)[i];
    case CDF_CHAR: return (double)((const unsigned char *)p)[i];
    case CDF_SHORT: return (double)((const int16_t *)p)[i];
    case CDF_INT: return (double)((const int32_t *)p)[i];
    case CDF_FLOAT: return (double)((const float *)p)[i];
    default: return ((const double *)p)[i];
  }
}
static void double_to_mem(double v, void *p, int64_t i, int memtype) {
  switch (memtype) {
    case CDF_BYTE: ((int8_t *)p)[i] = (int8_t)v; break;
    case CDF_CHAR: ((unsigned char *)p)[i] = (unsigned char)v; break;
    case CDF_SHORT: ((int16_t *)p)[i] = (int16_t)v; break;
    case CDF_INT: ((int32_t *)p)[i] = (int32_t)v; break;
    case CDF_FLOAT: ((float *)p)[i] = (float)v; break;
    default: ((double *)p)[i] = v; break;
  }
}

/* ---- growing byte buffer used to build / parse the header ---- */
typedef struct { unsigned char *p; size_t n, cap; int fail; } buf;
static void buf_put(buf *b, const void *src, size_t n) {
  if (b->n + n > b->cap) {
    size_t cap = b->cap ? b->cap * 2 : 1024;
    while (cap < b->n + n) cap *= 2;
    unsigned char *q = (unsigned char *)realloc(b->p, cap);
    if (!q) { b->fail = 1; return; }
    b->p = q; b->cap = cap;
  }
  memcpy(b->p + b->n, src, n);
  b->n += n;
}
static void buf_u32(buf *b, uint32_t v) { unsigned char t[4]; be_store(t, v, 4); buf_put(b, t, 4); }
static void buf_u64(buf *b, uint64_t v) { unsigned char t[8]; be_store(t, v, 8); buf_put(b, t, 8); }
static void buf_name(buf *b, const char *s) {
  static const unsigned char zero[4] = {0, 0, 0, 0};
  size_t n = strlen(s);
  buf_u32(b, (uint32_t)n);
  buf_put(b, s, n);
  buf_put(b, zero, (size_t)(pad4((int64_t)n) - (int64_t)n));
}
static void buf_atts(buf *b, const cdf_att *a, int n) {
  static const unsigned char zero[4] = {0, 0, 0, 0};
  if (n == 0) { buf_u32(b, 0); buf_u32(b, 0); return; }
  buf_u32(b, 0x0C); buf_u32(b, (uint32_t)n);
  for (int i = 0; i < n; ++i) {
    const int ts = type_size(a[i].xtype);
    buf_name(b, a[i].name);
    buf_u32(b, (uint32_t)a[i].xtype);
    buf_u32(b, (uint32_t)a[i].n);
    for (int64_t k = 0; k < a[i].n; ++k) {
      unsigned char t[8];
      double_to_ext(mem_to_double(a[i].data, k, a[i].xtype), t, a[i].xtype);
      buf_put(b, t, (size_t)ts);
    }
    buf_put(b, zero, (size_t)(pad4(a[i].n * ts) - a[i].n * ts));
  }
}

static int64_t var_slab_elems(const cdf_file *f, const cdf_var *v) {
  int64_t n = 1;
  for (int d = v->is_record ? 1 : 0; d < v->ndims; ++d) n *= f->dims[v->dimids[d]].len;
  return n;
}

/* Header bytes for the current definitions */
static int build_header(const cdf_file *f, buf *b) {
  unsigned char magic[4] = {'C', 'D', 'F', (unsigned char)f->version};
  buf_put(b, magic, 4);
  buf_u32(b, (uint32_t)f->numrecs);
  if (f->ndims == 0) { buf_u32(b, 0); buf_u32(b, 0); }
  else {
    buf_u32(b, 0x0A); buf_u32(b, (uint32_t)f->ndims);
    for (int i = 0; i < f->ndims; ++i) { buf_name(b, f->dims[i].name); buf_u32(b, (uint32_t)f->dims[i].len); }
  }
  buf_atts(b, f->gatts, f->ngatts);
  if (f->nvars == 0) { buf_u32(b, 0); buf_u32(b, 0); }
  else {
    buf_u32(b, 0x0B); buf_u32(b, (uint32_t)f->nvars);
    for (int i = 0; i < f->nvars; ++i) {
      const cdf_var *v = &f->vars[i];
      buf_name(b, v->name);
      buf_u32(b, (uint32_t)v->ndims);
      for (int d = 0; d < v->ndims; ++d) buf_u32(b, (uint32_t)v->dimids[d]);
      buf_atts(b, v->atts, v->natts);
      buf_u32(b, (uint32_t)v->xtype);
      buf_u32(b, (uint32_t)(v->vsize > 0xffffffffLL ? 0xffffffffLL : v->vsize));
      if (f->version == 2) buf_u64(b, (uint64_t)v->begin); else buf_u32(b, (uint32_t)v->begin);
    }
  }
  return b->fail ? CDF_ENOMEM : CDF_NOERR;
}

/* vsize, begin, recsize for all variables (header_size known) */
static void layout(cdf_file *f, int64_t header_size) {
  int nrecvars = 0;
  int64_t off = header_size;
  for (int i = 0; i < f->nvars; ++i) {
    cdf_var *v = &f->vars[i];
    v->is_record = v->ndims > 0 && f->dims[v->dimids[0]].len == 0;
    v->vsize = pad4(var_slab_elems(f, v) * type_size(v->xtype));
    if (v->is_record) ++nrecvars;
  }
  for (int i = 0; i < f->nvars; ++i)
    if (!f->vars[i].is_record) { f->vars[i].begin = off; off += f->vars[i].vsize; }
  f->rec_begin = off;
  f->recsize = 0;
  for (int i = 0; i < f->nvars; ++i) {
    cdf_var *v = &f->vars[i];
    if (!v->is_record) continue;
    if (nrecvars == 1) v->vsize = var_slab_elems(f, v) * type_size(v->xtype);   /* no padding: spec special case */
    v->begin = off;
    off += v->vsize;
    f->recsize += v->vsize;
  }
}

static void free_atts(cdf_att *a, int n) {
  for (int i = 0; i < n; ++i) free(a[i].data);
  free(a);
}
static void free_file(cdf_file *f) {
  if (!f) return;
  for (int i = 0; i < f->nvars; ++i) { free(f->vars[i].dimids); free_atts(f->vars[i].atts, f->vars[i].natts); }
  free(f->vars);
  free(f->dims);
  free_atts(f->gatts, f->ngatts);
  free(f);
}

static int new_slot(cdf_file *f, int *ncid) {
  for (int i = 0; i < MAX_FILES; ++i)
    if (!g_files[i]) { g_files[i] = f; *ncid = i; return CDF_NOERR; }
  return CDF_ENOMEM;
}

/* ---- reading the header ---- */
typedef struct { FILE *fp; int fail; } rd;
static uint32_t rd_u32(rd *r) {
  unsigned char t[4];
  if (fread(t, 1, 4, r->fp) != 4) { r->fail = 1; return 0; }
  return (uint32_t)be_load(t, 4);
}
static uint64_t rd_u64(rd *r) {
  unsigned char t[8];
  if (fread(t, 1, 8, r->fp) != 8) { r->fail = 1; return 0; }
  return be_load(t, 8);
}
static void rd_name(rd *r, char *out) {
  uint32_t n = rd_u32(r);
  unsigned char tmp[CDF_MAX_NAME + 4];
  out[0] = 0;
  if (r->fail || n > CDF_MAX_NAME) { r->fail = 1; return; }
  size_t np = (size_t)pad4(n);
  if (fread(tmp, 1, np, r->fp) != np) { r->fail = 1; return; }
  memcpy(out, tmp, n);
  out[n] = 0;
}
static int rd_atts(rd *r, cdf_att **out, int *nout) {
  uint32_t tag = rd_u32(r), n = rd_u32(r);
  *out = NULL; *nout = 0;
  if (r->fail) return CDF_ENOTNC;
  if (tag == 0 && n == 0) return CDF_NOERR;
  if (tag != 0x0C || n > 65536) return CDF_ENOTNC;
  cdf_att *a = (cdf_att *)calloc(n ? n : 1, sizeof(cdf_att));
  if (!a) return CDF_ENOMEM;
  for (uint32_t i = 0; i < n; ++i) {
    rd_name(r, a[i].name);
    a[i].xtype = (int)rd_u32(r);
    a[i].n = (int64_t)rd_u32(r);
    const int ts = type_size(a[i].xtype);
    if (r->fail || ts == 0) { free_atts(a, (int)n); return CDF_ENOTNC; }
    const size_t bytes = (size_t)pad4(a[i].n * ts);
    unsigned char *raw = (unsigned char *)malloc(bytes ? bytes : 1);
    a[i].data = malloc((size_t)(a[i].n * ts) + 8);
    if (!raw || !a[i].data || fread(raw, 1, bytes, r->fp) != bytes) { free(raw); free_atts(a, (int)n); r->fail = 1; return CDF_ENOTNC; }
    for (int64_t k = 0; k < a[i].n; ++k) double_to_mem(ext_to_double(raw + k * ts, a[i].xtype), a[i].data, k, a[i].xtype);
    free(raw);
  }
  *out = a; *nout = (int)n;
  return CDF_NOERR;
}

int cdf_open(const char *path, int *ncid) {
  FILE *fp = fopen(path, "rb");
  if (!fp) return errno ? errno : CDF_EIO;
  unsigned char magic[4];
  if (fread(magic, 1, 4, fp) != 4 || memcmp(magic, "CDF", 3) != 0 || (magic[3] != 1 && magic[3] != 2)) {
    fclose(fp);
    return CDF_ENOTNC;
  }
  cdf_file *f = (cdf_file *)calloc(1, sizeof(cdf_file));
  if (!f) { fclose(fp); return CDF_ENOMEM; }
  f->fp = fp;
  f->version = magic[3];
  rd r = {fp, 0};
  int rc = CDF_NOERR;
  uint32_t numrecs = rd_u32(&r);
  uint32_t tag = rd_u32(&r), n = rd_u32(&r);
  if (r.fail || !((tag == 0 && n == 0) || tag == 0x0A) || n > 65536) { rc = CDF_ENOTNC; goto fail; }
  f->ndims = (int)n;
  f->dims = (cdf_dim *)calloc(n ? n : 1, sizeof(cdf_dim));
  for (uint32_t i = 0; i < n; ++i) { rd_name(&r, f->dims[i].name); f->dims[i].len = (int64_t)rd_u32(&r); }
  rc = rd_atts(&r, &f->gatts, &f->ngatts);
  if (rc) goto fail;
  tag = rd_u32(&r); n = rd_u32(&r);
  if (r.fail || !((tag == 0 && n == 0) || tag == 0x0B) || n > 65536) { rc = CDF_ENOTNC; goto fail; }
  f->nvars = (int)n;
  f->vars = (cdf_var *)calloc(n ? n : 1, sizeof(cdf_var));
  for (uint32_t i = 0; i < n; ++i) {
    cdf_var *v = &f->vars[i];
    rd_name(&r, v->name);
    v->ndims = (int)rd_u32(&r);
    if (r.fail || v->ndims < 0 || v->ndims > CDF_MAX_VAR_DIMS) { rc = CDF_ENOTNC; goto fail; }
    v->dimids = (int *)calloc(v->ndims ? v->ndims : 1, sizeof(int));
    for (int d = 0; d < v->ndims; ++d) {
      v->dimids[d] = (int)rd_u32(&r);
      if (v->dimids[d] < 0 || v->dimids[d] >= f->ndims) { rc = CDF_ENOTNC; goto fail; }
    }
    rc = rd_atts(&r, &v->atts, &v->natts);
    if (rc) goto fail;
    v->xtype = (int)rd_u32(&r);
    v->vsize = (int64_t)rd_u32(&r);
    v->begin = f->version == 2 ? (int64_t)rd_u64(&r) : (int64_t)rd_u32(&r);
    if (r.fail || type_size(v->xtype) == 0) { rc = CDF_ENOTNC; goto fail; }
    v->is_record = v->ndims > 0 && f->dims[v->dimids[0]].len == 0;
  }
  /* record geometry from the header's own vsize (honours the one-record-variable rule) */
  {
    int nrecvars = 0;
    f->recsize = 0; f->rec_begin = 0;
    for (int i = 0; i < f->nvars; ++i)
      if (f->vars[i].is_record) {
        if (!nrecvars || f->vars[i].begin < f->rec_begin) f->rec_begin = f->vars[i].begin;
        ++nrecvars;
      }
    for (int i = 0; i < f->nvars; ++i) {
      cdf_var *v = &f->vars[i];
      const int64_t exact = var_slab_elems(f, v) * type_size(v->xtype);
      if (v->is_record) {
        v->vsize = (nrecvars == 1) ? exact : pad4(exact);
        f->recsize += v->vsize;
      } else {
        v->vsize = pad4(exact);    /* the 32-bit field saturates for huge variables */
      }
    }
  }
  f->numrecs = (numrecs == 0xffffffffu) ? 0 : (int64_t)numrecs;
  if (numrecs == 0xffffffffu && f->recsize > 0) {      /* "streaming": derive from the file size */
    fseeko(fp, 0, SEEK_END);
    f->numrecs = ((int64_t)ftello(fp) - f->rec_begin) / f->recsize;
  }
  rc = new_slot(f, ncid);
  if (rc) goto fail;
  return CDF_NOERR;
fail:
  fclose(fp);
  free_file(f);
  return rc;
}

int cdf_create(const char *path, int noclobber, int *ncid) {
  if (noclobber) {
    FILE *t = fopen(path, "rb");
    if (t) { fclose(t); return CDF_EEXIST; }
  }
  FILE *fp = fopen(path, "w+b");
  if (!fp) return errno ? errno : CDF_EIO;
  cdf_file *f = (cdf_file *)calloc(1, sizeof(cdf_file));
  if (!f) { fclose(fp); return CDF_ENOMEM; }
  f->fp = fp;
  f->writable = 1;
  f->define_mode = 1;
  f->version = 1;
  int rc = new_slot(f, ncid);
  if (rc) { fclose(fp); free_file(f); }
  return rc;
}

static int name_ok(const char *name) {
  size_t n = name ? strlen(name) : 0;
  return n > 0 && n <= CDF_MAX_NAME;
}

int cdf_def_dim(int ncid, const char *name, int64_t len, int *dimid) {
  cdf_file *f = lookup(ncid);
  if (!f) return CDF_EBADID;
  if (!f->writable) return CDF_EPERM;
  if (!f->define_mode) return CDF_ENOTINDEFINE;
  if (!name_ok(name) || len < 0 || len > 0x7fffffffLL) return CDF_EINVAL;
  for (int i = 0; i < f->ndims; ++i) {
    if (!strcmp(f->dims[i].name, name)) return CDF_ENAMEINUSE;
    if (len == 0 && f->dims[i].len == 0) return CDF_EUNLIMIT;
  }
  cdf_dim *d = (cdf_dim *)realloc(f->dims, sizeof(cdf_dim) * (size_t)(f->ndims + 1));
  if (!d) return CDF_ENOMEM;
  f->dims = d;
  memset(&d[f->ndims], 0, sizeof(cdf_dim));
  strcpy(d[f->ndims].name, name);
  d[f->ndims].len = len;
  if (dimid) *dimid = f->ndims;
  ++f->ndims;
  return CDF_NOERR;
}

int cdf_def_var(int ncid, const char *name, int xtype, int ndims, const int *dimids, int *varid) {
  cdf_file *f = lookup(ncid);
  if (!f) return CDF_EBADID;
  if (!f->writable) return CDF_EPERM;
  if (!f->define_mode) return CDF_ENOTINDEFINE;
  if (!name_ok(name) || ndims < 0 || ndims > CDF_MAX_VAR_DIMS) return CDF_EINVAL;
  if (type_size(xtype) == 0) return CDF_EBADTYPE;
  for (int i = 0; i < f->nvars; ++i) if (!strcmp(f->vars[i].name, name)) return CDF_ENAMEINUSE;
  for (int d = 0; d < ndims; ++d) {
    if (dimids[d] < 0 || dimids[d] >= f->ndims) return CDF_EBADDIM;
    if (d > 0 && f->dims[dimids[d]].len == 0) return CDF_EUNLIMIT;   /* record dimension must come first */
  }
  cdf_var *v = (cdf_var *)realloc(f->vars, sizeof(cdf_var) * (size_t)(f->nvars + 1));
  if (!v) return CDF_ENOMEM;
  f->vars = v;
  v += f->nvars;
  memset(v, 0, sizeof(cdf_var));
  strcpy(v->name, name);
  v->xtype = xtype;
  v->ndims = ndims;
  v->dimids = (int *)calloc(ndims ? ndims : 1, sizeof(int));
  if (!v->dimids) return CDF_ENOMEM;
  for (int d = 0; d < ndims; ++d) v->dimids[d] = dimids[d];
  if (varid) *varid = f->nvars;
  ++f->nvars;
  return CDF_NOERR;
}

static int att_list(cdf_file *f, int varid, cdf_att ***list, int **n) {
  if (varid == CDF_GLOBAL) { *list = &f->gatts; *n = &f->ngatts; return CDF_NOERR; }
  if (varid < 0 || varid >= f->nvars) return CDF_ENOTVAR;
  *list = &f->vars[varid].atts; *n = &f->vars[varid].natts;
  return CDF_NOERR;
}

int cdf_put_att(int ncid, int varid, const char *name, int xtype, int64_t n, const void *values) {
  cdf_file *f = lookup(ncid);
  if (!f) return CDF_EBADID;
  if (!f->writable) return CDF_EPERM;
  if (!f->define_mode) return CDF_ENOTINDEFINE;   /* attributes fix the header size */
  if (!name_ok(name) || n < 0) return CDF_EINVAL;
  const int ts = type_size(xtype);
  if (!ts) return CDF_EBADTYPE;
  cdf_att **list; int *cnt;
  int rc = att_list(f, varid, &list, &cnt);
  if (rc) return rc;
  cdf_att *a = NULL;
  for (int i = 0; i < *cnt; ++i) if (!strcmp((*list)[i].name, name)) a = &(*list)[i];
  if (!a) {
    cdf_att *q = (cdf_att *)realloc(*list, sizeof(cdf_att) * (size_t)(*cnt + 1));
    if (!q) return CDF_ENOMEM;
    *list = q;
    a = &q[*cnt];
    memset(a, 0, sizeof(cdf_att));
    strcpy(a->name, name);
    ++*cnt;
  }
  free(a->data);
  a->data = malloc((size_t)(n * ts) + 8);
  if (!a->data) return CDF_ENOMEM;
  memcpy(a->data, values, (size_t)(n * ts));
  a->xtype = xtype;
  a->n = n;
  return CDF_NOERR;
}

/* Fill value of a variable: its _FillValue attribute or the format default */
static double fill_value(const cdf_var *v) {
  for (int i = 0; i < v->natts; ++i)
    if (!strcmp(v->atts[i].name, "_FillValue") && v->atts[i].n >= 1 && v->atts[i].xtype == v->xtype)
      return mem_to_double(v->atts[i].data, 0, v->xtype);
  switch (v->xtype) {
    case CDF_BYTE: return -127.0;
    case CDF_CHAR: return 0.0;
    case CDF_SHORT: return -32767.0;
    case CDF_INT: return -2147483647.0;
    default: return 9.9692099683868690e+36;
  }
}

/* Write n fill elements of variable v at offset off */
static int write_fill(cdf_file *f, const cdf_var *v, int64_t off, int64_t bytes) {
  const int ts = type_size(v->xtype);
  unsigned char pat[8], chunk[8192];
  double_to_ext(fill_value(v), pat, v->xtype);
  for (size_t i = 0; i < sizeof(chunk); i += (size_t)ts) memcpy(chunk + i, pat, (size_t)ts);
  if (fseeko(f->fp, (off_t)off, SEEK_SET)) return CDF_EIO;
  while (bytes > 0) {
    size_t n = bytes > (int64_t)sizeof(chunk) ? sizeof(chunk) : (size_t)bytes;
    if (fwrite(chunk, 1, n, f->fp) != n) return CDF_EIO;
    bytes -= (int64_t)n;
  }
  return CDF_NOERR;
}

static int write_header(cdf_file *f) {
  buf b = {0};
  int rc = build_header(f, &b);
  if (!rc) {
    if (fseeko(f->fp, 0, SEEK_SET) || fwrite(b.p, 1, b.n, f->fp) != b.n) rc = CDF_EIO;
  }
  free(b.p);
  return rc;
}

int cdf_enddef(int ncid) {
  cdf_file *f = lookup(ncid);
  if (!f) return CDF_EBADID;
  if (!f->writable) return CDF_EPERM;
  if (!f->define_mode) return CDF_ENOTINDEFINE;
  /* the header size does not depend on the offsets except through the
   * version (width of `begin`): lay out with CDF-1, switch to CDF-2 if any
   * offset does not fit in 31 bits */
  for (int pass = 0; pass < 2; ++pass) {
    buf b = {0};
    layout(f, 0);
    int rc = build_header(f, &b);
    const int64_t hsize = (int64_t)b.n;
    free(b.p);
    if (rc) return rc;
    layout(f, pad4(hsize));
    int64_t last = f->rec_begin + f->recsize;
    if (f->version == 1 && last > 0x7fffffffLL) { f->version = 2; continue; }
    break;
  }
  int rc = write_header(f);
  if (rc) return rc;
  for (int i = 0; i < f->nvars; ++i)
    if (!f->vars[i].is_record) {
      rc = write_fill(f, &f->vars[i], f->vars[i].begin, f->vars[i].vsize);
      if (rc) return rc;
    }
  f->define_mode = 0;
  return CDF_NOERR;
}

int cdf_close(int ncid) {
  cdf_file *f = lookup(ncid);
  if (!f) return CDF_EBADID;
  int rc = CDF_NOERR;
  if (f->writable) {
    if (f->define_mode) rc = cdf_enddef(ncid);
    if (!rc && f->numrecs_dirty) {
      unsigned char t[4];
      be_store(t, (uint64_t)f->numrecs, 4);
      if (fseeko(f->fp, 4, SEEK_SET) || fwrite(t, 1, 4, f->fp) != 4) rc = CDF_EIO;
    }
  }
  if (fclose(f->fp) && !rc) rc = CDF_EIO;
  g_files[ncid] = NULL;
  free_file(f);
  return rc;
}

int cdf_inq_dimid(int ncid, const char *name, int *dimid) {
  cdf_file *f = lookup(ncid);
  if (!f) return CDF_EBADID;
  for (int i = 0; i < f->ndims; ++i)
    if (!strcmp(f->dims[i].name, name)) { if (dimid) *dimid = i; return CDF_NOERR; }
  return CDF_EBADDIM;
}

int cdf_inq_dim(int ncid, int dimid, char *name, int64_t *len) {
  cdf_file *f = lookup(ncid);
  if (!f) return CDF_EBADID;
  if (dimid < 0 || dimid >= f->ndims) return CDF_EBADDIM;
  if (name) strcpy(name, f->dims[dimid].name);
  if (len) *len = f->dims[dimid].len ? f->dims[dimid].len : f->numrecs;
  return CDF_NOERR;
}

int cdf_inq_varid(int ncid, const char *name, int *varid) {
  cdf_file *f = lookup(ncid);
  if (!f) return CDF_EBADID;
  for (int i = 0; i < f->nvars; ++i)
    if (!strcmp(f->vars[i].name, name)) { if (varid) *varid = i; return CDF_NOERR; }
  return CDF_ENOTVAR;
}

int cdf_inq_var(int ncid, int varid, char *name, int *xtype, int *ndims, int *dimids, int *natts) {
  cdf_file *f = lookup(ncid);
  if (!f) return CDF_EBADID;
  if (varid < 0 || varid >= f->nvars) return CDF_ENOTVAR;
  const cdf_var *v = &f->vars[varid];
  if (name) strcpy(name, v->name);
  if (xtype) *xtype = v->xtype;
  if (ndims) *ndims = v->ndims;
  if (dimids) for (int d = 0; d < v->ndims; ++d) dimids[d] = v->dimids[d];
  if (natts) *natts = v->natts;
  return CDF_NOERR;
}

static const cdf_att *find_att(cdf_file *f, int varid, const char *name, int *rc) {
  cdf_att **list; int *cnt;
  *rc = att_list(f, varid, &list, &cnt);
  if (*rc) return NULL;
  for (int i = 0; i < *cnt; ++i) if (!strcmp((*list)[i].name, name)) return &(*list)[i];
  *rc = CDF_ENOTATT;
  return NULL;
}

int cdf_inq_att(int ncid, int varid, const char *name, int *xtype, int64_t *len) {
  cdf_file *f = lookup(ncid);
  if (!f) return CDF_EBADID;
  int rc;
  const cdf_att *a = find_att(f, varid, name, &rc);
  if (!a) return rc;
  if (xtype) *xtype = a->xtype;
  if (len) *len = a->n;
  return CDF_NOERR;
}

int cdf_get_att(int ncid, int varid, const char *name, int memtype, void *values, int64_t maxn) {
  cdf_file *f = lookup(ncid);
  if (!f) return CDF_EBADID;
  int rc;
  const cdf_att *a = find_att(f, varid, name, &rc);
  if (!a) return rc;
  const int64_t n = a->n < maxn ? a->n : maxn;
  if ((memtype == CDF_CHAR) != (a->xtype == CDF_CHAR)) return CDF_EBADTYPE;   /* text <-> number */
  if (memtype == CDF_CHAR) { memcpy(values, a->data, (size_t)n); return CDF_NOERR; }
  if (type_size(memtype) == 0) return CDF_EBADTYPE;
  for (int64_t k = 0; k < n; ++k) double_to_mem(mem_to_double(a->data, k, a->xtype), values, k, memtype);
  return CDF_NOERR;
}

/* Common hyperslab walker: calls back once per contiguous run of the fastest
 * dimension with (file offset, element index in the caller's buffer, run length) */
typedef int (*run_fn)(cdf_file *, const cdf_var *, int64_t off, int64_t first, int64_t n, int memtype, void *mem);

static int walk(cdf_file *f, int varid, const int64_t *start, const int64_t *count, int memtype, void *mem,
                run_fn fn, int writing) {
  if (varid < 0 || varid >= f->nvars) return CDF_ENOTVAR;
  const cdf_var *v = &f->vars[varid];
  if (type_size(memtype) == 0) return CDF_EBADTYPE;
  if ((memtype == CDF_CHAR) != (v->xtype == CDF_CHAR)) return CDF_EBADTYPE;
  const int nd = v->ndims, ts = type_size(v->xtype);
  int64_t st[32], ct[32], len[32], stride[32], idx[32];
  if (nd > 32) return CDF_EINVAL;
  int64_t total = 1;
  for (int d = 0; d < nd; ++d) {
    len[d] = f->dims[v->dimids[d]].len;
    const int rec = (d == 0 && v->is_record);
    if (rec) len[d] = f->numrecs;
    st[d] = start ? start[d] : 0;
    ct[d] = count ? count[d] : len[d];
    if (st[d] < 0 || ct[d] < 0) return CDF_EINVALCOORDS;
    if (rec && writing) {
      /* the record dimension grows on write */
    } else {
      if (st[d] > len[d] || (ct[d] > 0 && st[d] >= len[d])) return CDF_EINVALCOORDS;
      if (st[d] + ct[d] > len[d]) return CDF_EEDGE;
    }
    total *= ct[d];
  }
  if (total == 0) return CDF_NOERR;
  /* element strides inside one (record) slab */
  int64_t s = 1;
  for (int d = nd - 1; d >= 0; --d) {
    stride[d] = s;
    if (!(d == 0 && v->is_record)) s *= f->dims[v->dimids[d]].len;
  }
  if (v->is_record && writing && st[0] + ct[0] > f->numrecs) {
    /* fill the new records of every record variable, then extend */
    const int64_t newrecs = st[0] + ct[0];
    for (int64_t r = f->numrecs; r < newrecs; ++r)
      for (int i = 0; i < f->nvars; ++i)
        if (f->vars[i].is_record) {
          int rc = write_fill(f, &f->vars[i], f->vars[i].begin + r * f->recsize, f->vars[i].vsize);
          if (rc) return rc;
        }
    f->numrecs = newrecs;
    f->numrecs_dirty = 1;
  }
  if (nd == 0) return fn(f, v, v->begin, 0, 1, memtype, mem);
  const int64_t run = ct[nd - 1];
  for (int d = 0; d < nd; ++d) idx[d] = 0;
  int64_t done = 0;
  while (done < total) {
    int64_t off = v->begin;
    for (int d = 0; d < nd; ++d) {
      const int64_t coord = st[d] + idx[d];
      if (d == 0 && v->is_record) off += coord * f->recsize;
      else off += coord * stride[d] * ts;
    }
    /* a 1-D record variable has runs of one element per record */
    const int64_t n = (nd == 1 && v->is_record) ? 1 : run;
    int rc = fn(f, v, off, done, n, memtype, mem);
    if (rc) return rc;
    done += n;
    /* advance the odometer (fastest dimension handled by the run) */
    int d = (nd == 1 && v->is_record) ? 0 : nd - 2;
    for (; d >= 0; --d) {
      if (++idx[d] < ct[d]) break;
      idx[d] = 0;
    }
    if (d < 0) break;
  }
  return CDF_NOERR;
}

static int read_run(cdf_file *f, const cdf_var *v, int64_t off, int64_t first, int64_t n, int memtype, void *mem) {
  const int ts = type_size(v->xtype);
  unsigned char stackbuf[8192];
  if (fseeko(f->fp, (off_t)off, SEEK_SET)) return CDF_EIO;
  while (n > 0) {
    const int64_t m = n < (int64_t)(sizeof(stackbuf) / 8) ? n : (int64_t)(sizeof(stackbuf) / 8);
    if (fread(stackbuf, (size_t)ts, (size_t)m, f->fp) != (size_t)m) return CDF_EIO;
    for (int64_t k = 0; k < m; ++k) double_to_mem(ext_to_double(stackbuf + k * ts, v->xtype), mem, first + k, memtype);
    first += m; n -= m;
  }
  return CDF_NOERR;
}

static int write_run(cdf_file *f, const cdf_var *v, int64_t off, int64_t first, int64_t n, int memtype, void *mem) {
  const int ts = type_size(v->xtype);
  unsigned char stackbuf[8192];
  if (fseeko(f->fp, (off_t)off, SEEK_SET)) return CDF_EIO;
  while (n > 0) {
    const int64_t m = n < (int64_t)(sizeof(stackbuf) / 8) ? n : (int64_t)(sizeof(stackbuf) / 8);
    for (int64_t k = 0; k < m; ++k) double_to_ext(mem_to_double(mem, first + k, memtype), stackbuf + k * ts, v->xtype);
    if (fwrite(stackbuf, (size_t)ts, (size_t)m, f->fp) != (size_t)m) return CDF_EIO;
    first += m; n -= m;
  }
  return CDF_NOERR;
}

int cdf_get_vara(int ncid, int varid, const int64_t *start, const int64_t *count, int memtype, void *out) {
  cdf_file *f = lookup(ncid);
  if (!f) return CDF_EBADID;
  if (f->define_mode) return CDF_EINDEFINE;
  if (f->writable) fflush(f->fp);
  return walk(f, varid, start, count, memtype, out, read_run, 0);
}

int cdf_put_vara(int ncid, int varid, const int64_t *start, const int64_t *count, int memtype, const void *in) {
  cdf_file *f = lookup(ncid);
  if (!f) return CDF_EBADID;
  if (!f->writable) return CDF_EPERM;
  if (f->define_mode) return CDF_EINDEFINE;
  return walk(f, varid, start, count, memtype, (void *)in, write_run, 1);
}

const char *cdf_strerror(int status) {
  switch (status) {
    case CDF_NOERR: return "No error";
    case CDF_EBADID: return "NetCDF: Not a valid ID";
    case CDF_EEXIST: return "NetCDF: File exists && NC_NOCLOBBER";
    case CDF_EINVAL: return "NetCDF: Invalid argument";
    case CDF_EPERM: return "NetCDF: Write to read only";
    case CDF_ENOTINDEFINE: return "NetCDF: Operation not allowed in data mode";
    case CDF_EINDEFINE: return "NetCDF: Operation not allowed in define mode";
    case CDF_EINVALCOORDS: return "NetCDF: Index exceeds dimension bound";
    case CDF_ENAMEINUSE: return "NetCDF: String match to name in use";
    case CDF_ENOTATT: return "NetCDF: Attribute not found";
    case CDF_EBADTYPE: return "NetCDF: Not a valid data type or _FillValue type mismatch";
    case CDF_EBADDIM: return "NetCDF: Invalid dimension ID or name";
    case CDF_EUNLIMIT: return "NetCDF: NC_UNLIMITED size already in use";
    case CDF_ENOTVAR: return "NetCDF: Variable not found";
    case CDF_ENOTNC: return "NetCDF: Unknown file format (only classic CDF-1/CDF-2 is supported by this library)";
    case CDF_EEDGE: return "NetCDF: Start+count exceeds dimension bound";
    case CDF_ENOMEM: return "NetCDF: Memory allocation (malloc) failure";
    case CDF_EIO: return "NetCDF: I/O failure";
    default: return status > 0 ? strerror(status) : "NetCDF: Unknown error";
  }
}
