! netcdf.F90 - a stand-in for netcdf-fortran's `module netcdf`, limited to the
! nf90_* symbols that the reference's utilities/easy_netcdf.F90 uses (SURVEY.md
! section 8f row 3), over the classic-format engine in cdf_classic.c.
!
! With it the reference's untouched driver (driver/spartacus_surface_driver.F90,
! _read_input.F90), radsurf/radsurf_save.F90 and utilities/easy_netcdf.F90 link
! and run where netcdf-fortran is not installed: inputs in netCDF classic /
! 64-bit-offset format are read, outputs are written in classic format
! (NF90_HDF5 in the creation mode is accepted and ignored).
!
! Semantics follow the netCDF Fortran 90 interface: ids are 1-based
! (NF90_GLOBAL = 0), dimension order is Fortran order (fastest first), an
! absent `start` means 1, an absent `count` means the shape of `values` padded
! with ones, numeric types are converted between memory and file.
! stride= and map= are not provided (easy_netcdf does not use them).
module netcdf

  use, intrinsic :: iso_c_binding

  implicit none
  private

  integer, parameter, public :: nf90_noerr = 0, nf90_nowrite = 0, nf90_write = 1, &
       &  nf90_clobber = 0, nf90_noclobber = 4, nf90_64bit_offset = 512, &
       &  nf90_netcdf4 = 4096, nf90_hdf5 = 4096, nf90_classic_model = 256, &
       &  nf90_global = 0, nf90_unlimited = 0, &
       &  nf90_byte = 1, nf90_int1 = 1, nf90_char = 2, nf90_short = 3, nf90_int2 = 3, &
       &  nf90_int = 4, nf90_int4 = 4, nf90_float = 5, nf90_real = 5, nf90_real4 = 5, &
       &  nf90_double = 6, nf90_real8 = 6, &
       &  nf90_max_var_dims = 1024, nf90_max_dims = 1024, nf90_max_name = 256, &
       &  nf90_fill = 0, nf90_nofill = 256
  integer, parameter, public :: nf90_ebadid = -33, nf90_eexist = -35, nf90_einval = -36, &
       &  nf90_eperm = -37, nf90_enotindefine = -38, nf90_eindefine = -39, &
       &  nf90_einvalcoords = -40, nf90_enameinuse = -42, nf90_enotatt = -43, &
       &  nf90_ebadtype = -45, nf90_ebaddim = -46, nf90_enotvar = -49, nf90_enotnc = -51, &
       &  nf90_eedge = -57
  real(c_float),  parameter, public :: nf90_fill_float  = 9.9692099683868690e+36
  real(c_double), parameter, public :: nf90_fill_double = 9.9692099683868690d+36

  public :: nf90_open, nf90_create, nf90_close, nf90_enddef, nf90_strerror, &
       &    nf90_def_dim, nf90_def_var, nf90_def_var_fill, nf90_inq_varid, nf90_inq_dimid, &
       &    nf90_inquire_variable, nf90_inquire_dimension, nf90_inquire_attribute, &
       &    nf90_get_att, nf90_put_att, nf90_get_var, nf90_put_var

  interface
    integer(c_int) function cdf_open(path, ncid) bind(C, name="cdf_open")
      import :: c_int, c_char
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int), intent(out) :: ncid
    end function
    integer(c_int) function cdf_create(path, noclobber, ncid) bind(C, name="cdf_create")
      import :: c_int, c_char
      character(kind=c_char), intent(in) :: path(*)
      integer(c_int), value :: noclobber
      integer(c_int), intent(out) :: ncid
    end function
    integer(c_int) function cdf_enddef(ncid) bind(C, name="cdf_enddef")
      import :: c_int
      integer(c_int), value :: ncid
    end function
    integer(c_int) function cdf_close(ncid) bind(C, name="cdf_close")
      import :: c_int
      integer(c_int), value :: ncid
    end function
    integer(c_int) function cdf_def_dim(ncid, name, len, dimid) bind(C, name="cdf_def_dim")
      import :: c_int, c_char, c_int64_t
      integer(c_int), value :: ncid
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int64_t), value :: len
      integer(c_int), intent(out) :: dimid
    end function
    integer(c_int) function cdf_def_var(ncid, name, xtype, ndims, dimids, varid) bind(C, name="cdf_def_var")
      import :: c_int, c_char
      integer(c_int), value :: ncid, xtype, ndims
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int), intent(in) :: dimids(*)
      integer(c_int), intent(out) :: varid
    end function
    integer(c_int) function cdf_put_att(ncid, varid, name, xtype, n, values) bind(C, name="cdf_put_att")
      import :: c_int, c_char, c_int64_t, c_ptr
      integer(c_int), value :: ncid, varid, xtype
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int64_t), value :: n
      type(c_ptr), value :: values
    end function
    integer(c_int) function cdf_inq_dimid(ncid, name, dimid) bind(C, name="cdf_inq_dimid")
      import :: c_int, c_char
      integer(c_int), value :: ncid
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int), intent(out) :: dimid
    end function
    integer(c_int) function cdf_inq_dim(ncid, dimid, name, len) bind(C, name="cdf_inq_dim")
      import :: c_int, c_char, c_int64_t
      integer(c_int), value :: ncid, dimid
      character(kind=c_char), intent(out) :: name(*)
      integer(c_int64_t), intent(out) :: len
    end function
    integer(c_int) function cdf_inq_varid(ncid, name, varid) bind(C, name="cdf_inq_varid")
      import :: c_int, c_char
      integer(c_int), value :: ncid
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int), intent(out) :: varid
    end function
    integer(c_int) function cdf_inq_var(ncid, varid, name, xtype, ndims, dimids, natts) bind(C, name="cdf_inq_var")
      import :: c_int, c_char
      integer(c_int), value :: ncid, varid
      character(kind=c_char), intent(out) :: name(*)
      integer(c_int), intent(out) :: xtype, ndims, natts
      integer(c_int), intent(out) :: dimids(*)
    end function
    integer(c_int) function cdf_inq_att(ncid, varid, name, xtype, len) bind(C, name="cdf_inq_att")
      import :: c_int, c_char, c_int64_t
      integer(c_int), value :: ncid, varid
      character(kind=c_char), intent(in) :: name(*)
      integer(c_int), intent(out) :: xtype
      integer(c_int64_t), intent(out) :: len
    end function
    integer(c_int) function cdf_get_att(ncid, varid, name, memtype, values, maxn) bind(C, name="cdf_get_att")
      import :: c_int, c_char, c_int64_t, c_ptr
      integer(c_int), value :: ncid, varid, memtype
      character(kind=c_char), intent(in) :: name(*)
      type(c_ptr), value :: values
      integer(c_int64_t), value :: maxn
    end function
    integer(c_int) function cdf_get_vara(ncid, varid, start, count, memtype, values) bind(C, name="cdf_get_vara")
      import :: c_int, c_int64_t, c_ptr
      integer(c_int), value :: ncid, varid, memtype
      integer(c_int64_t), intent(in) :: start(*), count(*)
      type(c_ptr), value :: values
    end function
    integer(c_int) function cdf_put_vara(ncid, varid, start, count, memtype, values) bind(C, name="cdf_put_vara")
      import :: c_int, c_int64_t, c_ptr
      integer(c_int), value :: ncid, varid, memtype
      integer(c_int64_t), intent(in) :: start(*), count(*)
      type(c_ptr), value :: values
    end function
    type(c_ptr) function cdf_strerror(status) bind(C, name="cdf_strerror")
      import :: c_int, c_ptr
      integer(c_int), value :: status
    end function
    integer(c_size_t) function c_strlen(s) bind(C, name="strlen")
      import :: c_size_t, c_ptr
      type(c_ptr), value :: s
    end function
  end interface

  interface nf90_def_var
    module procedure def_var_many, def_var_one, def_var_scalar
  end interface
  interface nf90_def_var_fill
    module procedure def_var_fill_r8, def_var_fill_r4, def_var_fill_i4, def_var_fill_i2, def_var_fill_i1
  end interface
  interface nf90_put_att
    module procedure put_att_text, put_att_r8, put_att_r4, put_att_i4, put_att_i2, put_att_i1, &
         &           put_att_r8_vec, put_att_r4_vec, put_att_i4_vec
  end interface
  interface nf90_get_att
    module procedure get_att_text, get_att_r8, get_att_r4, get_att_i4, get_att_r8_vec, get_att_i4_vec
  end interface
  interface nf90_get_var
    module procedure get_var_r8_0d, get_var_r8_1d, get_var_r8_2d, get_var_r8_3d, &
         &  get_var_r8_4d, get_var_r8_5d, get_var_r4_0d, get_var_r4_1d, get_var_r4_2d, &
         &  get_var_r4_3d, get_var_r4_4d, get_var_i4_0d, get_var_i4_1d, get_var_i4_2d, &
         &  get_var_i4_3d, get_var_i4_4d, get_var_i2_0d, get_var_i2_1d, get_var_i2_2d, &
         &  get_var_i1_0d, get_var_i1_1d, get_var_i1_2d
  end interface
  interface nf90_put_var
    module procedure put_var_r8_0d, put_var_r8_1d, put_var_r8_2d, put_var_r8_3d, &
         &  put_var_r8_4d, put_var_r8_5d, put_var_r4_0d, put_var_r4_1d, put_var_r4_2d, &
         &  put_var_r4_3d, put_var_r4_4d, put_var_i4_0d, put_var_i4_1d, put_var_i4_2d, &
         &  put_var_i4_3d, put_var_i4_4d, put_var_i2_0d, put_var_i2_1d, put_var_i2_2d, &
         &  put_var_i1_0d, put_var_i1_1d, put_var_i1_2d
  end interface

contains

  ! ---- helpers ----
  pure function cstr(s) result(c)
    character(len=*), intent(in) :: s
    character(kind=c_char) :: c(len_trim(s) + 1)
    integer :: i
    do i = 1, len_trim(s)
      c(i) = s(i:i)
    end do
    c(len_trim(s) + 1) = c_null_char
  end function cstr

  ! Fortran variable / attribute-owner id -> engine id (NF90_GLOBAL = 0 -> -1)
  pure integer(c_int) function cvar(varid)
    integer, intent(in) :: varid
    cvar = int(varid - 1, c_int)
  end function cvar

  function nf90_strerror(status) result(msg)
    integer, intent(in) :: status
    character(len=80) :: msg
    type(c_ptr) :: p
    character(kind=c_char), pointer :: s(:)
    integer :: i, n
    msg = ' '
    p = cdf_strerror(int(status, c_int))
    if (.not. c_associated(p)) return
    n = min(int(c_strlen(p)), len(msg))
    call c_f_pointer(p, s, [n])
    do i = 1, n
      msg(i:i) = s(i)
    end do
  end function nf90_strerror

  ! ---- files ----
  function nf90_open(path, mode, ncid, chunksize) result(status)
    character(len=*), intent(in) :: path
    integer, intent(in) :: mode
    integer, intent(out) :: ncid
    integer, intent(inout), optional :: chunksize
    integer :: status
    integer(c_int) :: id
    if (iand(mode, nf90_write) /= 0) then
      status = nf90_eperm      ! existing files are opened read-only
      ncid = -1
      return
    end if
    status = cdf_open(cstr(path), id)
    ncid = id
  end function nf90_open

  function nf90_create(path, cmode, ncid, initialsize, chunksize) result(status)
    character(len=*), intent(in) :: path
    integer, intent(in) :: cmode
    integer, intent(out) :: ncid
    integer, intent(in), optional :: initialsize
    integer, intent(inout), optional :: chunksize
    integer :: status
    integer(c_int) :: id, noclobber
    noclobber = 0
    if (iand(cmode, nf90_noclobber) /= 0) noclobber = 1
    status = cdf_create(cstr(path), noclobber, id)
    ncid = id
  end function nf90_create

  function nf90_close(ncid) result(status)
    integer, intent(in) :: ncid
    integer :: status
    status = cdf_close(int(ncid, c_int))
  end function nf90_close

  function nf90_enddef(ncid, h_minfree, v_align, v_minfree, r_align) result(status)
    integer, intent(in) :: ncid
    integer, intent(in), optional :: h_minfree, v_align, v_minfree, r_align
    integer :: status
    status = cdf_enddef(int(ncid, c_int))
  end function nf90_enddef

  ! ---- definitions ----
  function nf90_def_dim(ncid, name, len, dimid) result(status)
    integer, intent(in) :: ncid, len
    character(len=*), intent(in) :: name
    integer, intent(out) :: dimid
    integer :: status
    integer(c_int) :: id
    status = cdf_def_dim(int(ncid, c_int), cstr(name), int(len, c_int64_t), id)
    dimid = id + 1
  end function nf90_def_dim

  function def_var_many(ncid, name, xtype, dimids, varid) result(status)
    integer, intent(in) :: ncid, xtype, dimids(:)
    character(len=*), intent(in) :: name
    integer, intent(out) :: varid
    integer :: status, n, k
    integer(c_int) :: id, cdims(max(1, size(dimids)))
    n = size(dimids)
    do k = 1, n                       ! file order = reversed Fortran order
      cdims(k) = int(dimids(n + 1 - k) - 1, c_int)
    end do
    status = cdf_def_var(int(ncid, c_int), cstr(name), int(xtype, c_int), int(n, c_int), cdims, id)
    varid = id + 1
  end function def_var_many

  function def_var_one(ncid, name, xtype, dimids, varid) result(status)
    integer, intent(in) :: ncid, xtype, dimids
    character(len=*), intent(in) :: name
    integer, intent(out) :: varid
    integer :: status
    status = def_var_many(ncid, name, xtype, [dimids], varid)
  end function def_var_one

  function def_var_scalar(ncid, name, xtype, varid) result(status)
    integer, intent(in) :: ncid, xtype
    character(len=*), intent(in) :: name
    integer, intent(out) :: varid
    integer :: status
    integer :: none(0)
    status = def_var_many(ncid, name, xtype, none, varid)
  end function def_var_scalar

  ! _FillValue of the variable's own type (what nc_def_var_fill records)
  function def_var_fill_r8(ncid, varid, no_fill, fill) result(status)
    integer, intent(in) :: ncid, varid, no_fill
    real(c_double), intent(in) :: fill
    integer :: status
    integer(c_int) :: xtype, nd, na, dimids(nf90_max_var_dims)
    character(kind=c_char) :: nm(nf90_max_name + 1)
    status = cdf_inq_var(int(ncid, c_int), cvar(varid), nm, xtype, nd, dimids, na)
    if (status /= nf90_noerr .or. no_fill /= 0) return
    select case (xtype)
    case (nf90_double); status = put_att_r8(ncid, varid, "_FillValue", fill)
    case (nf90_float);  status = put_att_r4(ncid, varid, "_FillValue", real(fill, c_float))
    case (nf90_int);    status = put_att_i4(ncid, varid, "_FillValue", int(fill, c_int))
    case (nf90_short);  status = put_att_i2(ncid, varid, "_FillValue", int(fill, c_short))
    case (nf90_byte);   status = put_att_i1(ncid, varid, "_FillValue", int(fill, c_signed_char))
    case default;       status = nf90_ebadtype
    end select
  end function def_var_fill_r8
  function def_var_fill_r4(ncid, varid, no_fill, fill) result(status)
    integer, intent(in) :: ncid, varid, no_fill
    real(c_float), intent(in) :: fill
    integer :: status
    status = def_var_fill_r8(ncid, varid, no_fill, real(fill, c_double))
  end function def_var_fill_r4
  function def_var_fill_i4(ncid, varid, no_fill, fill) result(status)
    integer, intent(in) :: ncid, varid, no_fill
    integer(c_int), intent(in) :: fill
    integer :: status
    status = def_var_fill_r8(ncid, varid, no_fill, real(fill, c_double))
  end function def_var_fill_i4
  function def_var_fill_i2(ncid, varid, no_fill, fill) result(status)
    integer, intent(in) :: ncid, varid, no_fill
    integer(c_short), intent(in) :: fill
    integer :: status
    status = def_var_fill_r8(ncid, varid, no_fill, real(fill, c_double))
  end function def_var_fill_i2
  function def_var_fill_i1(ncid, varid, no_fill, fill) result(status)
    integer, intent(in) :: ncid, varid, no_fill
    integer(c_signed_char), intent(in) :: fill
    integer :: status
    status = def_var_fill_r8(ncid, varid, no_fill, real(fill, c_double))
  end function def_var_fill_i1

  ! ---- inquiries ----
  function nf90_inq_varid(ncid, name, varid) result(status)
    integer, intent(in) :: ncid
    character(len=*), intent(in) :: name
    integer, intent(out) :: varid
    integer :: status
    integer(c_int) :: id
    id = -1
    status = cdf_inq_varid(int(ncid, c_int), cstr(name), id)
    varid = id + 1
  end function nf90_inq_varid

  function nf90_inq_dimid(ncid, name, dimid) result(status)
    integer, intent(in) :: ncid
    character(len=*), intent(in) :: name
    integer, intent(out) :: dimid
    integer :: status
    integer(c_int) :: id
    id = -1
    status = cdf_inq_dimid(int(ncid, c_int), cstr(name), id)
    dimid = id + 1
  end function nf90_inq_dimid

  subroutine copy_name(cname, name)
    character(kind=c_char), intent(in) :: cname(:)
    character(len=*), intent(out) :: name
    integer :: i
    name = ' '
    do i = 1, min(len(name), size(cname))
      if (cname(i) == c_null_char) exit
      name(i:i) = cname(i)
    end do
  end subroutine copy_name

  function nf90_inquire_variable(ncid, varid, name, xtype, ndims, dimids, natts) result(status)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(out), optional :: name
    integer, intent(out), optional :: xtype, ndims, natts
    integer, intent(out), optional :: dimids(:)
    integer :: status, k
    integer(c_int) :: xt, nd, na, cd(nf90_max_var_dims)
    character(kind=c_char) :: nm(nf90_max_name + 1)
    status = cdf_inq_var(int(ncid, c_int), cvar(varid), nm, xt, nd, cd, na)
    if (status /= nf90_noerr) return
    if (present(name)) call copy_name(nm, name)
    if (present(xtype)) xtype = xt
    if (present(ndims)) ndims = nd
    if (present(natts)) natts = na
    if (present(dimids)) then
      do k = 1, min(int(nd), size(dimids))
        dimids(k) = cd(nd + 1 - k) + 1
      end do
    end if
  end function nf90_inquire_variable

  function nf90_inquire_dimension(ncid, dimid, name, len) result(status)
    integer, intent(in) :: ncid, dimid
    character(len=*), intent(out), optional :: name
    integer, intent(out), optional :: len
    integer :: status
    integer(c_int64_t) :: n
    character(kind=c_char) :: nm(nf90_max_name + 1)
    status = cdf_inq_dim(int(ncid, c_int), int(dimid - 1, c_int), nm, n)
    if (status /= nf90_noerr) return
    if (present(name)) call copy_name(nm, name)
    if (present(len)) len = int(n)
  end function nf90_inquire_dimension

  function nf90_inquire_attribute(ncid, varid, name, xtype, len, attnum) result(status)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    integer, intent(out), optional :: xtype, len, attnum
    integer :: status
    integer(c_int) :: xt
    integer(c_int64_t) :: n
    status = cdf_inq_att(int(ncid, c_int), cvar(varid), cstr(name), xt, n)
    if (status /= nf90_noerr) return
    if (present(xtype)) xtype = xt
    if (present(len)) len = int(n)
    if (present(attnum)) attnum = 0
  end function nf90_inquire_attribute

  ! ---- attributes ----
  function put_att_text(ncid, varid, name, values) result(status)
    ! as netcdf-fortran: trailing blanks of the value are not stored
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name, values
    integer :: status, i
    character(kind=c_char), target :: buf(max(1, len_trim(values)))
    do i = 1, len_trim(values)
      buf(i) = values(i:i)
    end do
    status = cdf_put_att(int(ncid, c_int), cvar(varid), cstr(name), nf90_char, &
         &               int(len_trim(values), c_int64_t), c_loc(buf))
  end function put_att_text

  function put_att_r8(ncid, varid, name, values) result(status)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    real(c_double), intent(in), target :: values
    integer :: status
    status = cdf_put_att(int(ncid, c_int), cvar(varid), cstr(name), 6, 1_c_int64_t, c_loc(values))
  end function put_att_r8

  function put_att_r4(ncid, varid, name, values) result(status)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    real(c_float), intent(in), target :: values
    integer :: status
    status = cdf_put_att(int(ncid, c_int), cvar(varid), cstr(name), 5, 1_c_int64_t, c_loc(values))
  end function put_att_r4

  function put_att_i4(ncid, varid, name, values) result(status)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    integer(c_int), intent(in), target :: values
    integer :: status
    status = cdf_put_att(int(ncid, c_int), cvar(varid), cstr(name), 4, 1_c_int64_t, c_loc(values))
  end function put_att_i4

  function put_att_i2(ncid, varid, name, values) result(status)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    integer(c_short), intent(in), target :: values
    integer :: status
    status = cdf_put_att(int(ncid, c_int), cvar(varid), cstr(name), 3, 1_c_int64_t, c_loc(values))
  end function put_att_i2

  function put_att_i1(ncid, varid, name, values) result(status)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    integer(c_signed_char), intent(in), target :: values
    integer :: status
    status = cdf_put_att(int(ncid, c_int), cvar(varid), cstr(name), 1, 1_c_int64_t, c_loc(values))
  end function put_att_i1

  function put_att_r8_vec(ncid, varid, name, values) result(status)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    real(c_double), intent(in), target, contiguous :: values(:)
    integer :: status
    status = cdf_put_att(int(ncid, c_int), cvar(varid), cstr(name), 6, &
         &               int(size(values), c_int64_t), c_loc(values))
  end function put_att_r8_vec

  function put_att_r4_vec(ncid, varid, name, values) result(status)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    real(c_float), intent(in), target, contiguous :: values(:)
    integer :: status
    status = cdf_put_att(int(ncid, c_int), cvar(varid), cstr(name), 5, &
         &               int(size(values), c_int64_t), c_loc(values))
  end function put_att_r4_vec

  function put_att_i4_vec(ncid, varid, name, values) result(status)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    integer(c_int), intent(in), target, contiguous :: values(:)
    integer :: status
    status = cdf_put_att(int(ncid, c_int), cvar(varid), cstr(name), 4, &
         &               int(size(values), c_int64_t), c_loc(values))
  end function put_att_i4_vec

  function get_att_text(ncid, varid, name, values) result(status)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    character(len=*), intent(out) :: values
    integer :: status, i
    character(kind=c_char), target :: buf(max(1, len(values)))
    buf = ' '
    status = cdf_get_att(int(ncid, c_int), cvar(varid), cstr(name), nf90_char, c_loc(buf), &
         &               int(len(values), c_int64_t))
    values = ' '
    if (status /= nf90_noerr) return
    do i = 1, len(values)
      if (buf(i) == c_null_char) exit
      values(i:i) = buf(i)
    end do
  end function get_att_text

  function get_att_r8(ncid, varid, name, values) result(status)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    real(c_double), intent(out), target :: values
    integer :: status
    status = cdf_get_att(int(ncid, c_int), cvar(varid), cstr(name), 6, c_loc(values), 1_c_int64_t)
  end function get_att_r8

  function get_att_r4(ncid, varid, name, values) result(status)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    real(c_float), intent(out), target :: values
    integer :: status
    status = cdf_get_att(int(ncid, c_int), cvar(varid), cstr(name), 5, c_loc(values), 1_c_int64_t)
  end function get_att_r4

  function get_att_i4(ncid, varid, name, values) result(status)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    integer(c_int), intent(out), target :: values
    integer :: status
    status = cdf_get_att(int(ncid, c_int), cvar(varid), cstr(name), 4, c_loc(values), 1_c_int64_t)
  end function get_att_i4

  function get_att_r8_vec(ncid, varid, name, values) result(status)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    real(c_double), intent(out), target, contiguous :: values(:)
    integer :: status
    status = cdf_get_att(int(ncid, c_int), cvar(varid), cstr(name), 6, c_loc(values), &
         &               int(size(values), c_int64_t))
  end function get_att_r8_vec

  function get_att_i4_vec(ncid, varid, name, values) result(status)
    integer, intent(in) :: ncid, varid
    character(len=*), intent(in) :: name
    integer(c_int), intent(out), target, contiguous :: values(:)
    integer :: status
    status = cdf_get_att(int(ncid, c_int), cvar(varid), cstr(name), 4, c_loc(values), &
         &               int(size(values), c_int64_t))
  end function get_att_i4_vec

  ! ---- variables ----
  ! One hyperslab transfer.  shp = shape of the caller's array ([] for a
  ! scalar); start / count in Fortran order, defaults as in the netCDF F90
  ! interface: start = 1, count = shp padded with ones.
  function var_io(ncid, varid, shp, memtype, ptr, writing, start, count) result(status)
    integer, intent(in) :: ncid, varid, shp(:), memtype
    type(c_ptr), intent(in) :: ptr
    logical, intent(in) :: writing
    integer, intent(in), optional :: start(:), count(:)
    integer :: status, k, nd
    integer(c_int) :: xt, cnd, na, cd(nf90_max_var_dims)
    character(kind=c_char) :: nm(nf90_max_name + 1)
    integer(c_int64_t) :: fstart(nf90_max_var_dims), fcount(nf90_max_var_dims)
    integer(c_int64_t) :: cstart(nf90_max_var_dims), ccount(nf90_max_var_dims)
    status = cdf_inq_var(int(ncid, c_int), cvar(varid), nm, xt, cnd, cd, na)
    if (status /= nf90_noerr) return
    nd = cnd
    fstart = 1
    fcount = 1
    if (present(start)) then
      if (size(start) > nd) then
        status = nf90_einvalcoords
        return
      end if
      fstart(1:size(start)) = start
    end if
    if (present(count)) then
      if (size(count) > nd) then
        status = nf90_einvalcoords
        return
      end if
      fcount(1:size(count)) = count
    else
      if (size(shp) > nd) then
        ! an array of higher rank than the variable: legal only when the
        ! extra (outer) extents are 1
        if (any(shp(nd+1:) /= 1)) then
          status = nf90_eedge
          return
        end if
        fcount(1:nd) = shp(1:nd)
      else
        fcount(1:size(shp)) = shp
      end if
    end if
    do k = 1, nd
      cstart(k) = fstart(nd + 1 - k) - 1
      ccount(k) = fcount(nd + 1 - k)
    end do
    if (writing) then
      status = cdf_put_vara(int(ncid, c_int), cvar(varid), cstart, ccount, int(memtype, c_int), ptr)
    else
      status = cdf_get_vara(int(ncid, c_int), cvar(varid), cstart, ccount, int(memtype, c_int), ptr)
    end if
  end function var_io

  function get_var_r8_0d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_double), intent(out), target :: values
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, [integer ::], 6, c_loc(values), .false., start, count)
  end function get_var_r8_0d

  function put_var_r8_0d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_double), intent(in), target :: values
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, [integer ::], 6, c_loc(values), .true., start, count)
  end function put_var_r8_0d

  function get_var_r8_1d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_double), intent(out), target, contiguous :: values(:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 6, c_loc(values), .false., start, count)
  end function get_var_r8_1d

  function put_var_r8_1d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_double), intent(in), target, contiguous :: values(:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 6, c_loc(values), .true., start, count)
  end function put_var_r8_1d

  function get_var_r8_2d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_double), intent(out), target, contiguous :: values(:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 6, c_loc(values), .false., start, count)
  end function get_var_r8_2d

  function put_var_r8_2d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_double), intent(in), target, contiguous :: values(:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 6, c_loc(values), .true., start, count)
  end function put_var_r8_2d

  function get_var_r8_3d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_double), intent(out), target, contiguous :: values(:,:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 6, c_loc(values), .false., start, count)
  end function get_var_r8_3d

  function put_var_r8_3d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_double), intent(in), target, contiguous :: values(:,:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 6, c_loc(values), .true., start, count)
  end function put_var_r8_3d

  function get_var_r8_4d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_double), intent(out), target, contiguous :: values(:,:,:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 6, c_loc(values), .false., start, count)
  end function get_var_r8_4d

  function put_var_r8_4d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_double), intent(in), target, contiguous :: values(:,:,:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 6, c_loc(values), .true., start, count)
  end function put_var_r8_4d

  function get_var_r8_5d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_double), intent(out), target, contiguous :: values(:,:,:,:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 6, c_loc(values), .false., start, count)
  end function get_var_r8_5d

  function put_var_r8_5d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_double), intent(in), target, contiguous :: values(:,:,:,:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 6, c_loc(values), .true., start, count)
  end function put_var_r8_5d

  function get_var_r4_0d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_float), intent(out), target :: values
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, [integer ::], 5, c_loc(values), .false., start, count)
  end function get_var_r4_0d

  function put_var_r4_0d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_float), intent(in), target :: values
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, [integer ::], 5, c_loc(values), .true., start, count)
  end function put_var_r4_0d

  function get_var_r4_1d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_float), intent(out), target, contiguous :: values(:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 5, c_loc(values), .false., start, count)
  end function get_var_r4_1d

  function put_var_r4_1d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_float), intent(in), target, contiguous :: values(:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 5, c_loc(values), .true., start, count)
  end function put_var_r4_1d

  function get_var_r4_2d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_float), intent(out), target, contiguous :: values(:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 5, c_loc(values), .false., start, count)
  end function get_var_r4_2d

  function put_var_r4_2d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_float), intent(in), target, contiguous :: values(:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 5, c_loc(values), .true., start, count)
  end function put_var_r4_2d

  function get_var_r4_3d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_float), intent(out), target, contiguous :: values(:,:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 5, c_loc(values), .false., start, count)
  end function get_var_r4_3d

  function put_var_r4_3d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_float), intent(in), target, contiguous :: values(:,:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 5, c_loc(values), .true., start, count)
  end function put_var_r4_3d

  function get_var_r4_4d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_float), intent(out), target, contiguous :: values(:,:,:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 5, c_loc(values), .false., start, count)
  end function get_var_r4_4d

  function put_var_r4_4d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    real(c_float), intent(in), target, contiguous :: values(:,:,:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 5, c_loc(values), .true., start, count)
  end function put_var_r4_4d

  function get_var_i4_0d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_int), intent(out), target :: values
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, [integer ::], 4, c_loc(values), .false., start, count)
  end function get_var_i4_0d

  function put_var_i4_0d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_int), intent(in), target :: values
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, [integer ::], 4, c_loc(values), .true., start, count)
  end function put_var_i4_0d

  function get_var_i4_1d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_int), intent(out), target, contiguous :: values(:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 4, c_loc(values), .false., start, count)
  end function get_var_i4_1d

  function put_var_i4_1d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_int), intent(in), target, contiguous :: values(:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 4, c_loc(values), .true., start, count)
  end function put_var_i4_1d

  function get_var_i4_2d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_int), intent(out), target, contiguous :: values(:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 4, c_loc(values), .false., start, count)
  end function get_var_i4_2d

  function put_var_i4_2d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_int), intent(in), target, contiguous :: values(:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 4, c_loc(values), .true., start, count)
  end function put_var_i4_2d

  function get_var_i4_3d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_int), intent(out), target, contiguous :: values(:,:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 4, c_loc(values), .false., start, count)
  end function get_var_i4_3d

  function put_var_i4_3d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_int), intent(in), target, contiguous :: values(:,:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 4, c_loc(values), .true., start, count)
  end function put_var_i4_3d

  function get_var_i4_4d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_int), intent(out), target, contiguous :: values(:,:,:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 4, c_loc(values), .false., start, count)
  end function get_var_i4_4d

  function put_var_i4_4d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_int), intent(in), target, contiguous :: values(:,:,:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 4, c_loc(values), .true., start, count)
  end function put_var_i4_4d

  function get_var_i2_0d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_short), intent(out), target :: values
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, [integer ::], 3, c_loc(values), .false., start, count)
  end function get_var_i2_0d

  function put_var_i2_0d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_short), intent(in), target :: values
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, [integer ::], 3, c_loc(values), .true., start, count)
  end function put_var_i2_0d

  function get_var_i2_1d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_short), intent(out), target, contiguous :: values(:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 3, c_loc(values), .false., start, count)
  end function get_var_i2_1d

  function put_var_i2_1d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_short), intent(in), target, contiguous :: values(:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 3, c_loc(values), .true., start, count)
  end function put_var_i2_1d

  function get_var_i2_2d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_short), intent(out), target, contiguous :: values(:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 3, c_loc(values), .false., start, count)
  end function get_var_i2_2d

  function put_var_i2_2d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_short), intent(in), target, contiguous :: values(:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 3, c_loc(values), .true., start, count)
  end function put_var_i2_2d

  function get_var_i1_0d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_signed_char), intent(out), target :: values
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, [integer ::], 1, c_loc(values), .false., start, count)
  end function get_var_i1_0d

  function put_var_i1_0d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_signed_char), intent(in), target :: values
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, [integer ::], 1, c_loc(values), .true., start, count)
  end function put_var_i1_0d

  function get_var_i1_1d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_signed_char), intent(out), target, contiguous :: values(:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 1, c_loc(values), .false., start, count)
  end function get_var_i1_1d

  function put_var_i1_1d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_signed_char), intent(in), target, contiguous :: values(:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 1, c_loc(values), .true., start, count)
  end function put_var_i1_1d

  function get_var_i1_2d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_signed_char), intent(out), target, contiguous :: values(:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 1, c_loc(values), .false., start, count)
  end function get_var_i1_2d

  function put_var_i1_2d(ncid, varid, values, start, count) result(status)
    integer, intent(in) :: ncid, varid
    integer(c_signed_char), intent(in), target, contiguous :: values(:,:)
    integer, intent(in), optional :: start(:), count(:)
    integer :: status
    status = var_io(ncid, varid, shape(values), 1, c_loc(values), .true., start, count)
  end function put_var_i1_2d

end module netcdf
