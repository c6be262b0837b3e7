! nf90_roundtrip.F90 - exercises this repository's `module netcdf` stand-in
! through the Fortran 90 interface the way utilities/easy_netcdf.F90 of the
! reference does (every generic it uses, with and without start/count), writes
! argv(1), reads it back and checks every value.  The Python side
! (tests/test_netcdf_shim.py) then reads the same file with scipy.
program nf90_roundtrip
  use netcdf
  implicit none
  integer, parameter :: dp = kind(1.0d0), sp = kind(1.0)
  character(len=512) :: path
  integer :: ncid, st, d_col, d_lay, d_band, d_time, v3, v2, vs, vi, v0, vr, vf
  integer :: i, j, k, n, xtype, ndims, natts, dimids(nf90_max_var_dims), alen
  real(dp) :: a3(2,3,4), b3(2,3,4), a2(3,4), b2(3,4), slab(2,3), s0, r0, vec(3), fillv
  real(dp) :: rec(3,2), recb(3,2)
  real(sp) :: f4
  integer :: ivec(4), jvec(4), shortvals(4)
  character(len=64) :: text, name
  integer :: nfail

  nfail = 0
  call get_command_argument(1, path)
  do k = 1, 4
    do j = 1, 3
      do i = 1, 2
        a3(i,j,k) = 100*k + 10*j + i + 0.5_dp
      end do
      a2(j,k) = 10*k + j + 0.25_dp
    end do
    ivec(k) = 1000 * k
    shortvals(k) = k - 2
  end do

  call ok(nf90_create(trim(path), ior(nf90_clobber, nf90_hdf5), ncid), "create")
  call ok(nf90_def_dim(ncid, "column", 4, d_col), "def_dim column")
  call ok(nf90_def_dim(ncid, "layer", 3, d_lay), "def_dim layer")
  call ok(nf90_def_dim(ncid, "band", 2, d_band), "def_dim band")
  call ok(nf90_def_dim(ncid, "time", nf90_unlimited, d_time), "def_dim time")
  call ok(nf90_def_var(ncid, "a3", nf90_double, [d_band, d_lay, d_col], v3), "def_var a3")
  call ok(nf90_def_var(ncid, "a2", nf90_float, [d_lay, d_col], v2), "def_var a2")
  call ok(nf90_def_var(ncid, "surface_type", nf90_short, [d_col], vs), "def_var short")
  call ok(nf90_def_var(ncid, "ivec", nf90_int, [d_col], vi), "def_var int")
  call ok(nf90_def_var(ncid, "s0", nf90_double, dimids(1:0), v0), "def_var scalar")
  call ok(nf90_def_var(ncid, "rec", nf90_double, [d_lay, d_time], vr), "def_var record")
  call ok(nf90_def_var(ncid, "filled", nf90_float, [d_lay, d_col], vf), "def_var filled")
  call ok(nf90_put_att(ncid, v3, "long_name", "Three-dimensional test array      "), "put_att text")
  call ok(nf90_put_att(ncid, v3, "units", "W m-2"), "put_att units")
  call ok(nf90_put_att(ncid, v2, "_FillValue", real(-9999.0_dp, 4)), "put_att _FillValue r4")
  call ok(nf90_put_att(ncid, v3, "scale", 2.5_dp), "put_att r8")
  call ok(nf90_put_att(ncid, vi, "offset", 7), "put_att i4")
  call ok(nf90_put_att(ncid, vs, "_FillValue", int(-1, 2)), "put_att i2")
  call ok(nf90_def_var_fill(ncid, vf, 0, real(-9999.0_dp, 4)), "def_var_fill")
  call ok(nf90_put_att(ncid, nf90_global, "title", "round trip"), "put_att global")
  call ok(nf90_enddef(ncid), "enddef")

  call ok(nf90_put_var(ncid, v3, a3), "put a3")
  call ok(nf90_put_var(ncid, v2, a2(:,1:2), start=[1,1], count=[3,2]), "put a2 first half")
  call ok(nf90_put_var(ncid, v2, a2(:,3:4), start=[1,3], count=[3,2]), "put a2 second half")
  call ok(nf90_put_var(ncid, vs, shortvals), "put short from default integer")
  call ok(nf90_put_var(ncid, vi, ivec), "put ivec")
  call ok(nf90_put_var(ncid, v0, 42.125_dp), "put scalar")
  call ok(nf90_put_var(ncid, vi, 77, start=[3]), "put scalar element")
  ivec(3) = 77
  rec(:,1) = [1.0_dp, 2.0_dp, 3.0_dp]
  rec(:,2) = [4.0_dp, 5.0_dp, 6.0_dp]
  call ok(nf90_put_var(ncid, vr, rec(:,2), start=[1,2], count=[3,1]), "put record 2")
  call ok(nf90_put_var(ncid, vr, rec(:,1), start=[1,1], count=[3,1]), "put record 1")
  call ok(nf90_put_var(ncid, vf, a2(:,2), start=[1,2], count=[3,1]), "put one column of filled")
  call ok(nf90_close(ncid), "close")

  ! ---- read back ----
  call ok(nf90_open(trim(path), nf90_nowrite, ncid), "open")
  call expect(nf90_inq_varid(ncid, "nonexistent", n) == nf90_enotvar, "missing variable -> NF90_ENOTVAR")
  call ok(nf90_inq_varid(ncid, "a3", v3), "inq_varid a3")
  call ok(nf90_inquire_variable(ncid, v3, name=name, xtype=xtype, ndims=ndims, dimids=dimids, natts=natts), "inquire_variable")
  call expect(trim(name) == "a3" .and. xtype == nf90_double .and. ndims == 3 .and. natts == 3, "a3 metadata")
  call ok(nf90_inquire_dimension(ncid, dimids(1), name=name, len=n), "inquire_dimension 1")
  call expect(trim(name) == "band" .and. n == 2, "fastest dimension is band")
  call ok(nf90_inquire_dimension(ncid, dimids(3), len=n), "inquire_dimension 3")
  call expect(n == 4, "slowest dimension is column")
  call ok(nf90_inq_dimid(ncid, "time", d_time), "inq_dimid time")
  call ok(nf90_inquire_dimension(ncid, d_time, len=n), "inquire record dimension")
  call expect(n == 2, "two records")
  call ok(nf90_get_var(ncid, v3, b3), "get a3")
  call expect(all(b3 == a3), "a3 values")
  call ok(nf90_get_var(ncid, v3, slab, start=[1,1,3], count=[2,3,1]), "get slab")
  call expect(all(slab == a3(:,:,3)), "a3 slab values")
  call ok(nf90_get_var(ncid, v3, s0, start=[2,3,4]), "get element")
  call expect(s0 == a3(2,3,4), "a3 element")
  call ok(nf90_get_var(ncid, v3, vec, start=[2,1,2], count=[1,3,1]), "get strided-by-count vector")
  call expect(all(vec == a3(2,:,2)), "a3 vector along the middle dimension")
  call ok(nf90_inq_varid(ncid, "a2", v2), "inq_varid a2")
  call ok(nf90_get_var(ncid, v2, b2), "get a2 (float -> double)")
  call expect(all(b2 == real(real(a2, sp), dp)), "a2 values (float32 rounding only)")
  call ok(nf90_inq_varid(ncid, "surface_type", vs), "inq_varid short")
  call ok(nf90_get_var(ncid, vs, jvec), "get short -> integer")
  call expect(all(jvec == shortvals), "short values")
  call ok(nf90_inq_varid(ncid, "ivec", vi), "inq_varid ivec")
  call ok(nf90_get_var(ncid, vi, jvec), "get ivec")
  call expect(all(jvec == ivec), "ivec values")
  call ok(nf90_inq_varid(ncid, "s0", v0), "inq_varid s0")
  call ok(nf90_get_var(ncid, v0, r0), "get scalar")
  call expect(r0 == 42.125_dp, "scalar value")
  call ok(nf90_inq_varid(ncid, "rec", vr), "inq_varid rec")
  call ok(nf90_get_var(ncid, vr, recb), "get record variable")
  call expect(all(recb == rec), "record values")
  call ok(nf90_inq_varid(ncid, "filled", vf), "inq_varid filled")
  call ok(nf90_get_var(ncid, vf, b2), "get filled")
  call expect(all(b2(:,1) == -9999.0_dp) .and. all(b2(:,2) == real(real(a2(:,2), sp), dp)) &
       &      .and. all(b2(:,3:4) == -9999.0_dp), "unwritten elements hold _FillValue")
  call ok(nf90_inquire_attribute(ncid, v3, "long_name", len=alen), "inquire_attribute")
  call expect(alen == 28, "trailing blanks of a text attribute are not stored")
  call ok(nf90_get_att(ncid, v3, "long_name", text), "get_att text")
  call expect(trim(text) == "Three-dimensional test array", "text attribute")
  call ok(nf90_get_att(ncid, v3, "scale", s0), "get_att r8")
  call expect(s0 == 2.5_dp, "real attribute")
  call ok(nf90_get_att(ncid, v2, "_FillValue", fillv), "get_att r4 -> r8")
  call expect(fillv == -9999.0_dp, "float attribute")
  call ok(nf90_get_att(ncid, nf90_global, "title", text), "get_att global")
  call expect(trim(text) == "round trip", "global attribute")
  call expect(nf90_inquire_attribute(ncid, v3, "absent", len=alen) /= nf90_noerr, "missing attribute is an error")
  call expect(nf90_get_var(ncid, v3, slab, start=[1,1,5], count=[2,3,1]) /= nf90_noerr, "out-of-range start is an error")
  call expect(len_trim(nf90_strerror(nf90_enotvar)) > 0, "strerror")
  call ok(nf90_close(ncid), "close")
  f4 = 0.0_sp

  if (nfail == 0) then
    print '(a)', "nf90_roundtrip: OK"
  else
    print '(a,i0,a)', "nf90_roundtrip: ", nfail, " FAILURES"
    stop 1
  end if

contains

  subroutine ok(status, what)
    integer, intent(in) :: status
    character(len=*), intent(in) :: what
    if (status /= nf90_noerr) then
      print '(a,a,a,a)', "FAIL ", what, ": ", trim(nf90_strerror(status))
      nfail = nfail + 1
    end if
  end subroutine ok

  subroutine expect(cond, what)
    logical, intent(in) :: cond
    character(len=*), intent(in) :: what
    if (.not. cond) then
      print '(a,a)', "FAIL ", what
      nfail = nfail + 1
    end if
  end subroutine expect

end program nf90_roundtrip
