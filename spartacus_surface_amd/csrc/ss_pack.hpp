// ss_pack.hpp - the per-layer inputs of the column kernels in the tiling of the
// scratch records.
//
// In the caller's arrays ((nspec, ntotlay), layers of a column contiguous) the
// value a lane needs for its layer sits 8 * nspec * nlay bytes from its
// neighbour's, and a column kernel comes back to the same 128-byte line once per
// layer, ~10 us apart: the downward kernels fetched every such line many times
// over (~10 KB per column of their 23-30 KB).  The plan therefore re-lays them
// out once per upload, [tile of 64 units][layer][element pair][lane][2] like the
// scratch records, and the column kernels read 16-byte pairs from there.
// The layer kernels read them from there too: the address of a pack record needs
// no look-up of the column's layer offset, so the inputs arrive with the first
// round trip of the wave instead of the third.
//
//   SW (kPackSw = 16): dz, veg_fraction | veg_scale, veg_ext | veg_fsd, building_fraction |
//                      building_scale, veg_contact_fraction | roof_albedo, roof_albedo_dir |
//                      wall_albedo, air_ext | air_ssa, veg_ssa | wall_specular_frac, cos_sza of the column
//                      (kPackNoLayer where the column has no such layer)
//   LW (kPackLw = 18): the same four geometry pairs | air_ext, air_ssa | veg_ssa, veg_planck |
//                      veg_air_planck, clear_air_planck | roof_emissivity, roof_emission |
//                      wall_emissivity, wall_emission
// with the same "not read, zero" rules as load_layer_geom / SwDownIn / LwLayerIn.
#pragma once

#include "ss_common.hpp"
#include "ss_layer_matrices.hpp"

namespace ss {

struct PackLaunch {
  Inputs in;
  const int32_t* cols;   // columns of the class (sorted as the solver launches see them)
  int ncu;               // units of the class = columns * nspec
  int nspec;
  int maxlay;            // largest nlay of the class
  int lw, urban, nreg;
  double* pack;
};

SS_HD void pack_unit(const PackLaunch& P, int cu, int jlay) {
  const Inputs& in = P.in;
  const int col = P.cols[cu / P.nspec];
  const int jb = cu % P.nspec;
  if (jlay >= in.nlay[col]) {
    // a layer slot the column does not have: marked in the last shortwave element, which otherwise carries the
    // column's cos_sza - so that the layer kernel needs no look-up through the column index before it starts
    if (!P.lw) rec_pair_store(P.pack + rec_base(jlay, P.maxlay, kPackSw, cu) + 14 * kRecStride, 0.0, kPackNoLayer);
    else rec_pair_store(P.pack + rec_base(jlay, P.maxlay, kPackLw, cu), kPackNoLayer, 0.0);   // longwave: in dz's place
    return;
  }
  const int lay = in.istartlay[col] - 1 + jlay;
  const long sidx = (long)jb + (long)P.nspec * lay;
  const bool veg = P.nreg > 1, urban = P.urban != 0;
  double* r = P.pack + rec_base(jlay, P.maxlay, P.lw ? kPackLw : kPackSw, cu);
  constexpr long st = kRecStride;
  rec_pair_store(r + 0 * st, in.dz[lay], veg ? in.veg_fraction[lay] : 0.0);
  rec_pair_store(r + 2 * st, veg ? in.veg_scale[lay] : 0.0, veg ? in.veg_ext[lay] : 0.0);
  rec_pair_store(r + 4 * st, (P.nreg == 3) ? in.veg_fsd[lay] : 0.0, urban ? in.building_fraction[lay] : 0.0);
  rec_pair_store(r + 6 * st, urban ? in.building_scale[lay] : 0.0,
                 (urban && veg) ? in.veg_contact_fraction[lay] : 0.0);
  if (!P.lw) {
    rec_pair_store(r + 8 * st, urban ? in.roof_albedo[sidx] : 0.0, urban ? in.roof_albedo_dir[sidx] : 0.0);
    rec_pair_store(r + 10 * st, urban ? in.wall_albedo[sidx] : 0.0, in.air_ext[sidx]);
    rec_pair_store(r + 12 * st, in.air_ssa[sidx], veg ? in.veg_ssa[sidx] : 0.0);
    rec_pair_store(r + 14 * st, urban ? in.wall_specular_frac[sidx] : 0.0, in.cos_sza ? in.cos_sza[col] : 0.0);
  } else {
    rec_pair_store(r + 8 * st, in.air_ext[sidx], in.air_ssa[sidx]);
    rec_pair_store(r + 10 * st, veg ? in.veg_ssa[sidx] : 0.0, veg ? in.veg_planck[sidx] : 0.0);
    rec_pair_store(r + 12 * st, veg ? in.veg_air_planck[sidx] : 0.0, in.clear_air_planck[sidx]);
    rec_pair_store(r + 14 * st, urban ? in.roof_emissivity[sidx] : 0.0, urban ? in.roof_emission[sidx] : 0.0);
    rec_pair_store(r + 16 * st, urban ? in.wall_emissivity[sidx] : 0.0, urban ? in.wall_emission[sidx] : 0.0);
  }
}

// The pack record of layer jlay of unit cu of a solver launch
SS_HD const double* pack_rec(const Launch& L, int jlay, int cu, int nelem) {
  return L.pack + rec_base(jlay, L.pack_maxlay, nelem, L.pack_cu0 + cu);
}

// veg_fraction, building_fraction of that layer (the overlap matrices need nothing else)
SS_HD void pack_fractions(const double* prec, double& vf, double& bf) {
  vf = rec_pair_load(prec + 0 * kRecStride).y;
  bf = rec_pair_load(prec + 4 * kRecStride).y;
}

}  // namespace ss
