// Instantiation unit: lw solver kernels, forest tiles, 2 region(s), ns = 1, 2, 4, 8
#define SS_INST_URBAN false
#define SS_INST_NREG 2
#define SS_INST_LW 1
#define SS_INST_NAME launch_lw_forest_r2
#define SS_INST_FUSE_NAME launch_fuse_lw_forest_r2
#include "ss_inst_body.hpp"
