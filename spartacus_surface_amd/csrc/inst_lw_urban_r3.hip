// Instantiation unit: lw solver kernels, urban tiles, 3 region(s), ns = 1, 2, 4, 8
#define SS_INST_URBAN true
#define SS_INST_NREG 3
#define SS_INST_LW 1
#define SS_INST_NAME launch_lw_urban_r3
#define SS_INST_FUSE_NAME launch_fuse_lw_urban_r3
#include "ss_inst_body.hpp"
