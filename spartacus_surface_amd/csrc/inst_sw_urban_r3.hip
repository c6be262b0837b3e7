// Instantiation unit: sw solver kernels, urban tiles, 3 region(s), ns = 1, 2, 4, 8
#define SS_INST_URBAN true
#define SS_INST_NREG 3
#define SS_INST_LW 0
#define SS_INST_NAME launch_sw_urban_r3
#define SS_INST_FUSE_NAME launch_fuse_sw_urban_r3
#include "ss_inst_body.hpp"
