// Instantiation unit: the fused layer + upward kernel of the shortwave, forest tiles, 2 region(s)
#define SS_INST_URBAN false
#define SS_INST_NREG 2
#define SS_INST_LW 0
#define SS_INST_FUSE_NAME launch_fuse_sw_forest_r2
#include "ss_inst_fuse_body.hpp"
