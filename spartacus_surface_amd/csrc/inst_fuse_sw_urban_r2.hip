// Instantiation unit: the fused layer + upward kernel of the shortwave, urban tiles, 2 region(s)
#define SS_INST_URBAN true
#define SS_INST_NREG 2
#define SS_INST_LW 0
#define SS_INST_FUSE_NAME launch_fuse_sw_urban_r2
#include "ss_inst_fuse_body.hpp"
