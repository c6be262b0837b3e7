// Instantiation unit: the fused layer + upward kernel of the shortwave, urban tiles, 1 region(s)
#define SS_INST_URBAN true
#define SS_INST_NREG 1
#define SS_INST_LW 0
#define SS_INST_FUSE_NAME launch_fuse_sw_urban_r1
#include "ss_inst_fuse_body.hpp"
