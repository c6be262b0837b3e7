// Instantiation unit: the fused layer + upward kernel of the longwave, urban tiles, 2 region(s)
#define SS_INST_URBAN true
#define SS_INST_NREG 2
#define SS_INST_LW 1
#define SS_INST_FUSE_NAME launch_fuse_lw_urban_r2
#include "ss_inst_fuse_body.hpp"
