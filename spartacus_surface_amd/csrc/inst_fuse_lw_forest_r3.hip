// Instantiation unit: the fused layer + upward kernel of the longwave, forest tiles, 3 region(s)
#define SS_INST_URBAN false
#define SS_INST_NREG 3
#define SS_INST_LW 1
#define SS_INST_FUSE_NAME launch_fuse_lw_forest_r3
#include "ss_inst_fuse_body.hpp"
