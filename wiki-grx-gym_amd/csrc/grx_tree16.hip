// grx_tree16.hip -- the tree step kernel (grx_tree.h) with a group of SIXTEEN lanes per env: the same source as grx_kernels.hip's
// instance, compiled a second time with GRX_TREE_GDEV = 16 (four envs per wave).  The three passes of the articulated-body algorithm are
// bound by the tree's depth levels whatever the group size; what 16 lanes buy is TWICE THE WAVES -- grx_capi.cpp picks this kernel while
// those waves still have a SIMD each (4096 envs of the 32-DOF body on an MI355X, BASELINE.json config 5's per-GPU size: the 8-lane
// kernel leaves half the SIMDs idle there) -- and half the rounds of everything that goes round the group's lanes: the terrain
// contacts' work list, the self-collision's link pairs, the height scan, the link frames.
#define GRX_TREE16_TU
#define GRX_TREE_GDEV 16
#define grx_step_tree grx_step_tree16   // (a name of its own in the profiles)
#define grx_step_tree_trimesh grx_step_tree16_trimesh
#include "grx_kernels.hip"
