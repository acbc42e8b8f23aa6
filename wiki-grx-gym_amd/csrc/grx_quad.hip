// grx_quad.hip -- the fused step kernel with a lane QUAD per env (GRX_LPE = 4): the same source as grx_kernels.hip, compiled
// a second time with the lane <-> env mapping of grx_math.h switched.  Only the four-wave kernel is instantiated here
// (grx_launch_step_quad); grx_capi.cpp picks it while its 16-env blocks fit the device's CUs in one round.
#define GRX_LPE 4
#define GRX_QUAD_TU
#define grx_step_kernel grx_step_kernel_quad   // (a kernel of its own: the template's name is the symbol the HIP runtime registers)
#define grx_step_kernel_trimesh grx_step_kernel_quad_trimesh
#include "grx_kernels.hip"
