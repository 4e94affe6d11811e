// tree_lane_split2.hpp - host interface of the second translation unit (roboy_sim_split2.hip): the LEAN two-part instances of the
// split joint-tree kernels (tree_lane_split.hpp, RBL_LEAN = 1) for the committed upper body, compiled ahead of time from
// tree_lane_split2_baked.hpp.  A translation unit of their own because the generated header defines the same macros (RBL_NPARTS,
// RBL_PART_LDS, ...) as the five-wave form's, with other values.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace rbe { struct TreeEnvArgs; }

namespace rbs2 {
uint64_t text_hash();          // FNV-1a of the generated text the instances were compiled from (the library regenerates and compares)
int n_parts();
size_t lds_bytes();            // dynamic LDS of a workgroup (n_parts waves): two workgroups fit a CU
int n_q();
int n_t();
// one workgroup per 64 envs; integrator 0 = Euler, 1 = RK4
void launch_step(int integrator, unsigned groups, hipStream_t stream, float *q, float *qd, uint32_t *feas, const float *act,
                 float act_scale, float h, int nsub, long n);
void launch_env_step(int integrator, unsigned groups, hipStream_t stream, const rbe::TreeEnvArgs &args);
}  // namespace rbs2
