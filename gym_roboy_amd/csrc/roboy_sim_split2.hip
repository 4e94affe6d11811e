// roboy_sim_split2.hip - second translation unit of libroboy_sim.so: the lean two-part split kernels of the committed upper body
// (tree_lane_split.hpp with RBL_LEAN = 1 around tree_lane_split2_baked.hpp; interface: tree_lane_split2.hpp).  For batches between
// "one five-wave workgroup per CU" (16 384 envs) and "a wave on every SIMD" (65 536): two part waves per 64 envs, two workgroups per
// CU, every wave alone on its SIMD - 32 768 envs in one generation (roboy_sim.hip: tree_wants_split2).
#include <hip/hip_runtime.h>

#include "tree_lane_split2.hpp"
#include "tree_lane_defs.hpp"
#define RBL_NS rbl_split2_baked
#define RBL_LEAN 1
#include "tree_lane_split2_baked.hpp"
#include "tree_lane_split.hpp"
#undef RBL_NS

static_assert(rbl_split2_baked::SP_LDS_BYTES <= 80 * 1024, "the lean layout must admit two workgroups per CU (160 KB of LDS)");

namespace rbs2 {

uint64_t text_hash() { return RBL_SPLIT_TEXT_HASH; }
int n_parts() { return RBL_NPARTS; }
size_t lds_bytes() { return size_t(rbl_split2_baked::SP_LDS_BYTES); }
int n_q() { return RBL_NQ; }
int n_t() { return RBL_NT; }

namespace {
// more than 64 KB of dynamic LDS would have to be granted; the lean layout stays below (asserted where it is launched from)
constexpr unsigned THREADS = 64u * unsigned(rbl_split2_baked::SP_NWAVES);
}

void launch_step(int integrator, unsigned groups, hipStream_t stream, float *q, float *qd, uint32_t *feas, const float *act,
                 float act_scale, float h, int nsub, long n) {
    const size_t lds = lds_bytes();
    if (integrator == 0)
        hipLaunchKernelGGL(rbl_split2_baked::tree_split_step<0>, dim3(groups), dim3(THREADS), lds, stream, q, qd, feas, act, act_scale, h, nsub, n);
    else
        hipLaunchKernelGGL(rbl_split2_baked::tree_split_step<1>, dim3(groups), dim3(THREADS), lds, stream, q, qd, feas, act, act_scale, h, nsub, n);
}

void launch_env_step(int integrator, unsigned groups, hipStream_t stream, const rbe::TreeEnvArgs &args) {
    const size_t lds = lds_bytes();
    if (integrator == 0) hipLaunchKernelGGL(rbl_split2_baked::tree_split_env_step<0>, dim3(groups), dim3(THREADS), lds, stream, args);
    else hipLaunchKernelGGL(rbl_split2_baked::tree_split_env_step<1>, dim3(groups), dim3(THREADS), lds, stream, args);
}

}  // namespace rbs2
