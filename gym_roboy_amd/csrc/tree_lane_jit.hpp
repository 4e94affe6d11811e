// tree_lane_jit.hpp - run-time build of the env-per-lane joint-tree kernels (tree_lane.hpp) for ONE robot: the text
// tree_lane_gen.hpp writes for it, compiled with hiprtc (loaded with dlopen: msj_jit.hpp).  One program per kernel
// (step / env step of the handle's integrator), built by the first call that needs it: ~10 s each.  Host code.
#pragma once
#include "msj_jit.hpp"
#include "tree_lane_gen.hpp"

namespace rblj {

struct Kernel {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    int state = 0;            // 0 = not tried yet, 1 = ready, -1 = not available
    std::string why;
};

// kind: 0 = tree_lane_step, 1 = tree_lane_env_step
inline bool build(const rblg::Generated &g, int kind, int integ, Kernel &out) {
    const std::string src = "#include \"tree_lane_defs.hpp\"\n#define RBL_NS rbl_jit\n" + g.text + "#include \"tree_lane.hpp\"\n";
    const std::string name = std::string(kind == 0 ? "rbl_jit::tree_lane_step<" : "rbl_jit::tree_lane_env_step<") + (integ ? "1>" : "0>");
    const char *names[1] = {name.c_str()};
    hipFunction_t *slots[1] = {&out.fn};
    out.state = rbj::compile_and_load(src, "roboy_tree_lane_jit.hip", names, 1, out.mod, slots, out.why) ? 1 : -1;
    return out.state == 1;
}

inline void unload(Kernel &k) {
    if (k.mod) (void)hipModuleUnload(k.mod);
    k = Kernel();
}

}  // namespace rblj
