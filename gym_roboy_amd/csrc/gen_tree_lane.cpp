// gen_tree_lane.cpp - C entry of the lane-kernel source generator (tree_lane_gen.hpp) for the tools and the CPU
// tests: writes the generated header of a robot to `path`.  tools/gen_tree_lane_baked.py runs it on the committed
// upper body (tree_lane_baked.hpp, compiled into the library ahead of time); tests/test_tree_lane_gen.py runs it
// on random robots and checks the text, compiled with g++, against the fp64 oracle.  Built with g++.
#include <cstdio>
#include <string>

#include "tree_lane_gen.hpp"

extern "C" int rb_gen_tree_lane(const rb_robot_desc *d, int lds_c, const char *path, int *lds_slots, int *n_stmt,
                                unsigned long long *hash, int *flops, int *max_live) {
    rblg::Generated g;
    std::string err;
    const int rc = rblg::generate(d, lds_c != 0, g, err);
    if (rc) { std::fprintf(stderr, "rb_gen_tree_lane: %s\n", err.c_str()); return rc; }
    FILE *f = std::fopen(path, "w");
    if (!f) return RB_EINVAL;
    std::fputs(g.text.c_str(), f);
    // the hash of the text above: a handle whose robot generates the same text runs the instances compiled from this file
    std::fprintf(f, "#define RBL_TEXT_HASH 0x%llxull\n", (unsigned long long)g.hash);
    std::fclose(f);
    if (lds_slots) *lds_slots = g.lds_slots;
    if (n_stmt) *n_stmt = g.n_stmt;
    if (hash) *hash = g.hash;
    if (flops) *flops = g.flops;
    if (max_live) *max_live = g.max_live;
    return RB_OK;
}
