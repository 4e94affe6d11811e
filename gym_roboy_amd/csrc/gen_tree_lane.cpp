// gen_tree_lane.cpp - C entry of the lane-kernel source generator (tree_lane_gen.hpp) for the tools and the CPU
// tests: writes the generated header of a robot to `path`.  tools/gen_tree_lane_baked.py runs it on the committed
// upper body (tree_lane_baked.hpp, compiled into the library ahead of time); tests/test_tree_lane_gen.py runs it
// on random robots and checks the text, compiled with g++, against the fp64 oracle.  Built with g++.
#include <cstdio>
#include <cstdlib>
#include <string>

#include "tree_lane_gen.hpp"

// lds_c: bit 0 - park the velocity-product accelerations in LDS; bit 1 - do NOT write structurally identical subtrees as pair values
extern "C" int rb_gen_tree_lane(const rb_robot_desc *d, int lds_c, const char *path, int *lds_slots, int *n_stmt,
                                unsigned long long *hash, int *flops, int *max_live) {
    rblg::Generated g;
    std::string err;
    const int rc = rblg::generate(d, (lds_c & 1) != 0, g, err, (lds_c & 2) == 0);
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

// mates of the links and tendons (-1: none, -2: is some link's / tendon's mate), as generate() pairs them
extern "C" int rb_gen_tree_lane_mates(const rb_robot_desc *d, int *mate, int *tmate) {
    rblg::Robot rob;
    std::string err;
    if (int rc = rblg::build_robot(d, rob, err)) return rc;
    std::vector<int> m, tm;
    rblg::find_mates(rob, m, tm);
    for (int i = 0; i < rob.nq; ++i) mate[i] = m[i];
    for (int k = 0; k < rob.nt; ++k) tmate[k] = tm[k];
    return RB_OK;
}

// the split form (several waves per env group): writes the header, returns parts / slots / statement counts.
// max_helpers: helper waves at most (tendon helpers of the longest parts; 0: the three-wave form of round 3)
extern "C" int rb_gen_tree_lane_split_h(const rb_robot_desc *d, int max_parts, int max_helpers, const char *path, int *n_parts, int *part_lds,
                                        int *x_slots, int *max_stmt, int *n_stmt, int *part_of_joint, unsigned long long *hash,
                                        int *n_helpers, int *helper_stmt) {
    rblg::SplitGenerated g;
    std::string err;
    // max_helpers: low byte = helper waves at most; bits 8-15 = the helpers' share of their parts' tendons in percent (0: the default)
    // ... bit 16 = the backward pass in two sweeps (inertias before barrier T, bias forces behind it)
    const int share = (max_helpers >> 8) & 0xff;
    const bool two_sweeps = ((max_helpers >> 16) & 1) != 0;
    // ... bit 17 = the CUT form instead (generate_split_cut: max_helpers parts cut in two, share = the distal waves' share of the tendons)
    const bool cut = ((max_helpers >> 17) & 1) != 0;
    const bool share_trunk = ((max_helpers >> 18) & 1) != 0;        // bit 18: one part evaluates the trunk links' inertias for all
    const int rc = cut ? rblg::generate_split_cut(d, max_parts, g, err, max_helpers & 0xff, share)
                       : rblg::generate_split(d, max_parts, g, err, max_helpers & 0xff, share ? share : 45, two_sweeps, share_trunk);
    if (rc) { std::fprintf(stderr, "rb_gen_tree_lane_split: %s\n", err.c_str()); return rc; }
    FILE *f = std::fopen(path, "w");
    if (!f) return RB_EINVAL;
    std::fputs(g.text.c_str(), f);
    std::fprintf(f, "#define RBL_SPLIT_TEXT_HASH 0x%llxull\n", (unsigned long long)g.hash);
    std::fclose(f);
    if (n_parts) *n_parts = g.n_parts;
    if (part_lds) *part_lds = g.part_lds;
    if (x_slots) *x_slots = g.x_slots;
    if (max_stmt) *max_stmt = g.max_stmt;
    if (n_stmt) *n_stmt = g.n_stmt;
    if (part_of_joint) for (int i = 0; i < g.n_q; ++i) part_of_joint[i] = g.part_of_joint[i];
    if (hash) *hash = g.hash;
    if (n_helpers) *n_helpers = g.n_helpers;
    if (helper_stmt) *helper_stmt = g.helper_stmt;
    return RB_OK;
}
extern "C" int rb_gen_tree_lane_split(const rb_robot_desc *d, int max_parts, const char *path, int *n_parts, int *part_lds, int *x_slots,
                                      int *max_stmt, int *n_stmt, int *part_of_joint, unsigned long long *hash) {
    return rb_gen_tree_lane_split_h(d, max_parts, 0, path, n_parts, part_lds, x_slots, max_stmt, n_stmt, part_of_joint, hash, nullptr, nullptr);
}
