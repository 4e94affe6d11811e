// TEST HARNESS (not product code): every form of the source generator (tree_lane_gen.hpp) over one robot description, for a build with
// -fsanitize=address,undefined (tests/test_generator_sanitizers.py).  Returns the number of forms generated.
#include <string>

#include "tree_lane_gen.hpp"

extern "C" int gen_all_forms(const rb_robot_desc *d) {
    std::string err;
    int ok = 0;
    { rblg::Generated g; ok += rblg::generate(d, true, g, err) == 0; }
    { rblg::Generated g; ok += rblg::generate(d, false, g, err, false) == 0; }
    for (int helpers = 0; helpers <= 3; ++helpers)
        for (int share : {45, 80, 100})
            for (int two = 0; two < 2; ++two)
                for (int st = 0; st < 2; ++st) { rblg::SplitGenerated g; ok += rblg::generate_split(d, 4, g, err, helpers, share, two != 0, st != 0) == 0; }
    for (int cuts = 1; cuts <= 3; ++cuts)
        for (int share : {0, 40, 100}) { rblg::SplitGenerated g; ok += rblg::generate_split_cut(d, 4, g, err, cuts, share) == 0; }
    for (int parts = 2; parts <= 6; ++parts) { rblg::SplitGenerated g; ok += rblg::generate_split(d, parts, g, err, 2, 70, true, true) == 0; }
    return ok;
}
