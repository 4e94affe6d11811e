// TEST HARNESS (not product code): the host side of the joint-tree kernels (gym_roboy_amd/csrc/tree_build.hpp)
// compiled with g++, so that tests/test_tree_tables.py can check the tables the kernels run on without a GPU.
#include "../../gym_roboy_amd/csrc/tree_build.hpp"

extern "C" int tt_build(const rb_robot_desc *d, double step_size, int nsub, uint32_t *words, int max_words,
                        int *n_words, rbt::TreeDev *dev, int *waves, long *lds_bytes) {
    rbt::TreeHost h;
    std::string err;
    const int rc = rbt::tree_build(d, step_size, nsub, h, err);
    if (rc) return rc;
    if (int(h.words.size()) > max_words) return RB_ENOMEM;
    std::memcpy(words, h.words.data(), sizeof(uint32_t) * h.words.size());
    *n_words = int(h.words.size());
    *dev = h.dev;
    *waves = rbt::tree_pick_waves(h);
    *lds_bytes = long(rbt::tree_lds_bytes(h, *waves));
    return RB_OK;
}
extern "C" int tt_consts(int *out) {   // TREE_E, LS, REC1, REC5, XSLOT, TENDON_REC, CROSS_REC
    out[0] = rbt::TREE_E; out[1] = rbt::LS; out[2] = rbt::REC1; out[3] = rbt::REC5; out[4] = rbt::XSLOT;
    out[5] = rbt::TENDON_REC; out[6] = rbt::CROSS_REC;
    return 0;
}
