// gpt_ctx.h -- policy context shared by the fp32 and the fast (16-bit MFMA) forward paths.
#pragma once
#include <vector>

#include "common.h"

struct LayerOff {
    size_t ln1, attn_w, proj_w, ln2, fc_w, proj2_w;
};


struct mgpt_gpt {
    int L, nh, C, hs, block, max_rows;
    size_t n_params = 0;
    size_t off_wte = 0, off_wpe = 0, off_lnf = 0;
    std::vector<LayerOff> layers;
    float *params = nullptr;          // fp32 master copy, device
    std::vector<uint8_t> is_set;      // per parameter tensor
    bool finalized = false;
    // fp32-path workspace
    float *x = nullptr, *xn = nullptr, *qkv = nullptr, *hbuf = nullptr, *logits_tmp = nullptr;
    // fast-path state (gpt_fast.hip)
    void *fast = nullptr;
};

