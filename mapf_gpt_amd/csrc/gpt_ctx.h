// gpt_ctx.h -- policy context shared by the fp32 and the fast (16-bit MFMA) forward paths.
#pragma once
#include <vector>

#include "common.h"

struct LayerOff {
    size_t ln1, attn_w, proj_w, ln2, fc_w, proj2_w;
};

// GPTConfig.bias = True (model.py:14-17,29,31,79,81,115): LayerNorm and Linear bias vectors, offsets into mgpt_gpt::bias
struct BiasOff {
    size_t ln1, attn, proj, ln2, fc, proj2;
};


struct mgpt_gpt {
    int L, nh, C, hs, block, max_rows;
    size_t n_params = 0;
    size_t off_wte = 0, off_wpe = 0, off_lnf = 0;
    std::vector<LayerOff> layers;
    float *params = nullptr;          // fp32 master copy, device
    std::vector<uint8_t> is_set;      // per parameter tensor
    // bias = True checkpoints: allocated (zeroed) by the first *.bias tensor mgpt_gpt_set_param sees; carried by the exact-fp32 kernels only
    float *bias = nullptr;            // [ln_f C | per layer: ln_1 C, c_attn 3C, attn c_proj C, ln_2 C, c_fc 4C, mlp c_proj C], device
    size_t off_lnf_b = 0;
    std::vector<BiasOff> bias_layers;
    std::vector<uint8_t> bias_set;    // per bias tensor: [ln_f | 6 per layer]
    bool has_bias = false;
    bool finalized = false;
    // fp32-path workspace
    float *x = nullptr, *xn = nullptr, *qkv = nullptr, *hbuf = nullptr, *logits_tmp = nullptr;
    // fast-path state (gpt_fast.hip)
    void *fast = nullptr;
    float *embed_table = nullptr;     // C = 256: [256 positions][67 tokens][C] = wpe + wte, the rows layer 0's attention block starts from (gpt_fast.hip)
    uint64_t generation = 1;          // bumped when weight planes / workspaces are freed or rebuilt (common.h: gpt_generation)
    // precision envelope of the split-fp16 mode (gpt.hip: envelope_*; include/mapf_gpt_amd.h: mgpt_gpt_envelope)
    float env_max_w = 0.f, env_max_rms = 0.f;   // over the 2-D matrices of the blocks, computed by mgpt_gpt_finalize
    float env_probe_err = -1.f;                 // max |f16x3 - f32| over the probe rows' logits, both call regimes (-1: not probed)
    float env_probe_err_small = -1.f, env_probe_err_large = -1.f;   // ... by regime: the kernels of calls <= / > kSmallRows rows
    float env_probe_tol = 0.f, env_probe_max_logit = 0.f;           // the bar it was held to, and max |logit| of the fp32 path
    int env_policy = 0;                         // MGPT_ENVELOPE_FALLBACK / _REFUSE / _IGNORE
    int env_state = 0;                          // 0 not decided, 1 inside, 2 outside
    bool env_logged = false;
};


// calls of up to kSmallRows rows (one environment) run other kernels than larger ones (head-parallel attention, 32 x 32 x 16 MLP
// block, one-launch last layer): gpt_fast.hip, DESIGN.md section 3.2
constexpr int kSmallRows = 128;

// gpt_fast.hip: 16-bit-MFMA path (packed operand planes, own workspace)
int gpt_fast_finalize(mgpt_gpt *g);
void gpt_fast_destroy(mgpt_gpt *g);
// call_rows = rows of the whole C-ABI call this chunk belongs to: path choices that change the arithmetic (the head-parallel small-launch
// kernels) are made from it, so that every chunk of one call -- a ragged remainder included -- runs the same kernels
int gpt_fast_forward(mgpt_gpt *g, const uint8_t *d_tokens, int rows, float *d_logits, int precision, hipStream_t s, int call_rows);
int gpt_fast_debug_copy(mgpt_gpt *g, int precision, int which, void *d_out, int64_t nbytes, hipStream_t s);

// gpt.hip: final LayerNorm + tied lm_head on the last position of g->x (shared by both paths)
int gpt_launch_head(mgpt_gpt *g, int rows, float *d_logits, hipStream_t s);
int gpt_launch_head_at(mgpt_gpt *g, const float *xsrc, int64_t row_stride, int64_t row_offset, int rows, float *d_logits,
                       hipStream_t s);
