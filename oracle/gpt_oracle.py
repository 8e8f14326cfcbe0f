"""CPU restatement of the reference policy forward -- TEST INFRASTRUCTURE ONLY.

Functional PyTorch-CPU port of mapf_gpt/model.py (GPT.forward model.py:167-189, GPT.act :244-260),
written from the weight dict up (no nn.Module), used as
  * the fp32 / fp64 checker of the HIP forward (tests, smoke), and
  * bench.py's cpu_baseline leg (kind "port": same PyTorch CPU ops the reference executes).
PINNED: tests/test_gpt_oracle_golden.py checks it against logits produced by the real
mapf_gpt/model.py (imported in the build container by tests/golden/make_golden.py).
The product package never imports this module.
"""
import math

import torch
import torch.nn.functional as F


def to_torch(sd, dtype=torch.float32):
    return {k: torch.as_tensor(v).to(dtype) for k, v in sd.items()}


def forward_logits(sd, args, idx, dtype=torch.float32, return_layers=False):
    """idx: integer tensor/array (B, T<=256) of token ids -> logits (B, 67) of the LAST position
    (model.py:186).  `sd` is a dict of tensors/arrays with the reference's state_dict keys."""
    w = to_torch(sd, dtype)
    idx = torch.as_tensor(idx).long()
    B, T = idx.shape
    L, nh, C = args["n_layer"], args["n_head"], args["n_embd"]
    hs = C // nh
    b = lambda key: w.get(key)      # bias vectors of a GPTConfig.bias = True checkpoint (model.py:14-17,29,31,79,81), else None
    x = w["transformer.wte.weight"][idx] + w["transformer.wpe.weight"][:T]          # model.py:171-175
    layers = [x.clone()] if return_layers else None
    for l in range(L):
        p = f"transformer.h.{l}."
        h = F.layer_norm(x, (C,), w[p + "ln_1.weight"], b(p + "ln_1.bias"), 1e-5)    # model.py:20,102
        qkv = F.linear(h, w[p + "attn.c_attn.weight"], b(p + "attn.c_attn.bias"))   # model.py:50
        q, k, v = qkv.split(C, dim=2)
        q = q.view(B, T, nh, hs).transpose(1, 2)                                    # model.py:51-53
        k = k.view(B, T, nh, hs).transpose(1, 2)
        v = v.view(B, T, nh, hs).transpose(1, 2)
        y = F.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False)   # model.py:58-60
        y = y.transpose(1, 2).contiguous().view(B, T, C)                            # model.py:68
        x = x + F.linear(y, w[p + "attn.c_proj.weight"], b(p + "attn.c_proj.bias"))  # model.py:71,102
        h = F.layer_norm(x, (C,), w[p + "ln_2.weight"], b(p + "ln_2.bias"), 1e-5)
        h = F.gelu(F.linear(h, w[p + "mlp.c_fc.weight"], b(p + "mlp.c_fc.bias")))   # model.py:85-86 exact-erf GELU
        x = x + F.linear(h, w[p + "mlp.c_proj.weight"], b(p + "mlp.c_proj.bias"))   # model.py:87,103
        if return_layers:
            layers.append(x.clone())
    x = F.layer_norm(x, (C,), w["transformer.ln_f.weight"], b("transformer.ln_f.bias"), 1e-5)   # model.py:178
    logits = x[:, -1, :] @ w["lm_head.weight"].t()                                  # model.py:186
    return (logits, layers) if return_layers else logits


def act_probs(logits):
    """model.py:250-254: only the first five logits survive the mask; softmax over them."""
    return torch.softmax(logits[:, :5], dim=-1)


def act_greedy(logits):
    """model.py:258-259 (do_sample=False)."""
    return torch.argmax(logits[:, :5], dim=-1)
