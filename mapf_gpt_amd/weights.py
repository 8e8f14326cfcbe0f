"""Checkpoint layout of the reference policy and a deterministic synthetic-weight generator.

The released checkpoints are unreachable offline (mapf_gpt/inference.py:54-55 downloads them), so
tests, goldens and the bench use seeded synthetic weights with the reference's exact key list and
shapes (model.py:126-138; VERIFIED key list in SURVEY.md section 8a-M0):

    transformer.wte.weight (67,C)   [tied to lm_head.weight, model.py:138]
    transformer.wpe.weight (block,C)
    transformer.h.{l}.ln_1.weight (C)        transformer.h.{l}.attn.c_attn.weight (3C,C)
    transformer.h.{l}.attn.c_proj.weight (C,C)  transformer.h.{l}.ln_2.weight (C)
    transformer.h.{l}.mlp.c_fc.weight (4C,C) transformer.h.{l}.mlp.c_proj.weight (C,4C)
    transformer.ln_f.weight (C)      lm_head.weight (67,C)
    bias=True (model.py:115; no released config): + ln_1 / ln_2 / ln_f .bias (C), c_attn.bias (3C), c_proj.bias (C), c_fc.bias (4C)

`load_checkpoint` accepts the reference's on-disk dict {"model": state_dict, "model_args": {...}}
including the `_orig_mod.` prefixes torch.compile leaves behind (inference.py:33-44,72-78).
"""
import math

import numpy as np

MODEL_SHAPES = {  # experiment_setup/config-{2M,6M,85M}.py:6-8
    "2M": dict(n_layer=5, n_head=5, n_embd=160),
    "6M": dict(n_layer=8, n_head=8, n_embd=256),
    "85M": dict(n_layer=12, n_head=12, n_embd=768),
    "tiny": dict(n_layer=2, n_head=2, n_embd=64),
}
VOCAB = 67
BLOCK = 256


def model_args(name_or_args):
    a = dict(MODEL_SHAPES[name_or_args]) if isinstance(name_or_args, str) else dict(name_or_args)
    a.setdefault("block_size", BLOCK)
    a.setdefault("vocab_size", VOCAB)
    a.setdefault("bias", False)
    a.setdefault("dropout", 0.0)
    return a


def synthetic_state_dict(name_or_args, seed=0, scale=1.0, ln_jitter=0.1):
    """numpy PCG64 -> dict[str, float32 ndarray].  N(0, 0.02*scale) for 2-D weights, c_proj scaled by
    1/sqrt(2L) (model.py:141-145); LayerNorm gains 1 + ln_jitter*N(0,1) so that gains are exercised.  bias=True: every bias vector
    N(0, 0.02*scale) (the reference initialises them to zero, model.py:152-153; zeros would exercise nothing), drawn AFTER all the weights so
    that the weights of a seed do not depend on the flag."""
    a = model_args(name_or_args)
    L, C, V, T = a["n_layer"], a["n_embd"], a["vocab_size"], a["block_size"]
    rng = np.random.Generator(np.random.PCG64(seed))

    def normal(shape, std):
        return (rng.standard_normal(shape) * std).astype(np.float32)

    sd = {}
    sd["transformer.wte.weight"] = normal((V, C), 0.02 * scale)
    sd["transformer.wpe.weight"] = normal((T, C), 0.02 * scale)
    for l in range(L):
        p = f"transformer.h.{l}."
        sd[p + "ln_1.weight"] = (1.0 + ln_jitter * rng.standard_normal(C)).astype(np.float32)
        sd[p + "attn.c_attn.weight"] = normal((3 * C, C), 0.02 * scale)
        sd[p + "attn.c_proj.weight"] = normal((C, C), 0.02 * scale / math.sqrt(2 * L))
        sd[p + "ln_2.weight"] = (1.0 + ln_jitter * rng.standard_normal(C)).astype(np.float32)
        sd[p + "mlp.c_fc.weight"] = normal((4 * C, C), 0.02 * scale)
        sd[p + "mlp.c_proj.weight"] = normal((C, 4 * C), 0.02 * scale / math.sqrt(2 * L))
    sd["transformer.ln_f.weight"] = (1.0 + ln_jitter * rng.standard_normal(C)).astype(np.float32)
    sd["lm_head.weight"] = sd["transformer.wte.weight"]  # tied
    if a["bias"]:
        for l in range(L):
            p = f"transformer.h.{l}."
            sd[p + "ln_1.bias"] = normal((C,), 0.02 * scale)
            sd[p + "attn.c_attn.bias"] = normal((3 * C,), 0.02 * scale)
            sd[p + "attn.c_proj.bias"] = normal((C,), 0.02 * scale)
            sd[p + "ln_2.bias"] = normal((C,), 0.02 * scale)
            sd[p + "mlp.c_fc.bias"] = normal((4 * C,), 0.02 * scale)
            sd[p + "mlp.c_proj.bias"] = normal((C,), 0.02 * scale)
        sd["transformer.ln_f.bias"] = normal((C,), 0.02 * scale)
    return sd


def strip_prefix(state_dict, prefix="_orig_mod."):
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in state_dict.items()}


def load_checkpoint(path, map_location="cpu", allow_pickle=False):
    """-> (model_args dict, state_dict of float32 numpy arrays).  Same dict layout the reference's
    train.py:300-310 writes and inference.py:72-85 reads.  The released files hold tensors and a plain
    dict only, so the safe unpickler suffices; allow_pickle=True is an explicit opt-in for files that
    carry other objects (unpickling can execute code -- only for checkpoints you trust)."""
    import torch
    ckpt = torch.load(path, map_location=map_location, weights_only=not allow_pickle)
    sd = strip_prefix(ckpt["model"])
    args = model_args(ckpt["model_args"])
    out = {k: v.detach().to(torch.float32).cpu().numpy() for k, v in sd.items() if hasattr(v, "detach")}
    if "transformer.wte.weight" not in out and "lm_head.weight" in out:
        out["transformer.wte.weight"] = out["lm_head.weight"]
    return args, out
