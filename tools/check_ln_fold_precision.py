#!/usr/bin/env python3
"""tools/check_ln_fold_precision.py -- the folded LayerNorm of the bf16 GEMM chain against ln_pack_kernel (MGPT_LN_FOLD=0) as the
per-token mean of the residual stream grows: a constant added to wte, C = 512, 2 layers, logits against the fp64 port."""
import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from mapf_gpt_amd import weights
from mapf_gpt_amd.model import GPT, GPTConfig
from oracle import gpt_oracle
from tests.helpers import load_tok
rows = load_tok("mazes000")["tokens"][7, :3]
args = weights.model_args(dict(n_layer=2, n_head=8, n_embd=512))
for off in (0.0, 0.1, 0.5, 2.0):
    sd = weights.synthetic_state_dict(args, seed=11, scale=2.0)
    sd = {k: np.array(v, copy=True) for k, v in sd.items()}
    sd["transformer.wte.weight"] = sd["transformer.wte.weight"] + np.float32(off); sd["lm_head.weight"] = sd["transformer.wte.weight"]  #                     # per-token mean of the residual stream = off (lm_head is tied: it moves too)
    ref = gpt_oracle.forward_logits(sd, args, rows, dtype=torch.float64).numpy()
    x0 = np.asarray(sd["transformer.wte.weight"][0] + sd["transformer.wpe.weight"][0])
    out = []
    for fold in ("1", "0"):
        os.environ["MGPT_LN_FOLD"] = fold
        net = GPT(GPTConfig(**args), max_rows=4, precision="bf16"); net.load_state_dict(sd)
        out.append(np.abs(net.logits_tokens(torch.from_numpy(rows).cuda()).cpu().numpy() - ref).max())
        del net
    print(f"wte offset {off:4.1f}: |mean|/std of an embedding row {abs(x0.mean()) / x0.std():6.1f}; max |dlogit| vs fp64: fold {out[0]:.3e}, ln_pack {out[1]:.3e}; |logits| max {np.abs(ref).max():.2f}")
