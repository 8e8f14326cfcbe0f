#!/usr/bin/env python3
"""Golden logits of GPTConfig.bias = True models from the REAL reference policy (build container only; see make_golden.py).

No released config has bias vectors (experiment_setup/config-*.py), but mapf_gpt/model.py:14-17,29,31,79,81,115 supports them and a
checkpoint trained that way loads into the reference; include/mapf_gpt_amd.h (mgpt_gpt_set_param) accepts them too.  Inputs: the token
rows of the committed gpt_{tiny,2M,6M}_s1.npz goldens; weights: weights.synthetic_state_dict(..., bias=True) -- the same seeded
weights plus N(0, 0.02) bias vectors; expected output: logits and greedy actions of the imported model.py.
Run:  python tests/golden/make_golden_bias.py
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

from mapf_gpt_amd import weights  # noqa: E402


def main():
    import torch
    sys.path.insert(0, "/root/reference")
    lg = types.ModuleType("loguru")
    lg.logger = type("L", (), {"__getattr__": lambda s, k: (lambda *a, **kw: None)})()
    sys.modules["loguru"] = lg
    from mapf_gpt.model import GPT, GPTConfig
    torch.manual_seed(0)
    torch.set_num_threads(8)
    for name in ("tiny", "2M", "6M"):
        args = dict(weights.model_args(name), bias=True)
        sd = weights.synthetic_state_dict(args, seed=0, scale=1.0)
        assert any(k.endswith(".bias") for k in sd)
        net = GPT(GPTConfig(**args)).eval()
        res = net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        rows = np.load(os.path.join(OUT, f"gpt_{name}_s1.npz"))["tokens"]
        idx = torch.from_numpy(rows.astype(np.int64))
        with torch.no_grad():
            logits, _ = net(idx)                                    # model.py:167-189
            greedy = net.act(idx, do_sample=False)                  # model.py:244-260
        out = dict(tokens=rows, logits=logits[:, 0, :].numpy().astype(np.float32), greedy=greedy.numpy().astype(np.int64),
                   seed=np.array(0), scale=np.array(1.0))
        np.savez_compressed(os.path.join(OUT, f"gptbias_{name}_s1.npz"), **out)
        plain = np.load(os.path.join(OUT, f"gpt_{name}_s1.npz"))["logits"]
        print(f"gptbias_{name}_s1", out["logits"].shape, "max|logit|", float(np.abs(out["logits"]).max()),
              "max |logit - the bias-free model's|", float(np.abs(out["logits"] - plain).max()), res)


if __name__ == "__main__":
    main()
