#!/usr/bin/env python3
"""Golden logits of GPT.forward on rows SHORTER than 256 tokens and of a block_size < 256 model, from the REAL reference policy
(build container only; see make_golden.py).

mapf_gpt/model.py:167-175 accepts any t <= block_size (positions 0 .. t-1, non-causal attention over the t tokens, logits of position t-1);
inference never does this (the tokenizer pads every row to 256, observation_generator.cpp:386-387), the interface does -- include/mapf_gpt_amd.h,
mgpt_gpt_forward_t.  Inputs: the first T tokens of the committed gpt_{tiny,2M}_s1.npz rows; weights: weights.synthetic_state_dict.
Run:  python tests/golden/make_golden_short.py
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
    out = {}
    # (model, block_size, row lengths): multiples of 32, ragged lengths, one token, one short of the block
    cases = [("tiny", 256, (1, 7, 32, 100, 129, 255)), ("2M", 256, (33, 64, 200)), ("6M", 256, (95,)), ("tiny", 100, (100, 64, 31)), ("85M", 256, (40,))]
    for name, block, lens in cases:
        args = dict(weights.model_args(name), block_size=block)
        sd = weights.synthetic_state_dict(args, seed=0, scale=1.0)
        net = GPT(GPTConfig(**args)).eval()
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        rows = np.load(os.path.join(OUT, f"gpt_{name}_s1.npz"))["tokens"][:5]
        for T in lens:
            idx = torch.from_numpy(rows[:, :T].astype(np.int64))
            with torch.no_grad():
                logits, _ = net(idx)                                # model.py:167-189
                greedy = net.act(idx, do_sample=False)              # model.py:244-260
            key = f"{name}_b{block}_t{T}"
            out[key + "_tokens"] = rows[:, :T]
            out[key + "_logits"] = logits[:, 0, :].numpy().astype(np.float32)
            out[key + "_greedy"] = greedy.numpy().astype(np.int64)
            print(key, "max|logit|", float(np.abs(out[key + "_logits"]).max()))
    np.savez_compressed(os.path.join(OUT, "gptshort.npz"), **out)


if __name__ == "__main__":
    main()
