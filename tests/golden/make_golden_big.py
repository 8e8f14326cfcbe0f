#!/usr/bin/env python3
"""Round-2 policy goldens from the REAL reference model (build container only; needs /root/reference).

For every released shape (2M, 6M, 85M) and weight scale (x1, x4: the x4 set mimics trained magnitudes, SURVEY.md
appendix B; x8 for 2M / 6M since round 4: record-only, beyond trained magnitudes) this writes tests/golden/gptbig_<shape>_s<scale>.npz with
    tokens       uint8 [256, 256]   256 REAL observation rows taken from the tokenizer goldens (all five eval maps)
    logits_f32   float32 [256, 67]  mapf_gpt/model.py GPT.forward in fp32               (model.py:167-189)
    logits_f64   float64 [256, 67]  the same module after .double()                      (the accuracy yardstick)
    logits_bf16  float32 [256, 67]  the same module under torch.autocast(bfloat16)       (train.py:66-70's regime)
Weights: the repo's seeded synthetic generator (released checkpoints need network).  The generated files are data
only; nothing at test time reads /root/reference.
Run:  python tests/golden/make_golden_big.py [shape ...]
"""
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

from mapf_gpt_amd import weights  # noqa: E402


def import_reference_model():
    sys.path.insert(0, "/root/reference")
    lg = types.ModuleType("loguru")
    lg.logger = type("L", (), {"__getattr__": lambda s, k: (lambda *a, **kw: None)})()
    sys.modules["loguru"] = lg
    from mapf_gpt.model import GPT, GPTConfig
    return GPT, GPTConfig


def real_rows(n=256):
    """n observation rows spread over the committed tokenizer goldens (reference-generated tokens)."""
    pool = []
    for tag in ("warehouse", "berlin", "mazes000", "random000", "puzzle00", "rect50x160", "rect140x70"):
        t = np.load(os.path.join(OUT, f"tok_{tag}.npz"))["tokens"]
        pool.append(t.reshape(-1, 256))
    pool = np.concatenate(pool)
    idx = np.random.Generator(np.random.PCG64(2)).permutation(len(pool))[:n]
    return np.ascontiguousarray(pool[np.sort(idx)], dtype=np.uint8)


def main():
    import torch
    GPT, GPTConfig = import_reference_model()
    torch.set_num_threads(8)
    rows = real_rows()
    idx = torch.from_numpy(rows.astype(np.int64))
    argv = sys.argv[1:]
    scales_arg = [float(a[2:]) for a in argv if a.startswith("-s")]        # e.g. -s8: only that scale
    shapes = [a for a in argv if not a.startswith("-s")] or ["2M", "6M", "85M"]
    for name in shapes:
        args = weights.model_args(name)
        # x8 (round 4, 2M / 6M only): maps where the ABSOLUTE 1e-5 bar breaks as logits grow, before real weights arrive
        for scale in (scales_arg or ((1.0, 4.0, 8.0) if name != "85M" else (1.0, 4.0))):
            sd = weights.synthetic_state_dict(name, seed=0, scale=scale)
            net = GPT(GPTConfig(**args)).eval()
            net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
            t0 = time.time()
            outs = {}
            with torch.no_grad():
                outs["logits_f32"] = torch.cat([net(idx[i:i + 32])[0][:, 0, :] for i in range(0, len(idx), 32)]).numpy().astype(np.float32)
                with torch.autocast(device_type="cpu", dtype=torch.bfloat16):           # train.py:66-70
                    outs["logits_bf16"] = torch.cat([net(idx[i:i + 32])[0][:, 0, :].float() for i in range(0, len(idx), 32)]).numpy().astype(np.float32)
                net64 = net.double()
                outs["logits_f64"] = torch.cat([net64(idx[i:i + 32])[0][:, 0, :] for i in range(0, len(idx), 32)]).numpy().astype(np.float64)
            tag = f"gptbig_{name}_s{int(scale)}"
            np.savez_compressed(os.path.join(OUT, tag + ".npz"), tokens=rows, scale=np.array(scale), seed=np.array(0), **outs)
            e32 = np.abs(outs["logits_f32"] - outs["logits_f64"]).max()
            e16 = np.abs(outs["logits_bf16"] - outs["logits_f64"]).max()
            print(f"{tag}: max|logit| {np.abs(outs['logits_f64']).max():.3f}  fp32-fp64 {e32:.2e}  autocast-fp64 {e16:.2e}  ({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
