import sys; sys.path.insert(0, '/root/repo')
import numpy as np, torch
from mapf_gpt_amd import weights
from mapf_gpt_amd.model import build_model
for name in ("2M", "6M"):
    for tail in (1.0, 20.0, 100.0):
        sd = weights.synthetic_state_dict(name, seed=3)
        rng = np.random.Generator(np.random.PCG64(5))
        for k, v in sd.items():
            if v.ndim == 2 and "wte" not in k and "wpe" not in k:
                m = rng.random(v.shape) < 0.01
                v[m] *= tail
        rows = 256
        tok = torch.from_numpy(rng.integers(0, 67, (rows, 256)).astype(np.uint8)).cuda()
        a = build_model(name, precision="f32", max_rows=rows, state_dict=sd).logits_tokens(tok).cpu().numpy()
        b = build_model(name, precision="f16x3", max_rows=rows, state_dict=sd).logits_tokens(tok).cpu().numpy()
        print(f"{name} 1% of weights x{tail:5.0f}: |logits| max {np.abs(a).max():8.3f}   max |f16x3 - f32| {np.abs(a-b).max():.3e}   (first 5 logits: {np.abs(a[:, :5]-b[:, :5]).max():.3e})")
