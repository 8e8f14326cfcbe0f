import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
from mapf_gpt_amd.model import build_model
for name in ("6M", "2M"):
    net = build_model(name, seed=0, max_rows=64, precision="f16x3")
    tok = torch.from_numpy(np.random.default_rng(0).integers(0, 67, (32, 256)).astype(np.uint8)).cuda()
    for _ in range(10): net.logits_tokens(tok)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): net.logits_tokens(tok)
    torch.cuda.synchronize(); print(name, "32 rows forward:", round((time.perf_counter() - t0) / 200 * 1e3, 4), "ms")
