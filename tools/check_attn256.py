"""tools/check_attn256.py -- attn256_kernel's y planes of layer 0 (2-layer C=256 model) against an fp64 host computation."""
import ctypes, numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mapf_gpt_amd import _lib, weights
from mapf_gpt_amd.model import GPT, GPTConfig
args = weights.model_args(dict(n_layer=2, n_head=8, n_embd=256))
sd = weights.synthetic_state_dict(args, seed=3, scale=float(sys.argv[1]) if len(sys.argv) > 1 else 1.0)
rows = 4
mode = sys.argv[2] if len(sys.argv) > 2 else ""
if "zqk" in mode: sd["transformer.h.0.attn.c_attn.weight"][:512] = 0
if "bigqk" in mode: sd["transformer.h.0.attn.c_attn.weight"][:512] *= 8
net = GPT(GPTConfig(**args), max_rows=rows, precision="f16x3"); net.load_state_dict(sd)
rng = np.random.Generator(np.random.PCG64(5))
tok = rng.integers(0, 67, (rows, 256)).astype(np.uint8)
net.logits_tokens(torch.from_numpy(tok).cuda())
M, C, NP, KS = rows * 256, 256, 2, 16
raw = torch.empty(M * C * NP, dtype=torch.float16, device="cuda")
_lib.check(_lib.lib().mgpt_gpt_debug_copy_raw(net._h, 1, 5, _lib.ptr(raw), raw.numel() * 2, _lib.stream_ptr()))
pk = raw.cpu().numpy().astype(np.float64)
m = np.arange(M)[:, None]; n = np.arange(C)[None, :]
def idx(pl): return ((((m >> 5) * KS + (n >> 4)) * NP + pl) << 9) + (((m & 31) + ((n & 8) << 2)) << 3) + (n & 7)
y = pk[idx(0)] + pk[idx(1)]
w = {k: torch.as_tensor(v).double() for k, v in sd.items()}
x = w["transformer.wte.weight"][torch.from_numpy(tok.astype(np.int64))] + w["transformer.wpe.weight"][:256]
h = torch.nn.functional.layer_norm(x, (C,), w["transformer.h.0.ln_1.weight"], None, 1e-5)
qkv = h @ w["transformer.h.0.attn.c_attn.weight"].t()
q, k, v = qkv.split(C, dim=2)
B, T, nh, hs = rows, 256, 8, 32
q = q.view(B, T, nh, hs).transpose(1, 2); k = k.view(B, T, nh, hs).transpose(1, 2); v = v.view(B, T, nh, hs).transpose(1, 2)
att = torch.softmax(q @ k.transpose(-1, -2) / np.sqrt(hs), dim=-1)
ref = (att @ v).transpose(1, 2).reshape(M, C).numpy()
err = np.abs(y - ref)
print("max|y| %.4f  max err %.3e" % (np.abs(ref).max(), err.max()))
print("per head:", " ".join("%.1e" % err[:, 32 * i:32 * i + 32].max() for i in range(8)))
print("per wave (token tile):", " ".join("%.1e" % err.reshape(rows, 8, 32, C)[:, i].max() for i in range(8)))
print("per d within head:", " ".join("%.0e" % err.reshape(M, 8, 32)[:, :, d].max() for d in range(32)))
