#!/usr/bin/env python3
"""tools/diag_fast.py [f16x3|bf16] -- stage-by-stage error report of the 16-bit forward on an L=1 model
(GPU box).  Prints max abs error of every intermediate against an fp64 torch port; nothing asserts."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F

from mapf_gpt_amd import _lib, weights
from mapf_gpt_amd.model import GPT, GPTConfig
from tests.helpers import load_tok

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
P = _lib.PRECISIONS[prec]
NP = 2 if prec == "f16x3" else 1
for shape in (dict(n_layer=1, n_head=5, n_embd=160), dict(n_layer=1, n_head=2, n_embd=64), dict(n_layer=1, n_head=8, n_embd=256),
              dict(n_layer=1, n_head=2, n_embd=128)):
    args = weights.model_args(shape)
    C, nh = args["n_embd"], args["n_head"]
    hs = C // nh
    sd = weights.synthetic_state_dict(args, seed=3, scale=4.0)
    B = 2
    net = GPT(GPTConfig(**args), max_rows=B, precision=prec)
    net.load_state_dict(sd)
    rows = load_tok("mazes000")["tokens"][5, :B]
    tokens = torch.from_numpy(rows).cuda()
    logits = net.logits_tokens(tokens).cpu().numpy()
    T = 256
    M = B * T
    w = {k: torch.as_tensor(v).double() for k, v in sd.items()}
    idx = torch.from_numpy(rows.astype(np.int64))
    x0 = w["transformer.wte.weight"][idx] + w["transformer.wpe.weight"][:T]
    h1 = F.layer_norm(x0, (C,), w["transformer.h.0.ln_1.weight"], None, 1e-5)
    qkv = h1 @ w["transformer.h.0.attn.c_attn.weight"].t()
    q, k, v = [t.view(B, T, nh, hs).transpose(1, 2) for t in qkv.split(C, dim=2)]
    att = torch.softmax(q @ k.transpose(-2, -1) / np.sqrt(hs), -1)
    y = (att @ v).transpose(1, 2).reshape(B, T, C)
    x1 = x0 + y @ w["transformer.h.0.attn.c_proj.weight"].t()
    h2 = F.layer_norm(x1, (C,), w["transformer.h.0.ln_2.weight"], None, 1e-5)
    hid = F.gelu(h2 @ w["transformer.h.0.mlp.c_fc.weight"].t())
    x2 = x1 + hid @ w["transformer.h.0.mlp.c_proj.weight"].t()
    xf = F.layer_norm(x2, (C,), w["transformer.ln_f.weight"], None, 1e-5)
    ref_logits = (xf[:, -1] @ w["lm_head.weight"].t()).numpy()

    def raw(which, n16):
        out = torch.empty(n16, dtype=torch.int16, device="cuda")
        _lib.check(_lib.lib().mgpt_gpt_debug_copy_raw(net._h, P, which, _lib.ptr(out), n16 * 2, _lib.stream_ptr()))
        return out

    def planes(wh, n):
        hi = raw(wh, n)
        dt = torch.float16 if prec == "f16x3" else torch.bfloat16
        val = hi.view(dt).double()
        if NP == 2:
            val = val + raw(wh + 1, n).view(dt).double()
        return val.cpu()

    def dbgx():
        out = torch.empty(M * C, dtype=torch.float32, device="cuda")
        _lib.check(_lib.lib().mgpt_gpt_debug_copy(net._h, 0, _lib.ptr(out), M * C, _lib.stream_ptr()))
        return out.cpu().double().view(B, T, C)

    def err(a, b):
        return float((a - b).abs().max()), float(b.abs().max())

    def where(name, a, b, thr=2e-5):
        d = (a - b).abs().reshape(B, T, -1)
        bad = d > thr
        if bad.any():
            toks = bad.any(2).nonzero()
            cols = bad.any(1).any(0).nonzero().flatten()
            print(f"    {name}: {int(bad.sum())}/{bad.numel()} elems > {thr}; bad tokens {toks.shape[0]} e.g. {toks[:6].tolist()} ... {toks[-3:].tolist()}; "
                  f"bad cols {cols.numel()} e.g. {cols[:12].tolist()}")

    print(f"== {prec} C={C} nh={nh} hs={hs}")
    st = raw(0, M * 4).view(torch.float32).view(M, 2).cpu().double()      # stats after the LAST producer (proj2): of x2
    mean2, var2 = x2.view(M, C).mean(1), x2.view(M, C).var(1, unbiased=False)
    print("  stats(final x) mean err %.2e rstd err %.2e" % (err(st[:, 0], mean2)[0], err(st[:, 1], (var2 + 1e-5).rsqrt())[0]))
    qk = planes(1, 2 * M * C).view(2, B, nh, T, hs)
    print("  q  err %.2e (max %.2f)" % err(qk[0], q))
    print("  k  err %.2e (max %.2f)" % err(qk[1], k))
    vt = planes(3, M * C).view(B, nh, hs, T)
    print("  vT err %.2e (max %.2f)" % err(vt, v.transpose(2, 3)))
    yy = planes(5, M * C).view(B, T, C)
    print("  y  err %.2e (max %.2f)" % err(yy, y))
    where("y", yy, y)
    dt_ = torch.float16 if prec == "f16x3" else torch.bfloat16
    yh = raw(5, M * C).view(dt_).double().cpu().view(B, T, C)
    yl = raw(6, M * C).view(dt_).double().cpu().view(B, T, C) if NP == 2 else torch.zeros_like(yh)
    bad = ((yy - y).abs() > 2e-5).nonzero()
    for bi in bad[:6]:
        b_, t_, c_ = [int(v) for v in bi]
        print(f"      y[{b_},{t_},{c_}] ref {float(y[b_,t_,c_]):.9f} hi {float(yh[b_,t_,c_]):.9f} lo {float(yl[b_,t_,c_]):.3e} sum {float(yy[b_,t_,c_]):.9f} "
              f"ref-hi {float(y[b_,t_,c_]-yh[b_,t_,c_]):.3e}  neighbours lo {[float(v) for v in yl[b_,t_,max(0,c_-2):c_+3]]}")
    hh = planes(7, 4 * M * C).view(B, T, 4 * C)
    print("  h  err %.2e (max %.2f)" % err(hh, hid))
    where("h", hh, hid)
    print("  x2 err %.2e (max %.2f)" % err(dbgx(), x2))
    where("x2", dbgx(), x2)
    print("  logits err %.2e (max %.2f)" % err(torch.from_numpy(logits).double(), torch.from_numpy(ref_logits)))
