import sys; sys.path.insert(0, '/root/repo')
import torch
from mapf_gpt_amd import _lib
from mapf_gpt_amd.model import build_model
for prec in ("bf16", "f16x3"):
    net = build_model("85M", precision=prec, max_rows=1024)
    tok = torch.randint(0, 67, (1024, 256), dtype=torch.uint8, device="cuda")
    for _ in range(2): net.logits_tokens(tok)
    torch.cuda.synchronize()
    out = torch.zeros(16, dtype=torch.float32, device="cuda")
    _lib.check(_lib.lib().mgpt_gpt_debug_copy_raw(net._h, _lib.PRECISIONS[prec], 0, _lib.ptr(out), 64, _lib.stream_ptr()))
    v = out.cpu().numpy()[0:12:2]
    ks = v[5]
    print(prec, "FC GEMM, per k-step cycles: wait+barrier %.0f | issue DMA %.0f | fetch issue %.0f | MFMA rounds %.0f | epilogue total %.0f (k-steps %d)" % (v[0]/ks, v[1]/ks, v[2]/ks, v[3]/ks, v[4], ks))
    del net
