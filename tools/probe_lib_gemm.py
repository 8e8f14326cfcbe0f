"""What the vendor library's bf16 GEMM sustains on the 85M chain's shapes (a bar for gemm_pk16_kernel, not a product path).

torch.matmul -> hipBLASLt / rocBLAS; random normal operands (the package power limit makes operand values matter).
Prints TFLOP/s per shape; run under rocprofv3 --kernel-trace --stats to see the kernels (macro tile in the name) the library picked."""
import sys
import torch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096 * 256
dev = torch.device("cuda:0")
shapes = [("c_fc", 768, 3072), ("mlp c_proj", 3072, 768), ("attn c_proj", 768, 768), ("q|k", 768, 1536), ("q|k|v", 768, 2304)]
for name, K, N in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(a, w.t(), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 10
    e0.record()
    for _ in range(it):
        torch.matmul(a, w.t(), out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    print(f"{name:12s} M={M} K={K} N={N}: {ms:8.3f} ms  {2.0 * M * K * N / ms / 1e9:8.1f} TFLOP/s", flush=True)
    del a, w, out
