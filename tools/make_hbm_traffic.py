#!/usr/bin/env python3
"""tools/make_hbm_traffic.py fetch_summary.txt write_summary.txt out.json [fetch_cfg5.txt write_cfg5.txt] -- HBM bytes per launch of the kernels bench.py's
`roofline.traffic` looks up, from two tools/pmc.sh passes (FETCH_SIZE, WRITE_SIZE; counters in KiB; FETCH_SIZE doubled per the
gfx950 note of MI355X_MICROARCH.md: 64 B counted per 128-B request)."""
import json
import re
import sys


def table(path):
    out = {}
    for line in open(path):
        m = re.match(r"(.*?grid=\d+)\s+(\d+)\s+([0-9.e+]+)\s*$", line.rstrip())
        if m:
            out[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)) * 1024.0)
    return out


KEYS = [  # (json key, kernel substring, grid, note)
    ("cfg3_f16x3_gpt_mlp_fused", "mlp256q_kernel<mgpt::fastk::F16T, 2, 0, 4>", 131072,
     "mlp256q_kernel (mlp256p_kernel on the 16 x 16 x 32 MFMA; persistent, 256 workgroups x 512 threads): x rows read by the producer for LayerNorm and again by the consumer for the "
     "residual add (2 x 3.22 GB; the second read comes ~90 us after the first, inside the 256-MiB memory-side cache's reach, and is still "
     "counted: FETCH_SIZE tallies L2 <-> fabric requests), the 2.2 MB cyclic weight stream per 128-token block served by L2, one write of x"),
    ("cfg3_f16x3_gpt_attention", "attn256q_kernel<mgpt::fastk::F16T, 2, 0, false>", 131072,
     "attn256q_kernel (attn256o_kernel with its projections and tail on the 16 x 16 x 32 MFMA; persistent, whole attention block; layers 1..6 of a forward -- layer 0 is the <.., EMB> instance, which reads the embedding table instead of x for its LayerNorm): x read for LayerNorm (3.22 GB) and again for the residual add, x written; the y planes "
     "go to the 56-MiB spill slab (written and read back by the same wave, L2 / memory-side cache) instead of a 3.22-GB y matrix + GEMM"),
    ("cfg3_f16x3_gpt_attention_last_layer", "attn_last1_kernel<256, 32>", 1572864,
     "last layer (attn_last1_kernel): all of x read once, only token 255's new row written (compact)"),
    ("cfg3_tok_generate_observations", "tokens_kernel<3, 4>", 196608, "cfg3's own launch: 64 instances x 192 agents"),
    ("cfg4_tok_generate_observations_65536_rows", "tokens_kernel<2, 4>", 1048576, "cfg4 per-GPU shard: 512 instances x 128 agents, per-instance maps"),
    ("tok_generate_observations_524160_rows", "tokens_kernel<3, 8>", 4193280, "2730 instances x 192 agents on the warehouse map (round 6: three candidate passes, 8 rows per wavefront)"),
]

KEYS5 = [  # from the cfg5 passes (bench.py --workload cfg5 --precision bf16, launches of 4096 rows = 1 048 576 tokens since round 6)
    ("cfg5_bf16_gpt_gemm_mlp_fc", "gemm_pk16_kernel<mgpt::fastk::BF16T, 3, 8, true>", 25165824,
     "c_fc of the 85M bf16 chain (LayerNorm folded): raw operand planes of 1 048 576 tokens x 768 in, hidden planes x 3072 out, weight tiles from L2"),
    ("cfg5_bf16_gpt_gemm_mlp_proj", "gemm_pk16_kernel<mgpt::fastk::BF16T, 2, 8, false>", 6291456,
     "residual GEMMs of the 85M bf16 chain (attention out-projection K = 768 and mlp.c_proj K = 3072 share this instantiation: the average mixes both)"),
]

f, w = table(sys.argv[1]), table(sys.argv[2])
out = {"source": f"{sys.argv[1]} + {sys.argv[2]}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (counters + kernel trace only) of "
                 "`python bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 1` (cfg3, f16x3, plus the two tokenizer roofline legs); counters in KiB; "
                 "FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md (64 B counted per 128-B request)",
       "units": "bytes per launch; a cfg3 GPT launch = all 12,288 rows of the step = 3,145,728 tokens, C = 256 (x plane 3,221,225,472 B)"}
for key, sub, grid, note in KEYS:
    fk = [k for k in f if sub in k and k.endswith(f"grid={grid}")]
    wk = [k for k in w if sub in k and k.endswith(f"grid={grid}")]
    if not fk:
        continue
    fr = f[fk[0]][1]
    wr = w[wk[0]][1] if wk else 0.0
    out[key] = {"kernel": sub, "grid_threads": grid, "dispatches": f[fk[0]][0], "fetch_raw": fr, "fetch_corrected_x2": 2 * fr, "write": wr, "note": note}
    if key.startswith("cfg3_f16x3_gpt"):
        out[key]["rows_per_launch"] = 12288
if len(sys.argv) >= 6:
    f5, w5 = table(sys.argv[4]), table(sys.argv[5])
    for key, sub, grid, note in KEYS5:
        fk = [k for k in f5 if sub in k and k.endswith(f"grid={grid}")]
        wk = [k for k in w5 if sub in k and k.endswith(f"grid={grid}")]
        if not fk:
            continue
        fr = f5[fk[0]][1]
        wr = w5[wk[0]][1] if wk else 0.0
        out[key] = {"kernel": sub, "grid_threads": grid, "dispatches": f5[fk[0]][0], "fetch_raw": fr, "fetch_corrected_x2": 2 * fr, "write": wr, "note": note,
                    "rows_per_launch": 4096}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: (round(v["fetch_corrected_x2"] / 1e6, 1), round(v["write"] / 1e6, 1)) for k, v in out.items() if isinstance(v, dict)}, indent=1))
