#!/usr/bin/env python3
"""Evaluation entry point in the role of the reference's benchmark.py (benchmark.py:20-50): for every folder
<eval-root>/<folder>/ holding `<folder>.yaml` (+ optional `maps.yaml`) run all algorithms on the grid of
environments and write `<folder>/<algorithm>.json` + the tabular views.

    python benchmark.py                                   # eval_configs/00-smoke (synthetic weights)
    python benchmark.py --eval-root /path/to/MAPF-GPT/eval_configs --folders 01-random 02-mazes \
                        --weights MAPF-GPT-2M=weights/MAPF-GPT-2M.pt
    python -m torch.distributed.run --nproc-per-node 8 benchmark.py ...   # instances sharded over the GPUs

All instances of one (algorithm, num_agents) group run as one device-resident batch (mapf_gpt_amd/evaluation.py).
"""
import argparse
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--eval-root", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "eval_configs"))
    ap.add_argument("--folders", nargs="*", default=None, help="default: every sub-folder of --eval-root")
    ap.add_argument("--weights", nargs="*", default=[], help="ALGORITHM=path overrides of path_to_weights")
    ap.add_argument("--precision", default="f16x3", choices=["f32", "f16x3", "bf16"],
                    help="policy arithmetic of every algorithm entry (the adapter's own default is f32)")
    a = ap.parse_args()

    import torch
    from mapf_gpt_amd import evaluation as ev
    from mapf_gpt_amd.inference import MAPFGPTInference

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # MGPT_BENCH_BACKEND=gloo + MGPT_BENCH_SHARE_GPU=1: dry run of the sharded path on a one-GPU box (as in bench.py)
        backend = os.environ.get("MGPT_BENCH_BACKEND", "nccl")
        local = 0 if os.environ.get("MGPT_BENCH_SHARE_GPU") else int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    MAPFGPTInference.build()                                                   # benchmark.py:26
    overrides = dict(w.split("=", 1) for w in a.weights)
    folders = a.folders or sorted(d for d in os.listdir(a.eval_root) if os.path.isdir(os.path.join(a.eval_root, d)))
    for folder in folders:
        reg = ev.MapRegistry()
        mp = os.path.join(a.eval_root, folder, "maps.yaml")
        if os.path.exists(mp):
            reg.register_maps(ev.load_yaml(mp))                                # benchmark.py:38-41
        cfg = ev.load_yaml(os.path.join(a.eval_root, folder, f"{os.path.basename(folder)}.yaml"))
        for name, algo in cfg["algorithms"].items():
            if name in overrides:
                algo["path_to_weights"] = overrides[name]
            if world > 1:
                algo["device"] = f"cuda:{0 if os.environ.get('MGPT_BENCH_SHARE_GPU') else int(os.environ.get('LOCAL_RANK', '0'))}"
        if rank == 0:
            print(f"=== {folder}")
        ev.evaluation(cfg, eval_dir=os.path.join(a.eval_root, folder), registry=reg, precision=a.precision, rank=rank, world=world)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
