"""bench.py's own control flow on the GPU box: the one-JSON-line contract at N = 1 and the N > 1 path (two ranks sharing
the one GPU over gloo -- the RCCL path differs only in the backend string and the device of the collective tensors)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_share_one_gpu():
    env = dict(os.environ, MGPT_BENCH_BACKEND="gloo", MGPT_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", "cfg4", "--instances", "6", "--precision", "f16x3"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    rows = 2 * 6 * 128                                   # both ranks' instances x 128 agents
    assert "12 total" in j["config"]["workload"] and f"{rows} rows/step" in j["config"]["workload"]
    assert abs(j["value"] - rows * 3 / (j["ms_per_step"] * 3e-3)) <= 1e-6 * j["value"]
    assert 0.0 <= j["config"]["mean_ISR_after_run"] <= 1.0
    assert j["roofline"]["unit"] == "TFLOP/s" and 0 < j["roofline"]["frac"] < 1
    assert "cpu_baseline" not in j                       # rank 0 at N = 1 only


def test_bench_eight_ranks_dry_run_on_one_gpu():
    """world = 8 (the driver's largest launch) as a dry run: 8 ranks share the one GPU, collectives over gloo; the instance
    ids 0 .. 31 of cfg4 are dealt 4 per rank, and the one JSON line reports the whole job."""
    env = dict(os.environ, MGPT_BENCH_BACKEND="gloo", MGPT_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--workload", "cfg4", "--instances", "4", "--precision", "f16x3", "--no-prof"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 8 and j["steps"] == 2 and j["scaling"] == "weak" and j["collective_backend"] == "gloo"
    assert "32 total" in j["config"]["workload"] and f"{8 * 4 * 128} rows/step" in j["config"]["workload"]
    assert abs(j["value"] - 8 * 4 * 128 * 2 / (j["ms_per_step"] * 2e-3)) <= 1e-6 * j["value"]
    assert j["rccl_ranks"] is None                       # only the RCCL ("nccl") backend reports its rank count
    # the N > 1 line explains itself -- every rank's own time, clock, power and the PCI address it bound to
    pr = j["per_rank"]
    assert [r["rank"] for r in pr] == list(range(8)) and all(r["ms_per_step"] > 0 and len(r["pci"]) == 12 for r in pr), pr
    assert max(r["ms_per_step"] for r in pr) <= j["ms_per_step"] * (1 + 1e-6) + 1e-6        # the headline is the max over ranks
    assert len({r["pci"] for r in pr}) == 1                                                 # the dry run shares one GPU -- and the line shows it
    assert all("sclk_mhz_mean" in r and "socket_power_w_mean" in r for r in pr)


def test_bench_gpus_n_without_a_launcher_launches_itself():
    """`python bench.py --gpus 2` with no torchrun around it re-executes as the contract's launch line (two ranks; here sharing the one
    GPU over gloo) instead of failing on the world-size assertion."""
    env = dict(os.environ, MGPT_BENCH_BACKEND="gloo", MGPT_BENCH_SHARE_GPU="1", MASTER_PORT="29549")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "cfg4",
           "--instances", "3", "--no-prof"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 2 and j["steps"] == 2 and "6 total" in j["config"]["workload"] and len(j["per_rank"]) == 2


def test_bench_single_rank_line():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--instances", "4",
           "--no-cpu-baseline", "--no-tokenizer-leg", "--no-secondary"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 1 and j["config"]["workload"].startswith("cfg3: wfi_warehouse, 192 agents, MAPF-GPT-6M")
    assert j["unit"] == "agent-steps/s" and j["dtype"] == "f16x3" and j["vs_baseline"] is None
    assert j["roofline"]["bound"] == "mfma" and j["roofline_tokenizer"]["bound"] == "hbm"
    # the episode boundary in a reported number and what the f16x3 request actually ran
    c = j["config"]
    assert c["reset_ms_per_episode"] > 0 and abs(c["amortised_reset_ms_per_step"] * c["max_episode_steps"] - c["reset_ms_per_episode"]) < 1e-9
    assert c["timed_region_crosses_episode_boundary"] is False
    assert j["effective_precision"] == "f16x3" and j["precision_envelope"]["state"] == "inside"
    assert len(j["per_rank"]) == 1 and j["per_rank"][0]["rank"] == 0


def test_bench_rccl_branch_with_one_rank():
    """the backend "nccl" (= RCCL) branch -- init_process_group(device_id=...), the barriers, the
    max-over-ranks all_reduce and the metrics all_gather on DEVICE tensors -- executed for real on a one-GPU box: bench.py under
    torch.distributed.run with one rank and MGPT_BENCH_FORCE_COLLECTIVE=1 takes the N > 1 code path with a one-rank communicator."""
    env = dict(os.environ, MGPT_BENCH_FORCE_COLLECTIVE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MGPT_BENCH_BACKEND", None)
    env.pop("MGPT_BENCH_SHARE_GPU", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29551", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--workload", "cfg4", "--instances", "6", "--precision", "f16x3", "--no-cpu-baseline", "--no-secondary", "--no-tokenizer-leg"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _last_json(r.stdout)
    assert j["n_gpus"] == 1 and j["collective_backend"] == "nccl" and j["rccl_ranks"] == 1
    assert j["rccl_version"] and j["rccl_version"][0].isdigit()
    assert j["metrics_gathered_on"].startswith("cuda")
    assert 0.0 <= j["config"]["mean_ISR_after_run"] <= 1.0
