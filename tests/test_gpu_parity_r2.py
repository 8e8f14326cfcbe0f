"""Round-2 parity additions (VERDICT r01 "close the parity holes"), all through the C ABI on the GPU:
  (a) the 16-rows-per-wave tokens kernel (launches with >= 4096 instance-chunks) vs the C oracle;
  (b) the cfg4 workload (per-instance random / maze maps, 128 agents) -> tokens vs oracle, 6M logits vs the fp64 port;
  (c) the bf16 mode vs the reference's own bf16-autocast logits (2M, 6M, 85M);
  (d) f32 and f16x3 on 256 real token rows per shape, x1 and x4 weights, vs the reference's fp32 / fp64 logits;
  (e) released-checkpoint layout ({"model": {"_orig_mod....": tensor}, "model_args"}) through MAPFGPTInference;
  (f) sharded sampling reproduces the unsharded draws (global row key).
Goldens gptbig_*.npz come from the REAL mapf_gpt/model.py (tests/golden/make_golden_big.py)."""
import os

import numpy as np
import pytest
import torch

from mapf_gpt_amd import maps, weights
from oracle import gpt_oracle
from oracle import oracle as orc
from tests.helpers import GOLDEN, record_parity

pytestmark = pytest.mark.gpu
TOL = 1e-5            # BASELINE.json north_star: logits within 1e-5 of the reference PyTorch forward


def _dev(a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


# ------------------------------------------------------------------------------------------------------------------
# (a) tokens_kernel<KP, 16>: chosen by mgpt_tokenizer_generate_observations when n_inst * ceil(n_agents / 64) >= 4096
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_inst,n_agents,kind", [(4096, 64, "mazes000"), (2048, 128, "random40"), (1400, 192, "warehouse")])
def test_large_launch_kernel_vs_oracle(n_inst, n_agents, kind):
    """64-agents-per-block / 16-rows-per-wave instantiation (KP = 1, 2 and 4): 96 distinct base instances are checked row
    for row against the C oracle over 3 steps (random intended actions, executed by the oracle env); all other instance
    slots replicate a base instance and must equal it bit for bit."""
    from mapf_gpt_amd.observation_generator import BatchedTokenizer
    assert n_inst * ((n_agents + 63) // 64) >= 4096
    if kind == "mazes000":
        grid, s_ok, g_ok = maps.load_named("validation-mazes-seed-000")
    elif kind == "warehouse":
        grid, s_ok, g_ok = maps.load_named("wfi_warehouse")
    else:
        grid, s_ok, g_ok = maps.pad(maps.random_map(40, 40, 0.2, 77)), None, None
    nb = 96
    comp = maps.largest_component(grid == 0)
    pg = [maps.place_agents(grid, n_agents, 1000 + i, s_ok, g_ok, component=comp) for i in range(nb)]
    pos = np.stack([p for p, _ in pg]).astype(np.int32)
    goal = np.stack([g for _, g in pg]).astype(np.int32)
    rep = np.arange(n_inst) % nb
    rng = np.random.Generator(np.random.PCG64(n_inst + n_agents))
    gens = [orc.OracleGenerator(grid) for _ in range(nb)]
    tok = BatchedTokenizer(grid, n_inst, n_agents)
    last = np.full((nb, n_agents), -1, np.int32)
    for t in range(3):
        dp, dg, da = _dev(pos[rep], torch.int16), _dev(goal[rep], torch.int16), _dev(last[rep], torch.int32)
        if t == 0:
            tok.create_agents(dp, dg)
            for i in range(nb):
                gens[i].create_agents(pos[i], goal[i])
        tok.update_agents(dp, dg, da, goals_may_change=False)
        got = tok.generate_observations().cpu().numpy().reshape(n_inst, n_agents, 256)
        for i in range(nb):
            gens[i].update_agents(pos[i], goal[i], last[i])
            assert np.array_equal(got[i], gens[i].generate_observations()), f"step {t} base instance {i}"
        assert np.array_equal(got, got[rep]), f"step {t}: replicas differ from their base instance"
        last = rng.integers(0, 5, (nb, n_agents)).astype(np.int32)
        for i in range(nb):
            pos[i], _ = orc.env_step(grid, pos[i], goal[i], last[i])


# ------------------------------------------------------------------------------------------------------------------
# (b) cfg4: BASELINE.json configs[3] -- per-instance random / maze maps 40 x 40, 128 agents, MAPF-GPT-6M
# ------------------------------------------------------------------------------------------------------------------
def _cfg4_instances(lo, hi, n_agents=128):
    import bench
    return bench.cfg4_instances(lo, hi, n_agents)


def test_cfg4_workload_tokens_and_6M_logits():
    """12 instances of bench.py's cfg4 generator (6 Bernoulli maps, 6 mazes, each instance its own map): 8 steps of the
    tokenizer vs the C oracle, then the 6M forward of those rows in f16x3 and f32 vs the fp64 torch port at 1e-5."""
    from mapf_gpt_amd.model import build_model
    from mapf_gpt_amd.observation_generator import BatchedTokenizer
    n_inst, n = 12, 128
    grids, pos_t, goal_t = _cfg4_instances(0, n_inst, n)
    pos, goal = pos_t.numpy().astype(np.int32), goal_t.numpy().astype(np.int32)
    assert len({g.tobytes() for g in grids}) == n_inst
    gens = [orc.OracleGenerator(grids[i]) for i in range(n_inst)]
    tok = BatchedTokenizer(grids, n_inst, n)
    rng = np.random.Generator(np.random.PCG64(44))
    last = np.full((n_inst, n), -1, np.int32)
    keep_rows = []
    for t in range(8):
        dp, dg, da = _dev(pos, torch.int16), _dev(goal, torch.int16), _dev(last, torch.int32)
        if t == 0:
            tok.create_agents(dp, dg)
            for i in range(n_inst):
                gens[i].create_agents(pos[i], goal[i])
        tok.update_agents(dp, dg, da, goals_may_change=False)
        got = tok.generate_observations().cpu().numpy().reshape(n_inst, n, 256)
        for i in range(n_inst):
            gens[i].update_agents(pos[i], goal[i], last[i])
            assert np.array_equal(got[i], gens[i].generate_observations()), f"step {t} instance {i}"
        keep_rows.append(got[:, ::16].reshape(-1, 256))             # 8 rows per instance and step
        last = rng.integers(0, 5, (n_inst, n)).astype(np.int32)
        for i in range(n_inst):
            pos[i], _ = orc.env_step(grids[i], pos[i], goal[i], last[i])
    rows = np.concatenate(keep_rows)[::8][:96]                      # 96 rows spread over steps / instances / maps
    sd, args = weights.synthetic_state_dict("6M", seed=0), weights.model_args("6M")
    ref = gpt_oracle.forward_logits(sd, args, rows, dtype=torch.float64).numpy()
    tokens = torch.from_numpy(np.ascontiguousarray(rows)).cuda()
    for prec in ("f16x3", "f32"):
        net = build_model("6M", seed=0, max_rows=64, precision=prec)
        err = np.abs(net.logits_tokens(tokens).cpu().numpy() - ref).max()
        assert err <= TOL, f"cfg4 rows, 6M {prec}: max |dlogit| = {err:.3e}"


# ------------------------------------------------------------------------------------------------------------------
# (c) + (d) 256 real rows per shape against the reference's own fp32 / fp64 / bf16-autocast logits
# ------------------------------------------------------------------------------------------------------------------
def _big(shape, scale):
    return np.load(os.path.join(GOLDEN, f"gptbig_{shape}_s{scale}.npz"))


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("shape,scale", [("2M", 1), ("2M", 4), ("6M", 1), ("6M", 4), ("85M", 1), ("85M", 4)])
def test_256_real_rows_within_1e5_of_reference(shape, scale, precision):
    """The north star's bar on every released shape, x1 and x4 (trained-magnitude) weights, 256 real observation rows:
    |ours - reference fp32 forward| <= 1e-5.  The fp32 reference itself sits e_ref = 7e-7 .. 4.4e-6 away from its own fp64 run
    on five of the six sets; on 85M x4 (|logit| up to 8.2) it is 3.6e-5 away, so no implementation can be within 1e-5 of both
    there.  The bar: e32 <= 1e-5, OR as close to fp64 as the reference's own fp32 run (e64 <= 1.05 e_ref:
    measured on the GPU in round 3, the headline f16x3 mode is 3.77e-5 from fp64 on that set against the reference's 3.64e-5,
    i.e. inside 4 % of the reference's own distance; round 2's docstring said 3.5e-5, which no record backed).  The f32 mode
    (v_mfma_f32_32x32x2_f32, a different accumulation order from the reference's CPU kernels) meets the strict bar on five
    sets and is held to its measured class (2 e_ref against both yardsticks) on 85M x4.  Every measured number goes to
    gpurun_out/parity_records.jsonl; the round's copy is profiles/r03_parity_records.jsonl."""
    from mapf_gpt_amd.model import build_model
    g = _big(shape, scale)
    # envelope="ignore": this test measures the split-fp16 ARITHMETIC; what a drop-in user gets on a checkpoint outside the
    # validated range is test_precision_envelope_is_a_runtime_property's subject
    net = build_model(shape, seed=0, scale=float(scale), max_rows=64 if shape == "85M" else 128, precision=precision, envelope="ignore")
    logits = net.logits_tokens(torch.from_numpy(g["tokens"]).cuda()).cpu().numpy().astype(np.float64)
    e_ref = np.abs(g["logits_f32"].astype(np.float64) - g["logits_f64"]).max()
    e32 = np.abs(logits - g["logits_f32"]).max()
    e64 = np.abs(logits - g["logits_f64"]).max()
    record_parity(test="256_real_rows", shape=shape, scale=scale, precision=precision, e32=e32, e64=e64, e_ref=e_ref,
                  max_abs_logit=np.abs(g["logits_f64"]).max(), rows=int(g["tokens"].shape[0]))
    msg = f"{shape} x{scale} {precision}: vs fp32 ref {e32:.3e}, vs fp64 ref {e64:.3e} (reference fp32 vs fp64 {e_ref:.3e})"
    if precision == "f32" and (shape, scale) == ("85M", 4):
        assert e64 <= 2 * e_ref and e32 <= 2 * e_ref, msg
    elif (shape, scale) == ("85M", 4):
        # the one set on which the absolute bar cannot hold (the reference's own fp32 run is 3.64e-5 from its
        # fp64 run) is held to what is MEASURED (profiles/r04_parity_records.jsonl: e32 5.66e-5, e64 3.77e-5 = 1.037 e_ref), not to
        # an open multiple of e_ref
        assert e32 <= 6.5e-5 and e64 <= 1.1 * e_ref, msg
    else:
        # e32 stays bounded in every case (it may exceed 1e-5 only where the reference's own fp32 run is further
        # than that from fp64, and then by at most twice that distance), and the fp64 yardstick gets real headroom: the recorded
        # 85M x4 f16x3 numbers are e64 = 3.77e-5 against e_ref = 3.64e-5 (ratio 1.036, profiles/r03_parity_records.jsonl) --
        # a 1.05 bound would flip on any change of accumulation order; 1.15 still means "as close to fp64 as the reference".
        assert e32 <= max(TOL, 2.0 * e_ref), msg
        assert e32 <= TOL or e64 <= 1.15 * e_ref, msg
        assert e64 <= max(TOL, 1.15 * e_ref), msg


@pytest.mark.parametrize("precision", ["f16x3", "f32"])
@pytest.mark.parametrize("shape", ["2M", "6M"])
def test_x8_weights_break_point_of_the_absolute_bar(shape, precision):
    """VERDICT r03 item 6 (record-only): x8 synthetic weights (|logits| up to 7 / 10) lie beyond trained magnitudes; there the
    reference's own fp32 forward is 1.5e-5 (2M) / 6.8e-5 (6M) away from its fp64 run, so the ABSOLUTE 1e-5 bar cannot hold for
    anyone.  The numbers go to parity_records (test = "x8_break_point"); the only assertion is the class: as close to fp64 as
    the reference's fp32 run, within a factor of two."""
    from mapf_gpt_amd.model import build_model
    g = _big(shape, 8)
    net = build_model(shape, seed=0, scale=8.0, max_rows=128, precision=precision, envelope="ignore")     # (x8 lies outside the envelope: rms 0.16)
    logits = net.logits_tokens(torch.from_numpy(g["tokens"]).cuda()).cpu().numpy().astype(np.float64)
    e_ref = np.abs(g["logits_f32"].astype(np.float64) - g["logits_f64"]).max()
    e32 = np.abs(logits - g["logits_f32"]).max()
    e64 = np.abs(logits - g["logits_f64"]).max()
    record_parity(test="x8_break_point", shape=shape, scale=8, precision=precision, e32=e32, e64=e64, e_ref=e_ref,
                  max_abs_logit=np.abs(g["logits_f64"]).max(), rows=int(g["tokens"].shape[0]))
    assert e64 <= 2.0 * e_ref and e32 <= 3.0 * e_ref, f"{shape} x8 {precision}: e32 {e32:.3e} e64 {e64:.3e} e_ref {e_ref:.3e}"


@pytest.mark.parametrize("shape", ["2M", "6M"])
def test_heavy_tailed_weights_x20(shape):
    """Headroom outside the goldens' N(0, 0.02) weights (tools/check_heavy_tails.py, now in the suite): 1 % of every 2-D weight
    multiplied by 20; f16x3 against our exact-fp32-MFMA path on 256 random rows stays inside the 1e-5 bar (measured 5.5e-6 for
    2M, 7.4e-6 for 6M at |logits| <= 5).  (With x100 outliers -- |logits| ~ 25 -- it is 1e-4 .. 1e-3: the absolute bar holds for
    weight distributions like the released models', not for arbitrarily heavy tails; bench.py's note says so.)"""
    from mapf_gpt_amd.model import build_model
    sd = weights.synthetic_state_dict(shape, seed=3)
    rng = np.random.Generator(np.random.PCG64(5))
    for k, v in sd.items():
        if v.ndim == 2 and "wte" not in k and "wpe" not in k:
            v[rng.random(v.shape) < 0.01] *= 20.0
    tok = torch.from_numpy(rng.integers(0, 67, (256, 256)).astype(np.uint8)).cuda()
    a = build_model(shape, precision="f32", max_rows=256, state_dict=sd).logits_tokens(tok).cpu().numpy()
    b = build_model(shape, precision="f16x3", max_rows=256, state_dict=sd, envelope="ignore").logits_tokens(tok).cpu().numpy()
    err = float(np.abs(a - b).max())
    record_parity(test="heavy_tails_x20", shape=shape, precision="f16x3 vs f32", err=err, max_abs_logit=float(np.abs(a).max()))
    assert err <= TOL, f"{shape}: max |f16x3 - f32| = {err:.3e} at |logits| <= {np.abs(a).max():.2f}"


@pytest.mark.parametrize("shape,scale", [("2M", 1), ("2M", 4), ("6M", 1), ("6M", 4), ("85M", 1), ("85M", 4)])
def test_bf16_mode_vs_reference_autocast(shape, scale):
    """bf16 mode (single-pass bf16 MFMA, fp32 accumulate, fp32 residual stream / LayerNorm / softmax) against the
    reference run under torch.autocast(bfloat16) (train.py:66-70).  Two bf16 pipelines round at different points, so the
    stated tolerance is relative to the reference's own autocast error e_ref = max|autocast - fp64| on the same rows:
        max|ours - fp64|      <= 1.5 * e_ref + 2e-3     (at least as accurate as the reference's regime)
        max|ours - autocast|  <= 2.5 * e_ref + 2e-3     (the two land in the same error ball)"""
    from mapf_gpt_amd.model import build_model
    g = _big(shape, scale)
    net = build_model(shape, seed=0, scale=float(scale), max_rows=64 if shape == "85M" else 128, precision="bf16")
    logits = net.logits_tokens(torch.from_numpy(g["tokens"]).cuda()).cpu().numpy().astype(np.float64)
    e_ref = np.abs(g["logits_bf16"] - g["logits_f64"]).max()
    e_ours = np.abs(logits - g["logits_f64"]).max()
    e_mut = np.abs(logits - g["logits_bf16"]).max()
    record_parity(test="bf16_vs_autocast", shape=shape, scale=scale, precision="bf16", e_ours_vs_fp64=e_ours, e_ours_vs_autocast=e_mut,
                  e_ref_autocast_vs_fp64=e_ref, max_abs_logit=np.abs(g["logits_f64"]).max())
    assert e_ours <= 1.5 * e_ref + 2e-3, f"{shape} x{scale}: ours-fp64 {e_ours:.3e} vs reference autocast-fp64 {e_ref:.3e}"
    assert e_mut <= 2.5 * e_ref + 2e-3, f"{shape} x{scale}: ours-autocast {e_mut:.3e} (reference autocast-fp64 {e_ref:.3e})"
    assert e_ours > 1e-6, "this must be the reduced-precision path"


# ------------------------------------------------------------------------------------------------------------------
# (e) released-checkpoint layout end to end (inference.py:33-44,72-85; train.py:300-310)
# ------------------------------------------------------------------------------------------------------------------
def test_checkpoint_roundtrip_through_adapter(tmp_path):
    """torch.save({"model": {"_orig_mod.<key>": tensor}, "model_args": {...}, ...}) -- the dict train.py writes and
    torch.compile prefixes -- loaded by MAPFGPTInference(path_to_weights=...) must give the same logits as the same
    weights handed over directly, and the safe unpickler must suffice."""
    from mapf_gpt_amd.inference import MAPFGPTInference, MAPFGPTInferenceConfig
    from mapf_gpt_amd.model import build_model
    args = weights.model_args("2M")
    sd = weights.synthetic_state_dict("2M", seed=5)
    ckpt = {"model": {"_orig_mod." + k: torch.from_numpy(v.copy()) for k, v in sd.items()},
            "model_args": {k: args[k] for k in ("n_layer", "n_head", "n_embd", "block_size", "bias", "vocab_size", "dropout")},
            "iter_num": 1000, "best_val_loss": 1.25, "config": {"dataset": "x", "batch_size": 32}}
    path = tmp_path / "MAPF-GPT-2M-test.pt"
    torch.save(ckpt, str(path))
    a2, sd2 = weights.load_checkpoint(str(path))
    assert a2["n_layer"] == 5 and a2["n_embd"] == 160 and set(sd2) == set(sd)
    assert all(np.array_equal(sd2[k], sd[k]) for k in sd)
    algo = MAPFGPTInference(MAPFGPTInferenceConfig(path_to_weights=str(path), batch_size=64, precision="f32"))
    rows = np.load(os.path.join(GOLDEN, "gptbig_2M_s1.npz"))["tokens"][:24]
    tokens = torch.from_numpy(rows).cuda()
    direct = build_model("2M", state_dict=sd, max_rows=64).logits_tokens(tokens).cpu().numpy()
    via = algo.net.logits_tokens(tokens).cpu().numpy()
    assert np.array_equal(via, direct)
    ref = gpt_oracle.forward_logits(sd, args, rows).numpy()
    assert np.abs(via - ref).max() <= TOL
    acts = algo.act_batch([rows[:12].astype(np.int64).tolist(), rows[12:].astype(np.int64).tolist()])   # pre-tokenised rows, inference.py:146
    assert [len(a) for a in acts] == [12, 12] and all(0 <= x <= 4 for a in acts for x in a)


# ------------------------------------------------------------------------------------------------------------------
# (f) shard-independent sampling: the RNG key is (seed, step, GLOBAL row)
# ------------------------------------------------------------------------------------------------------------------
def test_two_shards_reproduce_single_shard_actions():
    """a 2-shard run (row0 = first global row of the shard) must draw exactly what the unsharded run draws,
    step after step, through BatchedRunner (env + tokenizer + forward + device sampler)."""
    from mapf_gpt_amd.model import build_model
    from mapf_gpt_amd.runner import BatchedRunner, make_instances, shard_range
    grid, s_ok, g_ok = maps.load_named("validation-random-seed-000")
    n_inst, n = 6, 20
    net = build_model("tiny", seed=0, max_rows=64)
    pos, goal = make_instances(grid, n_inst, n, 0, s_ok, g_ok)
    full = BatchedRunner(grid, n_inst, n, net, max_episode_steps=16, seed=3, do_sample=True)
    full.reset(pos, goal)
    shards = []
    for r in range(2):
        lo, hi = shard_range(n_inst, r, 2)
        s = BatchedRunner(grid, hi - lo, n, net, max_episode_steps=16, seed=3, do_sample=True, row_offset=lo * n)
        s.reset(pos[lo:hi], goal[lo:hi])
        shards.append((lo, hi, s))
    differs = False
    for t in range(6):
        full.step()
        a_full = full.actions.cpu().numpy()
        for lo, hi, s in shards:
            s.step()
            assert np.array_equal(s.actions.cpu().numpy(), a_full[lo:hi]), f"step {t} shard [{lo},{hi})"
        differs |= bool((a_full[:3] != a_full[3:]).any())
    assert differs                                   # the draws are not degenerate
    m = torch.cat([s.metrics() for _, _, s in shards]).cpu().numpy()
    assert np.array_equal(m, full.metrics().cpu().numpy())


def test_generatorless_act_advances_between_calls():
    """GPT.act(idx) without a torch generator must not reuse the same uniform at every call."""
    from mapf_gpt_amd.model import build_model
    net = build_model("tiny", seed=0, max_rows=256)
    rows = np.load(os.path.join(GOLDEN, "gptbig_2M_s1.npz"))["tokens"][:200]
    idx = torch.from_numpy(rows.astype(np.int64)).cuda()
    torch.manual_seed(11)
    draws = np.stack([net.act(idx).cpu().numpy() for _ in range(4)])
    assert draws.min() >= 0 and draws.max() <= 4
    assert all((draws[i] != draws[j]).any() for i in range(4) for j in range(i))
    # re-seeding with the SAME seed replays the same draws (the reference's multinomial follows the global RNG) ...
    torch.manual_seed(11)
    again = np.stack([net.act(idx).cpu().numpy() for _ in range(4)])
    assert np.array_equal(draws, again)
    # ... a different seed does not, and the explicit control pins / forgets the seed
    torch.manual_seed(12)
    assert (net.act(idx).cpu().numpy() != draws[0]).any()
    net.reset_sampler(seed=5)
    a = net.act(idx).cpu().numpy()
    net.reset_sampler(seed=5)
    assert np.array_equal(a, net.act(idx).cpu().numpy())


@pytest.mark.envelope_fallback_ok
def test_precision_envelope_is_a_runtime_property():
    """the range on which precision="f16x3" meets the 1e-5 bar is checked at load time, per checkpoint
    (include/mapf_gpt_amd.h: MGPT_ENVELOPE_*): weight statistics at finalize, a probe of 8 fixed rows through both paths at the first
    f16x3 forward.  N(0, 0.02) weights: inside, the split path runs.  x100 outliers on 1 % of the entries (max|w| ~ 9: where the
    split path is 1e-4 .. 1e-3 off): outside; "fallback" serves the request with the exact-fp32 kernels (bit-identical to
    precision="f32", and as close to the fp64 port as torch's own fp32 forward), "refuse" raises, "ignore" runs the split path."""
    from mapf_gpt_amd.model import build_model
    from oracle import gpt_oracle
    rng = np.random.Generator(np.random.PCG64(5))
    rows = np.load(os.path.join(GOLDEN, "gptbig_6M_s1.npz"))["tokens"][:16]
    tok = torch.from_numpy(rows).cuda()
    ok = build_model("6M", seed=0, max_rows=16, precision="f16x3")
    assert ok.envelope()["state"] == "undecided"
    ok.logits_tokens(tok)
    e = ok.envelope()
    assert e["state"] == "inside" and e["effective_precision"] == "f16x3" and e["probe_err"] <= 1e-5 and e["max_abs_w"] < 0.2, e
    # the probe covers BOTH call regimes (the kernels of calls <= 128 rows and of larger calls differ in arithmetic), and its
    # bar is absolute (1e-5) only while the logits are of order one
    assert e["probe_err_small_calls"] is not None and e["probe_err_large_calls"] is not None, e
    assert e["probe_err"] == max(e["probe_err_small_calls"], e["probe_err_large_calls"]) and 0 < e["probe_max_logit"] < 3.0, e
    assert e["probe_tol"] == pytest.approx(max(1e-5, 64 * 2.0 ** -23 * e["probe_max_logit"]), rel=1e-5), e
    # ... and the large-call error it saw is the error a large call has: the same 16 rows as a chunk of a 200-row call
    big = build_model("6M", seed=0, max_rows=256, precision="f16x3", envelope="ignore")
    pad = torch.from_numpy(np.concatenate([rows] * 13)[:200]).cuda()
    exact = build_model("6M", seed=0, max_rows=256, precision="f32")
    d_large = float((big.logits_tokens(pad) - exact.logits_tokens(pad)).abs().max())
    assert d_large <= 1e-5 and e["probe_err_large_calls"] <= 1e-5, (d_large, e)
    record_parity(test="envelope_probe_regimes", shape="6M", probe_small=e["probe_err_small_calls"], probe_large=e["probe_err_large_calls"],
                  large_call_200_rows=d_large, probe_tol=e["probe_tol"], probe_max_logit=e["probe_max_logit"])
    sd = weights.synthetic_state_dict("6M", seed=3)
    for k, v in sd.items():
        if v.ndim == 2 and "wte" not in k and "wpe" not in k:
            v[rng.random(v.shape) < 0.01] *= 100.0
    guarded = build_model("6M", precision="f16x3", max_rows=16, state_dict=sd)
    a = guarded.logits_tokens(tok).cpu().numpy()
    e = guarded.envelope()
    assert e["state"] == "outside" and e["effective_precision"] == "f32" and e["max_abs_w"] > 2.5, e
    exact = build_model("6M", precision="f32", max_rows=16, state_dict=sd).logits_tokens(tok).cpu().numpy()
    assert np.array_equal(a, exact), "the guard must hand the request to the exact-fp32 kernels"
    raw = build_model("6M", precision="f16x3", max_rows=16, state_dict=sd, envelope="ignore").logits_tokens(tok).cpu().numpy()
    ref64 = gpt_oracle.forward_logits(sd, weights.model_args("6M"), rows, dtype=torch.float64).numpy()
    ref32 = gpt_oracle.forward_logits(sd, weights.model_args("6M"), rows).numpy().astype(np.float64)
    e_ref, e_guard, e_raw = np.abs(ref32 - ref64).max(), np.abs(a - ref64).max(), np.abs(raw - ref64).max()
    record_parity(test="envelope_x100", shape="6M", e_guarded_vs_fp64=e_guard, e_split_vs_fp64=e_raw, e_torch_fp32_vs_fp64=e_ref,
                  max_abs_logit=float(np.abs(ref64).max()), max_abs_w=e["max_abs_w"], max_rms_w=e["max_rms_w"])
    assert e_guard <= max(TOL, 2.0 * e_ref), f"guarded {e_guard:.3e}, torch fp32 {e_ref:.3e} (split path unguarded: {e_raw:.3e})"
    with pytest.raises(RuntimeError, match="envelope"):
        build_model("6M", precision="f16x3", max_rows=16, state_dict=sd, envelope="refuse").logits_tokens(tok)
    x8 = build_model("6M", seed=0, scale=8.0, precision="f16x3", max_rows=16)
    x8.logits_tokens(tok)
    assert x8.envelope()["state"] == "outside" and x8.envelope()["max_rms_w"] > 0.1


def test_envelope_bar_is_relative_once_the_logits_are_large():
    """a checkpoint whose logits are of order ten (what trained checkpoints produce) sits 1e-5 .. 1e-4 from ANY fp32 forward by
    rounding alone (the reference's own fp32 run is 3.6e-5 from its fp64 run at |logits| ~ 4).  The probe's bar is
    max(1e-5, 64 eps |logit|max): wte x 6 scales the tied head's logits without leaving the weight-statistics envelope (wte / wpe are not block
    matrices), the checkpoint stays inside, and the split path is as close to the fp64 port as torch's own fp32 forward."""
    from mapf_gpt_amd.model import build_model
    from oracle import gpt_oracle
    rows = np.load(os.path.join(GOLDEN, "gptbig_6M_s1.npz"))["tokens"][:16]
    tok = torch.from_numpy(rows).cuda()
    sd = {k: np.array(v, copy=True) for k, v in weights.synthetic_state_dict("6M", seed=0).items()}
    sd["transformer.wte.weight"] = sd["transformer.wte.weight"] * np.float32(6.0)
    sd["lm_head.weight"] = sd["transformer.wte.weight"]                      # tied
    net = build_model("6M", precision="f16x3", max_rows=16, state_dict=sd)
    got = net.logits_tokens(tok).cpu().numpy()
    e = net.envelope()
    ref64 = gpt_oracle.forward_logits(sd, weights.model_args("6M"), rows, dtype=torch.float64).numpy()
    ref32 = gpt_oracle.forward_logits(sd, weights.model_args("6M"), rows).numpy().astype(np.float64)
    e_ref, e_got = np.abs(ref32 - ref64).max(), np.abs(got - ref64).max()
    record_parity(test="envelope_relative_bar", shape="6M", max_abs_logit=float(np.abs(ref64).max()), probe_err=e["probe_err"], probe_tol=e["probe_tol"],
                  probe_max_logit=e["probe_max_logit"], e_split_vs_fp64=e_got, e_torch_fp32_vs_fp64=e_ref, state=e["state"])
    assert e["probe_max_logit"] > 2.0 and e["probe_tol"] > 1e-5, e
    assert e["state"] == "inside" and e["effective_precision"] == "f16x3", e
    assert e_got <= max(TOL, 3.0 * e_ref), f"split path {e_got:.3e} against the fp64 port, torch fp32 {e_ref:.3e}"


def test_first_f16x3_forward_refuses_a_stream_capture():
    """the envelope probe allocates and synchronises; inside a capture the call must fail with a clear state error instead of
    invalidating the capture, and the same forward works once the envelope is decided."""
    from mapf_gpt_amd.model import build_model
    rows = np.load(os.path.join(GOLDEN, "gptbig_6M_s1.npz"))["tokens"][:4]
    tok = torch.from_numpy(rows).cuda()
    net = build_model("tiny", seed=0, max_rows=4, precision="f16x3")
    out = torch.empty((4, 67), dtype=torch.float32, device="cuda")
    s = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with pytest.raises(RuntimeError, match="captur"):
            with torch.cuda.graph(g, stream=s):
                net.logits_tokens(tok, out=out)
    torch.cuda.synchronize()
    assert net.envelope()["state"] == "undecided"
    net.logits_tokens(tok, out=out)
    torch.cuda.synchronize()
    assert net.envelope()["state"] == "inside"


@pytest.mark.parametrize("offset", [0.5, 2.0])
def test_folded_layernorm_with_a_large_per_token_mean(offset, monkeypatch):
    """the 85M shape's bf16 chain folds LayerNorm into the GEMMs (DESIGN, HISTORY 11.9): the residual epilogues leave (sum, sum of
    squares) partials and ln_finalize_kernel turns them into (mean, rstd).  Since round 5 the partials are those of the SHIFTED row (x minus
    its mean at the LayerNorm before), so the one-pass variance is taken about a nearly centred value.  A constant added to wte gives the
    residual stream a per-token mean of 9 / 36 standard deviations (C = 768, 12 heads, 2 layers: the 85M chain); the folded path must stay in
    the class of ln_pack_kernel (MGPT_LN_FOLD=0, two-pass LayerNorm) against the fp64 port: measured 0.036 vs 0.026 at 9 and 0.103 vs 0.059 at 36
    standard deviations (|logits| up to 4.2 / 9.5) -- the difference is the bf16 rounding of x minus a shift that is one residual update old, not
    the variance formula (0.106 before the partials were shifted)."""
    from mapf_gpt_amd.model import GPT, GPTConfig
    from tests.helpers import load_tok
    rows = load_tok("mazes000")["tokens"][7, :3]
    args = weights.model_args(dict(n_layer=2, n_head=12, n_embd=768))
    sd = {k: np.array(v, copy=True) for k, v in weights.synthetic_state_dict(args, seed=11, scale=2.0).items()}
    sd["transformer.wte.weight"] = sd["transformer.wte.weight"] + np.float32(offset)
    sd["lm_head.weight"] = sd["transformer.wte.weight"]                      # tied
    ref = gpt_oracle.forward_logits(sd, args, rows, dtype=torch.float64).numpy()
    err = {}
    for fold in ("1", "0"):
        monkeypatch.setenv("MGPT_LN_FOLD", fold)
        net = GPT(GPTConfig(**args), max_rows=4, precision="bf16")
        net.load_state_dict(sd)
        err[fold] = float(np.abs(net.logits_tokens(torch.from_numpy(rows).cuda()).cpu().numpy() - ref).max())
        del net
    x0 = np.asarray(sd["transformer.wte.weight"][0] + sd["transformer.wpe.weight"][0])
    record_parity(test="ln_fold_large_mean", offset=offset, mean_over_std=float(abs(x0.mean()) / x0.std()), e_fold=err["1"], e_ln_pack=err["0"],
                  max_abs_logit=float(np.abs(ref).max()))
    assert err["1"] <= 2.0 * err["0"] + 5e-3, f"fold {err['1']:.3e} vs ln_pack {err['0']:.3e} at |mean|/std = {abs(x0.mean()) / x0.std():.1f}"
