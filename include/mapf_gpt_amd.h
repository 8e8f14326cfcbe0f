/*
 * mapf_gpt_amd.h -- C ABI of libmapf_gpt_amd.so, the MI355X (gfx950) implementation of
 * MAPF-GPT's per-step hot path:  env step -> observation tokenizer -> GPT forward -> action.
 *
 * This is the drop-in boundary.  Every entry point names the reference interface it replaces
 * (file:line in CognitiveAISystems/MAPF-GPT).  Conventions:
 *   - plain C types only; every `d_*` pointer is a DEVICE pointer (HBM) owned by the caller;
 *   - every call is stream-ordered on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) and returns without synchronising unless its comment says otherwise;
 *   - returns MGPT_OK or an error code; no exception crosses the ABI; mgpt_last_error() gives a
 *     thread-local message for the last failing call;
 *   - contexts own their internal device state; the caller owns every buffer it passes in;
 *   - a context may be used from one thread at a time; different contexts are independent
 *     (no hidden globals besides the per-thread error string);
 *   - the library never falls back to the CPU: with no HIP device every compute call fails.
 *
 * Coordinates are (row, col) int16 pairs in the PADDED map frame (5 obstacle cells on every
 * side), exactly what the reference hands to its tokenizer (inference.py:130-131).
 */
#ifndef MAPF_GPT_AMD_H
#define MAPF_GPT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGPT_OK 0
#define MGPT_ERR_ARG 1          /* bad argument (NULL, size, unsupported shape)              */
#define MGPT_ERR_HIP 2          /* a HIP runtime call failed (message has hipGetErrorString) */
#define MGPT_ERR_STATE 3        /* call order violated (e.g. tokens before create_agents)    */
#define MGPT_ERR_UNSUPPORTED 4  /* shape/precision outside what the kernels implement        */

#define MGPT_CONTEXT 256        /* tokens per observation row (observation_generator.cpp:386) */
#define MGPT_VOCAB 67           /* observation_generator.cpp:321-344                          */
#define MGPT_NUM_ACTIONS 5      /* model.py:250-252                                           */

/* thread-local text of the last error on this thread ("" if none) */
const char *mgpt_last_error(void);
/* ABI version (major*1000 + minor) */
int mgpt_abi_version(void);
/* number of visible HIP devices (0 and MGPT_OK when there is none) */
int mgpt_device_count(int *count);

/* ------------------------------------------------------------------------------------------
 * Tokenizer: replaces the pybind module `observation_generator`
 * (observation_generator.cpp:546-563), batched over many env instances.
 * ------------------------------------------------------------------------------------------ */

/* = struct InputParameters, observation_generator.h:22-40 / ctor args cpp:551.  All fields are honoured (kernel arguments: the
 * vocabulary -L..L -> 0..2L, -4L, -2L, +2L, six action letters, sixteen bit strings, "!" = 2L+26 of Encoder::Encoder cpp:321-350,
 * the (2 obs_radius + 1)^2 window cpp:288-311, the (2 agents_radius + 1)^2 neighbour scan and the num_agents records of
 * 5 + num_previous_actions tokens cpp:352-389, 487-512) within what the kernels' layouts hold:
 *   1 <= cost2go_value_limit <= 100, 1 <= num_agents <= 16, 0 <= num_previous_actions <= 5, 1 <= obs_radius <= 5,
 *   0 <= agents_radius <= min(5, cost2go_value_limit)   (beyond the limit the reference itself throws: int_vocab.at, cpp:358-359),
 *   (2 obs_radius + 1)^2 + num_agents (5 + num_previous_actions) <= 256 and context_size == 256: the reference pads every row to
 *   256 tokens whatever context_size says (cpp:386-387) and returns LONGER rows when they do not fit; this ABI's rows are 256 tokens.
 * Anything else -> MGPT_ERR_UNSUPPORTED.  Pinned by rows written by the compiled reference for six non-default sets
 * (tests/golden/tokp_*.npz).  The reference itself only ever passes (20, 13, 5, 256, 5, 5) (inference.py:15-29).
 * save_cost2go only steers the reference's CPU caching of distance fields (cpp:43-132) and is ignored.  grid_step (> 0,
 * and >= max(H, W) / 256) is the side of the reference's cost-to-go tiles: it changes no token except through the one
 * unseeded corner cell of an agent's cached 2*grid_step + 1 window (cpp:178-198), which is reproduced. */
typedef struct mgpt_input_parameters {
    int32_t cost2go_value_limit;
    int32_t num_agents;
    int32_t num_previous_actions;
    int32_t context_size;
    int32_t obs_radius;
    int32_t agents_radius;
    int32_t grid_step;
    int32_t save_cost2go;
} mgpt_input_parameters;

typedef struct mgpt_tokenizer mgpt_tokenizer;

/* = ObservationGenerator::ObservationGenerator(grid, cfg), observation_generator.h:112-119, for
 * n_inst instances x n_agents agents sharing one padded frame H x W.  `n_grids` distinct obstacle
 * maps are stored; instance i uses map i % n_grids (n_grids == 1: one shared map). */
int mgpt_tokenizer_create(mgpt_tokenizer **out, const mgpt_input_parameters *cfg,
                          int n_inst, int n_agents, int H, int W, int n_grids);
int mgpt_tokenizer_destroy(mgpt_tokenizer *tok);
/* = the size of Encoder's vocabulary (cpp:321-350): 2 cost2go_value_limit + 27 -- the integers -limit .. limit, the three sentinels,
 * six actions, sixteen direction strings and "!".  67 for the reference's limit of 20 = the vocab_size of every released model (model.py:110).
 * A policy whose embedding has fewer rows cannot take this tokenizer's rows: the reference's nn.Embedding raises IndexError there,
 * mgpt_step_create returns MGPT_ERR_UNSUPPORTED, and mgpt_gpt_forward does not look at the ids it is given. */
int mgpt_tokenizer_vocab_size(const mgpt_tokenizer *tok, int *out);

/* the `grid` ctor argument (h:112): d_grids = uint8 [n_grids, H, W], non-zero = blocked */
int mgpt_tokenizer_set_grids(mgpt_tokenizer *tok, const uint8_t *d_grids, void *stream);

/* = create_agents(positions, goals), cpp:391-410: history <- "n"x5, one BFS distance-to-goal field
 * per agent (cpp:200-286), greedy-direction bits (cpp:412-430).
 * d_pos, d_goal: int16 [n_inst, n_agents, 2]. */
int mgpt_tokenizer_create_agents(mgpt_tokenizer *tok, const int16_t *d_pos, const int16_t *d_goal,
                                 void *stream);

/* = update_agents(positions, goals, actions), cpp:432-485.  d_actions: int32 [n_inst, n_agents],
 * the policy's previous INTENDED actions (inference.py:140-144,168); values outside 0..4 append "n".
 * goals_may_change != 0 re-runs the BFS of every agent whose goal differs (cpp:464-468);
 * 0 skips that comparison (on_target = "nothing": goals never change). */
int mgpt_tokenizer_update_agents(mgpt_tokenizer *tok, const int16_t *d_pos, const int16_t *d_goal,
                                 const int32_t *d_actions, int goals_may_change, void *stream);
/* The same for a subset of the instances: d_active uint8 [n_inst] (NULL = all); instances with 0 keep their state (position,
 * action history, goal, distance field) untouched -- what the adapter needs when one act_batch call (inference.py:151-172)
 * presents only some of the environment slots that share a tokenizer context. */
int mgpt_tokenizer_update_agents_masked(mgpt_tokenizer *tok, const int16_t *d_pos, const int16_t *d_goal,
                                        const int32_t *d_actions, const uint8_t *d_active, int goals_may_change, void *stream);

/* = generate_observations(), cpp:516-528.  d_tokens: uint8 [n_inst * n_agents, 256], row-major,
 * row = inst * n_agents + agent.  Token ids are < 2 * cost2go_value_limit + 27 (67 by default) and fit a byte (the reference
 * widens them to int64 only for torch.nn.Embedding, inference.py:91,97).  Positions of one instance must be
 * pairwise distinct (always true for env states). */
int mgpt_tokenizer_generate_observations(mgpt_tokenizer *tok, uint8_t *d_tokens, void *stream);

/* test/debug read-backs (device pointers into the context's state, valid until destroy):
 * distance fields uint16 [n_inst, n_agents, H, W]; agent records 16 B each
 * {int16 pos_r,pos_c,goal_r,goal_c; uint8 hist[5]; uint8 next; uint8 org[2]}, org = origin (row / 64, col / 64) of the
 * partial cost-to-go window the reference would hold for the agent (observation_generator.cpp:204-207, 469-477). */
int mgpt_tokenizer_state(mgpt_tokenizer *tok, const uint16_t **d_dist, const void **d_records);
/* same state copied into caller-owned device buffers (either may be NULL) */
int mgpt_tokenizer_copy_state(mgpt_tokenizer *tok, uint16_t *d_dist_out, void *d_records_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Environment step: the part of the loop the reference delegates to POGEMA
 * (experiment_setup/create_env.py:36-46; call shape create_env.py:14-20).  POGEMA is not in the
 * reference tree: this implements the spec in DESIGN.md ("Env step spec"), parity unpinned.
 * ------------------------------------------------------------------------------------------ */
typedef struct mgpt_env mgpt_env;

int mgpt_env_create(mgpt_env **out, int n_inst, int n_agents, int H, int W, int n_grids,
                    int max_episode_steps);
int mgpt_env_destroy(mgpt_env *env);
int mgpt_env_set_grids(mgpt_env *env, const uint8_t *d_grids, void *stream);
/* = env.reset(): copies starts/goals (int16 [n_inst,n_agents,2]) and clears episode counters. */
int mgpt_env_reset(mgpt_env *env, const int16_t *d_pos, const int16_t *d_goal, void *stream);
/* = env.step(actions): d_actions int32 [n_inst, n_agents] in {0 wait,1 up,2 down,3 left,4 right}.
 * Instances that are already done ignore the actions. */
int mgpt_env_step(mgpt_env *env, const int32_t *d_actions, void *stream);
/* device views of the env state: pos/goal int16 [n_inst,n_agents,2]; done uint8 [n_inst]
 * (1 terminated = all on goal, 2 truncated = max_episode_steps reached); */
int mgpt_env_state(mgpt_env *env, const int16_t **d_pos, const int16_t **d_goal, const uint8_t **d_done);
/* the same state copied (stream-ordered, device to device) into caller-owned buffers; any may be NULL */
int mgpt_env_copy_state(mgpt_env *env, int16_t *d_pos_out, int16_t *d_goal_out, uint8_t *d_done_out, void *stream);
/* The reference-shaped list API (create_env.py:14-15: step(list) -> lists) in one call: h_actions int32 [n_inst * n_agents] on
 * the HOST (NULL: no step, just read the state back, e.g. after reset), state back into the HOST buffer
 * h_state_out = [pos int16 n*2 | goal int16 n*2 | done uint8 n_inst], n = n_inst * n_agents.  Synchronises `stream`. */
int mgpt_env_step_host(mgpt_env *env, const int32_t *h_actions, uint8_t *h_state_out, void *stream);
/* per-instance episode metrics, float32 [n_inst, 6] = {CSR, ISR, SoC, makespan, ep_length, avg_agents_density}
 * (the keys the reference's result tables use: eval_configs/05-puzzles/05-puzzles.yaml:49-58; avg_agents_density =
 * POGEMA's AgentsDensityWrapper of experiment_setup/create_env.py:38: mean over the reset observation and every step of
 * the agents' mean [agents in the 11 x 11 window / traversable cells of the window]). */
int mgpt_env_metrics(mgpt_env *env, float *d_metrics, void *stream);

/* Collision-rule switches (bit mask, default 0 = the spec of DESIGN.md section 4).  The two places where that spec rests on
 * recalled -- not verifiable offline -- POGEMA behaviour are switchable, so that pinning against fixtures captured from POGEMA
 * (tests/golden/make_golden_env.py) is a flag flip, not a rewrite:
 *   MGPT_ENV_RULE_NO_FOLLOW (1)    a move into a cell another agent occupies at the start of the step becomes wait even if
 *                                  that agent leaves it (default: following a leaving agent is allowed);
 *   MGPT_ENV_RULE_LOWEST_WINS (2)  a contested empty cell goes to the lowest-id mover (default: every claimant waits).
 * A captured step graph stays valid: the mask is a kernel argument of the next capture (the call bumps the env generation). */
#define MGPT_ENV_RULE_NO_FOLLOW 1
#define MGPT_ENV_RULE_LOWEST_WINS 2
int mgpt_env_set_rules(mgpt_env *env, int rules);

/* Lifelong mode (POGEMA on_target="restart", experiment_setup/create_env.py:28-32): d_goal_queue is int16
 * [n_inst][n_agents][queue_len][2] (padded coords); an agent that ends a step on its goal takes the next entry of its
 * queue (wrapping) and the arrival is counted; episodes then end by truncation only.  NULL / 0 switches back to
 * on_target="nothing".  Call before mgpt_env_reset.  d_reached_out: int32 [n_inst][n_agents] arrivals so far. */
int mgpt_env_set_lifelong(mgpt_env *env, const int16_t *d_goal_queue, int queue_len, void *stream);
int mgpt_env_lifelong_counts(mgpt_env *env, int32_t *d_reached_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Policy forward: replaces GPT.forward / GPT.act (mapf_gpt/model.py:167-189, 244-260).
 * ------------------------------------------------------------------------------------------ */
typedef struct mgpt_gpt mgpt_gpt;

#define MGPT_PREC_F32 0     /* exact fp32 MFMA (v_mfma_f32_32x32x2_f32): the 1e-5 parity path        */
#define MGPT_PREC_F16X3 1   /* split-fp16 (hi/lo, 3 MFMA passes, fp32 accumulate): ~fp32 accuracy     */
#define MGPT_PREC_BF16 2    /* single-pass bf16 MFMA, fp32 accumulate: the reference's autocast mode  */

/* = GPTConfig + GPT.__init__ (model.py:107-145): vocab 67; dropout is the identity at inference (inference.py:85 net.eval()).
 * bias (model.py:115; False in every released config): a checkpoint has bias vectors iff mgpt_gpt_set_param is handed any
 * "*.bias" tensor -- see there.  max_rows = largest batch one forward call will see (activation workspace is sized for it). */
int mgpt_gpt_create(mgpt_gpt **out, int n_layer, int n_head, int n_embd, int block_size, int max_rows);
int mgpt_gpt_destroy(mgpt_gpt *gpt);

/* = load_state_dict (inference.py:83).  `name` is the reference state_dict key
 * ("transformer.h.3.attn.c_attn.weight", ...; lm_head.weight is tied to transformer.wte.weight,
 * model.py:138, either name sets both).  data: float32, HOST or DEVICE pointer (is_device),
 * n_elem must match the parameter's size.  Synchronous.
 * GPTConfig.bias = True checkpoints (model.py:14-17 LayerNorm bias; model.py:29,31,79,81 nn.Linear bias): their seven kinds of
 * "*.bias" tensors ("transformer.ln_f.bias", "transformer.h.N.{ln_1,ln_2}.bias", "...attn.c_attn.bias", "...attn.c_proj.bias",
 * "...mlp.c_fc.bias", "...mlp.c_proj.bias") are accepted; once one is set, mgpt_gpt_finalize wants all of them.  Only the
 * MGPT_PREC_F32 kernels carry bias terms: MGPT_PREC_F16X3 requests follow the envelope policy (the checkpoint counts as outside:
 * fallback = served in fp32, refuse / ignore = MGPT_ERR_UNSUPPORTED), MGPT_PREC_BF16 requests return MGPT_ERR_UNSUPPORTED. */
int mgpt_gpt_set_param(mgpt_gpt *gpt, const char *name, const float *data, int64_t n_elem, int is_device);
/* call once after all parameters are set (builds the packed operand planes of the chosen precisions) */
int mgpt_gpt_finalize(mgpt_gpt *gpt);

/* = GPT.forward(idx)[0][:, -1, :] (model.py:167-189): d_tokens uint8 [rows, 256] ->
 * d_logits float32 [rows, 67] (logits of the LAST position, the only ones the reference computes
 * at inference, model.py:186). */
int mgpt_gpt_forward(mgpt_gpt *gpt, const uint8_t *d_tokens, int rows, float *d_logits,
                     int precision, void *stream);

/* = GPT.forward(idx) for idx of T <= block_size tokens per row (model.py:167-175, "Cannot forward sequence of length t, block size is only
 * block_size" :170): d_tokens uint8 [rows, T] -> d_logits float32 [rows, 67], the logits of position T - 1.  mgpt_gpt_create accepts
 * block_size 1 .. 256 (transformer.wpe.weight is [block_size, C]); models with block_size < 256 are served by this entry point only.  Exact-fp32
 * kernels whatever the model's usual precision: the hot path (tokenizer rows, inference.py:145) is 256 tokens and never comes here. */
int mgpt_gpt_forward_t(mgpt_gpt *gpt, const uint8_t *d_tokens, int rows, int T, float *d_logits, void *stream);

/* = GPT.act (model.py:244-260): softmax over logits[:5]; do_sample != 0 draws from it with the
 * library's counter-based RNG keyed by (seed, step, row0 + row) -- torch.multinomial's stream is
 * device-specific and not reproduced -- else arg-max.  row0 = GLOBAL id of this call's first row, so
 * that a shard of a larger job (instances split over GPUs) draws exactly what the unsharded job
 * draws for the same rows.  d_actions int32 [rows].
 * d_logits may be NULL, else float32 [rows, 67] is also written. */
int mgpt_gpt_act(mgpt_gpt *gpt, const uint8_t *d_tokens, int rows, int32_t *d_actions, float *d_logits,
                 int do_sample, uint64_t seed, uint64_t step, uint64_t row0, int precision, void *stream);

/* test/debug: copy an fp32-path workspace buffer (valid after mgpt_gpt_forward with MGPT_PREC_F32):
 * which 0 = residual stream x [rows*256, C], 1 = last LayerNorm output, 2 = q|k|v planes
 * [3][rows][n_head][256][hs], 3 = MLP hidden [rows*256, 4C]; n_elem floats from the start. */
int mgpt_gpt_debug_copy(mgpt_gpt *gpt, int which, float *d_out, int64_t n_elem, void *stream);

/* test/debug: raw bytes of a 16-bit-path workspace buffer (valid after a forward in `precision`):
 * which 0 = LayerNorm stats float2[rows*256]; 1/2 = q|k planes hi/lo; 3/4 = v^T planes hi/lo;
 * 5/6 = attention output planes hi/lo; 7/8 = MLP hidden planes hi/lo (lo only for MGPT_PREC_F16X3). */
int mgpt_gpt_debug_copy_raw(mgpt_gpt *gpt, int precision, int which, void *d_out, int64_t nbytes, void *stream);

/* Precision envelope of MGPT_PREC_F16X3 (split-fp16 MFMA passes, fp32 accumulate).  The 1e-5 logit bar of that mode is established
 * on checkpoints whose block matrices stay within max|w| <= MGPT_ENVELOPE_MAX_W and rms(w) <= MGPT_ENVELOPE_MAX_RMS (N(0, 0.02)-family
 * weights up to x4 in scale and x20 outliers on 1 % of the entries: tests/test_gpu_parity_r2.py, profiles/r0*_parity_records.jsonl);
 * beyond that (x8 scale, x100 outliers) it is 2e-5 .. 1e-3.  So the envelope is a run-time property of the loaded checkpoint:
 * mgpt_gpt_finalize measures max|w| and rms(w) of every 2-D block matrix, and the first MGPT_PREC_F16X3 forward runs
 * MGPT_ENVELOPE_PROBE_ROWS fixed pseudo-random token rows through the exact-fp32 path and through the split path in BOTH of its call
 * regimes (the kernels that serve calls of <= 128 rows and those of larger calls differ in arithmetic) and compares the logits.  The bar
 * is max(MGPT_ENVELOPE_PROBE_TOL, MGPT_ENVELOPE_PROBE_REL * max |fp32 logit|): 1e-5 absolute while the logits are of order one, where
 * the 1e-5 of the reference comparison was established; relative beyond (64 eps_f32: the fp32 forward itself moves by several
 * eps * |logit| between summation orders -- the reference's own fp32 run is 3.6e-5 from its fp64 run at |logits| ~ 4, SURVEY.md
 * appendix B).  The first such forward synchronises its stream and cannot sit inside a stream capture (MGPT_ERR_STATE).
 * A checkpoint outside either test is served according to the policy:
 *   MGPT_ENVELOPE_FALLBACK (default)  MGPT_PREC_F16X3 requests run the MGPT_PREC_F32 kernels; one line on stderr says so
 *   MGPT_ENVELOPE_REFUSE              MGPT_PREC_F16X3 requests fail with MGPT_ERR_UNSUPPORTED
 *   MGPT_ENVELOPE_IGNORE              no test, the split path runs (parity experiments)
 * Follows reference mapf_gpt/inference.py:72-85 (a loaded checkpoint is served as is, in fp32). */
#define MGPT_ENVELOPE_FALLBACK 0
#define MGPT_ENVELOPE_REFUSE 1
#define MGPT_ENVELOPE_IGNORE 2
#define MGPT_ENVELOPE_MAX_W 2.5f
#define MGPT_ENVELOPE_MAX_RMS 0.1f
#define MGPT_ENVELOPE_PROBE_ROWS 8
#define MGPT_ENVELOPE_PROBE_TOL 1e-5f
#define MGPT_ENVELOPE_PROBE_REL 7.62939453125e-6f      /* 64 * 2^-23 */
int mgpt_gpt_set_envelope_policy(mgpt_gpt *gpt, int policy);
/* out[0] = max|w|, out[1] = max rms(w) over the block matrices (valid after finalize), out[2] = probe error (-1 before the first
 * MGPT_PREC_F16X3 forward under a policy other than IGNORE); *state: 0 not decided yet, 1 inside, 2 outside */
int mgpt_gpt_envelope(mgpt_gpt *gpt, float *out3, int *state);
/* the probe in detail: out[0] = error of the small-call kernels, out[1] = of the large-call kernels (-1 each before the probe),
 * out[2] = the bar they were held to, out[3] = max |logit| of the fp32 path over the probe rows */
int mgpt_gpt_envelope_probe(mgpt_gpt *gpt, float *out4);

/* test/debug: event counters of the policy kernels since the last reset (synchronises the device).
 * which 0 = waves of the C = 256 / C = 160 attention kernels that threw a head of the pipelined key-tile loop (one softmax reference
 * per query and head) away and redid it with the exact running-maximum loop, because a half-row sum of exp2(s - ref) left the
 * fp16 range of the P planes (mapf_gpt_amd/csrc/gpt_kernels_c256a.h).  reset != 0 zeroes the counter after reading it. */
int mgpt_gpt_debug_counter(int which, uint64_t *value, int reset);

/* mgpt_gpt_act with the RNG step read from device memory when the kernel runs (*d_step): for callers that replay the call
 * from a captured hipGraph, where a per-step scalar argument would be frozen */
int mgpt_gpt_act_dev(mgpt_gpt *gpt, const uint8_t *d_tokens, int rows, int32_t *d_actions, float *d_logits,
                     int do_sample, uint64_t seed, const uint64_t *d_step, uint64_t row0, int precision, void *stream);

/* sampling alone (same RNG and key as mgpt_gpt_act), for callers that already hold logits */
int mgpt_sample_actions(const float *d_logits, int rows, int32_t *d_actions, int do_sample,
                        uint64_t seed, uint64_t step, uint64_t row0, void *stream);

/* ------------------------------------------------------------------------------------------
 * One whole environment step = the body of the reference's episode loop (example.py:63-65 around
 * inference.py:151-172):  update_agents(pos, goal, last actions) -> generate_observations -> act -> env.step.
 * The launch sequence is static, so after one eager step it is captured as a hipGraph and replayed: one call, no
 * per-launch host cost.  The contexts stay owned by the caller and must outlive the step object.
 *   create: rows = n_inst * n_agents of the tokenizer/env pair; precision / do_sample / seed / row0 as in mgpt_gpt_act.
 *   run:    d_tokens uint8 [rows, 256] scratch (holds this step's observation rows afterwards); d_actions int32 [rows]
 *           IN: the previous intended actions (-1 on the first step, inference.py:140), OUT: this step's actions, already
 *           executed by the env.  goals_may_change as in mgpt_tokenizer_update_agents.  use_graph = 0 forces eager launches
 *           (identical results); eager is also used while the timing hooks are enabled.
 *   reset:  sets the RNG step counter (call with 0 when an episode starts; the k-th run after it draws with step k) and
 *           drops the captured graph -- call it after anything that re-allocates inside the contexts (mgpt_env_set_lifelong).
 * ------------------------------------------------------------------------------------------ */
typedef struct mgpt_step mgpt_step;
int mgpt_step_create(mgpt_step **out, mgpt_tokenizer *tok, mgpt_gpt *gpt, mgpt_env *env, int rows, int precision,
                     int do_sample, uint64_t seed, uint64_t row0);
int mgpt_step_destroy(mgpt_step *step);
int mgpt_step_reset(mgpt_step *step, uint64_t step0, void *stream);
int mgpt_step_run(mgpt_step *step, uint8_t *d_tokens, int32_t *d_actions, int goals_may_change, int use_graph, void *stream);

/* ------------------------------------------------------------------------------------------
 * Dataset-side bulk tokenizer: replaces dataset/tokenizer/generate_observations.py:8-92 with its two native modules
 * (cost2go.cpp:33-88 precompute_cost2go / generate_cost2go_obs, encoder.cpp:88-127 Encoder::encode) -- every
 * (agent, timestep) of a logged episode becomes one 256-token row.
 *   create:   d_grid uint8 [H][W] PADDED map (non-zero = blocked); builds the all-pairs BFS table of the map
 *             (= the reference's cost2go_data dict, cached per map_name at :47-54).  Synchronises `stream` once.
 *   tokenize: d_paths int16 [n_agents][n_steps][2] = cells visited (padded coords, get_agent_paths :159-177; the goal of an
 *             agent is its last cell); d_tokens uint8 [n_agents][n_steps][256], agent-major like the reference's append
 *             order (:70-90).  Lifelong logs (per-step goals, :55-60) are not supported.  Relative goals beyond +-20 are
 *             clamped (the reference's unordered_map::at throws there).
 * ------------------------------------------------------------------------------------------ */
typedef struct mgpt_dataset mgpt_dataset;
int mgpt_dataset_create(mgpt_dataset **out, const uint8_t *d_grid, int H, int W, void *stream);
int mgpt_dataset_destroy(mgpt_dataset *ds);
int mgpt_dataset_tokenize(mgpt_dataset *ds, int n_agents, int n_steps, const int16_t *d_paths, uint8_t *d_tokens, void *stream);
/* the same with the two options of the reference's generator:
 *   d_goals  int16 [n_agents][n_steps][2] or NULL: lifelong logs (generate_observations.py:55-60,143-153) -- the goal every
 *            agent pursued at every timestep (relative goal + greedy-direction bits of its records); the cost-to-go window
 *            stays that of the path's last cell, as in the reference (:75-78);
 *   only_obstacles != 0: the mask_cost2go ablation (cost2go.cpp:52-62) -- window cells become the integers 0 / 1 (blocked). */
int mgpt_dataset_tokenize_ex(mgpt_dataset *ds, int n_agents, int n_steps, const int16_t *d_paths, const int16_t *d_goals,
                             int only_obstacles, uint8_t *d_tokens, void *stream);

/* ------------------------------------------------------------------------------------------
 * Kernel timing hooks (bench.py's live roofline): when enabled, the library brackets every kernel
 * class with hipEvents on the launch stream.  mgpt_prof_read synchronises the device.
 * ------------------------------------------------------------------------------------------ */
#define MGPT_PROF_MAX 32
int mgpt_prof_enable(int on);
int mgpt_prof_reset(void);
/* names[i] (static strings), total_ms[i], launches[i] for i < *n (n in: capacity, out: used) */
int mgpt_prof_read(const char **names, float *total_ms, int64_t *launches, int *n);

#ifdef __cplusplus
}
#endif
#endif /* MAPF_GPT_AMD_H */
