// step.hip -- one whole environment step behind ONE entry point, replayed as a hipGraph:
//     tokenizer.update_agents(pos, goal, last_actions)   observation_generator.cpp:432-485
//     tokens  = tokenizer.generate_observations()        observation_generator.cpp:516-528
//     actions = policy.act(tokens)                       model.py:244-260
//     env.step(actions)                                  create_env.py:14-15
// (the loop of inference.py:151-172 + example.py:63-65).  The launch sequence of a step is static -- every data-dependent
// decision (dirty-flag BFS, done instances, rescale branches) lives inside the kernels -- so after one eager step (which
// builds the lazily packed weight planes) the sequence is captured once and replayed; the only per-step scalar, the RNG
// step counter, lives in device memory and is bumped by the graph's last node.  For small workloads (one 32-agent env:
// ~40 launches of a few microseconds each) this removes the per-launch host cost and the Python/ctypes round trips.
#include "common.h"

namespace mgpt {
bool prof_is_enabled();
}
using namespace mgpt;

// internal entry of gpt.hip: mgpt_gpt_act with the RNG step read from device memory
extern "C" int mgpt_gpt_act_dev(mgpt_gpt *g, const uint8_t *d_tokens, int rows, int32_t *d_actions, float *d_logits, int do_sample,
                                uint64_t seed, const uint64_t *d_step, uint64_t row0, int precision, void *stream);

struct mgpt_step {
    mgpt_tokenizer *tok = nullptr;
    mgpt_gpt *gpt = nullptr;
    mgpt_env *env = nullptr;
    int rows = 0, precision = 0, do_sample = 0;
    uint64_t seed = 0, row0 = 0;
    uint64_t *d_step = nullptr;                 // device: RNG step counter (model.py:257's generator state, in our counter form)
    const int16_t *d_pos = nullptr, *d_goal = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t cap_stream = nullptr;           // capture happens here (the caller's stream may be the NULL stream, which cannot capture)
    int eager_runs = 0;
    // what the captured graph was recorded with
    const uint8_t *cap_tokens = nullptr;
    const int32_t *cap_actions = nullptr;
    int cap_gmc = -1;
    bool capture_failed = false;
    uint64_t seen_gpt_gen = 0, seen_env_gen = 0; // generations of OUR contexts after our last step (0: never ran)
};

namespace {
__global__ void bump_kernel(uint64_t *ctr) { *ctr += 1; }
__global__ void set_kernel(uint64_t *ctr, uint64_t v) { *ctr = v; }

int step_body(mgpt_step *st, uint8_t *d_tokens, int32_t *d_actions, int gmc, hipStream_t s)
{
    int rc;
    if ((rc = mgpt_tokenizer_update_agents(st->tok, st->d_pos, st->d_goal, d_actions, gmc, s)) != MGPT_OK) return rc;
    if ((rc = mgpt_tokenizer_generate_observations(st->tok, d_tokens, s)) != MGPT_OK) return rc;
    if ((rc = mgpt_gpt_act_dev(st->gpt, d_tokens, st->rows, d_actions, nullptr, st->do_sample, st->seed, st->d_step, st->row0, st->precision,
                               s)) != MGPT_OK)
        return rc;
    if ((rc = mgpt_env_step(st->env, d_actions, s)) != MGPT_OK) return rc;
    hipLaunchKernelGGL(bump_kernel, dim3(1), dim3(1), 0, s, st->d_step);
    MGPT_LAUNCH_CHECK();
    return MGPT_OK;
}

void drop_graph(mgpt_step *st)
{
    if (st->exec) (void)hipGraphExecDestroy(st->exec);
    if (st->graph) (void)hipGraphDestroy(st->graph);
    st->exec = nullptr; st->graph = nullptr;
}
}  // namespace

extern "C" int mgpt_step_create(mgpt_step **out, mgpt_tokenizer *tok, mgpt_gpt *gpt, mgpt_env *env, int rows, int precision,
                                int do_sample, uint64_t seed, uint64_t row0)
{
    MGPT_REQUIRE(out && tok && gpt && env && rows > 0, MGPT_ERR_ARG, "bad argument");
    {   // a tokenizer with a larger value limit than the policy's vocabulary was built for emits ids the embedding does not have (the reference:
        // IndexError in nn.Embedding, model.py:126,172)
        int vocab = 0;
        const int rc = mgpt_tokenizer_vocab_size(tok, &vocab);
        if (rc != MGPT_OK) return rc;
        MGPT_REQUIRE(vocab <= MGPT_VOCAB, MGPT_ERR_UNSUPPORTED, "the tokenizer's vocabulary has %d tokens (cost2go_value_limit %d), the policy's embedding %d",
                     vocab, (vocab - 27) / 2, MGPT_VOCAB);
    }
    mgpt_step *st = new mgpt_step();
    st->tok = tok; st->gpt = gpt; st->env = env; st->rows = rows; st->precision = precision; st->do_sample = do_sample;
    st->seed = seed; st->row0 = row0;
    int rc = mgpt_env_state(env, &st->d_pos, &st->d_goal, nullptr);
    if (rc != MGPT_OK) { delete st; return rc; }
    hipError_t e = hipMalloc(&st->d_step, sizeof(uint64_t));
    if (e == hipSuccess) e = hipMemset(st->d_step, 0, sizeof(uint64_t));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&st->cap_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("hipMalloc / hipStreamCreate failed in mgpt_step_create: %s", hipGetErrorString(e));
        (void)hipFree(st->d_step);
        delete st;
        return MGPT_ERR_HIP;
    }
    *out = st;
    return MGPT_OK;
}

extern "C" int mgpt_step_destroy(mgpt_step *st)
{
    if (!st) return MGPT_OK;
    drop_graph(st);
    if (st->cap_stream) (void)hipStreamDestroy(st->cap_stream);
    (void)hipFree(st->d_step);
    delete st;
    return MGPT_OK;
}

extern "C" int mgpt_step_reset(mgpt_step *st, uint64_t step0, void *stream)
{
    MGPT_REQUIRE(st, MGPT_ERR_ARG, "NULL argument");
    drop_graph(st);          // between episodes the contexts may re-allocate (lifelong goal queues): next run re-captures
    hipLaunchKernelGGL(set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, st->d_step, step0);
    MGPT_LAUNCH_CHECK();
    return MGPT_OK;
}

extern "C" int mgpt_step_run(mgpt_step *st, uint8_t *d_tokens, int32_t *d_actions, int goals_may_change, int use_graph, void *stream)
{
    MGPT_REQUIRE(st && d_tokens && d_actions, MGPT_ERR_ARG, "NULL argument");
    hipStream_t s = (hipStream_t)stream;
    const int gmc = goals_may_change ? 1 : 0;
    // one of OUR contexts re-allocated or freed device memory since our last step (weights reloaded -> planes freed and rebuilt lazily,
    // goal queues replaced): the graph holds dead pointers, and the rebuild (hipMalloc, synchronous copies, null-stream pack
    // kernels) must not happen inside a capture -> drop the graph and run one eager step first
    if (gpt_generation(st->gpt) != st->seen_gpt_gen || env_generation(st->env) != st->seen_env_gen) {
        drop_graph(st);
        st->eager_runs = 0;
    }
    // eager: first step (lazy weight-plane build allocates), timing hooks on (their events belong to the eager stream), opt-out
    if (!use_graph || st->capture_failed || st->eager_runs < 1 || prof_is_enabled()) {
        st->eager_runs++;
        const int rc = step_body(st, d_tokens, d_actions, gmc, s);
        st->seen_gpt_gen = gpt_generation(st->gpt);   // (the lazy build inside this step bumped it)
        st->seen_env_gen = env_generation(st->env);
        return rc;
    }
    if (!st->exec || st->cap_tokens != d_tokens || st->cap_actions != d_actions || st->cap_gmc != gmc) {
        drop_graph(st);
        hipError_t e = hipStreamBeginCapture(st->cap_stream, hipStreamCaptureModeRelaxed);
        int rc = MGPT_OK;
        if (e == hipSuccess) {
            rc = step_body(st, d_tokens, d_actions, gmc, st->cap_stream);   // recorded, not executed
            e = hipStreamEndCapture(st->cap_stream, &st->graph);
        }
        if (e == hipSuccess && rc == MGPT_OK) e = hipGraphInstantiate(&st->exec, st->graph, nullptr, nullptr, 0);
        if (e != hipSuccess || rc != MGPT_OK) {          // fall back to eager launches for good; nothing has run yet for this step
            (void)hipGetLastError();
            drop_graph(st);
            st->capture_failed = true;
            return step_body(st, d_tokens, d_actions, gmc, s);
        }
        st->cap_tokens = d_tokens; st->cap_actions = d_actions; st->cap_gmc = gmc;
    }
    MGPT_HIP(hipGraphLaunch(st->exec, s));
    return MGPT_OK;
}
