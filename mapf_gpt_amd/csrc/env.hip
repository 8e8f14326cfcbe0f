// env.hip -- batched grid MAPF environment step on MI355X (the part of the loop the reference
// delegates to POGEMA: experiment_setup/create_env.py:36-46, example.py:41-50).
//
// PARITY UNPINNED: POGEMA is a pip dependency of the reference, absent from its tree.  This file
// implements the spec written in DESIGN.md ("Env step spec") and restated in oracle/mapf_oracle.c
// (orc_env_step); tests check it bit-for-bit against that restatement and through invariants.
//
//   actions: 0 wait, 1 up(-1,0), 2 down(+1,0), 3 left(0,-1), 4 right(0,+1)
//   "soft" collisions, order-independent fixpoint:
//     1. a move into a blocked / out-of-frame cell becomes wait;
//     2. two agents swapping along an edge both wait;
//     3. until stable: a moving agent whose target is claimed by >= 2 agents reverts to wait
//        (a waiting agent claims its own cell).
//   on_target = "nothing": agents stay on the grid; terminated when all stand on their goals,
//   truncated after max_episode_steps.
//   on_target = "restart" (lifelong, create_env.py:28-32): an agent that ends a step on its goal takes the next
//   goal of its own pre-generated queue (mgpt_env_set_lifelong; the queue wraps) and the arrival is counted;
//   the episode only ends by truncation.  The tokenizer sees the goal change through update_agents
//   (observation_generator.cpp:464-477: a new cost-to-go field for that agent).
//
//   avg_agents_density (POGEMA's AgentsDensityWrapper, wired in at create_env.py:38,49; result key of
//   eval_configs/05-puzzles/05-puzzles.yaml:55): after reset and after every step, each agent's local density =
//   (agents inside its (2r+1)^2 observation window, itself included) / (traversable cells of that window, out-of-map
//   cells counting as obstacles), averaged over the agents; the metric is the mean of those T+1 samples.  r = 5
//   (example.py:48; the eval YAMLs keep POGEMA's default).
//
// One workgroup per instance, one thread per agent; the per-agent current/target cell ids live in
// LDS and every conflict test is an O(n_agents) LDS scan (n_agents <= 1024), so no per-cell scratch
// grid is needed.  Traffic is ~13 B per agent-step: negligible next to the tokenizer and the policy.
#include "common.h"
#include <cstring>

using namespace mgpt;

namespace {

__global__ void env_reset_kernel(int16_t *__restrict__ pos, int16_t *__restrict__ goal, const int16_t *__restrict__ pos0,
                                 const int16_t *__restrict__ goal0, int32_t *__restrict__ arrive,
                                 int32_t *__restrict__ tcount, uint8_t *__restrict__ done, int n_inst, int n_agents)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int total = n_inst * n_agents;
    if (i < total) {
        const int16_t pr = pos0[2 * i], pc = pos0[2 * i + 1], gr = goal0[2 * i], gc = goal0[2 * i + 1];
        pos[2 * i] = pr; pos[2 * i + 1] = pc;
        goal[2 * i] = gr; goal[2 * i + 1] = gc;
        arrive[i] = (pr == gr && pc == gc) ? 0 : -1;
    }
    if (i < n_inst) { tcount[i] = 0; done[i] = 0; }
}

constexpr int kDensityRadius = 5;       // obs_radius of the reference's GridConfig (example.py:48)

// traversable cells of the (2r+1)^2 window around every cell (out-of-frame = obstacle), one byte per cell (<= 121)
__global__ void env_window_free_kernel(const uint8_t *__restrict__ grids, int n_grids, int H, int W, uint8_t *__restrict__ wfree)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n_grids * H * W) return;
    const int c = (int)(i % W), r = (int)((i / W) % H);
    const uint8_t *grid = grids + (i / ((size_t)H * W)) * H * W;
    int n = 0;
    for (int dr = -kDensityRadius; dr <= kDensityRadius; dr++)
        for (int dc = -kDensityRadius; dc <= kDensityRadius; dc++) {
            const int rr = r + dr, cc = c + dc;
            n += (rr >= 0 && rr < H && cc >= 0 && cc < W && grid[rr * W + cc] == 0) ? 1 : 0;
        }
    wfree[i] = (uint8_t)n;
}

// One density sample of an instance: cells[] (LDS, cell id per agent) -> mean over agents of in-window agents / in-window
// traversable cells, added to dens[inst].  Fixed reduction shape (wave shuffle tree, then wave partials in order) so that
// the sum does not depend on scheduling.  Must be called by every thread of the block; cells[] must be visible.
__device__ void density_sample(const int *cells, int n_agents, int W, const uint8_t *__restrict__ wfree, double *part,
                               double *__restrict__ dens)
{
    const int a = threadIdx.x;
    double d = 0.0;
    if (a < n_agents) {
        const int me = cells[a], r = me / W, c = me - r * W;
        int cnt = 0;
        for (int b = 0; b < n_agents; b++) {
            const int o = cells[b], orow = o / W, ocol = o - orow * W;
            cnt += (abs(orow - r) <= kDensityRadius && abs(ocol - c) <= kDensityRadius) ? 1 : 0;
        }
        d = (double)cnt / (double)wfree[me];
    }
    for (int off = 32; off > 0; off >>= 1) d += __shfl_down(d, off, 64);
    if ((a & 63) == 0) part[a >> 6] = d;
    __syncthreads();
    if (a == 0) {
        double sum = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) sum += part[w];
        *dens += sum / (double)n_agents;
    }
}

// the sample of the initial observation (AgentsDensityWrapper.reset)
__global__ __launch_bounds__(1024) void env_density0_kernel(const int16_t *__restrict__ pos, int n_agents, int n_grids, int H, int W,
                                                            const uint8_t *__restrict__ wfree, double *__restrict__ dens)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *cells = reinterpret_cast<int *>(smem);
    double *part = reinterpret_cast<double *>(cells + 2 * n_agents);
    const int inst = blockIdx.x, a = threadIdx.x;
    const size_t g = (size_t)inst * n_agents + a;
    if (a < n_agents) cells[a] = pos[2 * g] * W + pos[2 * g + 1];
    if (a == 0) dens[inst] = 0.0;
    __syncthreads();
    density_sample(cells, n_agents, W, wfree + (size_t)(inst % n_grids) * H * W, part, dens + inst);
}

__global__ __launch_bounds__(1024) void env_step_kernel(const uint8_t *__restrict__ grids, int n_grids, int n_agents,
                                                        int H, int W, int max_steps, int16_t *__restrict__ pos,
                                                        int16_t *__restrict__ goal,
                                                        const int32_t *__restrict__ actions, int32_t *__restrict__ arrive,
                                                        int32_t *__restrict__ tcount, uint8_t *__restrict__ done,
                                                        const int16_t *__restrict__ goal_queue, int queue_len,
                                                        int32_t *__restrict__ qnext, int32_t *__restrict__ reached,
                                                        const uint8_t *__restrict__ wfree, double *__restrict__ dens, int rules)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *cur = reinterpret_cast<int *>(smem);
    int *tgt = cur + n_agents;
    double *part = reinterpret_cast<double *>(tgt + n_agents);      // 2 * n_agents ints: 8-byte aligned
    const int inst = blockIdx.x;
    if (done[inst] != 0) return;                      // workgroup-uniform
    const uint8_t *grid = grids + (size_t)(inst % n_grids) * H * W;
    const int a = threadIdx.x;
    const bool valid = a < n_agents;
    const size_t g = (size_t)inst * n_agents + a;

    int my_cur = -1, my_tgt = -1, pr = 0, pc = 0;
    if (valid) {
        pr = pos[2 * g]; pc = pos[2 * g + 1];
        int act = actions[g];
        if (act < 0 || act > 4) act = 0;
        const int dr = (act == 1) ? -1 : (act == 2 ? 1 : 0);
        const int dc = (act == 3) ? -1 : (act == 4 ? 1 : 0);
        const int nr = pr + dr, nc = pc + dc;
        my_cur = pr * W + pc;
        my_tgt = (nr < 0 || nr >= H || nc < 0 || nc >= W || grid[nr * W + nc] != 0) ? my_cur : nr * W + nc;   // rule 1
        cur[a] = my_cur;
        tgt[a] = my_tgt;
    }
    __syncthreads();
    if (rules & MGPT_ENV_RULE_NO_FOLLOW) {            // (workgroup-uniform) a cell occupied at the start of the step cannot be entered
        bool occupied = false;
        if (valid && my_tgt != my_cur)
            for (int b = 0; b < n_agents; b++) occupied |= (b != a && cur[b] == my_tgt);
        __syncthreads();
        if (occupied) { my_tgt = my_cur; tgt[a] = my_cur; }
        __syncthreads();
    }
    bool swap = false;                                // rule 2, decided on the rule-1 targets
    if (valid && my_tgt != my_cur) {
        for (int b = 0; b < n_agents; b++)
            if (b != a && cur[b] == my_tgt && tgt[b] == my_cur) swap = true;
    }
    __syncthreads();
    if (swap) { my_tgt = my_cur; tgt[a] = my_cur; }
    __syncthreads();
    for (;;) {                                        // rule 3 (Jacobi: all reads see the previous round)
        int revert = 0;
        if (valid && my_tgt != my_cur) {
            if (rules & MGPT_ENV_RULE_LOWEST_WINS) {  // a staying agent or a mover with a smaller id claims my target
                for (int b = 0; b < n_agents; b++) revert |= (b != a && tgt[b] == my_tgt && (cur[b] == my_tgt || b < a)) ? 1 : 0;
            } else {
                int c = 0;
                for (int b = 0; b < n_agents; b++) c += (tgt[b] == my_tgt) ? 1 : 0;
                revert = c > 1;
            }
        }
        const int any = __syncthreads_or(revert);
        if (revert) { my_tgt = my_cur; tgt[a] = my_cur; }
        __syncthreads();
        if (!any) break;
    }
    const int t_new = tcount[inst] + 1;
    int on = 0;
    if (valid) {
        const int nr = my_tgt / W, nc = my_tgt - nr * W;
        pos[2 * g] = (int16_t)nr; pos[2 * g + 1] = (int16_t)nc;
        const int gr = goal[2 * g], gc = goal[2 * g + 1];
        on = (nr == gr && nc == gc) ? 1 : 0;
        if (goal_queue != nullptr) {                  // lifelong: count the arrival, take the next queued goal
            if (on) {
                reached[g] += 1;
                const int q = qnext[g];
                const int16_t *nx = goal_queue + ((size_t)g * queue_len + q) * 2;
                goal[2 * g] = nx[0]; goal[2 * g + 1] = nx[1];
                qnext[g] = (q + 1 == queue_len) ? 0 : q + 1;
            }
            on = 0;
        } else {
            const bool was_on = (pr == gr && pc == gc);
            if (on && !was_on) arrive[g] = t_new;     // time of the (latest) arrival
            if (!on) arrive[g] = -1;
        }
    }
    const int n_on = __syncthreads_count(on);
    density_sample(tgt, n_agents, W, wfree + (size_t)(inst % n_grids) * H * W, part, dens + inst);   // tgt[] = the new cells
    if (a == 0) {
        tcount[inst] = t_new;
        if (goal_queue == nullptr && n_on == n_agents) done[inst] = 1;
        else if (t_new >= max_steps) done[inst] = 2;
    }
}

// {CSR, ISR, SoC, makespan, ep_length, avg_agents_density} per instance (keys of eval_configs/*/*.yaml results_views)
__global__ void env_metrics_kernel(const int16_t *__restrict__ pos, const int16_t *__restrict__ goal,
                                   const int32_t *__restrict__ arrive, const int32_t *__restrict__ tcount,
                                   const double *__restrict__ dens, int n_inst, int n_agents, float *__restrict__ out)
{
    const int inst = blockIdx.x * blockDim.x + threadIdx.x;
    if (inst >= n_inst) return;
    const int t = tcount[inst];
    int on = 0, soc = 0, mk = 0;
    for (int a = 0; a < n_agents; a++) {
        const size_t g = (size_t)inst * n_agents + a;
        const bool o = pos[2 * g] == goal[2 * g] && pos[2 * g + 1] == goal[2 * g + 1];
        const int ta = o ? max(arrive[g], 0) : t;
        on += o ? 1 : 0;
        soc += ta;
        mk = max(mk, ta);
    }
    float *m = out + (size_t)inst * 6;
    m[5] = (float)(dens[inst] / (double)(t + 1));       // samples: the reset observation + one per step
    m[0] = (on == n_agents) ? 1.f : 0.f;
    m[1] = (float)on / (float)n_agents;
    m[2] = (float)soc;
    m[3] = (float)mk;
    m[4] = (float)t;
}

}  // namespace

// LDS of the per-instance kernels: cur[] + tgt[] cell ids, then 16 wave partials of the density reduction (8-byte aligned)
static size_t env_lds_bytes(int n_agents) { return (size_t)n_agents * 2 * sizeof(int) + 16 * sizeof(double); }

struct mgpt_env {
    int n_inst, n_agents, H, W, n_grids, max_steps;
    uint8_t *grids = nullptr;
    int16_t *pos = nullptr, *goal = nullptr;
    int32_t *arrive = nullptr, *tcount = nullptr;
    int32_t *pin_act = nullptr;                 // pinned host staging of mgpt_env_step_host: the step kernel reads the actions in place
    uint8_t *pin_state = nullptr;               // pinned host copy of state_blob (an asynchronous device-to-host copy lands here)
    uint8_t *state_blob = nullptr;              // [pos int16 total*2 | goal int16 total*2 | done u8 n_inst]: pos / goal / done point into it
    uint8_t *done = nullptr;
    uint8_t *wfree = nullptr;           // [n_grids][H][W] traversable cells of the observation window around each cell
    double *dens = nullptr;             // [n_inst] running sum of the per-sample mean agent densities
    int16_t *goal_queue = nullptr;      // lifelong: [n_inst][n_agents][queue_len][2], else NULL
    int32_t *qnext = nullptr, *reached = nullptr;
    int queue_len = 0;
    bool have_grids = false, have_reset = false;
    uint64_t generation = 1;            // bumped when the goal queues are replaced (common.h: env_generation)
    int rules = 0;                      // MGPT_ENV_RULE_* mask (mgpt_env_set_rules)
};

uint64_t mgpt::env_generation(const mgpt_env *e) { return e->generation; }

extern "C" int mgpt_env_create(mgpt_env **out, int n_inst, int n_agents, int H, int W, int n_grids, int max_episode_steps)
{
    MGPT_REQUIRE(out, MGPT_ERR_ARG, "NULL argument");
    MGPT_REQUIRE(n_inst > 0 && n_agents > 0 && H > 0 && W > 0 && n_grids > 0 && n_grids <= n_inst && max_episode_steps > 0,
                 MGPT_ERR_ARG, "bad sizes");
    MGPT_REQUIRE(n_agents <= 1024, MGPT_ERR_UNSUPPORTED, "n_agents=%d > 1024 (one thread per agent)", n_agents);
    mgpt_env *e = new mgpt_env();
    e->n_inst = n_inst; e->n_agents = n_agents; e->H = H; e->W = W; e->n_grids = n_grids; e->max_steps = max_episode_steps;
    const size_t total = (size_t)n_inst * n_agents;
    hipError_t err = hipMalloc(&e->grids, (size_t)n_grids * H * W);
    // pos | goal | done live in ONE allocation, in the order mgpt_env_step_host hands them to the host (one copy)
    if (err == hipSuccess) err = hipMalloc(&e->state_blob, total * 8 + (size_t)n_inst);
    if (err == hipSuccess) {
        e->pos = reinterpret_cast<int16_t *>(e->state_blob);
        e->goal = e->pos + total * 2;
        e->done = reinterpret_cast<uint8_t *>(e->goal + total * 2);
    }
    if (err == hipSuccess) err = hipMalloc(&e->arrive, total * sizeof(int32_t));
    if (err == hipSuccess) err = hipMalloc(&e->tcount, (size_t)n_inst * sizeof(int32_t));
    if (err == hipSuccess) err = hipMalloc(&e->wfree, (size_t)n_grids * H * W);
    if (err == hipSuccess) err = hipMalloc(&e->dens, (size_t)n_inst * sizeof(double));
    if (err != hipSuccess) {
        set_error("hipMalloc failed in mgpt_env_create: %s", hipGetErrorString(err));
        mgpt_env_destroy(e);
        return MGPT_ERR_HIP;
    }
    *out = e;
    return MGPT_OK;
}

extern "C" int mgpt_env_destroy(mgpt_env *e)
{
    if (!e) return MGPT_OK;
    (void)hipFree(e->grids); (void)hipFree(e->state_blob);
    (void)hipFree(e->arrive); (void)hipFree(e->tcount);
    (void)hipFree(e->wfree); (void)hipFree(e->dens); (void)hipHostFree(e->pin_act); (void)hipHostFree(e->pin_state);
    (void)hipFree(e->goal_queue); (void)hipFree(e->qnext); (void)hipFree(e->reached);
    delete e;
    return MGPT_OK;
}

extern "C" int mgpt_env_set_grids(mgpt_env *e, const uint8_t *d_grids, void *stream)
{
    MGPT_REQUIRE(e && d_grids, MGPT_ERR_ARG, "NULL argument");
    MGPT_HIP(hipMemcpyAsync(e->grids, d_grids, (size_t)e->n_grids * e->H * e->W, hipMemcpyDeviceToDevice,
                            (hipStream_t)stream));
    const size_t cells = (size_t)e->n_grids * e->H * e->W;
    hipLaunchKernelGGL(env_window_free_kernel, dim3((unsigned)cdiv64((int64_t)cells, 256)), dim3(256), 0, (hipStream_t)stream, e->grids,
                       e->n_grids, e->H, e->W, e->wfree);
    MGPT_LAUNCH_CHECK();
    e->have_grids = true;
    return MGPT_OK;
}

extern "C" int mgpt_env_reset(mgpt_env *e, const int16_t *d_pos, const int16_t *d_goal, void *stream)
{
    MGPT_REQUIRE(e && d_pos && d_goal, MGPT_ERR_ARG, "NULL argument");
    MGPT_REQUIRE(e->have_grids, MGPT_ERR_STATE, "mgpt_env_set_grids must precede reset");
    hipStream_t s = (hipStream_t)stream;
    const int total = e->n_inst * e->n_agents;
    hipLaunchKernelGGL(env_reset_kernel, dim3(cdiv(total > e->n_inst ? total : e->n_inst, 256)), dim3(256), 0, s, e->pos,
                       e->goal, d_pos, d_goal, e->arrive, e->tcount, e->done, e->n_inst, e->n_agents);
    MGPT_LAUNCH_CHECK();
    const int threads = cdiv(e->n_agents, 64) * 64;
    hipLaunchKernelGGL(env_density0_kernel, dim3(e->n_inst), dim3(threads), env_lds_bytes(e->n_agents), s, e->pos, e->n_agents,
                       e->n_grids, e->H, e->W, e->wfree, e->dens);
    MGPT_LAUNCH_CHECK();
    if (e->goal_queue != nullptr) {
        const size_t total_b = (size_t)total * sizeof(int32_t);
        MGPT_HIP(hipMemsetAsync(e->qnext, 0, total_b, s));
        MGPT_HIP(hipMemsetAsync(e->reached, 0, total_b, s));
    }
    e->have_reset = true;
    return MGPT_OK;
}

extern "C" int mgpt_env_set_rules(mgpt_env *e, int rules)
{
    MGPT_REQUIRE(e, MGPT_ERR_ARG, "NULL argument");
    MGPT_REQUIRE((rules & ~(MGPT_ENV_RULE_NO_FOLLOW | MGPT_ENV_RULE_LOWEST_WINS)) == 0, MGPT_ERR_ARG, "unknown rule bits in %d", rules);
    if (rules != e->rules) e->generation++;              // the mask is a kernel argument inside a captured step graph
    e->rules = rules;
    return MGPT_OK;
}

extern "C" int mgpt_env_set_lifelong(mgpt_env *e, const int16_t *d_goal_queue, int queue_len, void *stream)
{
    MGPT_REQUIRE(e, MGPT_ERR_ARG, "NULL argument");
    hipStream_t s = (hipStream_t)stream;
    e->generation++;                                     // a captured step graph holds the queue pointers (or their absence)
    (void)hipFree(e->goal_queue); (void)hipFree(e->qnext); (void)hipFree(e->reached);
    e->goal_queue = nullptr; e->qnext = nullptr; e->reached = nullptr; e->queue_len = 0;
    if (d_goal_queue == nullptr || queue_len <= 0) return MGPT_OK;          // back to on_target = "nothing"
    const size_t total = (size_t)e->n_inst * e->n_agents;
    int16_t *q = nullptr;
    int32_t *qn = nullptr, *rc = nullptr;
    hipError_t err = hipMalloc(&q, total * queue_len * 2 * sizeof(int16_t));
    if (err == hipSuccess) err = hipMalloc(&qn, total * sizeof(int32_t));
    if (err == hipSuccess) err = hipMalloc(&rc, total * sizeof(int32_t));
    if (err == hipSuccess) err = hipMemcpyAsync(q, d_goal_queue, total * queue_len * 2 * sizeof(int16_t), hipMemcpyDeviceToDevice, s);
    if (err == hipSuccess) err = hipMemsetAsync(qn, 0, total * sizeof(int32_t), s);
    if (err == hipSuccess) err = hipMemsetAsync(rc, 0, total * sizeof(int32_t), s);
    if (err != hipSuccess) {                                               // stay in on_target = "nothing" rather than half-configured
        set_error("mgpt_env_set_lifelong: %s", hipGetErrorString(err));
        (void)hipFree(q); (void)hipFree(qn); (void)hipFree(rc);
        return MGPT_ERR_HIP;
    }
    e->goal_queue = q; e->qnext = qn; e->reached = rc; e->queue_len = queue_len;
    return MGPT_OK;
}

extern "C" int mgpt_env_lifelong_counts(mgpt_env *e, int32_t *d_reached_out, void *stream)
{
    MGPT_REQUIRE(e && d_reached_out, MGPT_ERR_ARG, "NULL argument");
    MGPT_REQUIRE(e->reached != nullptr, MGPT_ERR_STATE, "mgpt_env_set_lifelong must precede lifelong_counts");
    MGPT_HIP(hipMemcpyAsync(d_reached_out, e->reached, (size_t)e->n_inst * e->n_agents * sizeof(int32_t), hipMemcpyDeviceToDevice,
                            (hipStream_t)stream));
    return MGPT_OK;
}

extern "C" int mgpt_env_step(mgpt_env *e, const int32_t *d_actions, void *stream)
{
    MGPT_REQUIRE(e && d_actions, MGPT_ERR_ARG, "NULL argument");
    MGPT_REQUIRE(e->have_reset, MGPT_ERR_STATE, "mgpt_env_reset must precede step");
    hipStream_t s = (hipStream_t)stream;
    const int threads = cdiv(e->n_agents, 64) * 64;
    ProfScope ps(P_ENV_STEP, s);
    hipLaunchKernelGGL(env_step_kernel, dim3(e->n_inst), dim3(threads), env_lds_bytes(e->n_agents), s, e->grids,
                       e->n_grids, e->n_agents, e->H, e->W, e->max_steps, e->pos, e->goal, d_actions, e->arrive,
                       e->tcount, e->done, e->goal_queue, e->queue_len, e->qnext, e->reached, e->wfree, e->dens, e->rules);
    MGPT_LAUNCH_CHECK();
    return MGPT_OK;
}

// Host-list step of the reference-shaped API (create_env.py:14-15: lists in, lists out) in ONE call: actions from a host buffer,
// the step, and positions / goals / done flags back into a host buffer [pos int16 n*2 | goal int16 n*2 | done u8 n_inst],
// n = n_inst * n_agents.  h_actions may be NULL (no step: just pull the state, e.g. after reset).  Synchronises the stream.
extern "C" int mgpt_env_step_host(mgpt_env *e, const int32_t *h_actions, uint8_t *h_state_out, void *stream)
{
    MGPT_REQUIRE(e && h_state_out, MGPT_ERR_ARG, "NULL argument");
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)e->n_inst * e->n_agents, nb = 8 * n + (size_t)e->n_inst;
    // pinned staging on both sides: the kernel reads the (few hundred bytes of) actions straight from host memory, the state
    // comes back by one truly asynchronous copy -- two runtime calls and one synchronisation per step instead of a blocking
    // pageable copy each way (65 -> 4x us per GridEnv.step, tools/time_gridenv.py)
    if (!e->pin_state) {
        MGPT_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->pin_act), n * sizeof(int32_t), hipHostMallocDefault));
        MGPT_HIP(hipHostMalloc(reinterpret_cast<void **>(&e->pin_state), nb, hipHostMallocDefault));
    }
    if (h_actions) {
        memcpy(e->pin_act, h_actions, n * sizeof(int32_t));
        void *d_act = nullptr;
        MGPT_HIP(hipHostGetDevicePointer(&d_act, e->pin_act, 0));
        int rc = mgpt_env_step(e, static_cast<const int32_t *>(d_act), stream);
        if (rc != MGPT_OK) return rc;
    }
    MGPT_HIP(hipMemcpyAsync(e->pin_state, e->state_blob, nb, hipMemcpyDeviceToHost, s));   // pos | goal | done
    MGPT_HIP(hipStreamSynchronize(s));
    memcpy(h_state_out, e->pin_state, nb);
    return MGPT_OK;
}

extern "C" int mgpt_env_state(mgpt_env *e, const int16_t **d_pos, const int16_t **d_goal, const uint8_t **d_done)
{
    MGPT_REQUIRE(e, MGPT_ERR_ARG, "NULL argument");
    if (d_pos) *d_pos = e->pos;
    if (d_goal) *d_goal = e->goal;
    if (d_done) *d_done = e->done;
    return MGPT_OK;
}

extern "C" int mgpt_env_metrics(mgpt_env *e, float *d_metrics, void *stream)
{
    MGPT_REQUIRE(e && d_metrics, MGPT_ERR_ARG, "NULL argument");
    MGPT_REQUIRE(e->have_reset, MGPT_ERR_STATE, "mgpt_env_reset must precede metrics");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(P_ENV_METRICS, s);
    hipLaunchKernelGGL(env_metrics_kernel, dim3(cdiv(e->n_inst, 64)), dim3(64), 0, s, e->pos, e->goal, e->arrive,
                       e->tcount, e->dens, e->n_inst, e->n_agents, d_metrics);
    MGPT_LAUNCH_CHECK();
    return MGPT_OK;
}

extern "C" int mgpt_env_copy_state(mgpt_env *e, int16_t *d_pos_out, int16_t *d_goal_out, uint8_t *d_done_out, void *stream)
{
    MGPT_REQUIRE(e, MGPT_ERR_ARG, "NULL argument");
    hipStream_t s = (hipStream_t)stream;
    const size_t nb = (size_t)e->n_inst * e->n_agents * 2 * sizeof(int16_t);
    if (d_pos_out) MGPT_HIP(hipMemcpyAsync(d_pos_out, e->pos, nb, hipMemcpyDeviceToDevice, s));
    if (d_goal_out) MGPT_HIP(hipMemcpyAsync(d_goal_out, e->goal, nb, hipMemcpyDeviceToDevice, s));
    if (d_done_out) MGPT_HIP(hipMemcpyAsync(d_done_out, e->done, (size_t)e->n_inst, hipMemcpyDeviceToDevice, s));
    return MGPT_OK;
}
