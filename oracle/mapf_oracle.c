/*
 * mapf_oracle.c -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * Plain-C CPU restatement of MAPF-GPT's per-step integer hot path, used as the
 * checker for the HIP kernels (tests/, __graft_entry__.smoke(), bench.py's
 * cpu_baseline leg).  Nothing under mapf_gpt_amd/ may call into this file.
 *
 * Parity status
 *   tokenizer (orc_gen_*)  : PINNED.  Checked bit-for-bit against the real reference
 *                            (mapf_gpt/observation_generator.cpp compiled into oracle/_ref by
 *                            oracle/build_ref.sh) in tests/test_oracle_vs_reference.py and
 *                            against the committed vectors in tests/golden/ (made by
 *                            tests/golden/make_golden.py from that same build).
 *   env step (orc_env_*)   : PARITY UNPINNED.  POGEMA is a pip dependency of the reference
 *                            (pyproject.toml:18), absent from /root/reference and from this
 *                            image; the reference holds no env arithmetic and no env tests.
 *                            orc_env_step states OUR spec (DESIGN.md "Env step spec").
 *
 * Each function cites the reference lines it restates (cpp = mapf_gpt/observation_generator.cpp,
 * h = mapf_gpt/observation_generator.h, inf = mapf_gpt/inference.py).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_UNREACH 65535
#define ORC_CTX 256       /* cpp:386 hard-codes 256 (InputParameters.context_size is never read by the reference) */
/* defaults of struct InputParameters (h:24; the values inference.py:15-19 passes); orc_gen_set_params changes them per generator */
#define ORC_NHIST 5       /* inf:16 num_previous_actions */
#define ORC_NAGENTS 13    /* inf:15 num_agents */
#define ORC_R 5           /* inf:18-19 agents_radius == cost2go_radius == 5 */
#define ORC_LIMIT 20      /* inf:17 cost2go_value_limit */

/* Vocabulary, Encoder::Encoder cpp:321-350, for cost2go_value_limit L: ints -L..L -> 0..2L, -4L -> 2L+1, -2L -> 2L+2, +2L -> 2L+3,
 * n,w,u,d,l,r -> 2L+4..2L+9, "0000".."1111" -> 2L+10..2L+25, "!" -> 2L+26.  (L = 20: 41, 42, 43, 44.., 50.., 66.) */
#define TOK_UNREACH(L) (2 * (L) + 1)
#define TOK_NEG(L) (2 * (L) + 2)
#define TOK_POS(L) (2 * (L) + 3)
#define TOK_N(L) (2 * (L) + 4)
#define TOK_BITS0(L) (2 * (L) + 10)
#define TOK_PAD(L) (2 * (L) + 26)

/* cpp:200-286 (+ cpp:134-176 for small grids).  The reference's tiled border/priority-queue
 * machinery yields the 4-connected BFS distance from the goal over free cells and 65535
 * elsewhere (SURVEY.md finding 4; re-verified in tests/test_oracle_vs_reference.py on
 * >64 grids, where the large-map branch cpp:222-279 is the one that runs) -- with ONE exception,
 * restated in orc_gen_generate_observations: the (right, bottom) corner cell of an agent's cached
 * 129 x 129 partial window is not among the seeded border cells (cpp:178-198: both loops stop one
 * short of it), so the flood fill gives it min(in-window neighbours) + 1.  It is visible only from
 * (left + 123, top + 123), reached without a recompute (cpp:469-477); the window origin is
 * therefore part of the agent state here (`org`).  Pinned by tests/golden/tok_corner_*.npz. */
#define ORC_STEP 64       /* inf:28 grid_step: the default; orc_gen_set_grid_step changes it per generator */
void orc_bfs(const uint8_t *grid, int H, int W, int gr, int gc, uint16_t *dist)
{
    int n = H * W;
    for (int i = 0; i < n; i++) dist[i] = ORC_UNREACH;
    if (gr < 0 || gr >= H || gc < 0 || gc >= W) return;
    int *queue = (int *)malloc(sizeof(int) * (size_t)n);
    int head = 0, tail = 0;
    /* cpp:157-159: the goal cell itself is seeded with 0 without looking at grid[goal]. */
    dist[gr * W + gc] = 0;
    queue[tail++] = gr * W + gc;
    static const int dr[4] = {-1, 1, 0, 0}, dc[4] = {0, 0, -1, 1};
    while (head < tail) {
        int cur = queue[head++];
        int r = cur / W, c = cur % W;
        for (int k = 0; k < 4; k++) {
            int nr = r + dr[k], nc = c + dc[k];
            if (nr < 0 || nr >= H || nc < 0 || nc >= W) continue;
            int idx = nr * W + nc;
            if (grid[idx] == 0 && dist[idx] == ORC_UNREACH) {
                dist[idx] = (uint16_t)(dist[cur] + 1);
                queue[tail++] = idx;
            }
        }
    }
    free(queue);
}

/* cpp:412-430 + vocabulary cpp:330-343: bits in the order u(-1,0) d(+1,0) l(0,-1) r(0,+1),
 * bit set iff the neighbour's distance is strictly smaller than the cell's own. */
static uint8_t next_action_token(const uint16_t *dist, int W, int r, int c, int bits0)
{
    int cur = dist[r * W + c];
    int u = dist[(r - 1) * W + c] < cur;
    int d = dist[(r + 1) * W + c] < cur;
    int l = dist[r * W + c - 1] < cur;
    int rr = dist[r * W + c + 1] < cur;
    return (uint8_t)(bits0 + 8 * u + 4 * d + 2 * l + rr);
}

typedef struct {
    int H, W, n;
    uint8_t *grid;      /* H*W, non-zero = blocked (h:112 copies the grid) */
    int32_t *occ;       /* agents_locations, h:116: -1 or agent id */
    int32_t *pos;       /* n*2 (row, col) padded coords */
    int32_t *goal;      /* n*2 */
    uint8_t *hist;      /* n*nhist tokens, oldest -> newest */
    uint8_t *next;      /* n */
    uint16_t *dist;     /* n*H*W */
    int32_t *org;       /* n*2: (left_border, top_border) of the agent's cached partial window, cpp:204-207 */
    int step;           /* cfg.grid_step, h:38 */
    int limit, nslots, nhist, obs_r, ag_r;   /* cfg.cost2go_value_limit, num_agents, num_previous_actions, obs_radius, agents_radius (h:32-37) */
} orc_gen;

orc_gen *orc_gen_create(const uint8_t *grid, int H, int W)
{
    orc_gen *g = (orc_gen *)calloc(1, sizeof(orc_gen));
    g->H = H; g->W = W; g->n = 0;
    g->step = ORC_STEP;
    g->limit = ORC_LIMIT; g->nslots = ORC_NAGENTS; g->nhist = ORC_NHIST; g->obs_r = ORC_R; g->ag_r = ORC_R;
    g->grid = (uint8_t *)malloc((size_t)H * W);
    memcpy(g->grid, grid, (size_t)H * W);
    g->occ = (int32_t *)malloc(sizeof(int32_t) * (size_t)H * W);
    for (int i = 0; i < H * W; i++) g->occ[i] = -1;           /* h:116 */
    return g;
}

void orc_gen_destroy(orc_gen *g)
{
    if (!g) return;
    free(g->grid); free(g->occ); free(g->pos); free(g->goal);
    free(g->hist); free(g->next); free(g->dist); free(g->org); free(g);
}

/* cpp:204-207: origin of the partial window computed for an agent standing at (r, c) */
static void window_origin(int r, int c, int step, int R, int32_t *org)
{
    org[0] = (r - R > 0 ? r - R : 0) / step * step;
    org[1] = (c - R > 0 ? c - R : 0) / step * step;
}

/* InputParameters.grid_step (h:38, cpp:551); call before create_agents */
void orc_gen_set_grid_step(orc_gen *g, int step) { if (step > 0) g->step = step; }

/* the other fields of InputParameters (h:22-40, pybind ctor cpp:551); call before create_agents.  Returns 0, or -1 for values
 * the reference itself cannot run: a row longer than 256 tokens is returned longer than 256 (cpp:386-387), agents_radius above
 * the limit makes int_vocab.at throw on a relative position (cpp:358-359). */
int orc_gen_set_params(orc_gen *g, int limit, int nslots, int nhist, int obs_r, int ag_r)
{
    if (limit < 1 || 2 * limit + 26 > 255 || nslots < 0 || nhist < 0 || obs_r < 0 || ag_r < 0 || ag_r > limit) return -1;
    if ((2 * obs_r + 1) * (2 * obs_r + 1) + nslots * (5 + nhist) > ORC_CTX) return -1;
    g->limit = limit; g->nslots = nslots; g->nhist = nhist; g->obs_r = obs_r; g->ag_r = ag_r;
    return 0;
}

/* cpp:391-410: history = "n" x5, distance field, greedy bits.  Does NOT touch agents_locations. */
void orc_gen_create_agents(orc_gen *g, int n, const int32_t *pos, const int32_t *goal)
{
    free(g->pos); free(g->goal); free(g->hist); free(g->next); free(g->dist); free(g->org);
    g->n = n;
    g->org = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)n);
    g->pos = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)n);
    g->goal = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)n);
    g->hist = (uint8_t *)malloc((size_t)n * g->nhist + 1);
    g->next = (uint8_t *)malloc((size_t)n);
    g->dist = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)n * g->H * g->W);
    memcpy(g->pos, pos, sizeof(int32_t) * 2 * (size_t)n);
    memcpy(g->goal, goal, sizeof(int32_t) * 2 * (size_t)n);
    memset(g->hist, TOK_N(g->limit), (size_t)n * g->nhist);  /* cpp:403-406 */
    for (int a = 0; a < n; a++) {
        uint16_t *d = g->dist + (size_t)a * g->H * g->W;
        orc_bfs(g->grid, g->H, g->W, goal[2 * a], goal[2 * a + 1], d);
        window_origin(pos[2 * a], pos[2 * a + 1], g->step, g->obs_r, g->org + 2 * a);   /* cpp:408 compute_cost2go_partial */
        g->next[a] = next_action_token(d, g->W, pos[2 * a], pos[2 * a + 1], TOK_BITS0(g->limit));
    }
}

/* cpp:432-485.  actions are the policy's previous *intended* actions (inf:140-144,168);
 * anything outside 0..4 (the -1 of the first call) appends "n". */
void orc_gen_update_agents(orc_gen *g, const int32_t *pos, const int32_t *goal, const int32_t *actions)
{
    int n = g->n, W = g->W;
    for (int a = 0; a < n; a++)                               /* cpp:434-435 clear ALL old cells first */
        g->occ[g->pos[2 * a] * W + g->pos[2 * a + 1]] = -1;
    for (int a = 0; a < n; a++) {
        g->occ[pos[2 * a] * W + pos[2 * a + 1]] = a;          /* cpp:440, id order: highest id wins a shared cell */
        g->pos[2 * a] = pos[2 * a];
        g->pos[2 * a + 1] = pos[2 * a + 1];
        int act = actions[a];
        if (g->nhist > 0) {
            uint8_t *h = g->hist + (size_t)a * g->nhist;
            memmove(h, h + 1, (size_t)g->nhist - 1);          /* cpp:463 pop_front */
            h[g->nhist - 1] = (uint8_t)((act >= 0 && act <= 4) ? TOK_N(g->limit) + 1 + act : TOK_N(g->limit));   /* cpp:442-462 */
        }                                                     /* (num_previous_actions = 0: push_back + pop_front leave the deque empty) */
        if (g->goal[2 * a] != goal[2 * a] || g->goal[2 * a + 1] != goal[2 * a + 1]) {       /* cpp:464-468 */
            g->goal[2 * a] = goal[2 * a];
            g->goal[2 * a + 1] = goal[2 * a + 1];
            orc_bfs(g->grid, g->H, g->W, goal[2 * a], goal[2 * a + 1], g->dist + (size_t)a * g->H * W);
            window_origin(pos[2 * a], pos[2 * a + 1], g->step, g->obs_r, g->org + 2 * a);
        } else {
            /* cpp:469-477: the observation window left the cached partial box -> recomputed around the new position.
             * The full-grid field needs no recompute; only the box origin moves (it decides the corner cell below). */
            int left = g->org[2 * a], top = g->org[2 * a + 1];
            int right = left + 2 * g->step < g->H - 1 ? left + 2 * g->step : g->H - 1;
            int bottom = top + 2 * g->step < W - 1 ? top + 2 * g->step : W - 1;
            int R = g->obs_r;
            if (pos[2 * a] - R < left || pos[2 * a] + R > right || pos[2 * a + 1] - R < top || pos[2 * a + 1] + R > bottom)
                window_origin(pos[2 * a], pos[2 * a + 1], g->step, R, g->org + 2 * a);
        }
    }
    for (int a = 0; a < n; a++)                               /* cpp:483-484 */
        g->next[a] = next_action_token(g->dist + (size_t)a * g->H * W, W, g->pos[2 * a], g->pos[2 * a + 1], TOK_BITS0(g->limit));
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* cpp:516-528 = cpp:288-311 (window) + cpp:487-514 (neighbours) + cpp:352-389 (encode). */
void orc_gen_generate_observations(const orc_gen *g, uint8_t *out /* n*256 */)
{
    int n = g->n, H = g->H, W = g->W;
    const int L = g->limit, R = g->obs_r, A = g->ag_r, win = 2 * g->obs_r + 1, rec = 5 + g->nhist;
    int *cand = (int *)malloc(sizeof(int) * (size_t)(2 * A + 1) * (2 * A + 1));
    int *key = (int *)malloc(sizeof(int) * (size_t)(2 * A + 1) * (2 * A + 1));
    for (int a = 0; a < n; a++) {
        uint8_t *row = out + (size_t)a * ORC_CTX;
        const uint16_t *d = g->dist + (size_t)a * g->H * W;
        int pr = g->pos[2 * a], pc = g->pos[2 * a + 1];
        memset(row, TOK_PAD(L), ORC_CTX);                     /* cpp:375-376, 386-387 */
        int mid = d[pr * W + pc];                             /* cpp:297 */
        /* the unseeded corner of the cached partial window (cpp:178-198), if it exists and is in view */
        int cr = g->org[2 * a] + 2 * g->step, cc = g->org[2 * a + 1] + 2 * g->step;
        int corner_in_view = cr <= H - 1 && cc <= W - 1 && pr + R == cr && pc + R == cc;
        for (int i = 0; i <= 2 * R; i++)
            for (int j = 0; j <= 2 * R; j++) {
                int v = d[(pr - R + i) * W + (pc - R + j)];
                if (corner_in_view && i == 2 * R && j == 2 * R && v != ORC_UNREACH && v != 0) {
                    /* reached only from its two in-window neighbours (exact border seeds): cpp:252-268 */
                    int n1 = d[(cr - 1) * W + cc], n2 = d[cr * W + cc - 1];
                    int m = n1 < n2 ? n1 : n2;
                    v = m == ORC_UNREACH ? ORC_UNREACH : m + 1;
                }
                uint8_t t;
                if (v == ORC_UNREACH) t = (uint8_t)TOK_UNREACH(L);   /* cpp:308-309: -4L */
                else {
                    v -= mid;                                 /* cpp:304 */
                    if (v > L) t = (uint8_t)TOK_POS(L);       /* +2L */
                    else if (v < -L) t = (uint8_t)TOK_NEG(L); /* -2L */
                    else t = (uint8_t)(v + L);
                }
                row[i * win + j] = t;
            }
        /* cpp:487-505: row-major scan of agents_locations, sort by (Manhattan, id) */
        int nc = 0;
        for (int i = -A; i <= A; i++)
            for (int j = -A; j <= A; j++) {
                int b = g->occ[(pr + i) * W + (pc + j)];
                if (b >= 0) {
                    int md = abs(g->pos[2 * b] - pr) + abs(g->pos[2 * b + 1] - pc);
                    cand[nc] = b; key[nc] = md * 65536 + b; nc++;
                }
            }
        for (int x = 1; x < nc; x++) {                        /* insertion sort on the unique key */
            int k = key[x], b = cand[x], y = x - 1;
            while (y >= 0 && key[y] > k) { key[y + 1] = key[y]; cand[y + 1] = cand[y]; y--; }
            key[y + 1] = k; cand[y + 1] = b;
        }
        int m = nc < g->nslots ? nc : g->nslots;              /* cpp:506 */
        for (int s = 0; s < m; s++) {
            int b = cand[s];
            uint8_t *o = row + win * win + rec * s;
            /* cpp:358-361: relative pos via int_vocab.at (always within +-A <= L), goal clamped to +-L */
            o[0] = (uint8_t)(g->pos[2 * b] - pr + L);
            o[1] = (uint8_t)(g->pos[2 * b + 1] - pc + L);
            o[2] = (uint8_t)(clampi(g->goal[2 * b] - pr, -L, L) + L);
            o[3] = (uint8_t)(clampi(g->goal[2 * b + 1] - pc, -L, L) + L);
            memcpy(o + 4, g->hist + (size_t)b * g->nhist, (size_t)g->nhist);   /* cpp:364-367 */
            o[4 + g->nhist] = g->next[b];                                       /* cpp:368 */
        }
    }
    free(cand); free(key);
}

/* read-back helpers for tests */
const uint16_t *orc_gen_dist(const orc_gen *g) { return g->dist; }
const uint8_t *orc_gen_next(const orc_gen *g) { return g->next; }
const uint8_t *orc_gen_hist(const orc_gen *g) { return g->hist; }

/* ------------------------------------------------------------------------------------------
 * Env step -- OUR spec, parity unpinned (see header).  MOVES = wait, up(-1,0), down(+1,0),
 * left(0,-1), right(0,+1)  (action ids as in cpp:442-462 / dataset/tokenizer/generate_observations.py:10-17).
 * "soft" collisions, order-independent fixpoint:
 *   1. a move into a blocked cell becomes wait;
 *   2. two agents swapping along an edge both wait;
 *   3. repeat until stable: every moving agent whose target cell is claimed by >= 2 agents
 *      (an agent that stays claims its own cell) reverts to wait.
 * Afterwards no two agents share a cell and no edge is swapped; following a leaving agent is allowed.
 * on_target = "nothing": agents stay on the grid and keep acting.
 * Returns the number of agents standing on their goal after the move.
 * ------------------------------------------------------------------------------------------ */
/* rules (bit mask; 0 = the spec above).  The two places where the spec rests on RECALLED pogema behaviour (box
 * under E0) are switchable so that pinning against captured fixtures (tests/golden/make_golden_env.py) is a flag flip:
 *   ORC_RULE_NO_FOLLOW (1)   a move into a cell that another agent occupies at the START of the step becomes wait, whether or
 *                            not that agent leaves it (default: following a leaving agent is allowed);
 *   ORC_RULE_LOWEST_WINS (2) a contested empty cell goes to the lowest-id mover instead of nobody: in rule 3 a mover reverts
 *                            only if a staying agent or a mover with a smaller id claims its target (default: all revert). */
int orc_env_step_rules(const uint8_t *grid, int H, int W, int n, int32_t *pos /* n*2, in/out */,
                       const int32_t *goal, const int32_t *actions, int rules)
{
    static const int dr[5] = {0, -1, 1, 0, 0}, dc[5] = {0, 0, 0, -1, 1};
    int *tgt = (int *)malloc(sizeof(int) * (size_t)n);
    int *cur = (int *)malloc(sizeof(int) * (size_t)n);
    int *who = (int *)malloc(sizeof(int) * (size_t)H * W);
    for (int i = 0; i < H * W; i++) who[i] = -1;
    for (int a = 0; a < n; a++) {
        cur[a] = pos[2 * a] * W + pos[2 * a + 1];
        who[cur[a]] = a;
    }
    for (int a = 0; a < n; a++) {
        int act = actions[a];
        if (act < 0 || act > 4) act = 0;
        int nr = pos[2 * a] + dr[act], nc = pos[2 * a + 1] + dc[act];
        int t = nr * W + nc;
        if (nr < 0 || nr >= H || nc < 0 || nc >= W || grid[t] != 0) t = cur[a];   /* rule 1 */
        else if ((rules & 1) && who[t] >= 0 && who[t] != a) t = cur[a];            /* ORC_RULE_NO_FOLLOW */
        tgt[a] = t;
    }
    int *swap = (int *)calloc((size_t)n, sizeof(int));
    for (int a = 0; a < n; a++) {                              /* rule 2, decided on the rule-1 targets */
        if (tgt[a] == cur[a]) continue;
        int b = who[tgt[a]];
        if (b >= 0 && b != a && tgt[b] == cur[a]) swap[a] = 1;
    }
    for (int a = 0; a < n; a++) if (swap[a]) tgt[a] = cur[a];
    free(swap);
    int *nt = (int *)malloc(sizeof(int) * (size_t)n);
    for (;;) {                                                 /* rule 3 (Jacobi rounds: every test reads the previous round) */
        int changed = 0;
        for (int a = 0; a < n; a++) {
            nt[a] = tgt[a];
            if (tgt[a] == cur[a]) continue;
            int revert = 0;
            for (int b = 0; b < n && !revert; b++) {
                if (b == a || tgt[b] != tgt[a]) continue;
                if (!(rules & 2)) revert = 1;                              /* any other claimant */
                else revert = (tgt[b] == cur[b]) || (b < a);               /* a stayer, or a mover with a smaller id */
            }
            if (revert) { nt[a] = cur[a]; changed = 1; }
        }
        memcpy(tgt, nt, sizeof(int) * (size_t)n);
        if (!changed) break;
    }
    free(nt);
    int on_goal = 0;
    for (int a = 0; a < n; a++) {
        pos[2 * a] = tgt[a] / W;
        pos[2 * a + 1] = tgt[a] % W;
        if (pos[2 * a] == goal[2 * a] && pos[2 * a + 1] == goal[2 * a + 1]) on_goal++;
    }
    free(tgt); free(cur); free(who);
    return on_goal;
}

int orc_env_step(const uint8_t *grid, int H, int W, int n, int32_t *pos, const int32_t *goal, const int32_t *actions)
{
    return orc_env_step_rules(grid, H, W, n, pos, goal, actions, 0);
}
