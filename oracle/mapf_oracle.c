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
#define ORC_CTX 256       /* cpp:386 hard-codes 256 */
#define ORC_NHIST 5       /* inf:16 num_previous_actions */
#define ORC_NAGENTS 13    /* inf:15 num_agents */
#define ORC_R 5           /* inf:18-19 agents_radius == cost2go_radius == 5 */
#define ORC_LIMIT 20      /* inf:17 cost2go_value_limit */

/* Vocabulary, cpp:321-350: ints -20..20 -> 0..40, -80 -> 41, -40 -> 42, +40 -> 43,
 * n,w,u,d,l,r -> 44..49, "0000".."1111" -> 50..65, "!" -> 66. */
enum { TOK_UNREACH = 41, TOK_NEG = 42, TOK_POS = 43, TOK_N = 44, TOK_BITS0 = 50, TOK_PAD = 66 };

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
static uint8_t next_action_token(const uint16_t *dist, int W, int r, int c)
{
    int cur = dist[r * W + c];
    int u = dist[(r - 1) * W + c] < cur;
    int d = dist[(r + 1) * W + c] < cur;
    int l = dist[r * W + c - 1] < cur;
    int rr = dist[r * W + c + 1] < cur;
    return (uint8_t)(TOK_BITS0 + 8 * u + 4 * d + 2 * l + rr);
}

typedef struct {
    int H, W, n;
    uint8_t *grid;      /* H*W, non-zero = blocked (h:112 copies the grid) */
    int32_t *occ;       /* agents_locations, h:116: -1 or agent id */
    int32_t *pos;       /* n*2 (row, col) padded coords */
    int32_t *goal;      /* n*2 */
    uint8_t *hist;      /* n*5 tokens, oldest -> newest */
    uint8_t *next;      /* n */
    uint16_t *dist;     /* n*H*W */
    int32_t *org;       /* n*2: (left_border, top_border) of the agent's cached partial window, cpp:204-207 */
    int step;           /* cfg.grid_step, h:38 */
} orc_gen;

orc_gen *orc_gen_create(const uint8_t *grid, int H, int W)
{
    orc_gen *g = (orc_gen *)calloc(1, sizeof(orc_gen));
    g->H = H; g->W = W; g->n = 0;
    g->step = ORC_STEP;
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
static void window_origin(int r, int c, int step, int32_t *org)
{
    org[0] = (r - ORC_R > 0 ? r - ORC_R : 0) / step * step;
    org[1] = (c - ORC_R > 0 ? c - ORC_R : 0) / step * step;
}

/* InputParameters.grid_step (h:38, cpp:551); call before create_agents */
void orc_gen_set_grid_step(orc_gen *g, int step) { if (step > 0) g->step = step; }

/* cpp:391-410: history = "n" x5, distance field, greedy bits.  Does NOT touch agents_locations. */
void orc_gen_create_agents(orc_gen *g, int n, const int32_t *pos, const int32_t *goal)
{
    free(g->pos); free(g->goal); free(g->hist); free(g->next); free(g->dist); free(g->org);
    g->n = n;
    g->org = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)n);
    g->pos = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)n);
    g->goal = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)n);
    g->hist = (uint8_t *)malloc((size_t)n * ORC_NHIST);
    g->next = (uint8_t *)malloc((size_t)n);
    g->dist = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)n * g->H * g->W);
    memcpy(g->pos, pos, sizeof(int32_t) * 2 * (size_t)n);
    memcpy(g->goal, goal, sizeof(int32_t) * 2 * (size_t)n);
    memset(g->hist, TOK_N, (size_t)n * ORC_NHIST);           /* cpp:403-406 */
    for (int a = 0; a < n; a++) {
        uint16_t *d = g->dist + (size_t)a * g->H * g->W;
        orc_bfs(g->grid, g->H, g->W, goal[2 * a], goal[2 * a + 1], d);
        window_origin(pos[2 * a], pos[2 * a + 1], g->step, g->org + 2 * a);   /* cpp:408 compute_cost2go_partial */
        g->next[a] = next_action_token(d, g->W, pos[2 * a], pos[2 * a + 1]);
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
        uint8_t *h = g->hist + (size_t)a * ORC_NHIST;
        int act = actions[a];
        memmove(h, h + 1, ORC_NHIST - 1);                     /* cpp:463 pop_front */
        h[ORC_NHIST - 1] = (uint8_t)((act >= 0 && act <= 4) ? TOK_N + 1 + act : TOK_N);   /* cpp:442-462 */
        if (g->goal[2 * a] != goal[2 * a] || g->goal[2 * a + 1] != goal[2 * a + 1]) {       /* cpp:464-468 */
            g->goal[2 * a] = goal[2 * a];
            g->goal[2 * a + 1] = goal[2 * a + 1];
            orc_bfs(g->grid, g->H, g->W, goal[2 * a], goal[2 * a + 1], g->dist + (size_t)a * g->H * W);
            window_origin(pos[2 * a], pos[2 * a + 1], g->step, g->org + 2 * a);
        } else {
            /* cpp:469-477: the observation window left the cached partial box -> recomputed around the new position.
             * The full-grid field needs no recompute; only the box origin moves (it decides the corner cell below). */
            int left = g->org[2 * a], top = g->org[2 * a + 1];
            int right = left + 2 * g->step < g->H - 1 ? left + 2 * g->step : g->H - 1;
            int bottom = top + 2 * g->step < W - 1 ? top + 2 * g->step : W - 1;
            if (pos[2 * a] - ORC_R < left || pos[2 * a] + ORC_R > right || pos[2 * a + 1] - ORC_R < top || pos[2 * a + 1] + ORC_R > bottom)
                window_origin(pos[2 * a], pos[2 * a + 1], g->step, g->org + 2 * a);
        }
    }
    for (int a = 0; a < n; a++)                               /* cpp:483-484 */
        g->next[a] = next_action_token(g->dist + (size_t)a * g->H * W, W, g->pos[2 * a], g->pos[2 * a + 1]);
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* cpp:516-528 = cpp:288-311 (window) + cpp:487-514 (neighbours) + cpp:352-389 (encode). */
void orc_gen_generate_observations(const orc_gen *g, uint8_t *out /* n*256 */)
{
    int n = g->n, H = g->H, W = g->W;
    for (int a = 0; a < n; a++) {
        uint8_t *row = out + (size_t)a * ORC_CTX;
        const uint16_t *d = g->dist + (size_t)a * g->H * W;
        int pr = g->pos[2 * a], pc = g->pos[2 * a + 1];
        memset(row, TOK_PAD, ORC_CTX);                        /* cpp:375-376, 386-387 */
        int mid = d[pr * W + pc];                             /* cpp:297 */
        /* the unseeded corner of the cached partial window (cpp:178-198), if it exists and is in view */
        int cr = g->org[2 * a] + 2 * g->step, cc = g->org[2 * a + 1] + 2 * g->step;
        int corner_in_view = cr <= H - 1 && cc <= W - 1 && pr + ORC_R == cr && pc + ORC_R == cc;
        for (int i = 0; i <= 2 * ORC_R; i++)
            for (int j = 0; j <= 2 * ORC_R; j++) {
                int v = d[(pr - ORC_R + i) * W + (pc - ORC_R + j)];
                if (corner_in_view && i == 2 * ORC_R && j == 2 * ORC_R && v != ORC_UNREACH && v != 0) {
                    /* reached only from its two in-window neighbours (exact border seeds): cpp:252-268 */
                    int n1 = d[(cr - 1) * W + cc], n2 = d[cr * W + cc - 1];
                    int m = n1 < n2 ? n1 : n2;
                    v = m == ORC_UNREACH ? ORC_UNREACH : m + 1;
                }
                uint8_t t;
                if (v == ORC_UNREACH) t = TOK_UNREACH;        /* cpp:308-309: -80 -> 41 */
                else {
                    v -= mid;                                 /* cpp:304 */
                    if (v > ORC_LIMIT) t = TOK_POS;           /* +40 -> 43 */
                    else if (v < -ORC_LIMIT) t = TOK_NEG;     /* -40 -> 42 */
                    else t = (uint8_t)(v + ORC_LIMIT);
                }
                row[i * (2 * ORC_R + 1) + j] = t;
            }
        /* cpp:487-505: row-major scan of agents_locations, sort by (Manhattan, id) */
        int cand[(2 * ORC_R + 1) * (2 * ORC_R + 1)], key[(2 * ORC_R + 1) * (2 * ORC_R + 1)], nc = 0;
        for (int i = -ORC_R; i <= ORC_R; i++)
            for (int j = -ORC_R; j <= ORC_R; j++) {
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
        int m = nc < ORC_NAGENTS ? nc : ORC_NAGENTS;          /* cpp:506 */
        for (int s = 0; s < m; s++) {
            int b = cand[s];
            uint8_t *o = row + 121 + 10 * s;
            /* cpp:358-361: relative pos via int_vocab.at (always within +-5), goal clamped to +-20 */
            o[0] = (uint8_t)(g->pos[2 * b] - pr + ORC_LIMIT);
            o[1] = (uint8_t)(g->pos[2 * b + 1] - pc + ORC_LIMIT);
            o[2] = (uint8_t)(clampi(g->goal[2 * b] - pr, -ORC_LIMIT, ORC_LIMIT) + ORC_LIMIT);
            o[3] = (uint8_t)(clampi(g->goal[2 * b + 1] - pc, -ORC_LIMIT, ORC_LIMIT) + ORC_LIMIT);
            memcpy(o + 4, g->hist + (size_t)b * ORC_NHIST, ORC_NHIST);   /* cpp:364-367 */
            o[9] = g->next[b];                                            /* cpp:368 */
        }
    }
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
