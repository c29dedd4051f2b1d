/*
 * go_oracle.c -- CPU restatement of the Transgo Go rules engine.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP rules kernels (transgo_amd/csrc).  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load it; the product path never does.
 *
 * It restates the OBSERVABLE behaviour of the reference engine /root/reference/GoEnv/cpp_src (cited per
 * function below as file:line) with a different mechanism: the reference keeps incremental per-block liberty
 * counts on linked lists (board.cc:217-428); this file recomputes groups and liberties by flood fill on every
 * call.  The two agree on every reachable position (SURVEY.md §7 hard part (ii)); tests/test_oracle_rules.py pins
 * this file against golden vectors captured from the compiled reference (tests/golden/rules_*.npz) and, where
 * /root/reference is present, against oracle/_ref/GoEnv/go_env.so directly.
 *
 * No globals: board size, komi, max_step and the plane count travel in og_cfg (the reference keeps them in
 * file-static variables, go_env.cc:9-12, and a compile-time BOARD_SIZE, go_comm.h:20).
 */
#include <stdint.h>
#include <string.h>

#define OG_MAXS 19
#define OG_MAXP (OG_MAXS * OG_MAXS)
#define OG_EMPTY 0
#define OG_BLACK 1
#define OG_WHITE 2
#define OG_PASS (-1)
#define OG_RESIGN (-2)
#define OG_INVALID (-3)

typedef struct {
    int32_t size;       /* board edge S (9 or 19) */
    int32_t max_step;   /* go_env.cc:11  */
    float komi;         /* go_env.cc:12  */
    int32_t encode_dim; /* 9, 10 or 13; go_env.cc:96-115 */
} og_cfg;

typedef struct {
    uint8_t color[OG_MAXP]; /* 0 empty, 1 black, 2 white (go_comm.h:32-35) */
    int16_t last_move1;     /* board.h:47 */
    int16_t last_move2;     /* board.h:48 */
    int16_t ko_location;    /* board.h:51 */
    int16_t ko_age;         /* board.h:53 */
    uint16_t step_count;    /* board.h:46 */
    uint8_t ko_color;       /* board.h:52 */
    uint8_t next_player;    /* board.h:45 */
    uint8_t terminated;     /* go_env.h:17 */
    uint8_t pad[3];
} og_state;

/* neighbour order L,U,R,D = go_comm.h:44-45 (only matters for nothing observable, kept anyway) */
static const int DX[4] = {-1, 0, 1, 0};
static const int DY[4] = {0, -1, 0, 1};
static const int GX[4] = {-1, -1, 1, 1}; /* diagonals go_comm.h:49-50 */
static const int GY[4] = {-1, 1, 1, -1};

static inline int opp(int p) { return OG_BLACK + OG_WHITE - p; }

typedef struct {
    int16_t label[OG_MAXP];  /* group id for stones (index of a representative), -1 for empty */
    int16_t libs[OG_MAXP];   /* liberties of the group, indexed by label */
    int16_t nstones[OG_MAXP];
} og_groups;

/* Groups + liberties by flood fill.  Equivalent observable of blocks[].liberties/num_stones (board.h:19-24). */
static void analyze(const og_cfg *cfg, const og_state *st, og_groups *g) {
    const int S = cfg->size, P = S * S;
    int16_t stack[OG_MAXP];
    uint8_t seen_lib[OG_MAXP];
    for (int i = 0; i < P; ++i) { g->label[i] = -1; g->libs[i] = 0; g->nstones[i] = 0; }
    for (int c0 = 0; c0 < P; ++c0) {
        if (st->color[c0] == OG_EMPTY || g->label[c0] >= 0) continue;
        const int col = st->color[c0];
        int sp = 0, nl = 0, ns = 0;
        memset(seen_lib, 0, (size_t)P);
        stack[sp++] = (int16_t)c0; g->label[c0] = (int16_t)c0;
        while (sp) {
            int c = stack[--sp]; ++ns;
            int x = c % S, y = c / S;
            for (int d = 0; d < 4; ++d) {
                int nx = x + DX[d], ny = y + DY[d];
                if (nx < 0 || nx >= S || ny < 0 || ny >= S) continue;
                int n = ny * S + nx;
                if (st->color[n] == OG_EMPTY) { if (!seen_lib[n]) { seen_lib[n] = 1; ++nl; } }
                else if (st->color[n] == col && g->label[n] < 0) { g->label[n] = (int16_t)c0; stack[sp++] = (int16_t)n; }
            }
        }
        g->libs[c0] = (int16_t)nl; g->nstones[c0] = (int16_t)ns;
    }
}

/* board.cc:130-158 isSuicideMove on the pre-move neighbourhood (board.cc:90-127). */
static int is_suicide(const og_cfg *cfg, const og_state *st, const og_groups *g, int c, int player) {
    const int S = cfg->size; int x = c % S, y = c / S;
    for (int d = 0; d < 4; ++d) {
        int nx = x + DX[d], ny = y + DY[d];
        if (nx < 0 || nx >= S || ny < 0 || ny >= S) continue;
        int n = ny * S + nx;
        if (st->color[n] == OG_EMPTY) return 0;
        int l = g->libs[g->label[n]];
        if (st->color[n] == player) { if (l > 1) return 0; }
        else if (l == 1) return 0;
    }
    return 1;
}

/* board.cc:198-200 */
static inline int ko_violation(const og_state *st, int c, int player) {
    return st->ko_location == c && st->ko_age == 0 && st->ko_color == player;
}

/* board.cc:432-464 TryPlay (board points only). */
static int legal_point(const og_cfg *cfg, const og_state *st, const og_groups *g, int c, int player) {
    if (st->color[c] != OG_EMPTY) return 0;
    if (ko_violation(st, c, player)) return 0;
    return !is_suicide(cfg, st, g, c, player);
}

/* go_env.cc:34-41 + board.cc:13-26 */
int og_reset(const og_cfg *cfg, og_state *st) {
    (void)cfg;
    memset(st, 0, sizeof(*st));
    st->next_player = OG_BLACK;
    st->last_move1 = OG_INVALID; st->last_move2 = OG_INVALID; st->ko_location = OG_INVALID;
    st->step_count = 1;
    return 1;
}

/* board.cc:665-714 */
static int is_true_eye(const og_cfg *cfg, const og_state *st, int c, int player) {
    const int S = cfg->size; int x = c % S, y = c / S;
    if (st->color[c] != OG_EMPTY) return 0;
    for (int d = 0; d < 4; ++d) {
        int nx = x + DX[d], ny = y + DY[d];
        if (nx < 0 || nx >= S || ny < 0 || ny >= S) continue;
        if (st->color[ny * S + nx] != player) return 0;
    }
    int nopp = 0, nwall = 0;
    for (int d = 0; d < 4; ++d) {
        int nx = x + GX[d], ny = y + GY[d];
        if (nx < 0 || nx >= S || ny < 0 || ny >= S) { ++nwall; continue; }
        if (st->color[ny * S + nx] == opp(player)) ++nopp;
    }
    int fake = (nwall > 0 && nopp >= 1) || (nwall == 0 && nopp >= 2);
    return !fake;
}

/* board.cc:731-817 GivenBlockLives for the group labelled `lab`. */
static int group_lives(const og_cfg *cfg, const og_state *st, const og_groups *g, int lab) {
    const int S = cfg->size, P = S * S;
    if (g->libs[lab] <= 1) return 0;
    const int col = st->color[lab];
    uint8_t cand[OG_MAXP]; int ncand = 0;
    memset(cand, 0, (size_t)P);
    for (int c = 0; c < P; ++c) {
        if (g->label[c] != lab) continue;
        int x = c % S, y = c / S;
        for (int d = 0; d < 4; ++d) {
            int nx = x + DX[d], ny = y + DY[d];
            if (nx < 0 || nx >= S || ny < 0 || ny >= S) continue;
            int n = ny * S + nx;
            if (!cand[n] && is_true_eye(cfg, st, n, col)) { cand[n] = 1; ++ncand; }
        }
    }
    if (ncand <= 1) return 0;
    int good = 0;
    for (int e = 0; e < P; ++e) {
        if (!cand[e]) continue;
        int x = e % S, y = e / S, nb = 0, nt = 0;
        for (int d = 0; d < 4; ++d) {
            int nx = x + GX[d], ny = y + GY[d];
            if (nx < 0 || nx >= S || ny < 0 || ny >= S) { ++nb; continue; }
            int n = ny * S + nx;
            if (st->color[n] == OG_EMPTY) { if (cand[n]) ++nt; }
            else if (st->color[n] == col) ++nt;
        }
        if ((nb >= 1 && nb + nt == 4) || (nb == 0 && nt >= 3)) ++good;
    }
    return good >= 2;
}

/* go_env.cc:44-80 Step_ -> board.cc:432-464 TryPlay2 -> board.cc:546-653 Play.
 * action in [0,P) = board point, P or -1 = pass, -2 = resign.  Returns done.  *ok (optional) = move was legal. */
int og_step_inplace(const og_cfg *cfg, og_state *st, int action, int *ok) {
    const int S = cfg->size, P = S * S;
    if (ok) *ok = 1;
    if (st->terminated) return 1;                      /* go_env.cc:52-55 */
    if (action == P) action = OG_PASS;                 /* go_env.cc:56-57 */
    const int player = st->next_player;
    if (action == OG_PASS || action == OG_RESIGN) {    /* board.cc:554-558: ko is NOT aged by a pass */
        st->next_player = (uint8_t)opp(player);
        st->last_move2 = st->last_move1; st->last_move1 = (int16_t)action; st->step_count++;
        int done = st->step_count > 1 &&               /* board.cc:656-661 */
                   ((st->last_move1 == OG_PASS && st->last_move2 == OG_PASS) || st->last_move1 == OG_RESIGN);
        if (done || st->step_count > cfg->max_step) { st->terminated = 1; return 1; }
        return 0;
    }
    og_groups g;
    if (action < 0 || action >= P) { if (ok) *ok = 0; return 0; }
    analyze(cfg, st, &g);
    if (!legal_point(cfg, st, &g, action, player)) { if (ok) *ok = 0; return 0; } /* go_env.cc:75-79 */
    const int c = action, x = c % S, y = c / S;
    /* board.cc:163-194 isGivingSimpleKo, evaluated on the pre-move neighbourhood */
    int self_lib = 0, own_nb = 0, n11 = 0, ko_at = OG_INVALID;
    for (int d = 0; d < 4; ++d) {
        int nx = x + DX[d], ny = y + DY[d];
        if (nx < 0 || nx >= S || ny < 0 || ny >= S) continue;
        int n = ny * S + nx;
        if (st->color[n] == OG_EMPTY) { ++self_lib; continue; }
        if (st->color[n] == player) { own_nb = 1; continue; }
        int lab = g.label[n];
        if (g.libs[lab] == 1 && g.nstones[lab] == 1) { ++n11; ko_at = n; }
    }
    if (self_lib == 0 && !own_nb && n11 == 1) {        /* board.cc:561-570 */
        st->ko_location = (int16_t)ko_at; st->ko_color = (uint8_t)opp(player); st->ko_age = 0;
    } else {
        st->ko_age++;
    }
    /* captures: enemy neighbour groups whose only liberty was c (board.cc:606-630) */
    for (int d = 0; d < 4; ++d) {
        int nx = x + DX[d], ny = y + DY[d];
        if (nx < 0 || nx >= S || ny < 0 || ny >= S) continue;
        int n = ny * S + nx;
        if (st->color[n] != opp(player)) continue;
        int lab = g.label[n];
        if (g.libs[lab] != 1) continue;
        for (int q = 0; q < P; ++q) if (g.label[q] == lab) st->color[q] = OG_EMPTY;
    }
    st->color[c] = (uint8_t)player;
    st->next_player = (uint8_t)opp(player);            /* board.cc:536-542 */
    st->last_move2 = st->last_move1; st->last_move1 = (int16_t)c; st->step_count++;
    if (st->step_count > cfg->max_step) { st->terminated = 1; return 1; } /* go_env.cc:67 */
    return 0;
}

/* go_env.cc:44-47 */
int og_step(const og_cfg *cfg, const og_state *st, og_state *next, int action, int *ok) {
    memcpy(next, st, sizeof(*st));
    return og_step_inplace(cfg, next, action, ok);
}

/* go_env.cc:85-89 */
int og_check_action(const og_cfg *cfg, const og_state *st, int action) {
    const int P = cfg->size * cfg->size;
    /* go_env.cc:84-88 hands the coordinate to TryPlay2 untranslated (only Step maps S*S to PASS): the internal PASS / RESIGN codes
     * are "playable" (board.cc:440-445), S*S itself is off the board (board.cc:448-451) */
    if (action == OG_PASS || action == OG_RESIGN) return 1;
    if (action < 0 || action >= P) return 0;
    og_groups g; analyze(cfg, st, &g);
    return legal_point(cfg, st, &g, action, st->next_player);
}

/* go_env.cc:154-164: ascending legal points, pass (= P) appended last.  The pass filter is Python's
 * (environment.py:121-129) and lives in the wrapper. */
int og_legal_actions(const og_cfg *cfg, const og_state *st, int32_t *out) {
    const int P = cfg->size * cfg->size; int n = 0;
    og_groups g; analyze(cfg, st, &g);
    for (int c = 0; c < P; ++c) if (legal_point(cfg, st, &g, c, st->next_player)) out[n++] = c;
    out[n++] = P;
    return n;
}

/* go_env.cc:171-181 + board.cc:492-517 */
int og_legal_no_eye(const og_cfg *cfg, const og_state *st, int32_t *out) {
    const int P = cfg->size * cfg->size; int n = 0;
    og_groups g; analyze(cfg, st, &g);
    for (int c = 0; c < P; ++c)
        if (legal_point(cfg, st, &g, c, st->next_player) && !is_true_eye(cfg, st, c, st->next_player)) out[n++] = c;
    out[n++] = P;
    return n;
}

/* go_env.cc:96-115 + board_feature.cc:17-253.  Plane list (encode10, board_feature.cc:213-223):
 * 0-2 own groups with 1/2/>=3 liberties, 3-5 opponent same, 6 last move, 7 ko point U suicide points of the side to
 * move, 8 own true eyes, 9 own alive groups.  encode9 drops 9; encode13 = {0-5, last1, last2, ko/suicide, own eyes,
 * opp eyes, own alive, opp alive} (board_feature.cc:228-253). */
int og_encode(const og_cfg *cfg, const og_state *st, float *out) {
    const int S = cfg->size, P = S * S, C = cfg->encode_dim;
    if (C != 9 && C != 10 && C != 13) return 0;
    memset(out, 0, sizeof(float) * (size_t)C * (size_t)P);
    og_groups g; analyze(cfg, st, &g);
    const int me = st->next_player, op = opp(me);
    for (int c = 0; c < P; ++c) {
        if (st->color[c] == OG_EMPTY) continue;
        int l = g.libs[g.label[c]];
        int base = (st->color[c] == me) ? 0 : 3;
        int k = l == 1 ? 0 : l == 2 ? 1 : l >= 3 ? 2 : -1;
        if (k >= 0) out[(base + k) * P + c] = 1.0f;
    }
    int pl_hist1 = 6, pl_hist2 = -1, pl_ko = 7, pl_eye = 8, pl_oeye = -1, pl_live = 9, pl_olive = -1;
    if (C == 9) pl_live = -1;
    if (C == 13) { pl_hist2 = 7; pl_ko = 8; pl_eye = 9; pl_oeye = 10; pl_live = 11; pl_olive = 12; }
    if (st->last_move1 >= 0 && st->last_move1 < P) out[pl_hist1 * P + st->last_move1] = 1.0f;
    if (pl_hist2 >= 0 && st->last_move2 >= 0 && st->last_move2 < P) out[pl_hist2 * P + st->last_move2] = 1.0f;
    for (int c = 0; c < P; ++c)     /* board_feature.cc:69-89, board.cc:520-533: suicide ignores ko */
        if (st->color[c] == OG_EMPTY && is_suicide(cfg, st, &g, c, me)) out[pl_ko * P + c] = 1.0f;
    if (st->ko_age == 0 && st->ko_location >= 0 && st->ko_location < P)   /* board.cc:205-213: ko_color ignored */
        out[pl_ko * P + st->ko_location] = 1.0f;
    for (int c = 0; c < P; ++c) {
        if (st->color[c] != OG_EMPTY) continue;
        if (is_true_eye(cfg, st, c, me)) out[pl_eye * P + c] = 1.0f;
        if (pl_oeye >= 0 && is_true_eye(cfg, st, c, op)) out[pl_oeye * P + c] = 1.0f;
    }
    for (int lab = 0; lab < P; ++lab) {
        if (st->color[lab] == OG_EMPTY || g.label[lab] != lab) continue;
        int pl = st->color[lab] == me ? pl_live : pl_olive;
        if (pl < 0) continue;
        if (group_lives(cfg, st, &g, lab))
            for (int c = 0; c < P; ++c) if (g.label[c] == lab) out[pl * P + c] = 1.0f;
    }
    return 1;
}

/* board.cc:822-958 getTTScore: Tromp-Taylor area count.  owner[] (optional): 1 black, 2 white, 3 dame.
 * Empty board: returns 0 (board.cc:932-935). */
static float tt_score(const og_cfg *cfg, const og_state *st, uint8_t *owner) {
    const int S = cfg->size, P = S * S;
    uint8_t own_local[OG_MAXP]; if (!owner) owner = own_local;
    int16_t queue[OG_MAXP]; uint8_t seen[OG_MAXP];
    int cnt[4] = {0, 0, 0, 0}, stones = 0;
    memset(seen, 0, (size_t)P);
    for (int c = 0; c < P; ++c) if (st->color[c] != OG_EMPTY) { owner[c] = st->color[c]; cnt[st->color[c]]++; ++stones; }
    for (int c0 = 0; c0 < P; ++c0) {
        if (st->color[c0] != OG_EMPTY || seen[c0]) continue;
        int qs = 0, qe = 0, touch = 0;
        queue[qe++] = (int16_t)c0; seen[c0] = 1;
        while (qs < qe) {
            int c = queue[qs++], x = c % S, y = c / S;
            for (int d = 0; d < 4; ++d) {
                int nx = x + DX[d], ny = y + DY[d];
                if (nx < 0 || nx >= S || ny < 0 || ny >= S) continue;
                int n = ny * S + nx;
                if (st->color[n] != OG_EMPTY) touch |= st->color[n];
                else if (!seen[n]) { seen[n] = 1; queue[qe++] = (int16_t)n; }
            }
        }
        int o = (touch == OG_BLACK) ? OG_BLACK : (touch == OG_WHITE) ? OG_WHITE : 3;
        for (int i = 0; i < qe; ++i) owner[queue[i]] = (uint8_t)o;
        if (o != 3) cnt[o] += qe;
    }
    if (stones == 0) return 0.0f;
    return (float)(cnt[OG_BLACK] - cnt[OG_WHITE]);
}

/* go_env.cc:126-130 */
float og_score(const og_cfg *cfg, const og_state *st) { return tt_score(cfg, st, 0) - cfg->komi; }

/* go_env.cc:136-149: +1 black / 0 dame / -1 white */
float og_territory(const og_cfg *cfg, const og_state *st, float *terr) {
    const int P = cfg->size * cfg->size;
    uint8_t owner[OG_MAXP];
    float raw = tt_score(cfg, st, owner);
    for (int c = 0; c < P; ++c) terr[c] = owner[c] == OG_BLACK ? 1.0f : owner[c] == OG_WHITE ? -1.0f : 0.0f;
    return raw - cfg->komi;
}

int og_player(const og_state *st) { return st->next_player; }     /* go_env.cc:208-210 */
int og_step_count(const og_state *st) { return st->step_count; }  /* go_env.cc:213-215 */
int og_terminated(const og_state *st) { return st->terminated; }  /* go_env.cc:91-93 */
int og_state_size(void) { return (int)sizeof(og_state); }

/* bulk form of og_check_action over every board point (test speed only) */
void og_check_all(const og_cfg *cfg, const og_state *st, uint8_t *out) {
    const int P = cfg->size * cfg->size;
    og_groups g; analyze(cfg, st, &g);
    for (int c = 0; c < P; ++c) out[c] = (uint8_t)legal_point(cfg, st, &g, c, st->next_player);
}
