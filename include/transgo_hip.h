/*
 * transgo_hip.h -- C ABI of libtransgo_hip.so, the MI355X-native drop-in for Transgo's self-play hot path.
 *
 * Plain C, plain pointers and sizes, no torch types.  Every entry point names the reference interface it replaces
 * (file:line under /root/reference).  Unless a parameter says "device", pointers are HOST memory owned by the caller;
 * the library never keeps a caller pointer after a call returns.  All functions return 0 on success or a negative
 * tg_status; tg_last_error() gives the message.  A context is single-owner (one host thread, one GPU).
 *
 * Four groups (the fourth = the reference's own 15 names, for binding-compatibility):
 *   1. tg_env_*   batched form of the 15 extern "C" functions of GoEnv/cpp_src/go_env.h:24-70 over opaque state blobs
 *                 (what GoEnv/environment.py:42-90 binds through ctypes today).
 *   2. tg_sp_*    the batched self-play engine: G concurrent games, WP_MCTS (self_play.py:575-875) as HIP tree
 *                 kernels + the policy/value network forward (model.py:79-114) on MFMA.
 *   3. tg_host_*  host-only helpers that must be bit-identical to NumPy's legacy MT19937 stream
 *                 (np.random.dirichlet / choice used at self_play.py:93, :709, :683).
 */
#ifndef TRANSGO_HIP_H_
#define TRANSGO_HIP_H_

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tg_ctx tg_ctx;

typedef enum tg_status {
    TG_OK = 0,
    TG_ERR_ARG = -1,        /* bad argument / unsupported configuration */
    TG_ERR_HIP = -2,        /* HIP runtime error (message has the HIP error string) */
    TG_ERR_NO_DEVICE = -3,  /* no usable GPU: the library has no CPU fallback */
    TG_ERR_ARENA = -4,      /* a game's tree hit its cap (cfg.arena_slots) or found the shared pool empty (results of that game are invalid) */
    TG_ERR_STATE = -5       /* call sequence violated (e.g. absorb without collect) */
} tg_status;

/* Replaces: compile-time BOARD_SIZE (go_comm.h:20), Init(history_dim, encode_dim, max_step, komi) (go_env.h:30,
 * environment.py:43-45) and the search fields of configure.py:9-37.  Zero-initialise, then set what you need;
 * tg_config_default() fills the reference defaults. */
typedef struct tg_config {
    int32_t board_size;         /* 9 or 19 */
    int32_t encode_dim;         /* 9, 10 or 13 feature planes (environment.py:36; go_env.cc:96-115) */
    int32_t max_step;           /* environment.py:37 (120) */
    float komi;                 /* environment.py:38 (7.5) */
    int32_t n_games;            /* concurrent boards G; 0 = rules-only context */
    int32_t num_simulation;     /* configure.py:29 */
    int32_t parallel_readouts;  /* configure.py:33 (4); 1..8 */
    int32_t wu_loss;            /* configure.py:32 (2) */
    double c_puct1;             /* configure.py:26 (3) */
    double c_puct2;             /* configure.py:27 (0.05) */
    int32_t arena_slots;        /* most 32-byte tree slots ONE game's tree may hold (per-game cap); 0 = (4*num_simulation + 256) blocks of
                                   the largest size.  Memory is set by pool_slots below, not by this */
    int32_t net_blocks;         /* residual blocks of the tower (BASELINE.json "N-block x F-filter") */
    int32_t net_filters;        /* channels F (multiple of 32) */
    int32_t device;             /* HIP device ordinal */
    int32_t net_precision;      /* 0 = f32 network (default); 1 = fp16 weights/activations, f32 accumulate (BASELINE config 5), f32
                                   residual stream; 2 = as 1 with the residual stream stored in fp16 too (a quarter less HBM traffic);
                                   3 = split precision, opt-in: every conv operand as fp16 hi + lo, three of the four partial products (lo*lo dropped) on the
                                   fp16 matrix cores, f32 accumulate and residual stream (fp32-level accuracy at 2.4-3.1x the
                                   simulations/s).  1 and 2 take attention-free towers of 128 / 256 filters; 3 also takes
                                   attention layers at 9x9 (the reference's MainNetwork), anything else is refused by tg_net_load */
    int32_t record_games;       /* 1 (default): every game's move record -- env.encode(root) bit-packed, raw visit counts, side to
                                   move; the three Python lists of self_play.py:917-926 -- is kept in HBM for tg_sp_harvest */
    int32_t pool_slots;         /* tree memory: 32-byte slots provisioned PER GAME ON AVERAGE in the pool all games of the context share
                                   (pool = n_games x this; a game takes chunks from it as its tree grows and returns them when it is
                                   re-rooted or restarted); 0 = (2*num_simulation + 128) blocks of the largest size.  tg_sp_pool_stats
                                   reports the fill; a game that finds the pool empty is parked like one that hits its cap */
    int32_t reserved[5];
} tg_config;

void tg_config_default(tg_config* cfg);

/* Lifetime.  tg_create fails with TG_ERR_NO_DEVICE when no GPU is present (there is no CPU path). */
int tg_create(const tg_config* cfg, tg_ctx** out);
void tg_destroy(tg_ctx* ctx);
const char* tg_last_error(const tg_ctx* ctx);       /* ctx may be NULL: last creation error */
int tg_sync(tg_ctx* ctx);                            /* wait for the context's stream */
int tg_version(void);

/* ---- 1. rules engine over caller-owned opaque states ---------------------------------------------------------------
 * A state is tg_state_size() bytes (48 at 9x9, 112 at 19x19); the reference's is a 1188-byte GoState
 * (go_env.h:15-18) that Python also treats as opaque (environment.py:93-103).  `states` arrays are n contiguous
 * blobs. */
int tg_state_size(const tg_ctx* ctx);

/* Reset (go_env.h:33, go_env.cc:34-41). */
int tg_env_reset(tg_ctx* ctx, void* states, int n);

/* Step (go_env.h:37, go_env.cc:44-80): out[i] = in[i] advanced by actions[i]; action S*S (or -1) is pass.
 * done[i] = game over (double pass or step_count > max_step).  An illegal action leaves the state unchanged with
 * done=0 and ok[i]=0 (the reference prints a message, go_env.cc:75-79); stepping a finished state returns done=1
 * unchanged (go_env.cc:52-55).  in may equal out (Step_, go_env.h:38).  ok may be NULL. */
int tg_env_step(tg_ctx* ctx, const void* in, void* out, const int32_t* actions, int n, uint8_t* done, uint8_t* ok);

/* One pass computing any subset of the read-only queries; NULL outputs are skipped.
 *   legal  u8[n][A]   getLegalAction (go_env.h:57, go_env.cc:154-164): 1 for every legal point, legal[A-1] (pass)
 *                     always 1 -- the pass filter of environment.py:121-129 is the caller's, as in the reference.
 *   noeye  u8[n][A]   getLegalNoEye (go_env.h:60, go_env.cc:171-181).
 *   obs    f32[n][C][S][S]  Encode (go_env.h:48, board_feature.cc:213-253).
 *   score  f32[n]     getScore = Tromp-Taylor - komi (go_env.h:51, go_env.cc:126-130).
 *   terr   f32[n][S*S] getTerritory: +1 black, 0 dame, -1 white (go_env.h:54, go_env.cc:136-149).
 *   player i32[n], step i32[n], terminated u8[n]: getPlayer / getStep / isTerminated (go_env.h:66,69,45). */
int tg_env_query(tg_ctx* ctx, const void* states, int n, uint8_t* legal, uint8_t* noeye, float* obs, float* score,
                 float* terr, int32_t* player, int32_t* step, uint8_t* terminated);

/* Show (go_env.h:63): prints the board of one state to stdout. */
int tg_env_show(tg_ctx* ctx, const void* state);


/* NumPy legacy RandomState stream: key/pos are get_state()[1] / get_state()[2]. */
typedef struct tg_mt19937 {
    uint32_t key[624];
    int32_t pos;
} tg_mt19937;

/* ---- 2. batched self-play engine (needs cfg.n_games > 0) -----------------------------------------------------------------
 * G games advance in lock step.  One move of every game is:
 *     tg_sp_begin_move -> { tg_sp_collect -> evaluate -> tg_sp_absorb } until no game is active
 *                      -> tg_sp_root_info (visit counts) -> host picks moves -> tg_sp_play -> evaluate -> tg_sp_expand_roots
 * "evaluate" is tg_sp_eval (network forward on the GPU) or, for parity tests with a stand-in evaluator,
 * tg_sp_batch_obs + tg_sp_set_eval.  tg_sp_search runs the whole inner loop with the network.
 * An evaluation batch is a compact array of rows; rows belong either to fresh roots or to the leaves of one wave. */

/* WP_MCTS.reset_root (self_play.py:595-605) for every game (mask NULL) or the games with mask[g] != 0:
 * empty board, np.random.seed(seeds[g]) for the game's stream.  Leaves a root batch pending. */
int tg_sp_reset(tg_ctx* ctx, const uint32_t* seeds, const uint8_t* mask);

/* WP_MCTS.select_action prologue (self_play.py:689-700): a fresh root at the given position for every game with
 * mask[g] != 0 (mask NULL = all); unmasked slots are parked and take no part in searches.  `states` = G blobs of
 * tg_state_size() bytes.  RNG streams are left as they are.  Leaves a root batch pending. */
int tg_sp_reset_from(tg_ctx* ctx, const void* states, const uint8_t* mask);
/* Current root position of every game (G blobs), e.g. to hand a game from one engine to another. */
int tg_sp_root_states(tg_ctx* ctx, void* states);

int tg_sp_batch_rows(tg_ctx* ctx, int32_t* n_rows);                        /* rows of the pending batch */
int tg_sp_batch_obs(tg_ctx* ctx, float* obs /*[n_rows][C][S][S]*/, int32_t n_rows);   /* env.encode of each row (self_play.py:798) */
int tg_sp_set_eval(tg_ctx* ctx, const float* policy /*[n_rows][A]*/, const float* value /*[n_rows]*/, int32_t n_rows);
int tg_sp_eval(tg_ctx* ctx);                                               /* model.main_prediction on the pending batch (self_play.py:777-786) */
int tg_sp_expand_roots(tg_ctx* ctx);                                       /* root.expand with raw priors (self_play.py:603-605, :867-870) */

/* get_action_probs prologue (self_play.py:659-663): Dirichlet(0.03) root noise when selfplay != 0, and the visit
 * target N0 + num_simulation (<= 0: cfg.num_simulation). */
int tg_sp_begin_move(tg_ctx* ctx, int selfplay, int num_simulation);

/* Front half of WP_MCTS.run (self_play.py:616-646) for every active game.  n_active = games still below their
 * target before this wave; n_rows = leaves awaiting evaluation. */
int tg_sp_collect(tg_ctx* ctx, int32_t* n_active, int32_t* n_rows);
/* Back half (self_play.py:651-654, :727-774). */
int tg_sp_absorb(tg_ctx* ctx);
/* while root.N < N0 + sims: run()  (self_play.py:662-664) with the network as evaluator. */
int tg_sp_search(tg_ctx* ctx, int32_t* n_waves);

/* visit counts per action (self_play.py:666-667), root visit total, side to move, ply counter, env.encode(root)
 * (self_play.py:685).  Any pointer may be NULL. */
int tg_sp_root_info(tg_ctx* ctx, int32_t* visits /*[G][A]*/, int32_t* root_n, int32_t* player, int32_t* step,
                    float* obs /*[G][C][S][S]*/);
/* One random_sample() from each game's stream: the draw inside np.random.choice(A, p=) (self_play.py:683). */
int tg_sp_draw_uniform(tg_ctx* ctx, double* u /*[G]*/, const uint8_t* mask);
int tg_sp_rng_state(tg_ctx* ctx, int game, tg_mt19937* out);
/* All G streams at once, and the way back (mask NULL = all): lets a game's stream move between contexts -- in policy_evaluate
 * (self_play.py:986-1040) the two agents of a game draw from ONE global stream, so it follows the game from the engine holding
 * the train model to the one holding the evaluation model and back, ply by ply. */
int tg_sp_rng_get(tg_ctx* ctx, tg_mt19937* out /*[G]*/);
int tg_sp_rng_set(tg_ctx* ctx, const tg_mt19937* in /*[G]*/, const uint8_t* mask /*[G] or NULL*/);
/* update_with_action (self_play.py:857-872) for every unfinished game; done[g] = 1 game over, 2 the game is parked in
 * error (tree cap reached / pool empty; see tg_sp_game_errors), 0 otherwise.  With cfg.record_games the move's record entry is
 * written first (self_play.py:917-926).  Leaves a root batch pending for the games whose new root was not yet expanded. */
int tg_sp_play(tg_ctx* ctx, const int32_t* actions /*[G]*/, uint8_t* done /*[G]*/);
/* A game whose tree outgrows its cap (cfg.arena_slots), or that finds the shared pool (cfg.pool_slots) empty, is parked: it takes no
 * further part in searches (its chunks stay its own until its slot is reset), tg_sp_play reports
 * 2 for it, every other game is unaffected, and tg_sp_reset with its mask bit starts a new game in the slot.  n_errors = games
 * parked right now (as of the last collect/play; no device round trip when err is NULL); err[g] = 0 or a bit set (1 arena,
 * 2 path depth, 4 action not among the root's children). */
int tg_sp_game_errors(tg_ctx* ctx, int32_t* n_errors, int32_t* err /*[G] or NULL*/);
/* The reference keeps the searched tree in unbounded Python memory (tree reuse, self_play.py:860); here a re-rooted tree may
 * keep at most arena_slots minus the room of one full search.  When the kept sub-tree is larger (very peaked policies, many
 * moves in a row) its deepest blocks are dropped: those nodes keep their statistics and become unexpanded leaves again.
 * blocks = how many were dropped so far over all games (0 = every search so far equals the reference's). */
int tg_sp_tree_truncations(tg_ctx* ctx, uint64_t* blocks);

/* Tree memory.  The reference keeps every tree in unbounded Python memory (Node_V objects, self_play.py:51-95); here all games of
 * a context share ONE pool of cfg.pool_slots x n_games slots.  pool_slots = its size; high_water_slots = the most that was in use
 * at once since creation; in_use_slots = now; exhausted = how often a game found no chunk left (that game is then parked in error
 * 1, tg_sp_game_errors, or its kept sub-tree truncated, tg_sp_tree_truncations -- 0 = never).  Any pointer may be NULL. */
int tg_sp_pool_stats(tg_ctx* ctx, uint64_t* pool_slots, uint64_t* high_water_slots, uint64_t* in_use_slots, uint64_t* exhausted);

/* ---- finished games -> training positions, on the device (self_play.py:929-967) ------------------------------------------
 * Games the LAST tg_sp_play finished, in ascending slot order, and the number of recorded positions (= moves) they hold. */
int tg_sp_finished(tg_ctx* ctx, int32_t* n_games, int32_t* n_positions);
/* Their positions as one position-major batch, game after game in that order, plies ascending -- the arrays
 * tg_replay_append(_dev) takes: obs_bits u32[n_positions][ceil(C*S*S/32)] (env.encode bit-packed, bit i = plane-major flat
 * index i), counts i32[n_positions][A] (raw visit counts; pi = counts with 1 -> 0, / sum, self_play.py:666-671), z
 * f32[n_positions] (+1 mover == winner else -1, :931-934), own i8[n_positions][S*S] (territory from the mover's side,
 * :938-940), player u8[n_positions] (may be NULL).  device_out != 0: those five are pointers into this GPU's memory (e.g. the
 * payload tensors of an RCCL gather) and nothing bulky crosses PCIe.  Per-game tables are host arrays and may be NULL:
 * slot / n_moves / winner (1 black, 2 white, environment.py:118-119) / score (getScore, go_env.cc:126-130) [n_games], terr
 * i8[n_games][S*S] (getTerritory, +1 black / 0 / -1 white).  Call before tg_sp_reset restarts the slots.  The 8-fold
 * augmentation (self_play.py:943-965) is applied by the consumer (tg_replay_sample; transgo_amd.self_play.game_targets). */
int tg_sp_harvest(tg_ctx* ctx, uint32_t* obs_bits, int32_t* counts, float* z, int8_t* own, uint8_t* player, int device_out,
                  int32_t* slot, int32_t* n_moves, int32_t* winner, float* score, int8_t* terr);
/* getScoreAndTerritory / getWinner of the current root position (self_play.py:932-937). */
int tg_sp_final(tg_ctx* ctx, float* score /*[G]*/, float* terr /*[G][S*S]*/, int32_t* winner /*[G]*/);
/* Aggregate counters: completed simulations, evaluated leaves, summed selection depth, RNG words drawn by tie
 * breaks, games in error, and max_slots = the most slots any single game's tree has held (whole chunks; the pool's own fill is
 * tg_sp_pool_stats). */
int tg_sp_stats(tg_ctx* ctx, uint64_t* sims, uint64_t* evals, uint64_t* depth_sum, uint64_t* tie_draws, int32_t* errors,
                int32_t* max_slots);

/* ---- network -------------------------------------------------------------------------------------------------------------
 * Replaces TransGoNetwork.set_weights / main_prediction (model.py:17-27).  The blob is the BatchNorm-folded, kernel-layout
 * packing of the model's state_dict produced by transgo_amd/model.py:pack_weights (layout documented there and in
 * DESIGN.md); tg_net_blob_floats gives its exact length.  Loading again later is the weight refresh of
 * self_play.py:913.  rows_cap: largest batch tg_net_predict will be asked for (engine contexts size it themselves). */
size_t tg_net_blob_floats(int board_size, int encode_dim, int filters, int blocks);
int tg_net_load(tg_ctx* ctx, const float* blob, size_t n_floats, int rows_cap);
/* Same with an explicit layer program: one letter per trunk layer, 'R' = pre-activation ResidualBlock (model.py:238-248),
 * 'A' = Self_Attention (model.py:288-315), optional "+P" = attention in the policy head (model.py:72,106).  The shipped
 * MainNetwork (model.py:49-76) is "RARRRARRRRAR+P"; tg_net_load uses cfg.net_blocks x 'R'.  Attention needs 9x9. */
size_t tg_net_blob_floats_arch(int board_size, int encode_dim, int filters, const char* arch);
int tg_net_load_arch(tg_ctx* ctx, const char* arch, const float* blob, size_t n_floats, int rows_cap);
/* Weight refresh that never stalls a search (the hand-off of trainer.py:76-79 -> self_play.py:913): there are two complete
 * weight sets; this call copies the blob to pinned memory and uploads + re-stages it into the idle set on a side stream, then
 * returns.  Searches keep running on the live set; the switch happens at a boundary only: the next tg_sp_begin_move (so one
 * move's search, and its recorded pi, never mixes two weight sets) or the next tg_net_predict that finds the upload complete.
 * tg_sp_eval / tg_sp_collect / tg_sp_absorb never switch: a host that drives those itself and never calls tg_sp_begin_move must
 * call tg_net_load_poll(ctx, 0, &pending) where it wants a finished refresh to take over.
 * Falls back to the synchronous load when no network of this architecture is loaded yet.  One refresh in flight at a time (a
 * second call first waits for the previous one).  tg_net_load_poll: pending = 1 while a refresh is not adopted yet; wait != 0
 * blocks until it is. */
int tg_net_load_async(tg_ctx* ctx, const char* arch, const float* blob, size_t n_floats);
/* The same with the blob in THIS GPU's memory (e.g. the buffer an RCCL broadcast filled): copied device -> device, no host
 * bounce.  Its contents must be complete when the call is made; the buffer is free again when the call returns.  Without a
 * network of this architecture loaded yet it falls back to the synchronous load (one copy through the host). */
int tg_net_load_async_dev(tg_ctx* ctx, const char* arch, const float* d_blob /*device*/, size_t n_floats);
int tg_net_load_poll(tg_ctx* ctx, int wait, int* pending);
/* main_prediction (model.py:17-20) on host buffers: obs f32[n][C][S][S] -> policy f32[n][A] (softmax), value f32[n]
 * (tanh), own f32[n][S*S] (tanh; may be NULL). */
int tg_net_predict(tg_ctx* ctx, const float* obs, int n_rows, float* policy, float* value, float* own);
/* Range report.  The reference network is f32 end to end (model.py:79-114) and accepts any trained checkpoint (model.py:23-27);
 * net_precision 1 / 2 / 3 carry activations as fp16 (3: as fp16 hi + lo) and lose f32's range: beyond +-65504 the fp16 copy is
 * inf and the layers behind it compute NaN.  Every kernel that rounds to fp16 checks what it rounds; fp16_overflows = sticky
 * count, since the network was created, of output tiles (convs) / boards (attention) that held such a value (0 = all forward
 * passes so far stayed in range; always 0 with net_precision 0); a non-zero count also leaves a message in tg_last_error.
 * weight_absmax = largest |w| of the live weight set's BatchNorm-folded blob (inf / NaN if it holds one).  Either may be NULL. */
int tg_net_range(tg_ctx* ctx, uint64_t* fp16_overflows, float* weight_absmax);
/* HIP-event timing of the dominant kernel (3x3 conv F->F) on the launch stream: enable, run, read totals. */
int tg_prof_enable(tg_ctx* ctx, int on, int max_launches);
int tg_prof_read(tg_ctx* ctx, double* conv_ms, int64_t* conv_launches, double* conv_flops);
/* tg_prof_read drains the event pool into running totals (call it between steps); launches that found the pool full are not in
 * the totals and are counted here (0 = the totals cover every launch since tg_prof_enable). */
int tg_prof_skipped(tg_ctx* ctx, int64_t* launches);
/* The same for the tree stage (k_collect = selection + leaf step + pseudo-expansion + feature planes, self_play.py:607-650;
 * k_absorb = complete_update + backup, :651-654, :727-764): event time per kind, number of waves, and the number of children
 * scored by PUCT summed over all selection levels since the last reset (mean fan-out = children_scored / depth_sum). */
int tg_prof_enable_tree(tg_ctx* ctx, int on, int max_waves);
int tg_prof_read_tree(tg_ctx* ctx, double* collect_ms, double* absorb_ms, int64_t* waves, uint64_t* children_scored);

/* ---- device-resident replay store + batch sampler (replaces replay_buffer.py:30-47 + the stacking of trainer.py:46-54) ----
 * Positions are stored once, un-augmented and compact; entry e of the reference's ring of augmented tuples is
 * (position e/8, symmetry e%8) in the reference's append order (self_play.py:943-965: for i in 1..4: rot90(i), fliplr of
 * that).  tg_replay_sample materialises entries as the four float32 arrays the trainer feeds the network. */
typedef struct tg_replay tg_replay;
int tg_replay_create(tg_ctx* ctx, int capacity_positions, tg_replay** out);
void tg_replay_destroy(tg_replay* rp);
int tg_replay_append(tg_replay* rp, const uint32_t* obs_bits /*[n][ceil(C*S*S/32)]*/, const int32_t* counts /*[n][A]*/,
                     const float* z /*[n]*/, const int8_t* own /*[n][S*S]*/, int n);
/* The same with the four arrays in this GPU's memory (tg_sp_harvest with device_out, or what an RCCL gather delivered). */
int tg_replay_append_dev(tg_replay* rp, const uint32_t* obs_bits, const int32_t* counts, const float* z, const int8_t* own, int n);
int tg_replay_info(const tg_replay* rp, long long* entries, long long* index, int* full);
int tg_replay_sample(tg_replay* rp, const long long* entry /*[B]*/, int B, float* state /*[B][C][S][S]*/, float* pi /*[B][A]*/,
                     float* z /*[B]*/, float* own /*[B][S*S]*/, int device_out);

/* ---- 3. host-only NumPy-legacy MT19937 helpers ------------------------------------------------------------------------
 * Same stream as np.random.RandomState: key/pos are get_state()[1] / get_state()[2].  No GPU needed. */

void tg_host_mt_seed(tg_mt19937* s, uint32_t seed);                 /* np.random.seed(seed) */
uint32_t tg_host_mt_next32(tg_mt19937* s);
double tg_host_mt_random_sample(tg_mt19937* s);                     /* random_sample(): the draw inside choice(A, p=) (self_play.py:683) */
int32_t tg_host_mt_choice_index(tg_mt19937* s, int32_t k);          /* index drawn by choice(list of k) (self_play.py:709) */
int tg_host_mt_dirichlet(tg_mt19937* s, double alpha, int32_t n, double* out);   /* dirichlet([alpha]*n) (self_play.py:93) */
/* getSubEncode (go_env.h:70, board.cc:1166-1271) with the board size as an argument: crops channels x S x S 32-bit planes into
 * cut_num (<= 5) channels x s x s windows -- the four corners, then the centre.  Host arrays; no GPU needed. */
void tg_host_sub_encode(int board_size, const void* encode, void* sub_encode, int sub_board_size, int channels, int cut_num);


/* ---- 4. the reference engine's own 15 entry points (GoEnv/cpp_src/go_env.h:24-70), same names and signatures -------------
 * One process-wide context (the reference keeps its configuration in file-static variables, go_env.cc:9-12); every call is
 * the batched GPU entry point with n = 1.  `State` is any caller buffer of at least tg_state_size() bytes -- the
 * reference's 1188-byte GoState / Python's 1196-byte c_GoState (environment.py:17-29) qualify, so GoEnv/environment.py binds
 * this library unchanged.  Board size: 9, or 19 with TRANSGO_BOARD_SIZE=19 in the environment (go_comm.h:20 is compile-time
 * in the reference).  Coord = int16_t, Stone = uint8_t (go_comm.h:13-15). */
#ifndef TRANSGO_NO_GOENV_NAMES
bool Init(int history_dim, int encode_dim, int max_step, float komi);        /* go_env.h:30 */
bool Reset(void* state);                                                      /* go_env.h:33 */
bool Step(const void* state, void* next_state, int16_t action);              /* go_env.h:37 */
bool Step_(void* state, int16_t action);                                      /* go_env.h:38 */
bool checkAction(const void* state, int16_t action);                          /* go_env.h:42 */
bool isTerminated(const void* state);                                         /* go_env.h:45 */
bool Encode(const void* state, float* encode_state);                          /* go_env.h:48 */
float getScore(const void* state);                                            /* go_env.h:51 */
float getTerritory(const void* state, float* territory);                      /* go_env.h:54 */
int getLegalAction(const void* state, int* actions);                          /* go_env.h:57 */
int getLegalNoEye(const void* state, int* actions);                           /* go_env.h:60 */
void Show(const void* state);                                                 /* go_env.h:63 */
uint8_t getPlayer(const void* state);                                         /* go_env.h:66 */
int getStep(const void* state);                                               /* go_env.h:69 */
void getSubEncode(int* encode_state, int* sub_encode_state, int sub_board_size, int encode_state_channels, int cut_num);  /* go_env.h:70 */
#endif

#ifdef __cplusplus
}
#endif
#endif /* TRANSGO_HIP_H_ */
