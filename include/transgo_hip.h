/*
 * transgo_hip.h -- C ABI of libtransgo_hip.so, the MI355X-native drop-in for Transgo's self-play hot path.
 *
 * Plain C, plain pointers and sizes, no torch types.  Every entry point names the reference interface it replaces
 * (file:line under /root/reference).  Unless a parameter says "device", pointers are HOST memory owned by the caller;
 * the library never keeps a caller pointer after a call returns.  All functions return 0 on success or a negative
 * tg_status; tg_last_error() gives the message.  A context is single-owner (one host thread, one GPU).
 *
 * Three groups:
 *   1. tg_env_*   batched form of the 15 extern "C" functions of GoEnv/cpp_src/go_env.h:24-70 over opaque state blobs
 *                 (what GoEnv/environment.py:42-90 binds through ctypes today).
 *   2. tg_sp_*    the batched self-play engine: G concurrent games, WP_MCTS (self_play.py:575-875) as HIP tree
 *                 kernels + the policy/value network forward (model.py:79-114) on MFMA.
 *   3. tg_host_*  host-only helpers that must be bit-identical to NumPy's legacy MT19937 stream
 *                 (np.random.dirichlet / choice used at self_play.py:93, :709, :683).
 */
#ifndef TRANSGO_HIP_H_
#define TRANSGO_HIP_H_

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
    TG_ERR_ARENA = -4,      /* a game's tree arena overflowed (results of that game are invalid) */
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
    int32_t arena_slots;        /* 32-byte tree slots per game per half arena; 0 = sized from num_simulation */
    int32_t net_blocks;         /* residual blocks of the tower (BASELINE.json "N-block x F-filter") */
    int32_t net_filters;        /* channels F (multiple of 32) */
    int32_t device;             /* HIP device ordinal */
    int32_t reserved[8];
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

#ifdef __cplusplus
}
#endif
#endif /* TRANSGO_HIP_H_ */
