// compat.cpp -- the 15 extern "C" functions of the reference engine (GoEnv/cpp_src/go_env.h:24-70) with their original
// names and signatures, forwarding to the batched GPU entry points with n = 1.  With this, the reference's own
// GoEnv/environment.py binds libtransgo_hip.so unchanged (symlink it to ./GoEnv/go_env.so): its c_GoState buffer
// (1196 B, environment.py:17-29) is only ever treated as an opaque blob, and ours needs 48 B of it.
// Configuration lives in one process-wide context, as the reference's does in file-static variables (go_env.cc:9-12).
// Board size is a run-time property here: TRANSGO_BOARD_SIZE=19 selects 19x19 (the reference recompiles, go_comm.h:20).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/transgo_hip.h"

namespace {
tg_ctx* g_ctx = nullptr;
tg_config g_cfg;
bool g_cfg_init = false;

void cfg_defaults() {
    if (g_cfg_init) return;
    tg_config_default(&g_cfg);
    g_cfg.max_step = 300;                                  // go_env.cc:11 default before Init()
    const char* s = getenv("TRANSGO_BOARD_SIZE");
    if (s && atoi(s) == 19) g_cfg.board_size = 19;
    const char* d = getenv("TRANSGO_DEVICE");
    if (d) g_cfg.device = atoi(d);
    g_cfg_init = true;
}
tg_ctx* ctx() {
    if (!g_ctx) {
        cfg_defaults();
        if (tg_create(&g_cfg, &g_ctx) != 0) { fprintf(stderr, "libtransgo_hip: %s\n", tg_last_error(nullptr)); abort(); }
    }
    return g_ctx;
}
int P() { cfg_defaults(); return g_cfg.board_size * g_cfg.board_size; }
}  // namespace

extern "C" {

bool Init(int history_dim, int encode_dim, int max_step, float komi) {          // go_env.h:30, go_env.cc:21-32
    if (history_dim > 1) { printf("history_dim is too large\n"); return false; }
    cfg_defaults();
    if (g_ctx) { tg_destroy(g_ctx); g_ctx = nullptr; }
    g_cfg.encode_dim = encode_dim; g_cfg.max_step = max_step; g_cfg.komi = komi;
    return ctx() != nullptr;
}

bool Reset(void* state) { return tg_env_reset(ctx(), state, 1) == 0; }         // go_env.h:33

bool isTerminated(const void* state) {                                           // go_env.h:45
    uint8_t t = 0;
    tg_env_query(ctx(), state, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &t);
    return t != 0;
}

bool Step(const void* state, void* next_state, int16_t action) {                // go_env.h:37, go_env.cc:44-80
    if (isTerminated(state)) {
        if (next_state != state) memcpy(next_state, state, (size_t)tg_state_size(ctx()));
        printf("Fail to Step: game is done!!\n\n");
        return true;
    }
    int32_t a = action;
    uint8_t done = 0, ok = 1;
    tg_env_step(ctx(), state, next_state, &a, 1, &done, &ok);
    if (!ok) printf("Fail to Step: invalid action\n\n");
    return done != 0;
}

bool Step_(void* state, int16_t action) { return Step(state, state, action); }  // go_env.h:38

bool checkAction(const void* state, int16_t action) {                            // go_env.h:42, board.cc:437-464
    if (action == -1 || action == -2) return true;
    if (action < 0 || action >= P()) return false;
    std::vector<uint8_t> legal((size_t)P() + 1);
    tg_env_query(ctx(), state, 1, legal.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    return legal[action] != 0;
}

bool Encode(const void* state, float* encode_state) {                            // go_env.h:48
    return tg_env_query(ctx(), state, 1, nullptr, nullptr, encode_state, nullptr, nullptr, nullptr, nullptr, nullptr) == 0;
}

float getScore(const void* state) {                                              // go_env.h:51
    float s = 0;
    tg_env_query(ctx(), state, 1, nullptr, nullptr, nullptr, &s, nullptr, nullptr, nullptr, nullptr);
    return s;
}

float getTerritory(const void* state, float* territory) {                        // go_env.h:54
    float s = 0;
    tg_env_query(ctx(), state, 1, nullptr, nullptr, nullptr, &s, territory, nullptr, nullptr, nullptr);
    return s;
}

static int list_actions(const void* state, int* actions, bool noeye) {
    std::vector<uint8_t> m((size_t)P() + 1);
    tg_env_query(ctx(), state, 1, noeye ? nullptr : m.data(), noeye ? m.data() : nullptr, nullptr, nullptr, nullptr, nullptr,
                 nullptr, nullptr);
    int n = 0;
    for (int a = 0; a <= P(); ++a) if (m[a]) actions[n++] = a;                   // ascending, pass (= S*S) last
    return n;
}
int getLegalAction(const void* state, int* actions) { return list_actions(state, actions, false); }   // go_env.h:57
int getLegalNoEye(const void* state, int* actions) { return list_actions(state, actions, true); }     // go_env.h:60

uint8_t getPlayer(const void* state) {                                           // go_env.h:66
    int32_t p = 0;
    tg_env_query(ctx(), state, 1, nullptr, nullptr, nullptr, nullptr, nullptr, &p, nullptr, nullptr);
    return (uint8_t)p;
}

int getStep(const void* state) {                                                 // go_env.h:69
    int32_t s = 0;
    tg_env_query(ctx(), state, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &s, nullptr);
    return s;
}

void Show(const void* state) {                                                   // go_env.h:63, go_env.cc:184-200
    tg_env_show(ctx(), state);
    const int p = getPlayer(state);
    printf("step: %d\nnext_player: %s(%s)\ndone: %s\n\n", getStep(state), p == 1 ? "Black" : "White", p == 1 ? "X" : "O",
           isTerminated(state) ? "true" : "false");
}

// go_env.h:70, board.cc:1166-1271: crop C x S x S planes into cut_num (4 or 5) C x s x s windows (corners, then centre).
// Pure 32-bit word movement between two caller-owned host arrays; the reference only reaches it from the dead sub_model
// branch (self_play.py:806).
void getSubEncode(int* encode_state, int* sub_encode_state, int sub_board_size, int encode_state_channels, int cut_num) {
    cfg_defaults();
    tg_host_sub_encode(g_cfg.board_size, encode_state, sub_encode_state, sub_board_size, encode_state_channels, cut_num);
}

// The same with the board size as an argument (the reference's is a compile-time constant): what GoEnv.subEncode
// (environment.py:110-113) calls.
void tg_host_sub_encode(int board_size, const void* encode, void* sub_encode, int sub_board_size, int channels, int cut_num) {
    const int32_t* encode_state = (const int32_t*)encode; int32_t* sub_encode_state = (int32_t*)sub_encode;
    const int S = board_size, s = sub_board_size, C = channels, iv = S - s, ter = iv / 2;
    const int ox[5] = {0, iv, 0, iv, ter}, oy[5] = {0, 0, iv, iv, ter};
    for (int i = 0; i < cut_num && i < 5; ++i)
        for (int c = 0; c < C; ++c)
            for (int y = 0; y < s; ++y)
                for (int x = 0; x < s; ++x)
                    sub_encode_state[((size_t)i * C + c) * s * s + y * s + x] = encode_state[(size_t)c * S * S + (y + oy[i]) * S + x + ox[i]];
}

}  // extern "C"
