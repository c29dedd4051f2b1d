// ctx.h -- host-side context shared by the translation units of libtransgo_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/transgo_hip.h"
#include "board_dev.h"

namespace tg {

struct DevBuf {                       // grow-only device scratch
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 2 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return -1;
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct Engine;                        // tree + network state (engine.hip)

}  // namespace tg

struct tg_ctx {
    tg_config cfg;
    int S = 9, P = 81, A = 82;
    int state_bytes = 48;
    hipStream_t stream = nullptr;
    std::string err;
    tg::RulesCfg rules;
    // env scratch
    tg::DevBuf env_in, env_out, env_act, env_flags, env_u8, env_f32, env_i32;
    tg::Engine* eng = nullptr;
};

#define TG_HIP(ctx, expr)                                                                       \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                     \
            return TG_ERR_HIP;                                                                  \
        }                                                                                       \
    } while (0)

#define TG_FAIL(ctx, code, msg) do { (ctx)->err = (msg); return (code); } while (0)
