// net.hip -- policy/value network forward (placeholder until the MFMA kernels land)
#include "engine.h"
extern "C" int tg_net_forward(tg_ctx* ctx, int) { ctx->err = "no network weights loaded (tg_net_load)"; return TG_ERR_STATE; }
extern "C" void tg_net_destroy(tg_ctx*) {}
