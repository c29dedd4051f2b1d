// engine.hip -- placeholder until the tree/network engine lands (next milestone)
#include "ctx.h"
extern "C" int tg_engine_create(tg_ctx* ctx) { ctx->err = "engine not built yet"; return TG_ERR_ARG; }
extern "C" void tg_engine_destroy(tg_ctx*) {}
