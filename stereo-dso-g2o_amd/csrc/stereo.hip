// placeholder until the stereo kernels land
#include "sdso_internal.h"
#define NI(ctx) return sdso::fail(ctx, SDSO_ERR_STATE, "not implemented")
extern "C" int sdso_immature_init_batch(sdso_ctx* ctx, int, int, const float*, const float*, float*, float*, float*, float*) { NI(ctx); }
extern "C" int sdso_trace_stereo_batch(sdso_ctx* ctx, int, const float*, float, int, sdso_trace_points_t*, uint8_t*) { NI(ctx); }
