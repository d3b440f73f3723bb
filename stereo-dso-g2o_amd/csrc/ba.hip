// placeholder until the BA kernels land
#include "sdso_internal.h"
namespace sdso { struct BaWindowDev {}; void release_all_windows(sdso_ctx*) {} }
#define NI(ctx) return sdso::fail(ctx, SDSO_ERR_STATE, "not implemented")
extern "C" int sdso_ba_upload_window(sdso_ctx* ctx, int, const sdso_ba_window_t*) { NI(ctx); }
extern "C" int sdso_ba_release_window(sdso_ctx* ctx, int) { NI(ctx); }
extern "C" int sdso_ba_linearize(sdso_ctx* ctx, int, double*) { NI(ctx); }
extern "C" int sdso_ba_get_linearization(sdso_ctx* ctx, int, float*, uint8_t*, float*, float*, float*, float*) { NI(ctx); }
extern "C" int sdso_ba_apply_res(sdso_ctx* ctx, int) { NI(ctx); }
extern "C" int sdso_ba_get_residual_state(sdso_ctx* ctx, int, uint8_t*, uint8_t*, float*) { NI(ctx); }
extern "C" int sdso_ba_accumulate(sdso_ctx* ctx, int) { NI(ctx); }
extern "C" int sdso_ba_accum_floats(int nf) { return nf * nf * 91 * 2 + nf * nf * nf * 64 + nf * nf * 32 + nf * nf * 8 + 16 + 4 + 2; }
extern "C" int sdso_ba_accum_dev(sdso_ctx* ctx, int, void**) { NI(ctx); }
extern "C" int sdso_ba_get_accumulators(sdso_ctx* ctx, int, float*) { NI(ctx); }
extern "C" int sdso_ba_get_point_terms(sdso_ctx* ctx, int, float*, float*, float*, float*, float*) { NI(ctx); }
extern "C" int sdso_ba_solve(sdso_ctx* ctx, int, int, double, double*, double*, double*, double*, double*) { NI(ctx); }
extern "C" int sdso_ba_get_point_steps(sdso_ctx* ctx, int, float*) { NI(ctx); }
extern "C" int sdso_ba_optimize(sdso_ctx* ctx, int, int, double*, float*, uint8_t*, sdso_ba_opt_result_t*) { NI(ctx); }
extern "C" int sdso_ba_marginalize_points(sdso_ctx* ctx, int, const uint8_t*, double*, double*) { NI(ctx); }
extern "C" int sdso_ba_get_tables(sdso_ctx* ctx, int, float*, double*, double*, float*) { NI(ctx); }
