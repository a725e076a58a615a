// Context lifetime and error reporting of the C ABI.
#include "common.hpp"

using namespace mistitch;

extern "C" int mistitch_version(void) { return 100; }

extern "C" const char* mistitch_last_error(void) { return last_error_buf(); }

extern "C" int mistitch_create(int device, void* stream, mistitch_ctx** out) {
  MS_REQUIRE(out, MISTITCH_ERR_ARG, "out is NULL");
  *out = nullptr;
  int count = 0;
  MS_HIP(hipGetDeviceCount(&count));
  MS_REQUIRE(device >= 0 && device < count, MISTITCH_ERR_ARG, "device %d not in [0, %d)", device, count);
  MS_HIP(hipSetDevice(device));
  mistitch_ctx* ctx = new mistitch_ctx();
  ctx->device = device;
  ctx->stream = reinterpret_cast<hipStream_t>(stream);
  *out = ctx;
  return MISTITCH_OK;
}

extern "C" int mistitch_destroy(mistitch_ctx* ctx) {
  if (!ctx) return MISTITCH_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& kv : ctx->plans)
    if (kv.second.blob) (void)hipFree(kv.second.blob);
  DevBuf* bufs[] = {&ctx->desc, &ctx->staging, &ctx->spectrum, &ctx->partial, &ctx->peaks,
                    &ctx->cand, &ctx->acc,     &ctx->misc,     &ctx->bins};
  for (DevBuf* b : bufs) b->release();
  delete ctx;
  return MISTITCH_OK;
}

extern "C" int mistitch_set_pair_chunk(mistitch_ctx* ctx, int pairs_per_pass) {
  MS_REQUIRE(ctx, MISTITCH_ERR_ARG, "ctx is NULL");
  MS_REQUIRE(pairs_per_pass >= 1, MISTITCH_ERR_ARG, "pairs_per_pass must be >= 1");
  ctx->pair_chunk = pairs_per_pass;
  return MISTITCH_OK;
}

extern "C" int mistitch_synchronize(mistitch_ctx* ctx) {
  MS_REQUIRE(ctx, MISTITCH_ERR_ARG, "ctx is NULL");
  MS_HIP(hipSetDevice(ctx->device));
  MS_HIP(hipStreamSynchronize(ctx->stream));
  return MISTITCH_OK;
}
