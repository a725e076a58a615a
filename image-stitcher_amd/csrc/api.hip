// Context lifetime and error reporting of the C ABI.
#include "common.hpp"

using namespace mistitch;

namespace mistitch {
int fuse_static_setup(int device);  // fusion.hip: address of this device's static workspace
}

extern "C" int mistitch_version(void) { return 101; }

extern "C" const char* mistitch_last_error(void) { return last_error_buf(); }

extern "C" int mistitch_create(int device, void* stream, mistitch_ctx** out) {
  MS_REQUIRE(out, MISTITCH_ERR_ARG, "out is NULL");
  *out = nullptr;
  int count = 0;
  MS_HIP(hipGetDeviceCount(&count));
  MS_REQUIRE(device >= 0 && device < count, MISTITCH_ERR_ARG, "device %d not in [0, %d)", device, count);
  MS_HIP(hipSetDevice(device));
  int rc = fuse_static_setup(device);
  if (rc != MISTITCH_OK) return rc;
  mistitch_ctx* ctx = new mistitch_ctx();
  ctx->device = device;
  ctx->stream = reinterpret_cast<hipStream_t>(stream);
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->n_cus = cus;
  *out = ctx;
  return MISTITCH_OK;
}

extern "C" int mistitch_destroy(mistitch_ctx* ctx) {
  if (!ctx) return MISTITCH_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& kv : ctx->plans)
    if (kv.second.blob) (void)hipFree(kv.second.blob);
  DevBuf* bufs[] = {&ctx->desc,  &ctx->staging, &ctx->spectrum, &ctx->slabs, &ctx->partial,
                    &ctx->peaks, &ctx->cand,    &ctx->acc,      &ctx->misc,  &ctx->bins,    &ctx->flags,   &ctx->ups};
  for (DevBuf* b : bufs) b->release();
  for (void* q : ctx->shape_allocs) (void)hipFree(q);
  ctx->prof.destroy();
  delete ctx;
  return MISTITCH_OK;
}

extern "C" int mistitch_set_pair_chunk(mistitch_ctx* ctx, int pairs_per_pass) {
  MS_REQUIRE(ctx, MISTITCH_ERR_ARG, "ctx is NULL");
  MS_REQUIRE(pairs_per_pass >= 1, MISTITCH_ERR_ARG, "pairs_per_pass must be >= 1");
  ctx->pair_chunk = pairs_per_pass;
  return MISTITCH_OK;
}

extern "C" int mistitch_set_staging_limit(mistitch_ctx* ctx, int64_t bytes) {
  MS_REQUIRE(ctx, MISTITCH_ERR_ARG, "ctx is NULL");
  MS_REQUIRE(bytes >= 0, MISTITCH_ERR_ARG, "staging limit must be >= 0");
  ctx->staging_limit = bytes;
  return MISTITCH_OK;
}

extern "C" int mistitch_set_pixel_max(mistitch_ctx* ctx, int pixel_max) {
  MS_REQUIRE(ctx, MISTITCH_ERR_ARG, "ctx is NULL");
  MS_REQUIRE(pixel_max >= 1 && pixel_max <= 65535, MISTITCH_ERR_ARG, "pixel_max must be within 1 .. 65535");
  ctx->pixel_max = pixel_max;
  return MISTITCH_OK;
}

extern "C" int mistitch_synchronize(mistitch_ctx* ctx) {
  MS_REQUIRE(ctx, MISTITCH_ERR_ARG, "ctx is NULL");
  MS_HIP(hipSetDevice(ctx->device));
  MS_HIP(hipStreamSynchronize(ctx->stream));
  return MISTITCH_OK;
}

extern "C" int mistitch_profile_enable(mistitch_ctx* ctx, int on) {
  MS_REQUIRE(ctx, MISTITCH_ERR_ARG, "ctx is NULL");
  MS_HIP(hipSetDevice(ctx->device));
  MS_HIP(hipStreamSynchronize(ctx->stream));
  ctx->prof.release();
  ctx->prof.on = on != 0;
  return MISTITCH_OK;
}

extern "C" int mistitch_profile_collect(mistitch_ctx* ctx, char* names, size_t names_len, double* total_ms,
                                        int* launches, int max_entries, int* n_entries) {
  MS_REQUIRE(ctx && names && total_ms && launches && n_entries, MISTITCH_ERR_ARG, "NULL argument");
  MS_HIP(hipSetDevice(ctx->device));
  MS_HIP(hipStreamSynchronize(ctx->stream));
  std::vector<std::string> order;
  std::map<std::string, std::pair<double, int>> agg;
  for (auto& r : ctx->prof.recs) {
    float ms = 0.f;
    MS_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
    auto it = agg.find(r.name);
    if (it == agg.end()) {
      order.push_back(r.name);
      agg[r.name] = {ms, 1};
    } else {
      it->second.first += ms;
      it->second.second += 1;
    }
  }
  ctx->prof.release();
  MS_REQUIRE((int)order.size() <= max_entries, MISTITCH_ERR_ARG, "%zu kernels do not fit %d entries", order.size(),
             max_entries);
  std::string joined;
  for (size_t i = 0; i < order.size(); ++i) {
    if (i) joined += "\n";
    joined += order[i];
    total_ms[i] = agg[order[i]].first;
    launches[i] = agg[order[i]].second;
  }
  MS_REQUIRE(joined.size() + 1 <= names_len, MISTITCH_ERR_ARG, "names buffer too small");
  memcpy(names, joined.c_str(), joined.size() + 1);
  *n_entries = (int)order.size();
  return MISTITCH_OK;
}
