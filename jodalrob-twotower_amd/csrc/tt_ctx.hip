// Context, error string, ABI version.
#include "tt_common.h"
#include "tt_gemm.h"
#include "tt_riders.h"

#include <mutex>

#include <string.h>

static thread_local char g_err[512] = "";
std::atomic<uint64_t> tt_launches{0};

namespace {
__global__ __launch_bounds__(kRiderThreads) void riders_kernel(tt_riders r) {
  if ((int)blockIdx.x < r.c_wg) compact_body(r.c, blockIdx.x);
  else finish2_body(r.f);
}
}  // namespace

uint32_t* tt_chain_for(tt_ctx* ctx, hipStream_t stream) {
  static std::mutex mu;
  if (!ctx || !ctx->chained || !ctx->chain) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < ctx->chain_used; ++i)
    if (ctx->chain_stream[i] == static_cast<void*>(stream)) return ctx->chain + (size_t)i * kChainWords;
  if (ctx->chain_used == kChainSlices) return nullptr;
  ctx->chain_stream[ctx->chain_used] = static_cast<void*>(stream);
  return ctx->chain + (size_t)(ctx->chain_used++) * kChainWords;
}

int tt_riders_flush(tt_ctx* ctx, hipStream_t st) {
  if (!ctx || !ctx->riders || (ctx->riders->c_wg == 0 && ctx->riders->f_wg == 0)) return TT_OK;
  riders_kernel<<<ctx->riders->c_wg + ctx->riders->f_wg, kRiderThreads, 0, st>>>(*ctx->riders);
  ctx->riders->c_wg = ctx->riders->f_wg = 0;
  TT_LAUNCH_CHECK();
  return TT_OK;
}

void tt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

int tt_abi_version(void) { return TT_ABI_VERSION; }

const char* tt_last_error_string(void) { return g_err; }

int tt_ctx_create(int device, tt_ctx** out) {
  TT_CHECK_ARG(out != nullptr, "tt_ctx_create: out is NULL");
  int n = 0;
  TT_HIP(hipGetDeviceCount(&n));
  TT_CHECK_ARG(device >= 0 && device < n, "tt_ctx_create: device %d out of range (%d devices)", device, n);
  hipDeviceProp_t prop;
  TT_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    tt_set_error("tt_ctx_create: device %d is %s; this library is built for gfx950 (MI355X) only", device,
                 prop.gcnArchName);
    return TT_ERR_UNSUPPORTED;
  }
  tt_ctx* c = new tt_ctx();
  c->device = device;
  c->num_cus = prop.multiProcessorCount;
  c->lds_per_block = prop.sharedMemPerBlock;
  c->lookup_stamps = nullptr;
  c->lookup_stamp_slots = 0;
  c->defer_slab_reduce = 0;
  c->deferred = nullptr;
  c->keyed_parts = 0;
  c->score_bwd_rows_min = 32768;
  c->defer_riders = 0;
  c->fp8_grad = 1;
  c->riders = new tt_riders();
  c->riders->c_wg = c->riders->f_wg = 0;
  c->chain = nullptr;
  c->chain_words = 0;
  c->chain_used = 0;
  c->chained = 1;
  c->chain_spin = 1 << 22;
  c->dev_err = nullptr;
  c->ho_exec = c->ho_node = c->ho_last = nullptr;
  {
    int prev = 0;
    TT_HIP(hipGetDevice(&prev));
    TT_HIP(hipSetDevice(device));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->chain), sizeof(uint32_t) * (kChainWords * kChainSlices + 64));
    if (e == hipSuccess) e = hipMemset(c->chain, 0, sizeof(uint32_t) * (kChainWords * kChainSlices + 64));
    if (e == hipSuccess) c->dev_err = c->chain + (size_t)kChainWords * kChainSlices;       // (its own 256-byte line behind the pool)
    (void)hipSetDevice(prev);
    if (e != hipSuccess) {
      tt_set_error("tt_ctx_create: %s", hipGetErrorString(e));
      delete c->riders;
      delete c;
      return TT_ERR_HIP;
    }
    c->chain_words = kChainWords;
  }
  *out = c;
  return TT_OK;
}

int tt_ctx_destroy(tt_ctx* ctx) {
  if (ctx && ctx->deferred) tt_gemm_tn_pending_destroy(ctx->deferred);
  if (ctx) delete ctx->riders;
  if (ctx && ctx->chain) (void)hipFree(ctx->chain);
  delete ctx;
  return TT_OK;
}

int tt_ctx_set_option(tt_ctx* ctx, int32_t option, int32_t value) {
  TT_CHECK_ARG(ctx != nullptr, "tt_ctx_set_option: NULL context");
  switch (option) {
    case TT_OPT_DEFER_SLAB_REDUCE: ctx->defer_slab_reduce = value != 0; break;
    case TT_OPT_KEYED_PARTS:
      TT_CHECK_ARG(value >= 0, "tt_ctx_set_option: TT_OPT_KEYED_PARTS needs a value >= 0");
      ctx->keyed_parts = value;
      break;
    case TT_OPT_SCORE_BWD_ROWS_MIN:
      TT_CHECK_ARG(value >= 1, "tt_ctx_set_option: TT_OPT_SCORE_BWD_ROWS_MIN needs a value >= 1");
      ctx->score_bwd_rows_min = value;
      break;
    case TT_OPT_DEFER_RIDERS:
      TT_CHECK_ARG(value >= 0 && value <= 3, "tt_ctx_set_option: TT_OPT_DEFER_RIDERS takes 0 .. 3");
      ctx->defer_riders = value == 1 ? 3 : value;      // 1 = both riders (as 3), 2 = the loss reduction only
      break;
    case TT_OPT_FP8_GRAD: ctx->fp8_grad = value != 0; break;
    case TT_OPT_CHAINED: ctx->chained = value != 0; break;
    case TT_OPT_LOOKUP_NT: ctx->lookup_nt = value != 0; break;
    case TT_OPT_CHAIN_SPIN:
      TT_CHECK_ARG(value >= 1, "tt_ctx_set_option: TT_OPT_CHAIN_SPIN needs a value >= 1");
      ctx->chain_spin = value;
      break;
    default: tt_set_error("tt_ctx_set_option: unknown option %d", option); return TT_ERR_INVALID_ARG;
  }
  return TT_OK;
}

int tt_ctx_check_device_errors(tt_ctx* ctx, tt_stream stream) {
  TT_CHECK_ARG(ctx != nullptr, "tt_ctx_check_device_errors: NULL context");
  if (!ctx->dev_err) return TT_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  uint32_t word = 0;
  TT_HIP(hipMemcpyAsync(&word, ctx->dev_err, sizeof(word), hipMemcpyDeviceToHost, st));
  TT_HIP(hipStreamSynchronize(st));
  if (word == 0) return TT_OK;
  // clear the word and every chain slice: a launch that gave up half way leaves ready bits and a ticket behind, which the next
  // chained launch on that stream would read as prefix sums
  TT_HIP(hipMemsetAsync(ctx->chain, 0, sizeof(uint32_t) * ((size_t)kChainWords * kChainSlices + 64), st));
  TT_HIP(hipStreamSynchronize(st));
  tt_set_error("device error word 0x%x:%s%s the steps since the last check are invalid (they must be rejected)", word,
               (word & TT_DEVERR_CHAIN_TIMEOUT) ? " a chained segment-head launch gave up waiting for a predecessor tile (tt_dedup_plan / tt_dedup_plan_runs);" : "",
               (word & TT_DEVERR_ROW_RANGE) ? " a lookup decoded rows outside its table (key offsets / vocabularies that belong to another table; tt_embed_lookup_fwd / "
                                              "tt_embed_lookup_rows_fwd / tt_batch_ingest_lookup) and read the last row instead;" : "");
  return TT_ERR_DEVICE;
}

int tt_handover_retarget(tt_ctx* ctx, void* graph_exec, void* node) {
  TT_CHECK_ARG(ctx != nullptr && ((graph_exec == nullptr) == (node == nullptr)), "tt_handover_retarget: NULL context / one of (graph_exec, node) NULL");
  ctx->ho_exec = graph_exec;
  ctx->ho_node = node;
  return TT_OK;
}

int tt_handover_captured_node(tt_ctx* ctx, void** node) {
  TT_CHECK_ARG(ctx != nullptr && node != nullptr, "tt_handover_captured_node: NULL argument");
  *node = ctx->ho_last;
  ctx->ho_last = nullptr;
  return TT_OK;
}

int tt_flush_deferred(tt_ctx* ctx, tt_stream stream) {
  TT_CHECK_ARG(ctx != nullptr, "tt_flush_deferred: NULL context");
  if (int rc = tt_riders_flush(ctx, reinterpret_cast<hipStream_t>(stream))) return rc;
  return tt_gemm_deferred_flush(ctx, reinterpret_cast<hipStream_t>(stream));
}

int tt_flush_deferred_slabs(tt_ctx* ctx, tt_stream stream) {
  TT_CHECK_ARG(ctx != nullptr, "tt_flush_deferred_slabs: NULL context");
  return tt_gemm_deferred_flush(ctx, reinterpret_cast<hipStream_t>(stream));
}

int tt_deferred_pending(const tt_ctx* ctx) {
  if (!ctx) return 0;
  return ((ctx->deferred && ctx->deferred->n > 0) ? 1 : 0) | ((ctx->riders && (ctx->riders->c_wg > 0 || ctx->riders->f_wg > 0)) ? 2 : 0);
}

int tt_ctx_num_cus(const tt_ctx* ctx) { return ctx ? ctx->num_cus : 0; }

uint64_t tt_launch_count(void) { return tt_launches.load(std::memory_order_relaxed); }

}  // extern "C"
