// Shared internals of libtwotower_hip.so (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>

#include "twotower.h"

struct tt_ctx {
  int device;
  int num_cus;
  size_t lds_per_block;
  unsigned long long* lookup_stamps;   // optional device ring (tt_embed_lookup_set_profile)
  int lookup_stamp_slots;
  // TT_OPT_DEFER_SLAB_REDUCE: tt_towers_mlp_bwd leaves the split-K slab reduction of its weight gradients queued here; the
  // next tt_embed_grad_bwd on a planned workspace runs it inside its own launch, tt_flush_deferred / the Adam entries otherwise
  int defer_slab_reduce;
  struct TnPending* deferred;
  int keyed_parts;          // TT_OPT_KEYED_PARTS: workgroups per key of the keyed dedup plan (0 = chosen from the batch)
  int score_bwd_rows_min;   // TT_OPT_SCORE_BWD_ROWS_MIN: rows from which tt_score_bwd_bf16 takes the workgroup-staged form
  int defer_riders;         // TT_OPT_DEFER_RIDERS, as a mask: 1 plan compaction, 2 score loss reduction queue in `riders` (tt_riders.h)
  int fp8_grad;             // TT_OPT_FP8_GRAD: tt_score_bwd_fp8 forms the gradient products from e4m3 operands too (default 1)
  struct tt_riders* riders;
  // chained single-launch scans (segment heads, owner routing): small device buffers that are all-zero between launches -- word 0
  // a ticket, the rest per-workgroup aggregates with a ready bit; the last workgroup through clears what it used.  One slice
  // per stream that calls in (launches of different streams may overlap: the test rigs that run several virtual ranks on one
  // GPU do), handed out from a pool allocated with the context -- nothing is allocated at call time (stream capture)
  uint32_t* chain;
  int chain_words;          // per slice
  void* chain_stream[16];
  int chain_used;
  int chained;              // TT_OPT_CHAINED: use them (default 1); 0 = the multi-launch forms (same results; tests compare)
  uint32_t* dev_err;        // sticky device-side error word (TT_DEVERR_*): behind the chain pool; tt_ctx_check_device_errors
  int chain_spin;           // TT_OPT_CHAIN_SPIN: polls before a chained tile gives up (default 2^22)
  int lookup_nt;            // TT_OPT_LOOKUP_NT: the fused hand-over + lookup launch stores its bf16 rows non-temporally (default 0)
  // hand-over launches as nodes of a captured graph (tt_handover_retarget): while ho_exec is set, tt_batch_ingest* re-point node
  // ho_node of that executable graph at their arguments instead of launching; ho_last = the node the last tt_batch_ingest* call left
  // in the capture its stream was in
  void* ho_exec;
  void* ho_node;
  void* ho_last;
};

constexpr uint32_t kChainReady = 0x80000000u;
constexpr int kChainWords = 1 << 16, kChainSlices = 16;
// the calling stream's slice, or nullptr (pool exhausted / chaining off): the caller then takes its multi-launch form
uint32_t* tt_chain_for(tt_ctx* ctx, hipStream_t stream);

void tt_set_error(const char* fmt, ...);

#define TT_CHECK_ARG(cond, ...)         \
  do {                                  \
    if (!(cond)) {                      \
      tt_set_error(__VA_ARGS__);        \
      return TT_ERR_INVALID_ARG;        \
    }                                   \
  } while (0)

#define TT_HIP(expr)                                                                   \
  do {                                                                                 \
    hipError_t _e = (expr);                                                            \
    if (_e != hipSuccess) {                                                            \
      tt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return TT_ERR_HIP;                                                               \
    }                                                                                  \
  } while (0)

// (every kernel launch of the library is followed by this check: it also counts them -- tt_launch_count(), a measurement aid)
extern std::atomic<uint64_t> tt_launches;
#define TT_LAUNCH_CHECK()                                    \
  do {                                                       \
    tt_launches.fetch_add(1, std::memory_order_relaxed);     \
    TT_HIP(hipGetLastError());                               \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize of one kernel instantiation, set once per device (a static per expansion site; the
// kernel goes last because its template arguments carry commas).  Needs `ctx` in scope.
#define TT_LDS_ONCE(bytes, ...)                                                                                     \
  do {                                                                                                              \
    static std::atomic<uint64_t> tt_lds_done{0};                                                                    \
    const uint64_t tt_bit = 1ull << (ctx->device & 63);                                                             \
    if (!(tt_lds_done.load(std::memory_order_acquire) & tt_bit)) {                                                  \
      TT_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
      tt_lds_done.fetch_or(tt_bit, std::memory_order_release);                                                      \
    }                                                                                                               \
  } while (0)

static inline int64_t tt_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline bool tt_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// float -> bf16 round-to-nearest-even; NaN stays NaN (a plain cast lowers to v_cvt_pk_bf16_f32).
__device__ __forceinline__ uint16_t tt_f2bf(float x) {
  __bf16 b = static_cast<__bf16>(x);
  return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float tt_bf2f(uint16_t h) {
  return __builtin_bit_cast(float, static_cast<uint32_t>(h) << 16);
}

// counter-based uniform in [0,1): one value per (seed, element index) -- dropout masks are
// regenerated in the backward pass instead of being stored.  32-bit arithmetic only: a multiply-xorshift
// finaliser (the "lowbias32" constants) over the low index word xor the seed, the high words folded in by one multiply.  The 64-bit splitmix this
// replaces cost ~30 quarter-rate v_mul_*_u32 per element -- 3 us of each fused tail kernel at 4 elements per thread.
__device__ __forceinline__ uint32_t tt_mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float tt_uniform01(uint64_t seed, uint64_t idx) {
  const uint32_t lo = static_cast<uint32_t>(idx), hi = static_cast<uint32_t>(idx >> 32);
  const uint32_t h = tt_mix32(lo ^ static_cast<uint32_t>(seed) ^ ((hi ^ static_cast<uint32_t>(seed >> 32)) * 0x9E3779B9U));
  return static_cast<float>(h >> 8) * (1.0f / 16777216.0f);
}
