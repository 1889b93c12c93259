// Internal GEMM launchers (f32 in / f32 accumulate on v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered
// fmaf chain, so results track the reference's fp32 CPU arithmetic to rounding).
// Every launcher takes up to TT_MAX_SIDES independent problems (the towers) and covers them with ONE launch.
#pragma once
#include "tt_common.h"

// C[M,N] = act(alpha * A[M,K] . W[N,K]^T + bias)  (nn.Linear forward; alpha = 1/T for score matrices)
struct GemmNT {
  const float* A; int64_t lda; const float* W; int64_t ldw; const float* bias; float* C; int64_t ldc;
  int64_t M, N, K; bool relu; float alpha; bool bf16 = false;   // bf16: operands rounded to bf16, f32 accumulate
  void* workspace = nullptr; size_t workspace_bytes = 0;         // optional: enables split-K (tt_gemm_nt_workspace_bytes)
  // bf16 path only: A / C actually point at bf16 elements (lda / ldc in elements); the rounding the MFMA operand
  // needs anyway then happens where the tensor is produced, and the tensor costs half the HBM bytes
  bool a_bf16 = false, c_bf16 = false;
  // optional (all problems of a launch or none): the caller finishes a split-K launch itself (bias included) from the
  // slabs described here instead of this launcher's slab-reduction pass; splits == 0 means C was written directly
  struct NtDeferred* defer = nullptr;
};
struct NtDeferred { const float* slabs; int64_t slab_stride; int splits; };
size_t tt_gemm_nt_workspace_bytes(int64_t M, int64_t N, int64_t K);
int tt_gemm_nt_batched(hipStream_t st, const GemmNT* items, int n);
int tt_gemm_nt(hipStream_t st, const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C,
               int64_t ldc, int64_t M, int64_t N, int64_t K, bool relu, float alpha = 1.f);

// C[M,N] = A[M,K] . W[K,N]                        (data gradient: dX = dY . W)
struct GemmNN {
  const float* A; int64_t lda; const float* W; int64_t ldw; float* C; int64_t ldc; int64_t M, N, K; bool bf16 = false;
  bool c_bf16 = false;                                           // bf16 path only: C holds bf16 elements
};
int tt_gemm_nn_batched(hipStream_t st, const GemmNN* items, int n);

// C[M,N] = A[R,M]^T . B[R,N]  reduced over the R (batch) rows in deterministic split-K slabs
// (weight gradient: dW = dY^T . X).  colsum_out (optional, [M]) = sum_r A[r][m] (the bias gradient) from the same pass.
size_t tt_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t R);
struct GemmTN {
  const float* A; int64_t lda; const float* B; int64_t ldb; float* C; int64_t ldc; int64_t M, N, R;
  void* workspace; size_t workspace_bytes; float* colsum_out; bool bf16 = false;
  bool a_bf16 = false, b_bf16 = false;                           // bf16 path only: A / B hold bf16 elements
};
// `pending` (optional): the split-K slab reductions are queued there instead of being launched, and
// tt_gemm_tn_flush() covers everything queued with ONE launch (each small launch in a dependent chain costs ~5 us
// on this part).  Every queued problem needs its own workspace until the flush.
struct TnPending;
TnPending* tt_gemm_tn_pending_create();
void tt_gemm_tn_pending_destroy(TnPending*);
int tt_gemm_tn_batched(hipStream_t st, const GemmTN* items, int n, TnPending* pending = nullptr);
int tt_gemm_tn_flush(hipStream_t st, TnPending* pending);
