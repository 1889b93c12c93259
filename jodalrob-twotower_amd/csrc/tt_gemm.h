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
struct SlabArgs {
  const float* slabs; int64_t slab_stride; int splits; float* C; int64_t ldc; int M, N;
  const float* colsum_slab; float* colsum_out;
  const float* bias; int relu;          // split-K forward GEMMs finish bias / ReLU here
  int c_bf16 = 0;                        // C holds bf16 elements
  // projection bias item (gemm_back): no slabs of its own; colsum_out [proj_h0] = proj_w[:, 0:proj_h0]^T . (sum of the
  // proj_colsum_splits slabs colsum_slab [.][M]), M = H <= 256
  const float* proj_w = nullptr; int proj_ldw = 0, proj_h0 = 0, proj_colsum_splits = 0;
};
constexpr int kSlabItems = 16;
struct SlabBatch { SlabArgs a[kSlabItems]; };
struct TnPending {
  SlabBatch sb;
  int n = 0;
  int64_t maxtotal = 1;
};
inline int tt_slab_blocks_x(const TnPending* p) {
  int64_t b = (p->maxtotal + 255) / 256;
  return (int)(b > 1024 ? 1024 : b);
}
// the slab reduction as a ROLE of another launch (tt_embed_grad_bwd's): a flat run of workgroups, `nbx` per ordinary item and ONE
// per projection-bias item (a single workgroup's work).  One element per thread, as the stand-alone launch: a thread that owns four
// lives 11 us (its 4 x 28 loads go out eight at a time), and the registers that would hold them all in flight (122 - 218) cost the row
// role its occupancy (seg_reduce_chunk_slab anatomy, profiles/NOTES.md)
#ifndef TT_SLAB_ROLE_PER_THREAD
#define TT_SLAB_ROLE_PER_THREAD 1
#endif
inline int tt_slab_role_blocks_x(const TnPending* p) {
  int64_t b = (p->maxtotal + 256 * TT_SLAB_ROLE_PER_THREAD - 1) / (256 * TT_SLAB_ROLE_PER_THREAD);
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}
inline int tt_slab_role_blocks(const TnPending* p, int nbx) {
  int t = 0;
  for (int i = 0; i < p->n; ++i) t += p->sb.a[i].proj_w ? 1 : nbx;
  return t;
}

constexpr int kProjMaxH = 256;
#ifdef __HIPCC__
// colsum_out [proj_h0] = proj_w[:, 0:proj_h0]^T . (sum of the proj_colsum_splits slabs colsum_slab [.][M]): one workgroup.
// Every sum keeps its order (splits ascending, then h ascending); what changed in round 4 is how the loads are issued: 32 independent
// loads per thread, THEN the 32 dependent adds -- the loop that alternated them made this one workgroup the longest of the launch it
// rides in (10.4 us for a 64 x 128 matrix-vector product: profiles/NOTES.md, seg_reduce_chunk_slab anatomy).
__device__ __forceinline__ void proj_bias_finish(const SlabArgs& a, int bx) {
  if (bx != 0) return;
  __shared__ float dbs[kProjMaxH];
  const int H = a.M, t = threadIdx.x, nt = blockDim.x;
  constexpr int kAhead = 16;
  for (int h = t; h < H; h += nt) {
    float s = 0.f;
    for (int z0 = 0; z0 < a.proj_colsum_splits; z0 += kAhead) {
      float v[kAhead];
#pragma unroll
      for (int k = 0; k < kAhead; ++k) v[k] = z0 + k < a.proj_colsum_splits ? a.colsum_slab[(int64_t)(z0 + k) * H + h] : 0.f;
#pragma unroll
      for (int k = 0; k < kAhead; ++k)
        if (z0 + k < a.proj_colsum_splits) s += v[k];
    }
    dbs[h] = s;
  }
  __syncthreads();
  for (int i = t; i < a.proj_h0; i += nt) {
    float acc = 0.f;
    for (int h0 = 0; h0 < H; h0 += kAhead) {
      float w[kAhead];
#pragma unroll
      for (int k = 0; k < kAhead; ++k) w[k] = h0 + k < H ? a.proj_w[(int64_t)(h0 + k) * a.proj_ldw + i] : 0.f;
#pragma unroll
      for (int k = 0; k < kAhead; ++k)
        if (h0 + k < H) acc = fmaf(w[k], dbs[h0 + k], acc);
    }
    a.colsum_out[i] = acc;
  }
}

// workgroup `blk` of the slab role (tt_slab_role_blocks) -> (item, block of the item); false past the end
__device__ __forceinline__ bool slab_role_locate(const SlabBatch& batch, int n_items, int nbx, int blk, int& item, int& bx) {
  for (item = 0; item < n_items; ++item) {
    const int nb = batch.a[item].proj_w ? 1 : nbx;
    if (blk < nb) { bx = blk; return true; }
    blk -= nb;
  }
  return false;
}

// workgroup (bx of nbx, item by) of the slab reduction: the body of slab_reduce_kernel, also run by the first workgroups of
// tt_embed_grad_bwd's launch when the reduction was deferred (TT_OPT_DEFER_SLAB_REDUCE)
__device__ __forceinline__ void slab_reduce_block(const SlabBatch& batch, int bx, int nbx, int by) {
  const SlabArgs& a = batch.a[by];
  if (a.proj_w) {
    proj_bias_finish(a, bx);
    return;
  }
  const int64_t total = (int64_t)a.M * a.N;
  const int64_t all = total + (a.colsum_out ? a.M : 0);
  const int64_t stride = (int64_t)nbx * blockDim.x;
  for (int64_t i = (int64_t)bx * blockDim.x + threadIdx.x; i < all; i += stride) {
    float s = 0.f;
    if (i < total) {
#pragma unroll 8
      for (int z = 0; z < a.splits; ++z) s += a.slabs[(int64_t)z * a.slab_stride + i];
      const int64_t m = i / a.N, n = i - m * a.N;
      if (a.bias) s += a.bias[n];
      if (a.relu) s = fmaxf(s, 0.f);
      if (a.c_bf16) reinterpret_cast<uint16_t*>(a.C)[m * a.ldc + n] = tt_f2bf(s);
      else a.C[m * a.ldc + n] = s;
    } else {
      const int64_t m = i - total;
#pragma unroll 8
      for (int z = 0; z < a.splits; ++z) s += a.colsum_slab[(int64_t)z * a.M + m];
      a.colsum_out[m] = s;
    }
  }
}
#endif

TnPending* tt_gemm_tn_pending_create();
void tt_gemm_tn_pending_destroy(TnPending*);
int tt_gemm_tn_batched(hipStream_t st, const GemmTN* items, int n, TnPending* pending = nullptr);
int tt_gemm_tn_flush(hipStream_t st, TnPending* pending);
// move the queue into the context instead of flushing it (tt_ctx.defer_slab_reduce); launch whatever the context holds
int tt_gemm_tn_defer(tt_ctx* ctx, TnPending* pending);
int tt_gemm_deferred_flush(tt_ctx* ctx, hipStream_t st);

// First-block backward of up to TT_MAX_SIDES towers in ONE launch (bf16 operands, edge-free shapes): per tower
//   dW = d_pre^T . x (+ db = column sums of d_pre),  d_x[:, h0:] = d_pre . W[:, h0:],  G = d_pre^T . dense,
// and dW_proj = W[:, 0:h0]^T . G (each batch split's share multiplied in its own workgroup, summed by the pending
// slab-reduction launch), db_proj = W[:, 0:h0]^T . db  (= d_proj^T . dense and the column sums of d_proj = d_pre . W[:, 0:h0],
// without materialising d_proj: d_x[:, 0:h0] is NOT written).  H a multiple of 64, at most 256.  tt_gemm_back_supported() says whether the shapes qualify; otherwise use the TN / NN launchers.
struct GemmBack {
  const float* dpre; int H;                       // [B, H] f32, ld = H
  const void* x; int64_t ldx; int kx; bool x_bf16;
  const float* dense; int64_t ld_dense; int din;
  const float* w; int h0;                         // block weight [H, kx] f32, ld = kx
  float* dx; int64_t ld_dx; bool dx_bf16;
  float* dw; float* db; float* dwp; float* dbp;   // [H, kx], [H], [h0, din], [h0]
  void* ws_dw; size_t ws_dw_bytes;                // >= tt_gemm_tn_workspace_bytes(H, kx, B)
  void* ws_g; size_t ws_g_bytes;                  // >= tt_gemm_back_g_workspace_bytes(H, h0, din, B)
  int64_t B;
  const void* w16 = nullptr;                      // optional bf16 shadow of w (same shape): read by the row-gradient product
};
bool tt_gemm_back_supported(const GemmBack* items, int n);
size_t tt_gemm_back_g_workspace_bytes(int64_t H, int64_t h0, int64_t din, int64_t B);
int tt_gemm_back_batched(hipStream_t st, const GemmBack* items, int n, TnPending* pending);
