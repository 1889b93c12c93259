// Internal GEMM launchers (f32 in / f32 accumulate on v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered
// fmaf chain, so results track the reference's fp32 CPU arithmetic to rounding).
#pragma once
#include "tt_common.h"

// C[M,N] = act(alpha * A[M,K] . W[N,K]^T + bias)  (nn.Linear forward; alpha = 1/T for score matrices)
int tt_gemm_nt(hipStream_t st, const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C,
               int64_t ldc, int64_t M, int64_t N, int64_t K, bool relu, float alpha = 1.f);
// C[M,N] = A[M,K] . W[K,N]                        (data gradient: dX = dY . W)
int tt_gemm_nn(hipStream_t st, const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int64_t M,
               int64_t N, int64_t K);
// C[M,N] = A[R,M]^T . B[R,N]  reduced over the R (batch) rows in `splits` deterministic slabs
// (weight gradient: dW = dY^T . X).  colsum_out (optional, [M]) = sum_r A[r][m] (the bias gradient) from the same pass.
// workspace: tt_gemm_tn_workspace_bytes(M,N,R).
size_t tt_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t R);
int tt_gemm_tn(hipStream_t st, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
               int64_t N, int64_t R, void* workspace, size_t workspace_bytes, float* colsum_out = nullptr);
