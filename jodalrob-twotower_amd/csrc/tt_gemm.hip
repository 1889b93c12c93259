// f32 MFMA GEMM tiles for the tower MLP (tall-skinny: M = batch, N,K = layer widths).
// 64x64 output tile per 256-thread workgroup, 4 waves x one 32x32 accumulator
// (v_mfma_f32_32x32x2_f32), K staged 16 deep through LDS in k-major order so that both MFMA
// operand reads are conflict-free ds_read_b32 (lanes 0-31 -> k, lanes 32-63 -> k+1).
#include "tt_gemm.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 16, PAD = 4, THREADS = 256;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// MODE 0: operand is K-contiguous  : elem(x, k) = P[x*ld + k]   (x = m for A, n for B)
// MODE 1: operand is X-contiguous  : elem(x, k) = P[k*ld + x]
template <int MODE, bool VEC>
struct TileLoader {
  float r[4];
  __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int x0, int X, int k0, int kend, int t) {
    if (MODE == 0) {
      const int row = t >> 2, kq = (t & 3) * 4;
      const int x = x0 + row, k = k0 + kq;
      if (VEC && x < X && k + 3 < kend) {
        const float4 v = *reinterpret_cast<const float4*>(P + (int64_t)x * ld + k);
        r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = (x < X && k + j < kend) ? P[(int64_t)x * ld + k + j] : 0.f;
      }
    } else {
      const int k = k0 + (t >> 4), xq = (t & 15) * 4;
      const int x = x0 + xq;
      if (VEC && k < kend && x + 3 < X) {
        const float4 v = *reinterpret_cast<const float4*>(P + (int64_t)k * ld + x);
        r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = (k < kend && x + j < X) ? P[(int64_t)k * ld + x + j] : 0.f;
      }
    }
  }
  __device__ __forceinline__ void store(float (*S)[BM + PAD], int t) const {
    if (MODE == 0) {
      const int row = t >> 2, kq = (t & 3) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) S[kq + j][row] = r[j];
    } else {
      const int k = t >> 4, xq = (t & 15) * 4;
      *reinterpret_cast<float4*>(&S[k][xq]) = make_float4(r[0], r[1], r[2], r[3]);
    }
  }
};

// COLSUM (TN mode only): additionally accumulate sum_k A[k][m] (the bias gradient sum_batch dY) of this
// workgroup's k-range into colsum_slab[blockIdx.z][m] -- the A tiles are in registers anyway.
template <int MODE_A, int MODE_B, bool VEC, bool COLSUM = false>
__global__ __launch_bounds__(THREADS) void gemm_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                      int64_t ldb, float* __restrict__ C, int64_t ldc, int64_t slab_stride,
                                                      int M, int N, int K, int kchunk, const float* __restrict__ bias, int relu, float alpha,
                                                      float* __restrict__ colsum_slab = nullptr) {
  __shared__ __attribute__((aligned(16))) float As[BK][BM + PAD];
  __shared__ __attribute__((aligned(16))) float Bs[BK][BN + PAD];
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int kbeg = blockIdx.z * kchunk;
  const int kend = min(K, kbeg + kchunk);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  TileLoader<MODE_A, VEC> la;
  TileLoader<MODE_B, VEC> lb;
  if (kbeg < kend) {
    la.load(A, lda, m0, M, kbeg, kend, t);
    lb.load(B, ldb, n0, N, kbeg, kend, t);
  }
  const int li = lane & 31, lh = lane >> 5;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    la.store(As, t);
    lb.store(Bs, t);
    if (COLSUM) {
#pragma unroll
      for (int j = 0; j < 4; ++j) cs[j] += la.r[j];
    }
    __syncthreads();
    if (k0 + BK < kend) {
      la.load(A, lda, m0, M, k0 + BK, kend, t);
      lb.load(B, ldb, n0, N, k0 + BK, kend, t);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a = As[kk + lh][wr * 32 + li];
      const float b = Bs[kk + lh][wc * 32 + li];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  if (COLSUM && colsum_slab && blockIdx.y == 0) {          // 16 k-lanes per column quad -> fixed-order sum through LDS
    float(*red)[BM + PAD] = As;                            // (As is free: the loop ended with a barrier)
    *reinterpret_cast<float4*>(&red[t >> 4][(t & 15) * 4]) = make_float4(cs[0], cs[1], cs[2], cs[3]);
    __syncthreads();
    if (t < BM && m0 + t < M) {
      float sum = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) sum += red[q][t];
      colsum_slab[(int64_t)blockIdx.z * M + m0 + t] = sum;
    }
  }
  float* Cz = C + (int64_t)blockIdx.z * slab_stride;
  const int n = n0 + wc * 32 + li;
  const float bv = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (m < M && n < N) {
      float v = acc[r] * alpha + bv;
      if (relu) v = fmaxf(v, 0.f);
      Cz[(int64_t)m * ldc + n] = v;
    }
  }
}

__global__ __launch_bounds__(THREADS) void slab_reduce_kernel(const float* __restrict__ slabs, int64_t slab_stride, int splits,
                                                             float* __restrict__ C, int64_t ldc, int M, int N,
                                                             const float* __restrict__ colsum_slab, float* __restrict__ colsum_out) {
  const int64_t total = (int64_t)M * N;
  const int64_t all = total + (colsum_out ? M : 0);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < all; i += stride) {
    float s = 0.f;
    if (i < total) {
#pragma unroll 8
      for (int z = 0; z < splits; ++z) s += slabs[(int64_t)z * slab_stride + i];
      const int64_t m = i / N, n = i - m * N;
      C[m * ldc + n] = s;
    } else {
      const int64_t m = i - total;
#pragma unroll 8
      for (int z = 0; z < splits; ++z) s += colsum_slab[(int64_t)z * M + m];
      colsum_out[m] = s;
    }
  }
}

inline bool vec_ok(const float* p, int64_t ld) { return tt_aligned(p, 16) && (ld % 4 == 0); }

inline int tn_splits(int64_t M, int64_t N, int64_t R) {
  const int64_t tiles = tt_cdiv(M, BM) * tt_cdiv(N, BN);
  int64_t s = 256 / (tiles > 0 ? tiles : 1);
  int64_t maxs = tt_cdiv(R, 128);
  if (maxs > 32) maxs = 32;
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  return (int)s;
}

}  // namespace

int tt_gemm_nt(hipStream_t st, const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C,
               int64_t ldc, int64_t M, int64_t N, int64_t K, bool relu, float alpha) {
  if (M == 0 || N == 0) return TT_OK;
  dim3 grid((unsigned)tt_cdiv(M, BM), (unsigned)tt_cdiv(N, BN), 1);
  const int kchunk = (int)(tt_cdiv(K > 0 ? K : 1, BK) * BK);
  if (vec_ok(A, lda) && vec_ok(W, ldw))
    gemm_kernel<0, 0, true><<<grid, THREADS, 0, st>>>(A, lda, W, ldw, C, ldc, 0, (int)M, (int)N, (int)K, kchunk, bias, relu, alpha);
  else
    gemm_kernel<0, 0, false><<<grid, THREADS, 0, st>>>(A, lda, W, ldw, C, ldc, 0, (int)M, (int)N, (int)K, kchunk, bias, relu, alpha);
  TT_LAUNCH_CHECK();
  return TT_OK;
}

int tt_gemm_nn(hipStream_t st, const float* A, int64_t lda, const float* W, int64_t ldw, float* C, int64_t ldc, int64_t M,
               int64_t N, int64_t K) {
  if (M == 0 || N == 0) return TT_OK;
  dim3 grid((unsigned)tt_cdiv(M, BM), (unsigned)tt_cdiv(N, BN), 1);
  const int kchunk = (int)(tt_cdiv(K > 0 ? K : 1, BK) * BK);
  if (vec_ok(A, lda) && vec_ok(W, ldw))
    gemm_kernel<0, 1, true><<<grid, THREADS, 0, st>>>(A, lda, W, ldw, C, ldc, 0, (int)M, (int)N, (int)K, kchunk, nullptr, 0, 1.f);
  else
    gemm_kernel<0, 1, false><<<grid, THREADS, 0, st>>>(A, lda, W, ldw, C, ldc, 0, (int)M, (int)N, (int)K, kchunk, nullptr, 0, 1.f);
  TT_LAUNCH_CHECK();
  return TT_OK;
}

size_t tt_gemm_tn_workspace_bytes(int64_t M, int64_t N, int64_t R) {
  return sizeof(float) * (size_t)tn_splits(M, N, R) * ((size_t)M * (size_t)N + (size_t)M) + 256;
}

int tt_gemm_tn(hipStream_t st, const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
               int64_t N, int64_t R, void* workspace, size_t workspace_bytes, float* colsum_out) {
  if (M == 0 || N == 0) return TT_OK;
  const int splits = tn_splits(M, N, R);
  if (workspace_bytes < tt_gemm_tn_workspace_bytes(M, N, R) || !workspace) {
    tt_set_error("tt_gemm_tn: workspace %zu < required %zu", workspace_bytes, tt_gemm_tn_workspace_bytes(M, N, R));
    return TT_ERR_WORKSPACE;
  }
  float* slabs = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~uintptr_t(255));
  float* cslab = slabs + (size_t)splits * (size_t)M * (size_t)N;
  const int kchunk = (int)(tt_cdiv(tt_cdiv(R > 0 ? R : 1, splits), BK) * BK);
  dim3 grid((unsigned)tt_cdiv(M, BM), (unsigned)tt_cdiv(N, BN), (unsigned)splits);
  const int64_t slab_stride = M * N;
  const bool v = vec_ok(A, lda) && vec_ok(B, ldb);
  if (colsum_out) {
    if (v) gemm_kernel<1, 1, true, true><<<grid, THREADS, 0, st>>>(A, lda, B, ldb, slabs, N, slab_stride, (int)M, (int)N, (int)R, kchunk, nullptr, 0, 1.f, cslab);
    else gemm_kernel<1, 1, false, true><<<grid, THREADS, 0, st>>>(A, lda, B, ldb, slabs, N, slab_stride, (int)M, (int)N, (int)R, kchunk, nullptr, 0, 1.f, cslab);
  } else {
    if (v) gemm_kernel<1, 1, true><<<grid, THREADS, 0, st>>>(A, lda, B, ldb, slabs, N, slab_stride, (int)M, (int)N, (int)R, kchunk, nullptr, 0, 1.f);
    else gemm_kernel<1, 1, false><<<grid, THREADS, 0, st>>>(A, lda, B, ldb, slabs, N, slab_stride, (int)M, (int)N, (int)R, kchunk, nullptr, 0, 1.f);
  }
  TT_LAUNCH_CHECK();
  const int64_t total = M * N + (colsum_out ? M : 0);
  int blocks = (int)tt_cdiv(total, THREADS);
  if (blocks > 2048) blocks = 2048;
  slab_reduce_kernel<<<blocks, THREADS, 0, st>>>(slabs, slab_stride, splits, C, ldc, (int)M, (int)N, cslab, colsum_out);
  TT_LAUNCH_CHECK();
  return TT_OK;
}
