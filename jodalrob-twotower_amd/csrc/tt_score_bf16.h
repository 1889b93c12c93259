// Shared pieces of the bf16-operand score kernels (tt_score_bf16.hip, tt_score_sym.hip): vector types, the packed operand
// images' addressing, small device helpers.
#pragma once
#include "tt_common.h"

namespace ttscore {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kNegBig = -3.0e38f;

__host__ __device__ inline int64_t rup(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
inline int padded_d(int D) { return D <= 32 ? 32 : (D <= 64 ? 64 : (D <= 128 ? 128 : 256)); }

// row of accumulator register r in lane half lh of a 32x32 MFMA result
__device__ __forceinline__ int rowmap(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// single-instruction 3-input max (plain fmaxf on MFMA results makes hipcc insert canonicalising v_max first)
__device__ __forceinline__ float max3_asm(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// fragments of 32-row tile t of a rows image [tile][k-step][half][row][8]: a wave-instruction reads 1 KB contiguous
template <int KS>
__device__ __forceinline__ void load_bfrag(const __bf16* __restrict__ b_rows, int64_t t, int c, int h, bf16x8 (&bf)[KS]) {
  const __bf16* p = b_rows + ((t * KS * 2 + h) * 32 + c) * 8;
#pragma unroll
  for (int s = 0; s < KS; ++s) bf[s] = *reinterpret_cast<const bf16x8*>(p + s * 512);
}

struct PackedView {
  const __bf16* rows;
  const __bf16* frag;
};
inline PackedView view(const void* packed, int64_t R, int D) {
  const int64_t Rp = rup(R, 64);
  const int Dp = padded_d(D);
  const __bf16* base = reinterpret_cast<const __bf16*>(packed);
  return PackedView{base, base + Rp * Dp};
}

}  // namespace ttscore
