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

// one v_add_f32: keeps hipcc from SLP-packing neighbouring adds into v_pk_add_f32, which costs more than two scalar adds
// beside MFMAs (MI355X_MICROARCH.md, per-instruction constants)
__device__ __forceinline__ float add_asm(float a, float b) {
  float r;
  asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// fragments of 32-row tile t of a rows image [tile][k-step][half][row][8]: a wave-instruction reads 1 KB contiguous
template <int KS>
__device__ __forceinline__ void load_bfrag(const __bf16* __restrict__ b_rows, int64_t t, int c, int h, bf16x8 (&bf)[KS]) {
  const __bf16* p = b_rows + ((t * KS * 2 + h) * 32 + c) * 8;
#pragma unroll
  for (int s = 0; s < KS; ++s) bf[s] = *reinterpret_cast<const bf16x8*>(p + s * 512);
}

// ---- fp8 (OCP e4m3) operands of the S products: v_mfma_scale_f32_32x32x64_f8f6f4, K = 64 per instruction at twice the bf16
// rate.  Rows image [tile][k64-step][part 0..1][half][row][16 bytes]: lane (row c, half h) of a wave holds the 32 values
// k = 64 s + 32 h + 16 p + byte of its row -- the two operands only have to agree on which k sits in which byte (probe:
// tools/probe/mfma_fp8_layout.hip) -- and every 16-byte load of a wave is 1 KB contiguous.  The images hold
// fp8(64 * scale * x): unit rows sit around 1/16, most of that below e4m3's smallest normal 2^-6; the factor 2^6 goes back
// out through the instruction's block scales (e8m0 121 = 2^-6 on each operand), so the accumulator is scale_a * scale_b * x.y
// exactly as on the bf16 path.
using i32x8 = __attribute__((ext_vector_type(8))) int;
using i32x4 = __attribute__((ext_vector_type(4))) int;
using v2s16 = __attribute__((ext_vector_type(2))) short;
constexpr float kFp8Up = 64.f;
constexpr int kFp8ScaleE8M0 = 0x79797979;               // 2^-6 in every byte

inline int fp8_tile_bytes(int Dp) { return Dp * 32; }

template <int K64>
__device__ __forceinline__ void load_f8frag(const char* __restrict__ rows8, int64_t t, int c, int h, i32x8 (&f)[K64]) {
  const char* p = rows8 + (t * K64 * 4 + h) * 512 + c * 16;          // [(s * 2 + part) * 2 + half][row][16]
#pragma unroll
  for (int s = 0; s < K64; ++s) {
    const i32x4 lo = *reinterpret_cast<const i32x4*>(p + (s * 4) * 512);
    const i32x4 hi = *reinterpret_cast<const i32x4*>(p + (s * 4 + 2) * 512);
    f[s] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  }
}

__device__ __forceinline__ f32x16 mfma_f8(const i32x8& a, const i32x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, kFp8ScaleE8M0, 0, kFp8ScaleE8M0);
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
// fp8 packing: [fp8 rows image: Rp * Dp bytes | bf16 fragment image: Rp * Dp * 2 bytes | fp8 fragment image: Rp * Dp bytes];
// Dp is a multiple of 64 here.  The fp8 fragment image is the second operand of the gradient products with K = 64 rows per
// MFMA: [pair of 32-row tiles P][32-column block d][part 0..1][half h][column c][16 bytes], byte j of part p = element
// (row 64 P + 32 p + rowmap(j, h), column 32 d + c) -- the order in which a lane of the S accumulators holds the rows of the pair
// -- as fp8(64 * scale * x), like the rows image.
struct PackedView8 {
  const char* rows8;
  const __bf16* frag;
  const char* frag8;
};
inline int padded_d8(int D) { return D <= 64 ? 64 : (D <= 128 ? 128 : 256); }
inline PackedView8 view8(const void* packed, int64_t R, int D) {
  const int64_t Rp = rup(R, 64);
  const int Dp = padded_d8(D);
  const char* base = reinterpret_cast<const char*>(packed);
  return PackedView8{base, reinterpret_cast<const __bf16*>(base + Rp * Dp), base + 3 * Rp * Dp};
}

}  // namespace ttscore
