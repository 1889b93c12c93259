// Probe (run on the GPU box): operand lane maps of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands and unit scales.
// Fills A [32 x 64] and B [64 x 32] with small exact integers (fp8-representable), tries candidate lane -> (row, k) maps and
// prints the one whose result equals the integer matrix product.  cdna_hip_programming.md: "check the map with exact
// integer data before relying on it".
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
using i32x8 = __attribute__((ext_vector_type(8))) int;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__global__ void k(const i32x8* a, const i32x8* b, f32x16* c) {
  f32x16 acc = {};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
  c[threadIdx.x] = acc;
}

// e4m3 encodings of 0, 1, 2, 3, 4, -1, -2, 0.5
static const uint8_t enc[8] = {0x00, 0x38, 0x40, 0x44, 0x48, 0xB8, 0xC0, 0x30};
static const float val[8] = {0.f, 1.f, 2.f, 3.f, 4.f, -1.f, -2.f, 0.5f};

int kmap(int variant, int half, int j) {
  switch (variant) {
    case 0: return 32 * half + j;                       // 32 consecutive k per lane
    case 1: return 16 * half + (j & 15) + 32 * (j >> 4); // two K=32 halves, 16 per lane each
    case 2: return 8 * half + (j & 7) + 16 * (j >> 3);   // four K=16 quarters, 8 per lane each
    default: return 4 * half + (j & 3) + 8 * (j >> 2);
  }
}

int main() {
  std::vector<int> A(32 * 64), B(64 * 32);
  uint32_t s = 12345;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (int)((s >> 24) & 7); };
  for (auto& x : A) x = rnd();
  for (auto& x : B) x = rnd();
  std::vector<float> ref(32 * 32, 0.f);
  for (int i = 0; i < 32; ++i)
    for (int n = 0; n < 32; ++n) {
      float t = 0.f;
      for (int kk = 0; kk < 64; ++kk) t += val[A[i * 64 + kk]] * val[B[kk * 32 + n]];
      ref[i * 32 + n] = t;
    }
  i32x8 *da, *db;
  f32x16* dc;
  hipMalloc(&da, 64 * sizeof(i32x8)); hipMalloc(&db, 64 * sizeof(i32x8)); hipMalloc(&dc, 64 * sizeof(f32x16));
  for (int va = 0; va < 4; ++va)
    for (int vb = 0; vb < 4; ++vb) {
      std::vector<uint8_t> ha(64 * 32), hb(64 * 32);
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 32; ++j) {
          ha[l * 32 + j] = enc[A[(l & 31) * 64 + kmap(va, l >> 5, j)]];
          hb[l * 32 + j] = enc[B[kmap(vb, l >> 5, j) * 32 + (l & 31)]];
        }
      hipMemcpy(da, ha.data(), ha.size(), hipMemcpyHostToDevice);
      hipMemcpy(db, hb.data(), hb.size(), hipMemcpyHostToDevice);
      k<<<1, 64>>>(da, db, dc);
      std::vector<float> hc(64 * 16);
      hipMemcpy(hc.data(), dc, hc.size() * 4, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
          if (hc[l * 16 + r] != ref[row * 32 + col]) ++bad;
        }
      printf("A map %d, B map %d: %d mismatches%s\n", va, vb, bad, bad == 0 ? "   <== MATCH" : "");
    }
  return 0;
}
