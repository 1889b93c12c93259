// Probe (run on the GPU box): the two instructions the fp8 gradient products of the score backward rely on.
//  (1) v_cvt_scalef32_pk_fp8_f32: which way the scale goes (x / scale or x * scale), whether only the scale's exponent is
//      used, rounding (nearest even against an exhaustive host encoder) and what happens past 448.
//  (2) v_mfma_scale_f32_32x32x64_f8f6f4 with a PER-LANE scale of the first operand: which 32 of a row's 64 k-values the
//      scale of lane (row c, half h) multiplies.  With all-ones operands every assignment of the two scales to the two
//      blocks gives the same sum (that was the first version of this probe, and it let a wrong assumption through: "the 32
//      values the lane itself holds"); the variants with only bytes 0..15 / only bytes 16..31 of every lane non-zero tell
//      them apart: the scale of lane (c, 0) multiplies bytes 0..15 of BOTH lanes (c, 0) and (c, 1), the scale of lane (c, 1)
//      bytes 16..31 of both -- the instruction's k order is [half 0 bytes 0..15 | half 1 bytes 0..15 | half 0 bytes
//      16..31 | half 1 bytes 16..31], and a block is 32 consecutive k.
//  (3) v_permlane32_swap_b32: the cross-half exchange that brings a block's two lanes together.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
using i32x8 = __attribute__((ext_vector_type(8))) int;
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef short v2s __attribute__((ext_vector_type(2)));

__global__ void cvt_kernel(const float* x, const float* sc, int n, uint8_t* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  v2s r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, x[i], 0.f, sc[i], false);
  out[i] = (uint8_t)(r[0] & 0xFF);
}

__global__ void swap_kernel(const unsigned* x, unsigned* o) {
  const auto r = __builtin_amdgcn_permlane32_swap(x[threadIdx.x], x[threadIdx.x + 64], false, false);
  o[threadIdx.x] = r[0];
  o[threadIdx.x + 64] = r[1];
}

__global__ void mfma_kernel(const i32x8* a, const i32x8* b, const int* sa, f32x16* c) {
  f32x16 acc = {};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0, sa[threadIdx.x], 0, 0x7F7F7F7F);
  c[threadIdx.x] = acc;
}

static float dec(uint8_t c) {                       // OCP e4m3fn
  const int e = (c >> 3) & 15, m = c & 7;
  const float v = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.f + m / 8.f, e - 7);
  return (c & 0x80) ? -v : v;
}
static uint8_t enc_rne(float x) {                   // nearest finite code, ties to the even code, saturating
  const float ax = fabsf(x);
  int best = 0;
  double bd = 1e300;
  for (int c = 0; c <= 0x7E; ++c) {
    const double d = fabs((double)dec((uint8_t)c) - (double)ax);
    if (d < bd || (d == bd && (c & 1) == 0)) { bd = d; best = c; }
  }
  return (uint8_t)(best | (x < 0 ? 0x80 : 0));
}

int main() {
  // ---- (1)
  std::vector<float> x, sc;
  uint32_t s = 777;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0f / 16777216.0f); };
  const float scales[] = {1.f, 2.f, 0.25f, 1024.f, ldexpf(1.f, -20), 3.f, 1.5f, ldexpf(1.f, -40)};
  for (float k : scales)
    for (int i = 0; i < 4000; ++i) {
      const float mag = ldexpf(1.f + rnd(), (int)(rnd() * 20) - 12);      // 2^-12 .. 2^8 times the scale
      x.push_back((i & 1 ? -mag : mag) * k);
      sc.push_back(k);
    }
  for (float k : {1.f, 4.f}) {                                            // ties, saturation, the subnormal range
    for (float v : {448.f, 449.f, 464.f, 480.f, 512.f, 1000.f, 1e6f, 0.f, ldexpf(1.f, -9), ldexpf(1.f, -10), ldexpf(3.f, -11), 17.f, 19.f, 1.0625f, 1.1875f}) {
      x.push_back(v * k);
      sc.push_back(k);
    }
  }
  const int n = (int)x.size();
  float *dx, *ds;
  uint8_t* dout;
  hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dout, n);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(ds, sc.data(), n * 4, hipMemcpyHostToDevice);
  cvt_kernel<<<(n + 255) / 256, 256>>>(dx, ds, n, dout);
  std::vector<uint8_t> out(n);
  hipMemcpy(out.data(), dout, n, hipMemcpyDeviceToHost);
  int bad_div = 0, bad_mul = 0, bad_div_exp = 0;
  for (int i = 0; i < n; ++i) {
    const float p2 = ldexpf(1.f, (int)floorf(log2f(sc[i])));              // the scale's power of two
    if (out[i] != enc_rne(x[i] / sc[i])) ++bad_div;
    if (out[i] != enc_rne(x[i] * sc[i])) ++bad_mul;
    if (out[i] != enc_rne(x[i] / p2)) ++bad_div_exp;
  }
  printf("cvt_scalef32_pk_fp8_f32 over %d cases: mismatches vs rne(x / scale) %d, vs rne(x * scale) %d, vs rne(x / 2^floor(log2 scale)) %d\n",
         n, bad_div, bad_mul, bad_div_exp);
  for (int i = n - 30; i < n; ++i)
    printf("  x = %-12g scale = %-4g -> 0x%02x = %-8g (host rne(x / scale) 0x%02x)\n", x[i], sc[i], out[i], dec(out[i]), enc_rne(x[i] / sc[i]));
  // ---- (2)  B = all ones; A = ones in (variant 0) all bytes, (1) bytes 0..15 only, (2) bytes 16..31 only; scale of lane l: 2^(l % 7 - 3)
  std::vector<int> hs(64);
  for (int l = 0; l < 64; ++l) hs[l] = 127 + (l % 7) - 3;
  i32x8 *da, *db;
  int* dsa;
  f32x16* dc;
  hipMalloc(&da, 64 * 32); hipMalloc(&db, 64 * 32); hipMalloc(&dsa, 64 * 4); hipMalloc(&dc, 64 * 64);
  hipMemcpy(dsa, hs.data(), 64 * 4, hipMemcpyHostToDevice);
  for (int variant = 0; variant < 3; ++variant) {
    std::vector<uint8_t> ha(64 * 32, 0), hb(64 * 32, 0x38);
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 32; ++j)
        if (variant == 0 || (variant == 1) == (j < 16)) ha[l * 32 + j] = 0x38;
    hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice);
    mfma_kernel<<<1, 64>>>(da, db, dsa, dc);
    std::vector<float> hc(64 * 16);
    hipMemcpy(hc.data(), dc, 64 * 64, hipMemcpyDeviceToHost);
    int bad_own = 0, bad_blk = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        const float s0 = ldexpf(1.f, hs[row] - 127), s1 = ldexpf(1.f, hs[row + 32] - 127);
        // "own": a lane's scale multiplies the 32 bytes it holds;  "blk": lane (c, 0)'s scale multiplies bytes 0..15 of both halves
        const float own = variant == 0 ? 32.f * (s0 + s1) : 16.f * (s0 + s1);
        const float blk = variant == 0 ? 32.f * (s0 + s1) : (variant == 1 ? 32.f * s0 : 32.f * s1);
        if (hc[l * 16 + r] != own) ++bad_own;
        if (hc[l * 16 + r] != blk) ++bad_blk;
      }
    printf("variant %d: mismatches vs \"scale of the lane's own 32 bytes\" %d, vs \"lane (c,0) scales bytes 0..15 of both halves, lane (c,1) bytes 16..31\" %d\n",
           variant, bad_own, bad_blk);
  }
  // ---- (3)
  std::vector<unsigned> hx(128), ho(128);
  for (int i = 0; i < 128; ++i) hx[i] = i;
  unsigned *dx2, *do2;
  hipMalloc(&dx2, 512); hipMalloc(&do2, 512);
  hipMemcpy(dx2, hx.data(), 512, hipMemcpyHostToDevice);
  swap_kernel<<<1, 64>>>(dx2, do2);
  hipMemcpy(ho.data(), do2, 512, hipMemcpyDeviceToHost);
  int bad3 = 0;
  for (int l = 0; l < 64; ++l) {
    const unsigned want0 = l < 32 ? l : 64 + (l - 32);        // r[0]: lanes 0..31 keep a, lanes 32..63 get b's lanes 0..31
    const unsigned want1 = l < 32 ? 32 + l : 64 + l;          // r[1]: lanes 0..31 get a's lanes 32..63, lanes 32..63 keep b
    if (ho[l] != want0 || ho[64 + l] != want1) ++bad3;
  }
  printf("permlane32_swap(a, b): r[0] = [a.lo | b.lo], r[1] = [a.hi | b.hi]: %d mismatches (r[0][0,32,33] = %u %u %u, r[1][0,1,32] = %u %u %u)\n", bad3,
         ho[0], ho[32], ho[33], ho[64], ho[65], ho[96]);
  return 0;
}
