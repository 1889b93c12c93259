// How a SIMD overlaps a chain of four dependent v_mfma_f32_32x32x16_bf16 with the softmax-like vector work of a score tile
// (16 v_exp_f32 + 32 v_add_f32 on registers the MFMAs do not touch), for 1, 2 and 4 waves per SIMD, one workgroup per CU.
// MODE 0: MFMAs only; 1: vector work only; 2: both, MFMA chain first; 3: both, interleaved by hand (1 MFMA, 4 exp + 8 add, ...).
// Build: hipcc --offload-arch=gfx950 -O2 -w -o tools/probe/mix_rate tools/probe/mix_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define EXP(i) asm volatile("v_exp_f32 %0, %1" : "=v"(w[i]) : "v"(v[i]))
#define ADD(i) asm volatile("v_add_f32 %0, %1, %2" : "=v"(u[i]) : "v"(v[i]), "v"(u[i]))

template <int MODE>
__global__ void mix_kernel(float* out, unsigned long long* stamps, int trips) {
  float v[16], w[16], u[16];
  for (int i = 0; i < 16; ++i) { v[i] = -1.0f - 0.01f * (threadIdx.x + i); w[i] = 0.f; u[i] = 0.f; }
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * i); b[i] = (__bf16)(0.02f * i); }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int t = 0; t < trips; ++t) {
    if (MODE == 0 || MODE == 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    }
    if (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) EXP(i);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) ADD(i);
    }
    if (MODE == 3) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
        for (int i = 0; i < 4; ++i) EXP(4 * k + i);
#pragma unroll
        for (int i = 0; i < 8; ++i) ADD((8 * k + i) & 15);
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += w[i] + u[i] + acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    const int wv = blockIdx.x * 16 + (threadIdx.x >> 6);
    stamps[4 * wv] = c0; stamps[4 * wv + 1] = c1; stamps[4 * wv + 2] = r0; stamps[4 * wv + 3] = r1;
  }
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
  const int trips = 2000, blocks = 256, threads = 256 * waves_per_simd;
  float* out; unsigned long long* st;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&st, sizeof(unsigned long long) * 64 * blocks);
  mix_kernel<MODE><<<blocks, threads>>>(out, st, trips);
  mix_kernel<MODE><<<blocks, threads>>>(out, st, trips);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(64 * blocks);
  hipMemcpy(h.data(), st, sizeof(unsigned long long) * 64 * blocks, hipMemcpyDeviceToHost);
  double cyc = 0, ns = 0;
  const int nw = threads / 64;
  for (int b = 0; b < blocks; ++b) {
    unsigned long long c0 = ~0ull, c1 = 0, r0 = ~0ull, r1 = 0;
    for (int w = 0; w < nw; ++w) {
      const unsigned long long* q = &h[4 * (b * 16 + w)];
      c0 = q[0] < c0 ? q[0] : c0; c1 = q[1] > c1 ? q[1] : c1; r0 = q[2] < r0 ? q[2] : r0; r1 = q[3] > r1 ? q[3] : r1;
    }
    cyc += (double)(c1 - c0); ns += 10.0 * (double)(r1 - r0);
  }
  cyc /= blocks; ns /= blocks;
  const double n = (double)trips * waves_per_simd;   // tile-equivalents per SIMD
  printf("%-34s %d wave(s)/SIMD: %7.1f cycles per tile-equivalent and SIMD, %7.1f ns, clock %.2f GHz\n", name, waves_per_simd, cyc / n,
         ns / n, cyc / ns);
  hipFree(out); hipFree(st);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run<0>("4 chained MFMA", w);
    run<1>("16 exp + 32 add", w);
    run<2>("MFMA chain, then exp + add", w);
    run<3>("MFMA / exp + add interleaved", w);
  }
  return 0;
}
