// Issue cost of the vector instructions the score kernels are made of, one wave per SIMD and four waves per SIMD:
// N dependent-free instructions per loop trip on 8 registers, timed with s_memtime (shader cycles) and s_memrealtime (100 MHz).
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probe/valu_rate tools/probe/valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ void rate_kernel(float* out, unsigned long long* stamps, int trips) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = -1.0f - 0.01f * (threadIdx.x + i);
  float w[8];
  for (int i = 0; i < 8; ++i) w[i] = 0.5f + i;
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_exp_f32 %0, %1" : "=v"(w[i]) : "v"(v[i]));
        if (OP == 1) asm volatile("v_add_f32 %0, %1, %2" : "=v"(w[i]) : "v"(v[i]), "v"(w[i]));
        if (OP == 2) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(w[i]) : "v"(v[i]), "v"(v[(i + 1) & 7]));
        if (OP == 3) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(w[i]) : "v"(v[i]), "v"(v[(i + 1) & 7]), "v"(w[i]));
        if (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w[i]) : "v"(v[i]), "v"(v[(i + 1) & 7]));
        if (OP == 5) asm volatile("v_rcp_f32 %0, %1" : "=v"(w[i]) : "v"(v[i]));
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += w[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {           // every wave: the workgroup's span is max(end) - min(start) on the host
    const int wv = blockIdx.x * 16 + (threadIdx.x >> 6);
    stamps[4 * wv] = c0; stamps[4 * wv + 1] = c1; stamps[4 * wv + 2] = r0; stamps[4 * wv + 3] = r1;
  }
}

template <int OP>
void run(const char* name, int waves_per_simd) {
  const int trips = 2000, blocks = 256, threads = 256 * waves_per_simd;   // one workgroup per CU
  float* out; unsigned long long* st;
  hipMalloc(&out, sizeof(float) * blocks * threads);
  hipMalloc(&st, sizeof(unsigned long long) * 64 * blocks);
  rate_kernel<OP><<<blocks, threads>>>(out, st, trips);
  rate_kernel<OP><<<blocks, threads>>>(out, st, trips);
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
  const double n = (double)trips * 32 * waves_per_simd;   // instructions per SIMD
  printf("%-18s %d wave(s)/SIMD: %6.2f cycles per instruction and SIMD (s_memtime), %6.3f ns, clock %.2f GHz\n", name, waves_per_simd,
         cyc / n, ns / n, cyc / ns);
  hipFree(out); hipFree(st);
}

int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run<0>("v_exp_f32", w);
    run<5>("v_rcp_f32", w);
    run<1>("v_add_f32", w);
    run<2>("v_mul_f32", w);
    run<3>("v_max3_f32", w);
    run<4>("v_cvt_pk_bf16_f32", w);
  }
  return 0;
}
