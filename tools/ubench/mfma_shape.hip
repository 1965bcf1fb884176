// Round 4 microbenchmark: does the MFMA SHAPE matter at the 1400 W cap?  Bare bf16 MFMA loops on random operands (registers only,
// no LDS, no memory), 256-thread blocks, 3 blocks per CU (the conv kernels' residency), a 64 x 64 fp32 accumulator tile per wave:
//   mode 0: v_mfma_f32_32x32x16_bf16, 4 accumulator tiles, 12 MFMAs per step (the split-bf16 kernels' inner step: K = 16)
//   mode 1: v_mfma_f32_16x16x32_bf16, 16 accumulator tiles, 48 MFMAs per double step (K = 32): the same MACs per K
// 32x32x16 moves 8 KB of accumulator through the register file per 16384 MACs, 16x16x32 2 KB per 8192 MACs (half per MAC) and
// twice the operand bytes per MAC.  Long runs (>= 1 s each) so the clock settles at the cap; prints bf16 TFLOP/s.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_shape.hip -o mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, const bf16x8_t* __restrict__ rnd) {
  extern __shared__ char lds[];
  const int tid = blockIdx.x * 256 + threadIdx.x;
  bf16x8_t a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = rnd[(tid * 16 + i) & 0xfffff]; b[i] = rnd[(tid * 16 + 8 + i) & 0xfffff]; }
  float s = 0.f;
  if (MODE == 0) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int half = 0; half < 2; ++half)       // two K = 16 steps per iteration (= one K = 32 step of mode 1)
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i >> 1) + 2 * (g & 1) + 4 * half], b[(i & 1) + 2 * (g >> 1) + 4 * half], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i >> 2) + 4 * (g & 1)], b[(i & 3) + 4 * (g >> 1)], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  }
  out[tid] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  bf16x8_t* rnd;
  const size_t nr = 1 << 20;
  hipMalloc(&rnd, nr * 16);
  std::vector<unsigned short> h(nr * 8);
  srand(1);
  for (auto& v : h) {                                 // random bf16 in (-2, 2): sign, exponent 0x3f or 0x3e.., 7 random mantissa bits
    const unsigned sign = rand() & 1, man = rand() & 0x7f, ex = 0x7b + (rand() & 3);
    v = (unsigned short)((sign << 15) | (ex << 7) | man);
  }
  hipMemcpy(rnd, h.data(), nr * 16, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000, blocks = 768;
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 2; ++mode) {
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 48 * 1024, 0, out, iters, rnd);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 48 * 1024, 0, out, iters, rnd);
      };
      for (int i = 0; i < 20; ++i) launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      const int n = 150;
      for (int i = 0; i < n; ++i) launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      // per iteration and wave: 64 x 64 x 32 MACs x 3 products
      const double flops = 2.0 * 64 * 64 * 32 * 3 * (double)iters * blocks * 4 * n;
      printf("mode %d (%s): %.1f ms for %d launches, %.0f bf16 TFLOP/s = %.0f split-bf16-equivalent\n", mode,
             mode == 0 ? "32x32x16, 4 acc tiles" : "16x16x32, 16 acc tiles", ms, n, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 3);
    }
  return 0;
}
