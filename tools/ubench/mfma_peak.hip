// microbenchmark: bare v_mfma_f32_32x32x16_bf16 rate with the conv kernels' shape (256 threads, 4 accumulator tiles,
// 12 MFMAs per iteration), optionally with 8 ds_read_b128 and a barrier per iteration.  Build: hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

__device__ inline void lds_dma16(const float* gsrc, unsigned lds_dst_uniform) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst_uniform) : "memory");
}
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int dyn, const float* wsrc) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8_t a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)(float)(lane + i + j); b[i][j] = (__bf16)(float)(lane * 2 + i + j); }
  if (MODE >= 1) {
    for (int i = threadIdx.x; i < 12288; i += 256) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
  }
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 1) {
      const char* base = lds + ((it & 3) << 13) + lane * 64;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a[i] = *reinterpret_cast<const bf16x8_t*>(base + i * 16 + (threadIdx.x >> 6) * 4096 % 8192);
        b[i] = *reinterpret_cast<const bf16x8_t*>(base + 32768 % 16384 + i * 16);
      }
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + g) & 3], b[(i >> 1) + g & 3], acc[i], 0, 0, 0);
    if (MODE >= 3) {      // MODE 3: +2, MODE 4: +4 LDS-DMA pieces (1 KiB each, 64-B segments 9 KiB apart, L2 resident) per iteration
      const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
      const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
      for (int j = 0; j < (MODE == 3 ? 2 : 4); ++j) {
        const float* p = wsrc + (size_t)((blockIdx.x & 1) * 128 + wave * 32 + j * 16 + (lane >> 2)) * 2304 + ((it * 16) % 2304) + (lane & 3) * 4;
        lds_dma16(p, lds_base + 16384 + ((it & 1) << 13) + (wave * 4 + j) * 1024);
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MODE == 3 ? 2 : 4) : "memory");
    }
    if (MODE >= 2) __builtin_amdgcn_s_barrier();
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  float* wsrc;
  hipMalloc(&wsrc, 256 * 2304 * 4 + 65536);
  hipMemset(wsrc, 0, 256 * 2304 * 4 + 65536);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 5; ++mode)
    for (int lds_kb : {48, 76}) {        // 3, 2, 1 blocks per CU
      for (int blocks : {256 * 6}) {
        auto launch = [&]() {
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), lds_kb * 1024, 0, out, iters, 0, wsrc);
          if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), lds_kb * 1024, 0, out, iters, 0, wsrc);
          if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), lds_kb * 1024, 0, out, iters, 0, wsrc);
          if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), lds_kb * 1024, 0, out, iters, 0, wsrc);
          if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), lds_kb * 1024, 0, out, iters, 0, wsrc);
        };
        hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        launch(); launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double flops = (double)blocks * 4 * iters * 12 * 2.0 * 32 * 32 * 16;
        printf("mode %d (0 bare, 1 +8 ds_read_b128, 2 +barrier, 3 +2 DMA, 4 +4 DMA)  lds %3d KB/block  %.3f ms  %.0f TFLOP/s bf16 (%.0f fp32-equivalent /3)  err=%s\n", mode, lds_kb, ms,
               flops / ms / 1e9, flops / ms / 1e9 / 3, hipGetErrorString(hipGetLastError()));
      }
    }
  return 0;
}
