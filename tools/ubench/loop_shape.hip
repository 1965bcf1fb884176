// microbenchmark: the MFMA rate a conv inner-loop SHAPE can reach on gfx950, independent of any real kernel.
// One "iteration" = what a wave does between two workgroup barriers:
//   NMFMA  v_mfma_f32_32x32x16_bf16 on NACC accumulator tiles,
//   NREAD  ds_read_b128 (operand fragments),
//   NDMA   1 KiB global_load_lds_dwordx4 pieces (L2-resident source) with a counted vmcnt wait,
//   NVALU  fp32 VALU instructions (operand transform / split work),
//   one s_barrier.
// Used to price the Winograd variants of the stride-1 3x3 conv against the shipping direct "patch" loop BEFORE building
// them: a variant's projected direct-conv-equivalent rate = measured MFMA rate x its MFMA saving.
// Build: hipcc --offload-arch=gfx950 -O3 loop_shape.hip -o loop_shape
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

template <int NMFMA, int NACC, int NREAD, int NDMA, int NVALU>
__global__ __launch_bounds__(256) void loop_kernel(float* out, int iters, const float* wsrc, int dma_slots) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  constexpr int NF = NREAD > 0 ? NREAD : 4;
  bf16x8_t frag[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) frag[i][j] = (__bf16)(float)((lane + i + j) & 7);
  for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<float*>(lds)[i] = (float)(i & 15);
  __syncthreads();
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
  float v0 = (float)lane, v1 = 1.0001f;
  for (int it = 0; it < iters; ++it) {
    // operand fragments: conflict-free 16-byte reads, a different 1 KiB window per read
#pragma unroll
    for (int i = 0; i < NREAD; ++i)
      frag[i] = *reinterpret_cast<const bf16x8_t*>(lds + (((it + i) & 15) << 10) + lane * 16);
    // transform / split work
#pragma unroll
    for (int i = 0; i < NVALU; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v0) : "v"(v1));
    // matrix work: consecutive MFMAs hit different accumulators
#pragma unroll
    for (int i = 0; i < NMFMA; ++i)
      acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag[i % NF], frag[(i * 7 + 3) % NF], acc[i % NACC], 0, 0, 0);
    // operand stream for a later iteration
    if (NDMA > 0) {
#pragma unroll
      for (int j = 0; j < NDMA; ++j) {
        const float* p = wsrc + (size_t)(((blockIdx.x & 3) * 64 + wave * 16 + (lane >> 2)) * 2304 + ((it * 16 + j * 64) % 2304) + (lane & 3) * 4);
        lds_dma16(p, lds_base + 32768 + ((((it & 1) * 4 * NDMA + wave * NDMA + j) % dma_slots) << 10));
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    }
    __builtin_amdgcn_s_barrier();
  }
  float s = v0;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NMFMA, int NACC, int NREAD, int NDMA, int NVALU>
static void run(const char* name, double direct_equiv_factor, int lds_kb, int blocks_per_cu, float* out, const float* wsrc) {
  const int iters = 1500, blocks = 256 * blocks_per_cu * 2;
  auto kern = loop_kernel<NMFMA, NACC, NREAD, NDMA, NVALU>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds_kb * 1024, 0, out, iters, wsrc, lds_kb - 32);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  const int reps = 12;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds_kb * 1024, 0, out, iters, wsrc, lds_kb - 32);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double flops = (double)blocks * 4 * iters * NMFMA * 2.0 * 32 * 32 * 16;
  const double tf = flops / ms / 1e9;
  printf("%-58s mfma %2d acc %2d reads %2d dma %2d valu %3d  lds %3d KB (%d blk/CU)  %7.0f TFLOP/s bf16 = %5.0f fp32-equivalent (x3 split) -> %5.0f direct-conv-equivalent  [%s]\n",
         name, NMFMA, NACC, NREAD, NDMA, NVALU, lds_kb, blocks_per_cu, tf, tf / 3.0, tf / 3.0 * direct_equiv_factor,
         hipGetErrorString(hipGetLastError()));
}

int main() {
  float *out, *wsrc;
  hipMalloc(&out, 8192 * 256 * 4);
  hipMalloc(&wsrc, 256 * 2304 * 4 + (1 << 20));
  hipMemset(wsrc, 0, 256 * 2304 * 4 + (1 << 20));
  // name, direct-conv-equivalent factor (direct MFMAs replaced per MFMA issued), LDS per block, blocks per CU
  run<12, 4, 8, 0, 0>("bare loop: 12 MFMA, 8 reads, barrier", 1.0, 48, 3, out, wsrc);
  run<12, 4, 8, 2, 14>("direct patch loop, BN=128 (shipping): per tap", 1.0, 48, 3, out, wsrc);
  run<24, 8, 12, 9, 40>("Winograd F(2,3) along W, j across waves, BN=128: per kh", 1.5, 138, 1, out, wsrc);
  run<12, 4, 8, 5, 40>("Winograd F(2,3) along W, j across waves, BN=64: per kh", 1.5, 74, 2, out, wsrc);
  run<6, 2, 6, 5, 50>("Winograd F(2x2,3x3), 32 tiles x BN=64: per position column", 2.25, 72, 2, out, wsrc);
  run<24, 8, 24, 17, 200>("Winograd F(2x2,3x3), 32 tiles x BN=64: whole K chunk", 2.25, 150, 1, out, wsrc);
  run<12, 4, 4, 2, 14>("direct, 2 taps per B tile read (fewer fragment bytes)", 1.0, 48, 3, out, wsrc);
  run<24, 4, 12, 4, 28>("direct, 2 taps per barrier", 1.0, 64, 2, out, wsrc);
  run<6, 2, 6, 1, 10>("direct patch loop, BN=64 (Cout = 64 convs, shipping): per tap", 1.0, 36, 3, out, wsrc);
  run<12, 2, 12, 2, 20>("direct patch loop, BN=64, two taps per barrier", 1.0, 48, 3, out, wsrc);
  run<18, 2, 18, 3, 30>("direct patch loop, BN=64, three taps per barrier", 1.0, 48, 3, out, wsrc);
  run<18, 6, 26, 3, 170>("row-tap wgrad loop (shipping): per 16-pixel chunk", 1.0, 40, 3, out, wsrc);
  run<18, 6, 26, 3, 60>("row-tap wgrad loop with a third of the VALU work", 1.0, 40, 3, out, wsrc);
  run<18, 6, 12, 3, 60>("row-tap wgrad loop, transposed 16-byte fragment reads, a third of the VALU", 1.0, 40, 3, out, wsrc);
  run<36, 6, 52, 6, 340>("row-tap wgrad loop, two chunks per barrier", 1.0, 64, 2, out, wsrc);
  run<4, 4, 4, 2, 8>("half mode (one product), BN=128: per tap", 3.0, 48, 3, out, wsrc);
  run<12, 4, 12, 6, 24>("half mode (one product), BN=128: three taps per barrier", 3.0, 72, 2, out, wsrc);
  return 0;
}
