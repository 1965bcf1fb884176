// Probe of ds_read_b64_tr_b16 semantics on gfx950: every lane supplies its own 8-byte-aligned LDS address; prints which
// LDS elements each lane receives.  Build: hipcc --offload-arch=gfx950 -O2 tr_read_probe.hip -o tr_read_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(int* out, int mode) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  const int q = l & 15, grp = l >> 4;
  int off;
  if (mode == 0) off = grp * 64 + (q >> 2) * 16 + (q & 3) * 4;                 // contiguous [4][16] block per 16-lane group
  else off = grp * 1000 + (q >> 2) * 200 + (q & 3) * 4 + ((q >> 2) & 1) * 16;  // arbitrary per-row placement: rows 200 apart, odd rows shifted
  __attribute__((address_space(3))) s16x4* p = (__attribute__((address_space(3))) s16x4*)(lds + off);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  int* d; hipMalloc(&d, 64 * 4 * sizeof(int));
  int h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l % 2) ? "\n" : "   |   ");
  }
  return 0;
}
