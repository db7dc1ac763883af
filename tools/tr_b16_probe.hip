// Probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read): LDS holds each half's own index; every lane reads one 8-byte address
// and prints which (row, col) elements it received.  mode 0: a 16-lane group addresses a [4 rows][16 cols] block (lane i -> row i / 4,
// columns 4 (i % 4) ..); mode 1: 16 rows x 4 columns.  Build and run on the GPU box: hipcc --offload-arch=gfx950 -O2 -w -o /tmp/tr
// tools/tr_b16_probe.hip && /tmp/tr   (for the V-row-major attention staging of DESIGN.md section 4)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void probe(int* out, int pitch_halfs, int mode) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 64; i += 64) lds[i] = (short)i;   // element value = its half index in LDS
  __syncthreads();
  // lane i of a 16-lane group: row (i / 4) and 4-column quad (i % 4) of a [4][16] block (mode 0), or row i (mode 1: 16 rows x 4 cols)
  const int g = lane >> 4, i = lane & 15;
  int off;
  if (mode == 0) off = (g * 4 + (i >> 2)) * pitch_halfs + (i & 3) * 4;
  else off = (g * 16 + i) * pitch_halfs;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + off));
  for (int k = 0; k < 4; ++k) out[lane * 4 + k] = (int)(unsigned short)v[k];
}
int main() {
  int* d; hipMalloc(&d, 64 * 4 * 4);
  int h[256];
  for (int mode = 0; mode < 2; ++mode) {
    const int pitch = mode == 0 ? 16 : 4;
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, pitch, mode);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("mode %d (pitch %d halfs): lane -> 4 values as (row, col) of the source block\n", mode, pitch);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int k = 0; k < 4; ++k) printf(" (%d,%d)", h[l * 4 + k] / pitch, h[l * 4 + k] % pitch);
      printf("\n");
    }
  }
  return 0;
}
