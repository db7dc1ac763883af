// Fused epilogue shared by the implicit GEMM and the halo 3x3 conv kernel.
#pragma once
#include "common.h"

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

__device__ __forceinline__ long out_row(const IGemm& g, int m) {
  if (g.out_linear) return m;
  int x = m % g.X;
  int t = m / g.X;
  int y = t % g.Y;
  t /= g.Y;
  int z = t % g.Z;
  int b = t / g.Z;
  return ((long)(b * g.OZ + z * g.ozm + g.ozo) * g.OY + (y * g.oym + g.oyo)) * g.OX + (x * g.oxm + g.oxo);
}

// v: accumulator for column n (and `gate` for column n+32 when geglu)
__device__ __forceinline__ void igemm_epilogue_store(const IGemm& g, int m, long orow, int n, float v, float gate) {
  v *= g.alpha;
  if (g.bias) v += g.bias[n];
  int ncol = n;
  if (g.geglu) {
    gate *= g.alpha;
    if (g.bias) gate += g.bias[n + 32];
    v = v * gelu_erf(gate);
    ncol = (n >> 6) * 32 + (n & 31);
  } else {
    if (g.rowbias) {
      int b = m / (g.Z * g.Y * g.X);
      v += g.rowbias[(long)b * g.rb_ld + n];
    }
    if (g.resid) {
      if (g.resid_f32) v += ((const float*)g.resid)[orow * g.ldr + n];
      else v += (float)((const half_t*)g.resid)[orow * g.ldr + n];
    }
    if (g.act == ACT_SILU) v = v / (1.0f + __expf(-v));
  }
  if (g.out_f32) ((float*)g.out)[orow * g.ldc + ncol] = v;
  else ((half_t*)g.out)[orow * g.ldc + ncol] = (half_t)v;
}

