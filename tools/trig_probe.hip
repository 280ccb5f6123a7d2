// trig_probe.hip -- accuracy of the hardware-transcendental Gabor path vs fp64
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../wire_amd/csrc/wire_dev.h"

__device__ __forceinline__ void sincos_hw(float x, float& sn, float& cs) {
  const float HI = 0.15915494f, LO = 6.42063833e-9f;
  float t = x * HI;
  float e = __builtin_fmaf(x, HI, -t);
  e = __builtin_fmaf(x, LO, e);
  float r = __builtin_amdgcn_fractf(t) + e;
  sn = __builtin_amdgcn_sinf(r);
  cs = __builtin_amdgcn_cosf(r);
}
__global__ void k(const float* x, float* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s, c, s2, c2;
  sincos_hw(x[i], s, c);
  wire_sincos(x[i], s2, c2);
  o[6 * i] = s; o[6 * i + 1] = c; o[6 * i + 2] = s2; o[6 * i + 3] = c2;
  o[6 * i + 4] = __builtin_amdgcn_exp2f(x[i] * 1.44269502f);   // uncompensated exp
  o[6 * i + 5] = wire_exp(x[i]);
}
int main() {
  const int n = 1 << 20;
  std::vector<float> hx(n), ho(6 * n);
  for (int i = 0; i < n; ++i) hx[i] = -200.0f + 400.0f * (i + 0.37f) / n;
  float *dx, *dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dout, 6 * n * 4);
  hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dout, n);
  hipMemcpy(ho.data(), dout, 6 * n * 4, hipMemcpyDeviceToHost);
  double e_hw = 0, e_poly = 0, e_exp_raw = 0, e_exp = 0;
  for (int i = 0; i < n; ++i) {
    double x = hx[i], s = sin(x), c = cos(x);
    e_hw = fmax(e_hw, fmax(fabs(ho[6 * i] - s), fabs(ho[6 * i + 1] - c)));
    e_poly = fmax(e_poly, fmax(fabs(ho[6 * i + 2] - s), fabs(ho[6 * i + 3] - c)));
    if (x < 3.0 && x > -87.0) {
      double ex = exp(x);
      // error relative to max(1, value): what matters for activations of magnitude <= e^2.25
      e_exp_raw = fmax(e_exp_raw, fabs(ho[6 * i + 4] - ex) / fmax(ex, 1.0));
      e_exp = fmax(e_exp, fabs(ho[6 * i + 5] - ex) / fmax(ex, 1.0));
    }
  }
  printf("sincos |x|<=200: hw path max abs err %.3e ; polynomial path %.3e\n", e_hw, e_poly);
  printf("exp x in (-87,3): raw v_exp max err/max(1,val) %.3e ; compensated %.3e\n", e_exp_raw, e_exp);
  return 0;
}
