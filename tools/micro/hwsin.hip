// Accuracy of the hardware v_sin_f32 / v_cos_f32 (input in revolutions) after a 2-term Cody-Waite reduction by 2*pi,
// against double-precision sin/cos, for arguments in [-40, 40] (the range of Fourier-feature arguments W.x).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__device__ __forceinline__ void hw_sincos(float a, float* s, float* c) {
  const float k = rintf(a * 0.15915494309189535f);
  float r = fmaf(-k, 6.28318548202514648f, a);          // 2pi hi (float)
  r = fmaf(-k, -1.74845553e-07f, r);                     // 2pi lo
  const float x = r * 0.15915494309189535f;             // revolutions in [-0.5, 0.5]
  *s = __builtin_amdgcn_sinf(x);
  *c = __builtin_amdgcn_cosf(x);
}
__global__ void k(const float* a, float* s, float* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) hw_sincos(a[i], &s[i], &c[i]);
}
int main() {
  const int n = 1 << 22;
  std::vector<float> a(n), s(n), c(n);
  unsigned long long st = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; a[i] = (float)((double)(st >> 11) / (double)(1ull << 53) * 80.0 - 40.0); }
  float *da, *ds, *dc;
  hipMalloc(&da, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
  hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(da, ds, dc, n);
  hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
  double es = 0, ec = 0, rs = 0; int is = 0;
  for (int i = 0; i < n; ++i) {
    const double S = sin((double)a[i]), C = cos((double)a[i]);
    const double e1 = fabs(s[i] - S), e2 = fabs(c[i] - C);
    if (e1 > es) { es = e1; is = i; }
    if (e2 > ec) ec = e2;
    if (fabs(S) > 1e-3) rs = fmax(rs, e1 / fabs(S));
  }
  printf("max abs err sin %.3e (arg %.6f) cos %.3e ; max rel err sin (|sin|>1e-3) %.3e ; float eps 5.96e-08\n", es, a[is], ec, rs);
  return 0;
}
