// v_mfma_f64_4x4x4_4b_f64 against v_mfma_f64_16x16x4_f64 on gfx950, for the four rows 16..19 of a 20 x 20 transition
// matrix (the second, 75 % padded M tile of the 20-state kernels, csrc/kernels_s20.hpp):
//   (1) layout: with lane = 16 q + n the 4x4x4 instruction reads A_b[i][k] from lane (k = q, b = n / 4, i = n % 4),
//       B_b[k][j] from lane (k = q, b = n / 4, j = n % 4) and leaves D_b[i][j] in lane (i = q, b, j) -- the B operand
//       and the D result are exactly the registers the 16x16x4 sequence uses (state row 4 ks + q, site n);
//   (2) bits: are the results identical to the 16x16x4 ones (same accumulation inside the instruction)?
//   (3) issue rate of the two instructions.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_4x4x4 tools/micro/mfma_4x4x4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef double v4d __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// P: [20][20] row-major; X: [20][16] (state, site); out16 / out4: [4][16] rows 16..19
__global__ void k_compare(const double * P, const double * X, double * out16, double * out4)
{
  const unsigned lane = threadIdx.x, q = lane >> 4, n = lane & 15;
  v4d acc = {0, 0, 0, 0};
  double acc4 = 0.0;
  for (int ks = 0; ks < 5; ++ks)
  {
    const double b = X[(4 * ks + q) * 16 + n];
    const double a16 = (n < 4) ? P[(16 + n) * 20 + 4 * ks + q] : 0.0;        // tile rows 16 + n, zero beyond row 19
    const double a4 = P[(16 + (n & 3)) * 20 + 4 * ks + q];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a16, b, acc, 0, 0, 0);
    acc4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a4, b, acc4, 0, 0, 0);
  }
  // 16x16x4 D: register v, lane (q, n) = row q + 4 v, column n -> register 0 holds rows 0..3 of the tile = 16..19
  out16[q * 16 + n] = acc[0];
  out4[q * 16 + n] = acc4;
}

template <int WHICH>
__global__ __launch_bounds__(256) void k_rate(double * out, int iters)
{
  v4d acc[4];
  double acc4[4];
  for (int i = 0; i < 4; ++i) { acc[i] = v4d{0, 0, 0, 0}; acc4[i] = 0.0; }
  double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it)
  {
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
      if (WHICH == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      else acc4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc4[i], 0, 0, 0);
    }
  }
  double s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + acc4[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int WHICH>
static void rate(int waves_per_simd, int cus)
{
  const int iters = 20000, blocks = cus * waves_per_simd;
  double * out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_rate<WHICH>, dim3(blocks), dim3(256), 0, 0, out, 100);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_rate<WHICH>, dim3(blocks), dim3(256), 0, 0, out, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double n = (double)blocks * 4 * iters * 4;
  printf("%s, CUs %3d, waves/SIMD %d: %.2f ms, %.1f ns per instruction per SIMD\n", WHICH ? "4x4x4_4b " : "16x16x4  ", cus, waves_per_simd,
         ms, ms * 1e6 / (n / (cus * 4.0)));
  CHECK(hipFree(out));
}

int main()
{
  double hP[400], hX[320], r16[64], r4[64], ref[64];
  unsigned long long s = 12345;
  auto rnd = [&]() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(s >> 11) / 9007199254740992.0; };
  int identical = 0, trials = 200;
  double worst = 0.0;
  double * dP, * dX, * d16, * d4;
  CHECK(hipMalloc(&dP, sizeof(hP))); CHECK(hipMalloc(&dX, sizeof(hX))); CHECK(hipMalloc(&d16, sizeof(r16))); CHECK(hipMalloc(&d4, sizeof(r4)));
  for (int t = 0; t < trials; ++t)
  {
    for (double & v : hP) v = rnd() * ((t & 1) ? 1.0 : exp(-20.0 * rnd()));
    for (double & v : hX) v = rnd() * ((t & 2) ? 1.0 : exp(-300.0 * rnd()));
    CHECK(hipMemcpy(dP, hP, sizeof(hP), hipMemcpyHostToDevice)); CHECK(hipMemcpy(dX, hX, sizeof(hX), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_compare, dim3(1), dim3(64), 0, 0, dP, dX, d16, d4);
    CHECK(hipMemcpy(r16, d16, sizeof(r16), hipMemcpyDeviceToHost)); CHECK(hipMemcpy(r4, d4, sizeof(r4), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 16; ++j)
      {
        long double a = 0;
        for (int k = 0; k < 20; ++k) a += (long double)hP[(16 + i) * 20 + k] * hX[k * 16 + j];
        ref[i * 16 + j] = (double)a;
        worst = fmax(worst, fabs(r4[i * 16 + j] - ref[i * 16 + j]) / fabs(ref[i * 16 + j]));
      }
    identical += memcmp(r16, r4, sizeof(r16)) == 0;
    if (t == 0)
      printf("sample: 16x16x4 %.17g  4x4x4 %.17g  long double %.17g\n", r16[5], r4[5], ref[5]);
  }
  printf("layout + bits: %d of %d random trials bit-identical between the two instructions; 4x4x4 vs long double: %.2e relative\n",
         identical, trials, worst);
  for (int w : {1, 2}) { rate<0>(w, 256); rate<1>(w, 256); }
  rate<0>(2, 64); rate<1>(2, 64);
  return 0;
}
