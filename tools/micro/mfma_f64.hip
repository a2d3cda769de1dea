// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950: the FP64 matrix roof the 61-state kernels
// are priced against.  build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64 tools/micro/mfma_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v4d __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int ACCS>
__global__ __launch_bounds__(256) void k(double * out, int iters)
{
  v4d acc[ACCS];
  for (int i = 0; i < ACCS; ++i) acc[i] = v4d{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it)
  {
#pragma unroll
    for (int i = 0; i < ACCS; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < ACCS; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int ACCS>
static void run(int waves_per_simd, int cus = 256)
{
  const int iters = 20000;
  const int blocks = cus * waves_per_simd;       // 256 threads = 4 waves = one per SIMD
  double * out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<ACCS>, dim3(blocks), dim3(256), 0, 0, out, 100);
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<ACCS>, dim3(blocks), dim3(256), 0, 0, out, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double mfmas = (double)blocks * 4 * iters * ACCS;
  printf("CUs %3d, accumulators %d, waves/SIMD %d: %.2f ms, %.1f TFLOP/s, %.1f ns per MFMA per SIMD\n", cus, ACCS, waves_per_simd, ms,
         mfmas * 2048 / (ms * 1e-3) / 1e12, ms * 1e6 / (mfmas / (cus * 4.0)));
  CHECK(hipFree(out));
}

int main()
{
  run<1>(1); run<4>(1); run<8>(1); run<4>(2); run<8>(2); run<8>(4);
  // fewer CUs busy: is the full-chip rate limited by power / clocks?
  run<8>(2, 8); run<8>(2, 32); run<8>(2, 64); run<8>(2, 128);
  return 0;
}
