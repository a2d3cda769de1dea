// Host-side cost of a kernel launch as a function of the size of its by-value arguments, its dynamic LDS
// request and its grid:  hipcc --offload-arch=gfx950 -O2 -o launch_cost launch_cost.hip && ./launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
template <unsigned N> struct Args { unsigned char b[N]; };
template <unsigned N> __global__ void k(Args<N> a, unsigned * out)
{
  extern __shared__ double lds[];
  if (threadIdx.x == 9999) { lds[0] = a.b[0]; out[0] = (unsigned)lds[1]; }
}
template <unsigned N> static void run(hipStream_t s, unsigned * d, dim3 grid, size_t lds)
{
  Args<N> a = {};
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k<N>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k<N>, grid, dim3(64), lds, s, a, d);
  (void)hipStreamSynchronize(s);
  const int reps = 5000;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<N>, grid, dim3(64), lds, s, a, d);
  auto t1 = std::chrono::steady_clock::now();
  (void)hipStreamSynchronize(s);
  auto t2 = std::chrono::steady_clock::now();
  printf("args %4u B, grid %5u x %3u, LDS %6zu B: %6.2f us per launch call (host), %6.2f us incl. drain\n", N, grid.x, grid.y, lds,
         std::chrono::duration<double, std::micro>(t1 - t0).count() / reps,
         std::chrono::duration<double, std::micro>(t2 - t0).count() / reps);
}
int main()
{
  hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned * d; (void)hipMalloc(&d, 64);
  run<16>(s, d, dim3(1), 0); run<1024>(s, d, dim3(1), 0); run<3600>(s, d, dim3(1), 0);
  run<3600>(s, d, dim3(256, 8), 0); run<3600>(s, d, dim3(256, 8), 40 * 1024); run<3600>(s, d, dim3(256, 8), 160 * 1024);
  run<16>(s, d, dim3(256, 8), 160 * 1024); run<16>(s, d, dim3(2048, 14), 40 * 1024);
  return 0;
}
