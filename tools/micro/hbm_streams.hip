// Practical HBM roof for the access mixes of the partials kernels: pure write,
// 1 read + 1 write, 2 reads + 1 write, with 16-byte-per-lane accesses, grid-stride.
// build: hipcc --offload-arch=gfx950 -O3 -o hbm_streams tools/micro/hbm_streams.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>   // 0: write, 1: copy, 2: a*b -> c, 3: read-only sum of 2
__global__ __launch_bounds__(256) void k(const double2 * a, const double2 * b, double2 * c, size_t n, double * sink)
{
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
  {
    if (MODE == 0) c[i] = make_double2(1.0, 2.0);
    if (MODE == 1) c[i] = a[i];
    if (MODE == 2) { const double2 x = a[i], y = b[i]; c[i] = make_double2(x.x * y.x, x.y * y.y); }
    if (MODE == 3) { const double2 x = a[i], y = b[i]; acc += x.x * y.x + x.y * y.y; }
  }
  if (MODE == 3 && acc == 12345.678) *sink = acc;
}

// same mixes, but every workgroup owns ONE contiguous range of the buffers (a wave walks its
// share in 1 KiB steps) instead of the grid-stride interleaving above
template <int MODE>
__global__ __launch_bounds__(256) void kr(const double2 * a, const double2 * b, double2 * c, size_t n, double * sink)
{
  const size_t per = (n + gridDim.x - 1) / gridDim.x;
  const size_t beg = per * blockIdx.x, end = beg + per < n ? beg + per : n;
  double acc = 0.0;
  for (size_t i = beg + threadIdx.x; i < end; i += 256)
  {
    if (MODE == 0) c[i] = make_double2(1.0, 2.0);
    if (MODE == 1) c[i] = a[i];
    if (MODE == 2) { const double2 x = a[i], y = b[i]; c[i] = make_double2(x.x * y.x, x.y * y.y); }
    if (MODE == 3) { const double2 x = a[i], y = b[i]; acc += x.x * y.x + x.y * y.y; }
  }
  if (MODE == 3 && acc == 12345.678) *sink = acc;
}

template <int MODE>
static void run_ranges(const char * name, double2 * a, double2 * b, double2 * c, size_t n, double bytes_per_elem, int grid, double * sink)
{
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kr<MODE>, dim3(grid), dim3(256), 0, 0, a, b, c, n, sink);
  CHECK(hipEventRecord(e0));
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kr<MODE>, dim3(grid), dim3(256), 0, 0, a, b, c, n, sink);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-22s grid %6d  %8.1f us  %7.1f GB/s  (contiguous range per workgroup)\n", name, grid, ms / reps * 1e3, bytes_per_elem * n / (ms / reps * 1e-3) / 1e9);
}

template <int MODE>
static void run(const char * name, double2 * a, double2 * b, double2 * c, size_t n, double bytes_per_elem, int grid, double * sink)
{
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, a, b, c, n, sink);
  CHECK(hipEventRecord(e0));
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, a, b, c, n, sink);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-22s grid %6d  %8.1f us  %7.1f GB/s\n", name, grid, ms / reps * 1e3, bytes_per_elem * n / (ms / reps * 1e-3) / 1e9);
}

int main(int argc, char ** argv)
{
  const size_t mb = argc > 1 ? atoi(argv[1]) : 640;
  const size_t n = mb * 1000 * 1000 / 16;
  double2 * a, * b, * c; double * sink;
  CHECK(hipMalloc(&a, n * 16)); CHECK(hipMalloc(&b, n * 16)); CHECK(hipMalloc(&c, n * 16)); CHECK(hipMalloc(&sink, 8));
  CHECK(hipMemset(a, 0, n * 16)); CHECK(hipMemset(b, 0, n * 16));
  printf("buffers of %zu MB\n", mb);
  for (int grid : {1024, 2048, 4096, 16384, 65536})
  {
    run<0>("write", a, b, c, n, 16, grid, sink);
    run<1>("1 read + 1 write", a, b, c, n, 32, grid, sink);
    run<2>("2 reads + 1 write", a, b, c, n, 48, grid, sink);
    run<3>("2 reads", a, b, c, n, 32, grid, sink);
  }
  for (int grid : {256, 512, 1024, 4096})
  {
    run_ranges<0>("write", a, b, c, n, 16, grid, sink);
    run_ranges<1>("1 read + 1 write", a, b, c, n, 32, grid, sink);
    run_ranges<2>("2 reads + 1 write", a, b, c, n, 48, grid, sink);
  }
  return 0;
}
