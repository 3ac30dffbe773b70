// Probe: what HBM rate does the box sustain when a stream touches only PIECE bytes of every STRIDE bytes
// (the channel-sliced channels-last views K2 writes: 128-B S pieces / 512-B T pieces of a 1280 / 4224 / 3328-B pixel row)?
//   hipcc -O3 --offload-arch=gfx950 tools/pattern_probe.hip -o gpurun_out/pattern_probe && gpurun_out/pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// mode 0: write pieces; 1: read pieces; 2: dense read -> piece write (copy); 3: piece read -> dense write
template <int MODE>
__global__ __launch_bounds__(256) void piece_k(const f4* __restrict__ src, f4* __restrict__ dst, float* __restrict__ sink,
                                               size_t nquads, int qpp /* 16-B quads per piece */, int stride_q /* stride in quads */,
                                               int off_q) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nquads) return;
  const size_t px = i / qpp;
  const int q = (int)(i - px * qpp);
  const size_t strided = px * stride_q + off_q + q;
  if (MODE == 0) { dst[strided] = f4{1.f, 2.f, 3.f, 4.f}; }
  else if (MODE == 1) { f4 v = src[strided]; if (v.x == 1234.5f) sink[0] = v.y; }
  else if (MODE == 2) { dst[strided] = src[i]; }
  else { dst[i] = src[strided]; }
}

template <typename F>
static float time_us(F f, int iters) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) f();
  (void)hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0.f; (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / iters;
}

int main() {
  const size_t big = (size_t)4 << 30;       // strided side
  const size_t dense = (size_t)600 << 20;
  f4 *a, *b; float* sink;
  CK(hipMalloc(&a, big)); CK(hipMalloc(&b, dense)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 0, big)); CK(hipMemset(b, 0, dense));
  const int pieces[] = {128, 512};
  const int strides[] = {0 /* dense */, 256, 512, 1024, 2048, 4096, 640, 1280, 3328, 4224};
  printf("piece stride  moved_MB   write_us GB/s | read_us GB/s | dense->piece us GB/s(total) | piece->dense us GB/s(total)\n");
  for (int piece : pieces)
    for (int stride : strides) {
      const int st = stride == 0 ? piece : stride;
      if (st < piece) continue;
      size_t moved = (size_t)520 << 20;                    // bytes moved on the strided side
      size_t npx = moved / piece;
      if (npx * (size_t)st > big) npx = big / st;
      moved = npx * piece;
      const int qpp = piece / 16, sq = st / 16;
      const size_t nq = npx * qpp;
      const unsigned grid = (unsigned)((nq + 255) / 256);
      float t0 = time_us([&] { hipLaunchKernelGGL((piece_k<0>), dim3(grid), dim3(256), 0, 0, b, a, sink, nq, qpp, sq, 0); }, 10);
      float t1 = time_us([&] { hipLaunchKernelGGL((piece_k<1>), dim3(grid), dim3(256), 0, 0, a, b, sink, nq, qpp, sq, 0); }, 10);
      float t2 = time_us([&] { hipLaunchKernelGGL((piece_k<2>), dim3(grid), dim3(256), 0, 0, b, a, sink, nq, qpp, sq, 0); }, 10);
      float t3 = time_us([&] { hipLaunchKernelGGL((piece_k<3>), dim3(grid), dim3(256), 0, 0, a, b, sink, nq, qpp, sq, 0); }, 10);
      printf("%5d %6d  %8.1f  %8.1f %6.0f | %8.1f %6.0f | %8.1f %6.0f | %8.1f %6.0f\n", piece, st, moved / 1048576.0, t0, moved / t0 / 1e3, t1,
             moved / t1 / 1e3, t2, 2.0 * moved / t2 / 1e3, t3, 2.0 * moved / t3 / 1e3);
    }
  // two interleaved roles into one 1280-B row: 128-B piece at offset 0 and 512-B piece at offset 128 written by different launches
  return 0;
}
