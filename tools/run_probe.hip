// Probe for the K1b (weight-gradient GEMM) X-operand access pattern on MI355X: a block streams a
// [128 channel rows][RUN bytes] tile per step out of an NCHW map (row pitch = HW*4 bytes), the next step
// takes the next RUN bytes of the same rows.  Question: how does the sustained HBM rate depend on RUN?
//   hipcc -O3 --offload-arch=gfx950 tools/run_probe.hip -o gpurun_out/run_probe && gpurun_out/run_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// LPT = float4 loads per thread per step = 128 rows * RUN / 16 / 256
template <int RUN>
__global__ __launch_bounds__(256, 2) void tile_read(const float* __restrict__ x, float* __restrict__ out, int C, int HW,
                                                    int frames_per_blk, int nslab) {
  extern __shared__ float pad[];   // occupancy control only
  constexpr int LPT = RUN / 32;
  constexpr int TPR = RUN / 16;            // threads per row
  constexpr int RPS = 256 / TPR;           // rows per load step
  const int slab = blockIdx.x % nslab, chunk = blockIdx.x / nslab;
  const int tid = threadIdx.x;
  const int r0 = tid / TPR, kq = tid % TPR;
  const int steps = (HW * 4 + RUN - 1) / RUN;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  f4 cur[LPT], nxt[LPT];
  auto load = [&](int frame, int step, f4 (&v)[LPT]) {
    const size_t base = ((size_t)frame * C + slab * 128) * HW;
    const int k = step * (RUN / 4) + 4 * kq;
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
      v[j] = f4{0.f, 0.f, 0.f, 0.f};
      if (k < HW) v[j] = *reinterpret_cast<const f4*>(x + base + (size_t)(r0 + RPS * j) * HW + k);
    }
  };
  int frame = chunk * frames_per_blk, step = 0;
  const int total = frames_per_blk * steps;
  load(frame, step, cur);
  for (int it = 0; it < total; ++it) {
    int nf = frame, ns = step + 1;
    if (ns == steps) { ns = 0; ++nf; }
    if (it + 1 < total) load(nf, ns, nxt);
#pragma unroll
    for (int j = 0; j < LPT; ++j) acc += cur[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < LPT; ++j) cur[j] = nxt[j];
    frame = nf; step = ns;
  }
  if (acc.x + acc.y + acc.z + acc.w == 1234.5f) out[0] = acc.x + pad[0];
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
  struct Case { int C, HW; } cases[] = {{256, 784}, {576, 196}};
  for (auto cs : cases) {
    const int frames = 448, C = cs.C, HW = cs.HW;
    const size_t n = (size_t)frames * C * HW;
    float *x, *out;
    CK(hipMalloc(&x, n * 4 + 4096)); CK(hipMalloc(&out, 4));
    CK(hipMemset(x, 0, n * 4 + 4096));
    const int nslab = C / 128;      // full slabs only
    const double bytes = (double)frames * nslab * 128 * HW * 4;
    for (int fpb : {4, 1}) {
      for (int lds_kb : {60, 30}) {
        const int grid = frames / fpb * nslab;
        const size_t lds = (size_t)lds_kb << 10;
        float t;
#define RUNCASE(R)                                                                                                   \
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(tile_read<R>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10)); \
        t = time_us([&] { hipLaunchKernelGGL((tile_read<R>), dim3(grid), dim3(256), lds, 0, x, out, C, HW, fpb, nslab); }, 10);    \
        printf("C %4d HW %3d frames/blk %d lds %2d KB  run %4d B  %8.1f us  %7.1f GB/s\n", C, HW, fpb, lds_kb, R, t, bytes / t / 1e3);
        RUNCASE(128) RUNCASE(256) RUNCASE(512) RUNCASE(1024)
      }
    }
    CK(hipFree(x)); CK(hipFree(out));
  }
  return 0;
}
