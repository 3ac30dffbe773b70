// Probe (round 5): fp32 contractions on the bf16 matrix pipe of MI355X -- numerics and rate.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_split_mfma.hip -o gpurun_out/probe_split_mfma && gpurun_out/probe_split_mfma
// Part 1 (numerics): C[ch][px] = sum_k W[ch][k] X[px][k] on v_mfma_f32_16x16x32_bf16 with both operands cut into three bf16 planes
//   (8 + 8 + 8 significand bits = the fp32 value exactly; every product of two planes is exact in fp32), 6 / 8 / 9 of the nine
//   plane products, in one, two or three accumulators, against fp64 on the host -- beside the fp32 pipe's own error
//   (v_mfma_f32_16x16x4_f32 chain, what pw_tdiff16_kernel multiplies with).  Reported: max and rms error over max |ref| and the
//   backward-error constant c of |err| <= c 2^-24 sum_k |w x|.
// Part 2 (one instruction): what v_mfma_f32_16x16x32_bf16 does with 32 products of very different size -- is the K = 32 sum
//   formed wide and rounded once, or product by product?
// Part 3 (rate): bare loops of the bf16 MFMA on random / zero / split-plane operands, every CU busy: the clock the chip holds.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// three bf16 planes of eight fp32 values (truncating cut: hi = upper 16 bits, remainder exact, again, again)
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& h, u32x4& m, u32x4& l) {
  unsigned hh[8], mm[8], ll[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned x = __float_as_uint(v[i]) & 0xffff0000u;
    const float r1 = v[i] - __uint_as_float(x);
    const unsigned y = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(y);
    hh[i] = x; mm[i] = y; ll[i] = __float_as_uint(r2) & 0xffff0000u;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = (hh[2 * i] >> 16) | hh[2 * i + 1];
    m[i] = (mm[2 * i] >> 16) | mm[2 * i + 1];
    l[i] = (ll[2 * i] >> 16) | ll[2 * i + 1];
  }
}
// round-to-nearest cut (residuals signed, |residual| <= half an ulp of the plane above)
__device__ __forceinline__ unsigned rne16(float f) {
  const unsigned u = __float_as_uint(f);
  return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
}
__device__ __forceinline__ void split8_rne(const float (&v)[8], u32x4& h, u32x4& m, u32x4& l) {
  unsigned hh[8], mm[8], ll[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned x = rne16(v[i]);
    const float r1 = v[i] - __uint_as_float(x);
    const unsigned y = rne16(r1);
    const float r2 = r1 - __uint_as_float(y);
    hh[i] = x; mm[i] = y; ll[i] = rne16(r2);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = (hh[2 * i] >> 16) | hh[2 * i + 1];
    m[i] = (mm[2 * i] >> 16) | mm[2 * i + 1];
    l[i] = (ll[2 * i] >> 16) | ll[2 * i + 1];
  }
}

#define MF(c, a, b) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)

// MODE: 0 fp32 pipe (16x16x4 chain, k ascending in steps of 4)
//       1 nine products, one accumulator, smallest first      2 nine, one accumulator, largest first
//       3 six products (terms below 2^-24 dropped), one acc   4 eight (ll dropped), one acc
//       5 nine, two accumulators (hh | the rest)              6 six, two accumulators
//       7 nine, three accumulators (hh | hm mh | the rest)    8 three products of a two-plane cut (the round-2 "bf16x3" mode)
//       9 six, three accumulators                             10 nine, one acc, RNE cut     11 six, one acc, RNE cut
//       12 six: the three small ones summed from zero per 32-k step and added to the accumulator once per step, the three large ones
//          (w_m x_h, w_h x_m, w_h x_h) accumulated directly -- two chains of three per tile (the kernels' form since the chain-latency probe)
//       13 six, all summed from zero per 32-k step and added once per step (the kernels' first form)
template <int MODE>
__global__ __launch_bounds__(64) void contract(const float* __restrict__ W, const float* __restrict__ X, float* __restrict__ out, int K, int NPX) {
  const int lane = threadIdx.x, i16 = lane & 15, g = lane >> 4;
  const int ch0 = blockIdx.x * 16, px0 = blockIdx.y * 16;
  const float* wrow = W + (size_t)(ch0 + i16) * K;
  const float* xrow = X + (size_t)(px0 + i16) * K;
  f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0;
  if (MODE == 0) {
    for (int k = 0; k < K; k += 4) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wrow[k + g], xrow[k + g], c0, 0, 0, 0);
  } else {
    for (int k = 0; k < K; k += 32) {
      float wv[8], xv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { wv[e] = wrow[k + 8 * g + e]; xv[e] = xrow[k + 8 * g + e]; }
      u32x4 wh, wm, wl, xh, xm, xl;
      if (MODE == 10 || MODE == 11) { split8_rne(wv, wh, wm, wl); split8_rne(xv, xh, xm, xl); }
      else { split8(wv, wh, wm, wl); split8(xv, xh, xm, xl); }
      if (MODE == 8) {      // two planes: hi truncated, lo = bf16_rne(x - hi)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float w0 = wv[2 * q] - __uint_as_float(__float_as_uint(wv[2 * q]) & 0xffff0000u), w1 = wv[2 * q + 1] - __uint_as_float(__float_as_uint(wv[2 * q + 1]) & 0xffff0000u);
          const float x0 = xv[2 * q] - __uint_as_float(__float_as_uint(xv[2 * q]) & 0xffff0000u), x1 = xv[2 * q + 1] - __uint_as_float(__float_as_uint(xv[2 * q + 1]) & 0xffff0000u);
          wm[q] = (rne16(w0) >> 16) | rne16(w1);
          xm[q] = (rne16(x0) >> 16) | rne16(x1);
        }
        MF(c0, wh, xm); MF(c0, wm, xh); MF(c0, wh, xh);
      } else if (MODE == 1 || MODE == 10) {
        MF(c0, wl, xl); MF(c0, wl, xm); MF(c0, wm, xl); MF(c0, wl, xh); MF(c0, wh, xl); MF(c0, wm, xm); MF(c0, wm, xh); MF(c0, wh, xm); MF(c0, wh, xh);
      } else if (MODE == 2) {
        MF(c0, wh, xh); MF(c0, wh, xm); MF(c0, wm, xh); MF(c0, wm, xm); MF(c0, wh, xl); MF(c0, wl, xh); MF(c0, wm, xl); MF(c0, wl, xm); MF(c0, wl, xl);
      } else if (MODE == 3 || MODE == 11) {
        MF(c0, wl, xh); MF(c0, wh, xl); MF(c0, wm, xm); MF(c0, wm, xh); MF(c0, wh, xm); MF(c0, wh, xh);
      } else if (MODE == 4) {
        MF(c0, wl, xm); MF(c0, wm, xl); MF(c0, wl, xh); MF(c0, wh, xl); MF(c0, wm, xm); MF(c0, wm, xh); MF(c0, wh, xm); MF(c0, wh, xh);
      } else if (MODE == 5) {
        MF(c1, wl, xl); MF(c1, wl, xm); MF(c1, wm, xl); MF(c1, wl, xh); MF(c1, wh, xl); MF(c1, wm, xm); MF(c1, wm, xh); MF(c1, wh, xm); MF(c0, wh, xh);
      } else if (MODE == 6) {
        MF(c1, wl, xh); MF(c1, wh, xl); MF(c1, wm, xm); MF(c1, wm, xh); MF(c1, wh, xm); MF(c0, wh, xh);
      } else if (MODE == 7) {
        MF(c2, wl, xl); MF(c2, wl, xm); MF(c2, wm, xl); MF(c2, wl, xh); MF(c2, wh, xl); MF(c2, wm, xm); MF(c1, wm, xh); MF(c1, wh, xm); MF(c0, wh, xh);
      } else if (MODE == 9) {
        MF(c2, wl, xh); MF(c2, wh, xl); MF(c2, wm, xm); MF(c1, wm, xh); MF(c1, wh, xm); MF(c0, wh, xh);
      } else if (MODE == 12) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        MF(t, wl, xh); MF(t, wh, xl); MF(t, wm, xm);
        MF(c0, wm, xh); MF(c0, wh, xm); MF(c0, wh, xh);
        c0 = c0 + t;
      } else if (MODE == 13) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        MF(t, wl, xh); MF(t, wh, xl); MF(t, wm, xm); MF(t, wm, xh); MF(t, wh, xm); MF(t, wh, xh);
        c0 = c0 + t;
      }
    }
    c0 = c0 + (c1 + c2);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) out[(size_t)(ch0 + 4 * g + r) * NPX + px0 + i16] = c0[r];
}

// ---- part 2: one instruction --------------------------------------------------------------------------------------------------
__global__ void one_mfma(const unsigned short* __restrict__ a, const unsigned short* __restrict__ b, const float* __restrict__ cin, float* out) {
  // A row 0 / B column 0 carry the 32 test products (k = 8 g + e <-> lane 16 g, element e); everything else is zero
  const int lane = threadIdx.x, g = lane >> 4;
  bf16x8 av, bv;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const unsigned short ua = (lane & 15) == 0 ? a[8 * g + e] : (unsigned short)0, ub = (lane & 15) == 0 ? b[8 * g + e] : (unsigned short)0;
    av[e] = __builtin_bit_cast(__bf16, ua); bv[e] = __builtin_bit_cast(__bf16, ub);
  }
  f32x4 c = {lane == 0 ? cin[0] : 0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c, 0, 0, 0);
  if (lane == 0) out[0] = c[0];
}

// ---- part 3: rate ----------------------------------------------------------------------------------------------------------
// NP products per "k step" into NACC accumulator tiles of 16x16 (SHAPE 0) or 32x32 (SHAPE 1); operands stay in registers
template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void rate_kernel(const unsigned* __restrict__ seed, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  u32x4 a[3], b[3];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) { a[p][q] = seed[(p * 4 + q) * 64 + lane]; b[p][q] = seed[(12 + p * 4 + q) * 64 + lane]; }
  if (SHAPE == 0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int pa = 0; pa < 3; ++pa)
#pragma unroll
          for (int pb = 0; pb < 3; ++pb) MF(acc[t], a[pa], b[pb]);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int t = 1; t < NACC; ++t) s += acc[t];
    if (s[0] == 123.456f) out[threadIdx.x] = s[1];
  } else {
    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int pa = 0; pa < 3; ++pa)
#pragma unroll
          for (int pb = 0; pb < 3; ++pb)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[pa]), __builtin_bit_cast(bf16x8, b[pb]), acc[t], 0, 0, 0);
    }
    f32x16 s = acc[0];
#pragma unroll
    for (int t = 1; t < NACC; ++t) s += acc[t];
    if (s[0] == 123.456f) out[threadIdx.x] = s[1];
  }
}
// the fp32 pipe beside it (16x16x4), same loop shape
template <int NACC>
__global__ __launch_bounds__(256) void rate_f32_kernel(const unsigned* __restrict__ seed, float* out, int iters) {
  const int lane = threadIdx.x & 63;
  float a[8], b[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { a[q] = __uint_as_float((seed[q * 64 + lane] & 0x007fffffu) | 0x3f800000u); b[q] = __uint_as_float((seed[(8 + q) * 64 + lane] & 0x007fffffu) | 0x3f000000u); }
  f32x4 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[q], acc[t], 0, 0, 0);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int t = 1; t < NACC; ++t) s += acc[t];
  if (s[0] == 123.456f) out[threadIdx.x] = s[1];
}

static unsigned short bf16_of(double v) {     // exact for the constants used below
  float f = (float)v; unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16);
}

template <int MODE>
static int run_mode(const char* name, const float* dW, const float* dX, float* dO, int CH, int NPX, int K, const std::vector<double>& ref,
                    const std::vector<double>& mag, double refmax) {
  hipLaunchKernelGGL(contract<MODE>, dim3(CH / 16, NPX / 16), dim3(64), 0, 0, dW, dX, dO, K, NPX);
  CK(hipGetLastError());
  std::vector<float> o((size_t)CH * NPX);
  CK(hipMemcpy(o.data(), dO, o.size() * 4, hipMemcpyDeviceToHost));
  double emax = 0, e2 = 0, cmax = 0, c2 = 0;
  for (size_t i = 0; i < o.size(); ++i) {
    const double e = fabs((double)o[i] - ref[i]);
    emax = fmax(emax, e); e2 += e * e;
    const double c = e / (mag[i] * ldexp(1.0, -24));
    cmax = fmax(cmax, c); c2 += c * c;
  }
  printf("    %-46s max %.3e  rms %.3e  | c = err / (2^-24 sum|wx|): max %.3f rms %.4f\n", name, emax / refmax, sqrt(e2 / o.size()) / refmax, cmax, sqrt(c2 / o.size()));
  return 0;
}

int main() {
  // ---------------- part 1 ----------------
  const int CH = 160, NPX = 1792;
  std::mt19937_64 rng(12345);
  std::normal_distribution<double> nd(0.0, 1.0);
  std::uniform_real_distribution<double> ud(-1.0, 1.0);
  const char* kinds[] = {"synth relu(N(0,1))", "full_mantissa (x pi/3)", "heavy_tail expm1(1.151 relu z)", "cancellation (rows of w sum to 0, x const per pixel)"};
  for (int K : {256, 1024}) {
    for (int kind = 0; kind < 4; ++kind) {
      std::vector<float> W((size_t)CH * K), X((size_t)NPX * K);
      const double bound = 1.0 / sqrt((double)K);
      for (auto& w : W) w = (float)(ud(rng) * bound);
      if (kind == 3) {
        for (int c = 0; c < CH; ++c) { double s = 0; for (int k = 0; k < K; ++k) s += W[(size_t)c * K + k]; for (int k = 0; k < K; ++k) W[(size_t)c * K + k] = (float)(W[(size_t)c * K + k] - s / K); }
        for (int p = 0; p < NPX; ++p) { const float a = (float)(expm1(1.151 * fmax(nd(rng), 0.0)) + 0.5); for (int k = 0; k < K; ++k) X[(size_t)p * K + k] = a; }
      } else {
        for (auto& x : X) {
          const double z = fmax(nd(rng), 0.0);
          x = kind == 0 ? (float)(float)z : kind == 1 ? (float)((double)(float)z * (M_PI / 3.0)) : (float)expm1((double)(float)z * 1.151);
        }
      }
      std::vector<double> ref((size_t)CH * NPX), mag((size_t)CH * NPX);
      double refmax = 0;
      for (int c = 0; c < CH; ++c)
        for (int p = 0; p < NPX; ++p) {
          double s = 0, m = 0;
          for (int k = 0; k < K; ++k) { const double t = (double)W[(size_t)c * K + k] * (double)X[(size_t)p * K + k]; s += t; m += fabs(t); }
          ref[(size_t)c * NPX + p] = s; mag[(size_t)c * NPX + p] = fmax(m, 1e-30); refmax = fmax(refmax, fabs(s));
        }
      if (kind == 3) refmax = 1.0;     // exact result ~ 0: report absolute error
      float *dW, *dX, *dO;
      CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dO, ref.size() * 4));
      CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
      printf("K = %d, %s, max|ref| %.3f%s\n", K, kinds[kind], refmax, kind == 3 ? " (absolute errors)" : "");
      if (run_mode<0>("fp32 pipe (16x16x4 chain)", dW, dX, dO, CH, NPX, K, ref, mag, refmax)) return 1;
      if (run_mode<8>("two planes, 3 products (old bf16x3)", dW, dX, dO, CH, NPX, K, ref, mag, refmax)) return 1;
      if (run_mode<3>("6 products, 1 acc, small first", dW, dX, dO, CH, NPX, K, ref, mag, refmax)) return 1;
      if (run_mode<11>("6 products, 1 acc, small first, RNE cut", dW, dX, dO, CH, NPX, K, ref, mag, refmax)) return 1;
      if (run_mode<6>("6 products, 2 acc (hh | rest)", dW, dX, dO, CH, NPX, K, ref, mag, refmax)) return 1;
      if (run_mode<9>("6 products, 3 acc (hh | hm mh | rest)", dW, dX, dO, CH, NPX, K, ref, mag, refmax)) return 1;
      if (run_mode<13>("6 products from zero per step, one add per step", dW, dX, dO, CH, NPX, K, ref, mag, refmax)) return 1;
      if (run_mode<12>("3 small from zero + add per step, 3 large direct", dW, dX, dO, CH, NPX, K, ref, mag, refmax)) return 1;
      if (run_mode<4>("8 products, 1 acc, small first", dW, dX, dO, CH, NPX, K, ref, mag, refmax)) return 1;
      if (run_mode<1>("9 products, 1 acc, small first", dW, dX, dO, CH, NPX, K, ref, mag, refmax)) return 1;
      if (run_mode<10>("9 products, 1 acc, small first, RNE cut", dW, dX, dO, CH, NPX, K, ref, mag, refmax)) return 1;
      if (run_mode<2>("9 products, 1 acc, large first", dW, dX, dO, CH, NPX, K, ref, mag, refmax)) return 1;
      if (run_mode<5>("9 products, 2 acc (hh | rest)", dW, dX, dO, CH, NPX, K, ref, mag, refmax)) return 1;
      if (run_mode<7>("9 products, 3 acc (hh | hm mh | rest)", dW, dX, dO, CH, NPX, K, ref, mag, refmax)) return 1;
      CK(hipFree(dW)); CK(hipFree(dX)); CK(hipFree(dO));
    }
  }

  // ---------------- part 2 ----------------
  {
    unsigned short *da, *db; float *dc, *dout;
    CK(hipMalloc(&da, 64)); CK(hipMalloc(&db, 64)); CK(hipMalloc(&dc, 4)); CK(hipMalloc(&dout, 4));
    struct Case { const char* what; double a[32], b[32], c, exact; };
    std::vector<Case> cases;
    auto mk = [&](const char* what, double c) { Case z; memset(&z, 0, sizeof z); z.what = what; z.c = c; return z; };
    { Case z = mk("2^20 + 1.0078125 - 2^20 (exact 1.0078125; a per-product fp32 chain gives 1.0)", 0.0);
      z.a[0] = 1024; z.b[0] = 1024; z.a[1] = 1.0078125; z.b[1] = 1; z.a[2] = -1024; z.b[2] = 1024; z.exact = 1.0078125; cases.push_back(z); }
    { Case z = mk("2^30 + 1 - 2^30 (exact 1; needs > 30 bits inside)", 0.0);
      z.a[0] = 32768; z.b[0] = 32768; z.a[1] = 1; z.b[1] = 1; z.a[2] = -32768; z.b[2] = 32768; z.exact = 1; cases.push_back(z); }
    { Case z = mk("same, the big pair in k group 0, the 1 in k group 3 (k = 24)", 0.0);
      z.a[0] = 32768; z.b[0] = 32768; z.a[24] = 1; z.b[24] = 1; z.a[1] = -32768; z.b[1] = 32768; z.exact = 1; cases.push_back(z); }
    { Case z = mk("c = 2^24, products 32 x 0.5 (exact 2^24 + 16; one by one every 0.5 is lost)", 16777216.0);
      for (int k = 0; k < 32; ++k) { z.a[k] = 0.5; z.b[k] = 1; } z.exact = 16777216.0 + 16; cases.push_back(z); }
    { Case z = mk("c = 2^24, products 4 x 0.5 in k 0..3 (exact 2^24 + 2)", 16777216.0);
      for (int k = 0; k < 4; ++k) { z.a[k] = 0.5; z.b[k] = 1; } z.exact = 16777216.0 + 2; cases.push_back(z); }
    { Case z = mk("c = 2^24, products 0.5 at k = 0, 8, 16, 24 (one per lane group; exact 2^24 + 2)", 16777216.0);
      for (int k = 0; k < 32; k += 8) { z.a[k] = 0.5; z.b[k] = 1; } z.exact = 16777216.0 + 2; cases.push_back(z); }
    { Case z = mk("c = 1, products 2^-25 x 32 (exact 1 + 2^-20)", 1.0);
      for (int k = 0; k < 32; ++k) { z.a[k] = ldexp(1.0, -25); z.b[k] = 1; } z.exact = 1.0 + ldexp(1.0, -20); cases.push_back(z); }
    { Case z = mk("c = -1, product 1 + 2^-7 squared = 1 + 2^-6 + 2^-14 (exact 2^-6 + 2^-14: is the product kept exact?)", -1.0);
      z.a[0] = 1.0078125; z.b[0] = 1.0078125; z.exact = ldexp(1.0, -6) + ldexp(1.0, -14); cases.push_back(z); }
    { Case z = mk("subnormal product: 2^-70 x 2^-70 with c = 0 (exact 2^-140, an fp32 subnormal)", 0.0);
      z.a[0] = ldexp(1.0, -70); z.b[0] = ldexp(1.0, -70); z.exact = ldexp(1.0, -140); cases.push_back(z); }
    { Case z = mk("subnormal bf16 input: 2^-130 x 2^10 (exact 2^-120)", 0.0);
      z.a[0] = ldexp(1.0, -130); z.b[0] = 1024; z.exact = ldexp(1.0, -120); cases.push_back(z); }
    printf("\none v_mfma_f32_16x16x32_bf16:\n");
    for (auto& z : cases) {
      unsigned short ha[32], hb[32];
      for (int k = 0; k < 32; ++k) { ha[k] = bf16_of(z.a[k]); hb[k] = bf16_of(z.b[k]); }
      float c = (float)z.c;
      CK(hipMemcpy(da, ha, 64, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb, 64, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, &c, 4, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, da, db, dc, dout);
      float r; CK(hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost));
      printf("    %-100s -> %.10g (exact %.10g)\n", z.what, (double)r, z.exact);
    }
  }

  // ---------------- part 3 ----------------
  {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("\nrate: %d CUs, clock %d kHz\n", ncu, prop.clockRate);
    unsigned* dseed; float* dout;
    CK(hipMalloc(&dseed, 24 * 64 * 4)); CK(hipMalloc(&dout, 4096));
    std::vector<unsigned> seed(24 * 64);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int data = 0; data < 3; ++data) {
      // 0: zeros; 1: random bf16 bit patterns of sane exponent (all planes alike); 2: split planes of random fp32 (hi / mid / lo differ by 2^-8 each)
      for (int i = 0; i < 24 * 64; ++i) {
        if (data == 0) seed[i] = 0;
        else {
          auto one = [&](int plane) { unsigned m = (unsigned)(rng() & 0x7f), s = (unsigned)(rng() & 1), e = 127 - (data == 2 ? 8 * plane : 0) - (unsigned)(rng() & 3); return (s << 15) | (e << 7) | m; };
          const int plane = (i / 64 / 4) % 3;
          seed[i] = one(plane) | (one(plane) << 16);
        }
      }
      CK(hipMemcpy(dseed, seed.data(), seed.size() * 4, hipMemcpyHostToDevice));
      const char* dn[] = {"zeros", "random bf16", "split planes"};
      for (int wpc = 1; wpc <= 2; ++wpc) {       // blocks of 256 threads per CU: 1 or 2 waves per SIMD
        const int iters = 20000;
        auto time_it = [&](auto kern, double flop_per_iter_wave, const char* nm) {
          hipLaunchKernelGGL(kern, dim3(ncu * wpc), dim3(256), 0, 0, dseed, dout, 200);
          (void)hipDeviceSynchronize();
          float best = 1e30f;
          for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(ncu * wpc), dim3(256), 0, 0, dseed, dout, iters);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = fminf(best, ms);
          }
          const double tf = flop_per_iter_wave * iters * 4.0 * ncu * wpc / (best * 1e-3) / 1e12;
          printf("    %-14s %-34s %d wave(s) / SIMD: %8.3f ms  %8.1f TF\n", dn[data], nm, wpc, best, tf);
        };
        time_it(rate_kernel<0, 4>, 4 * 9 * 16.0 * 16 * 32 * 2, "16x16x32 bf16, 4 acc x 9 products");
        time_it(rate_kernel<1, 2>, 2 * 9 * 32.0 * 32 * 16 * 2, "32x32x16 bf16, 2 acc x 9 products");
        if (data == 1) time_it(rate_f32_kernel<4>, 4 * 8 * 16.0 * 16 * 4 * 2, "16x16x4 f32, 4 acc x 8 k steps");
      }
    }
  }
  return 0;
}
