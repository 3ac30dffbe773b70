// Probe (round 5): what does a vector-memory instruction cost a wave that is issuing MFMAs -- the question behind the split-fp32 units
// kernel, whose K loop slowed from 3.7 k to 5.8 k cycles per step when its 13 loads per wave and step were switched on.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_vmem_issue.hip -o tools/_ab/probe_vmem_issue && tools/_ab/probe_vmem_issue
// Every CU runs two blocks of four waves (two waves per SIMD, as the kernel).  A "unit" = twelve v_mfma_f32_16x16x32_bf16 in two
// alternating chains (192 cycles of the pipe per wave, 384 for the two waves of a SIMD) + NLOAD buffer_load_dwordx4 (1 KB per wave
// and instruction) behind MFMA 5 (and 11, 2, 8).  Variants: the loads' target (registers / LDS-DMA), their source (a 16-KB region
// per block that stays in L1 / L2, or a stream), a block-wide barrier every 11 units (the kernel's K-tile), all waves at the same
// place or each wave of a block behind a different MFMA (uniform branch inside the asm statement).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// NLOAD loads per unit; TGT 0 registers, 1 LDS-DMA; STREAM 0 cached region, 1 streaming; BAR barrier every 11 units; STAG 0 same place, 1 per-wave place
template <int NLOAD, int TGT, int STREAM, int BAR, int STAG, int LW = 4, int ASYM = 0>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, float* out, int units, unsigned long long* cyc, long long region_bytes) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  u32x4 a[3], b[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) { a[q] = u32x4{0x3f803f80u + lane, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u}; b[q] = u32x4{0x3f803f80u, 0x3f003f00u + q, 0x3e803e80u, 0x3f803f80u}; }
  f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0, acc = t0;
  u32x4 ld[4] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
  const unsigned long long sa = reinterpret_cast<unsigned long long>(src);
  const i32x4 desc = {(int)(unsigned)sa, (int)(unsigned)(sa >> 32) & 0xffff, (int)region_bytes, 0x00020000};
  const int voff = lane * 16;
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds) + wave * 1024;
  // cached: 16 KB per block, re-read; stream: every (block, unit, load) its own KB
  // (32-bit scalar arithmetic only: a 64-bit modulo here cost ~150 cycles per load and WAS the first version's "load cost")
  unsigned pos = STREAM ? blockIdx.x * 4096u + wave * 1024u : (blockIdx.x & 255u) * 16384u + wave * 1024u;
  const unsigned cbase = (blockIdx.x & 255u) * 16384u + wave * 1024u, sstride = gridDim.x * 4096u;
  auto mf = [&](f32x4 c, const u32x4& x, const u32x4& y) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0);
  };
  auto load = [&](const int i, const int who) {
    if (ASYM && (blockIdx.x & 1)) return;       // ASYM: only every other block loads (block ids alternate over the XCDs: both kinds meet on a CU)
    const int so = (int)pos;
    pos = STREAM ? (pos + sstride) & 0x7fffefffu : cbase + ((pos + 4096u) & 12288u);
    if (LW == 1) { asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "+v"(ld[i][0]) : "v"(voff), "s"(desc), "s"(so) : "memory"); return; }
    if (LW == 2) { asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "+v"(*reinterpret_cast<unsigned long long*>(&ld[i])) : "v"(voff), "s"(desc), "s"(so) : "memory"); return; }
    if (TGT == 1) {
      if (who < 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_base), "v"(voff), "s"(desc), "s"(so) : "memory", "m0");
      else asm volatile("s_cmp_lg_u32 %4, %5\n\ts_cbranch_scc1 .Lsk%=\n\ts_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n.Lsk%=:"
                        :: "s"(lds_base), "v"(voff), "s"(desc), "s"(so), "s"(wave), "s"(who) : "memory", "m0", "scc");
    } else {
      if (who < 0) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(ld[i]) : "v"(voff), "s"(desc), "s"(so) : "memory");
      else asm volatile("s_cmp_lg_u32 %4, %5\n\ts_cbranch_scc1 .Lsk%=\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen\n.Lsk%=:"
                        : "+v"(ld[i]) : "v"(voff), "s"(desc), "s"(so), "s"(wave), "s"(who) : "memory", "scc");
    }
  };
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int u = 0; u < units; ++u) {
#define SB __builtin_amdgcn_sched_barrier(0)
#define SLOT(n)                                                                                              \
    if (STAG == 0) {                                                                                         \
      if (n == 5 && NLOAD >= 1) { load(0, -1); SB; }                                                         \
      if (n == 11 && NLOAD >= 2) { load(1, -1); SB; }                                                        \
      if (n == 2 && NLOAD >= 4) { load(2, -1); SB; }                                                         \
      if (n == 8 && NLOAD >= 4) { load(3, -1); SB; }                                                         \
    } else if (n % 3 == 2) {                                                                                 \
      if (NLOAD >= 1) load(0, n / 3);                                                                        \
      if (NLOAD >= 2) load(1, n / 3);                                                                        \
      if (NLOAD >= 4) { load(2, n / 3); load(3, n / 3); }                                                    \
      SB;                                                                                                    \
    }
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));      // (not loop-invariant for the compiler)
    t0 = mf(z, a[2], b[0]);  SB; SLOT(0)
    t1 = mf(z, a[1], b[0]);  SB; SLOT(1)
    t0 = mf(t0, a[0], b[2]); SB; SLOT(2)
    t1 = mf(t1, a[1], b[2]); SB; SLOT(3)
    t0 = mf(t0, a[1], b[1]); SB; SLOT(4)
    t1 = mf(t1, a[2], b[1]); SB; SLOT(5)
    t0 = mf(t0, a[1], b[0]); SB; SLOT(6)
    t1 = mf(t1, a[0], b[0]); SB; SLOT(7)
    t0 = mf(t0, a[0], b[1]); SB; SLOT(8)
    t1 = mf(t1, a[2], b[1]); SB; SLOT(9)
    t0 = mf(t0, a[0], b[0]); SB; SLOT(10)
    t1 = mf(t1, a[2], b[0]); SB; SLOT(11)
#undef SLOT
    acc += t0 + t1;
    asm volatile("" : "+v"(acc));
    if (BAR && u % 11 == 10) {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(ld[0]), "+v"(ld[1]), "+v"(ld[2]), "+v"(ld[3]) :: "memory");
      __syncthreads();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(ld[0]), "+v"(ld[1]), "+v"(ld[2]), "+v"(ld[3]) :: "memory");
  const unsigned long long c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) atomicAdd(cyc, c1 - c0);
  if (acc[0] == 123.25f && ld[0][0] + ld[1][1] + ld[2][2] + ld[3][3] == 77u) out[threadIdx.x] = acc[1];
}

template <int NLOAD, int TGT, int STREAM, int BAR, int STAG, int LW = 4, int ASYM = 0, int BPC = 2>
static int run(const char* what, const float* src, float* out, unsigned long long* cyc, long long region, int ncu) {
  const int units = 11 * 200, grid = BPC * ncu;
  auto kern = k<NLOAD, TGT, STREAM, BAR, STAG, LW, ASYM>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 65536, 0, src, out, 110, cyc, region);
  CK(hipDeviceSynchronize());
  CK(hipMemset(cyc, 0, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 65536, 0, src, out, units, cyc, region);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double per_unit = (double)c / grid / units;
  printf("  %-64s %7.1f cycles / unit (wave 0 of a block)   %8.3f ms   clock %.2f GHz\n", what, per_unit, ms, (double)c / grid / (ms * 1e-3) / 1e9);
  return 0;
}


// Producer / consumer: one block of EIGHT waves per CU.  Waves 0..3 only multiply (four chains: a dependent v_mfma_f32_16x16x32_bf16 is
// ~44 cycles away, two chains leave the pipe idle with one wave per SIMD); waves 4..7 stream NLD KB per "unit" from HBM into registers,
// cut nothing, and write 1.5 KB per KB to LDS (the plane images), throttled by s_waitcnt vmcnt(2 NLD).  BAR: one block barrier per 11 units.
template <int NLD, int BAR>
__global__ __launch_bounds__(512, 1) void kpc(const float* __restrict__ src, float* out, int units, unsigned long long* cyc, long long region_bytes) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned long long c0 = __builtin_readcyclecounter();
  if (wave < 4) {
    u32x4 a[3], b[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { a[q] = u32x4{0x3f803f80u + lane, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u}; b[q] = u32x4{0x3f803f80u, 0x3f003f00u + q, 0x3e803e80u, 0x3f803f80u}; }
    f32x4 t[4], acc = {0.f, 0.f, 0.f, 0.f};
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    auto mf = [&](f32x4 c, const u32x4& x, const u32x4& y) {
      return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0);
    };
    for (int u = 0; u < units; u += 2) {      // two units = 24 MFMAs in four chains
      asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));
#pragma unroll
      for (int c = 0; c < 4; ++c) { t[c] = mf(z, a[2], b[c % 3]); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int c = 0; c < 4; ++c) { t[c] = mf(t[c], a[s % 3], b[(s + c) % 3]); __builtin_amdgcn_sched_barrier(0); }
      acc += (t[0] + t[1]) + (t[2] + t[3]);
      asm volatile("" : "+v"(acc));
      if (BAR && u % 22 == 20) __syncthreads();
    }
    if (acc[0] == 123.25f) out[threadIdx.x] = acc[1];
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) atomicAdd(cyc, c1 - c0);
  } else {
    u32x4 ld[2][2] = {{u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}}, {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}}};
    const unsigned long long sa = reinterpret_cast<unsigned long long>(src);
    const i32x4 desc = {(int)(unsigned)sa, (int)(unsigned)(sa >> 32) & 0xffff, (int)region_bytes, 0x00020000};
    const int voff = lane * 16;
    unsigned pos = blockIdx.x * 4096u + (wave - 4) * 1024u;
    const unsigned sstride = gridDim.x * 4096u;
    char* dst = lds + (wave - 4) * 8192 + lane * 16;
    unsigned long long moved = 0;
    for (int u = 0; u < units; u += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (NLD >= 1) { asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(ld[h][0]) : "v"(voff), "s"(desc), "s"((int)pos) : "memory"); pos = (pos + sstride) & 0x7fffefffu; }
        if (NLD >= 2) { asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(ld[h][1]) : "v"(voff), "s"(desc), "s"((int)pos) : "memory"); pos = (pos + sstride) & 0x7fffefffu; }
        // the other half's loads (issued one unit ago) have landed: "cut" and write them
        if (NLD == 1) asm volatile("s_waitcnt vmcnt(1)" : "+v"(ld[h ^ 1][0]), "+v"(ld[h ^ 1][1]) :: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" : "+v"(ld[h ^ 1][0]), "+v"(ld[h ^ 1][1]) :: "memory");
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
          const u32x4 v = ld[h ^ 1][i];
          *reinterpret_cast<u32x4*>(dst + i * 2048) = u32x4{v.x & 0xffff0000u, v.y & 0xffff0000u, v.z, v.w};
          *reinterpret_cast<unsigned long long*>(dst + 1024 + i * 2048) = ((unsigned long long)v.x << 32) | v.y;
        }
        moved += NLD;
      }
      if (BAR && u % 22 == 20) __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ld[0][0]), "+v"(ld[0][1]), "+v"(ld[1][0]), "+v"(ld[1][1]) :: "memory");
    const unsigned long long c1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 255) == 0) { atomicAdd(cyc + 1, c1 - c0); atomicAdd(cyc + 2, moved); }
    if (ld[0][0][0] + ld[1][1][1] == 77u) out[threadIdx.x] = 1.f;
  }
}
template <int NLD, int BAR>
static int run_pc(const char* what, const float* src, float* out, unsigned long long* cyc, long long region, int ncu) {
  const int units = 22 * 100, grid = ncu;
  auto kern = kpc<NLD, BAR>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 65536, 0, src, out, 220, cyc, region);
  CK(hipDeviceSynchronize());
  CK(hipMemset(cyc, 0, 24));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 65536, 0, src, out, units, cyc, region);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long c[3]; CK(hipMemcpy(c, cyc, 24, hipMemcpyDeviceToHost));
  printf("  %-56s consumer %6.1f cycles / unit, producer loop %6.1f cycles / unit, %8.3f ms, stream %.2f TB/s\n", what,
         (double)c[0] / grid / units, (double)c[1] / grid / units, ms, (double)c[2] * 1024.0 / (ms * 1e-3) / 1e12);
  return 0;
}

// chains: one wave per SIMD, NCH independent accumulation chains round-robin, 6 MFMAs per chain per round (in-place accumulate)
template <int NCH>
__global__ __launch_bounds__(256, 1) void kchains(float* out, int rounds, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63;
  u32x4 a[3], b[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) { a[q] = u32x4{0x3f803f80u + lane, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u}; b[q] = u32x4{0x3f803f80u, 0x3f003f00u + q, 0x3e803e80u, 0x3f803f80u}; }
  f32x4 t[NCH], acc = {0.f, 0.f, 0.f, 0.f};
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int r = 0; r < rounds; ++r) {
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));
#pragma unroll
    for (int c = 0; c < NCH; ++c) { t[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[2]), __builtin_bit_cast(bf16x8, b[c % 3]), z, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
      for (int c = 0; c < NCH; ++c) { t[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[s % 3]), __builtin_bit_cast(bf16x8, b[(s + c) % 3]), t[c], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc += t[c];
    asm volatile("" : "+v"(acc));
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) atomicAdd(cyc, c1 - c0);
  if (acc[0] == 123.25f) out[threadIdx.x] = acc[1];
}
template <int NCH>
static int run_ch(float* out, unsigned long long* cyc, int ncu) {
  const int rounds = 2000;
  CK(hipMemset(cyc, 0, 8));
  hipLaunchKernelGGL(kchains<NCH>, dim3(ncu), dim3(256), 0, 0, out, rounds, cyc);
  CK(hipDeviceSynchronize());
  unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  printf("  %d chains x 6 MFMAs per round, one wave per SIMD: %6.2f cycles per MFMA (the adds of the %d scratch tiles at the end of a round included)\n", NCH,
         (double)c / ncu / rounds / (6.0 * NCH), NCH);
  return 0;
}

// the same with side work behind every MFMA: SIDE bit 0: two v_add_f32 on other registers; bit 1: one ds_read_b128 behind every third MFMA
// (operands of later MFMAs: consumed two rounds later); bit 2: the B operands come from those reads (rotating registers)
template <int NCH, int SIDE>
__global__ __launch_bounds__(256, 1) void kside(float* out, int rounds, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0x3f803f80u + (i & 7);
  __syncthreads();
  u32x4 a[3], b[3], xb[2][3];
#pragma unroll
  for (int q = 0; q < 3; ++q) { a[q] = u32x4{0x3f803f80u + lane, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u}; b[q] = u32x4{0x3f803f80u, 0x3f003f00u + q, 0x3e803e80u, 0x3f803f80u}; xb[0][q] = b[q]; xb[1][q] = b[q]; }
  f32x4 t[NCH], acc = {0.f, 0.f, 0.f, 0.f}, junk[4] = {acc, acc, acc, acc};
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const char* rd = lds + lane * 16;
  const unsigned long long c0 = __builtin_readcyclecounter();
  for (int r = 0; r < rounds; ++r) {
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]));
    const int par = r & 1;
    int n = 0;
#pragma unroll
    for (int s = 0; s < 6; ++s)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const u32x4& bb = (SIDE & 4) ? xb[0][(s + c) % 3] : b[(s + c) % 3];
        t[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[s % 3]), __builtin_bit_cast(bf16x8, bb), s == 0 ? z : t[c], 0, 0, 0);
        if (SIDE & 1) { junk[n & 3].x += junk[(n + 1) & 3].y; junk[n & 3].z += junk[(n + 1) & 3].w; asm volatile("" : "+v"(junk[n & 3])); }
        if ((SIDE & 2) && n % 3 == 0 && n / 3 < 3) xb[1][n / 3] = *reinterpret_cast<const u32x4*>(rd + (n / 3) * 1024 + par * 4096);
        __builtin_amdgcn_sched_barrier(0);
        ++n;
      }
    if (SIDE & 2) {
#pragma unroll
      for (int q = 0; q < 3; ++q) { const u32x4 tmp = xb[0][q]; xb[0][q] = xb[1][q]; xb[1][q] = tmp; }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) acc += t[c];
    asm volatile("" : "+v"(acc));
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) atomicAdd(cyc, c1 - c0);
  if (acc[0] + junk[0].x + junk[1].x + junk[2].x + junk[3].x == 123.25f) out[threadIdx.x] = acc[1];
}
template <int NCH, int SIDE>
static int run_side(float* out, unsigned long long* cyc, int ncu) {
  const int rounds = 2000;
  CK(hipMemset(cyc, 0, 8));
  hipLaunchKernelGGL((kside<NCH, SIDE>), dim3(ncu), dim3(256), 0, 0, out, rounds, cyc);
  CK(hipDeviceSynchronize());
  unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  printf("  %d chains, side work %d (1: two adds per MFMA, 2: an LDS read per three MFMAs, 4: operands from those reads): %6.2f cycles per MFMA\n", NCH, SIDE,
         (double)c / ncu / rounds / (6.0 * NCH));
  return 0;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  const long long region = 1ll << 31;
  float *src, *out; unsigned long long* cyc;
  CK(hipMalloc(&src, region)); CK(hipMemset(src, 0, region)); CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 8));
  printf("two blocks of four waves per CU; unit = 12 MFMAs per wave (384 pipe cycles per SIMD for its two waves)\n");
  if (run<0, 0, 0, 0, 0>("no loads", src, out, cyc, region, ncu)) return 1;
  if (run<0, 0, 0, 1, 0>("no loads, barrier every 11 units", src, out, cyc, region, ncu)) return 1;
  if (run<1, 0, 0, 0, 0>("1 load / unit, registers, cached", src, out, cyc, region, ncu)) return 1;
  if (run<2, 0, 0, 0, 0>("2 loads / unit, registers, cached", src, out, cyc, region, ncu)) return 1;
  if (run<4, 0, 0, 0, 0>("4 loads / unit, registers, cached", src, out, cyc, region, ncu)) return 1;
  if (run<1, 0, 0, 1, 0>("1 load / unit, registers, cached, barrier / 11", src, out, cyc, region, ncu)) return 1;
  if (run<2, 0, 0, 1, 0>("2 loads / unit, registers, cached, barrier / 11", src, out, cyc, region, ncu)) return 1;
  if (run<1, 0, 0, 1, 1>("1 load / unit, registers, cached, barrier / 11, staggered", src, out, cyc, region, ncu)) return 1;
  if (run<2, 0, 0, 1, 1>("2 loads / unit, registers, cached, barrier / 11, staggered", src, out, cyc, region, ncu)) return 1;
  if (run<1, 1, 0, 1, 0>("1 load / unit, LDS-DMA, cached, barrier / 11", src, out, cyc, region, ncu)) return 1;
  if (run<2, 1, 0, 1, 0>("2 loads / unit, LDS-DMA, cached, barrier / 11", src, out, cyc, region, ncu)) return 1;
  if (run<1, 0, 1, 1, 0>("1 load / unit, registers, stream, barrier / 11", src, out, cyc, region, ncu)) return 1;
  if (run<2, 0, 1, 1, 0>("2 loads / unit, registers, stream, barrier / 11", src, out, cyc, region, ncu)) return 1;
  if (run<1, 1, 1, 1, 0>("1 load / unit, LDS-DMA, stream, barrier / 11", src, out, cyc, region, ncu)) return 1;
  if (run<2, 0, 1, 0, 0>("2 loads / unit, registers, stream, no barrier", src, out, cyc, region, ncu)) return 1;
  if (run<2, 0, 0, 0, 0, 2>("2 loads / unit of 8 B per lane (512 B), cached", src, out, cyc, region, ncu)) return 1;
  if (run<2, 0, 0, 0, 0, 1>("2 loads / unit of 4 B per lane (256 B), cached", src, out, cyc, region, ncu)) return 1;
  if (run<4, 0, 0, 0, 0, 1>("4 loads / unit of 4 B per lane (256 B), cached", src, out, cyc, region, ncu)) return 1;
  if (run<2, 0, 0, 0, 0, 4, 1>("2 loads / unit, cached, only every other block loads", src, out, cyc, region, ncu)) return 1;
  if (run<4, 0, 0, 0, 0, 4, 1>("4 loads / unit, cached, only every other block loads", src, out, cyc, region, ncu)) return 1;
  printf("one block of four waves per CU (one wave per SIMD):\n");
  if (run<0, 0, 0, 0, 0, 4, 0, 1>("no loads", src, out, cyc, region, ncu)) return 1;
  if (run<2, 0, 0, 0, 0, 4, 0, 1>("2 loads / unit, cached", src, out, cyc, region, ncu)) return 1;
  if (run<4, 0, 0, 0, 0, 4, 0, 1>("4 loads / unit, cached", src, out, cyc, region, ncu)) return 1;
  unsigned long long* cyc3; CK(hipMalloc(&cyc3, 24));
  printf("producer / consumer, one block of eight waves per CU (consumer waves: four MFMA chains, 192 pipe cycles per unit):\n");
  if (run_pc<0, 0>("no producer loads", src, out, cyc3, region, ncu)) return 1;
  if (run_pc<1, 0>("producers stream 1 KB / wave / unit", src, out, cyc3, region, ncu)) return 1;
  if (run_pc<2, 0>("producers stream 2 KB / wave / unit", src, out, cyc3, region, ncu)) return 1;
  if (run_pc<1, 1>("producers stream 1 KB / wave / unit, barrier / 22 units", src, out, cyc3, region, ncu)) return 1;
  if (run_pc<2, 1>("producers stream 2 KB / wave / unit, barrier / 22 units", src, out, cyc3, region, ncu)) return 1;
  printf("dependent chains of v_mfma_f32_16x16x32_bf16:\n");
  if (run_ch<1>(out, cyc3, ncu) || run_ch<2>(out, cyc3, ncu) || run_ch<3>(out, cyc3, ncu) || run_ch<4>(out, cyc3, ncu) || run_ch<6>(out, cyc3, ncu)) return 1;
  if (run_side<6, 0>(out, cyc3, ncu) || run_side<6, 1>(out, cyc3, ncu) || run_side<6, 2>(out, cyc3, ncu) || run_side<6, 3>(out, cyc3, ncu) || run_side<6, 7>(out, cyc3, ncu)) return 1;
  if (run_side<3, 0>(out, cyc3, ncu) || run_side<3, 3>(out, cyc3, ncu)) return 1;
  return 0;
}
