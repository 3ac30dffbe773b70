// Probe (round 5): does a VALU write to the data registers of a wide buffer store, issued in the instruction right behind the store,
// reach memory?  The question behind wino_gemm_split.hip's one-wrong-item-in-1500: hipcc had re-used the FIRST data register of a
// buffer_store_dwordx4 (soffset in an SGPR) for the next address in the very next instruction, and lanes 12-15 of every sixteen stored
// the new value.  LLVM's hazard recogniser inserts the wait state only for wide stores WITHOUT an SGPR offset.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_store_hazard.hip -o tools/_ab/probe_store_hazard && tools/_ab/probe_store_hazard
// Every lane stores 16 bytes of a known pattern (its global lane index in all four dwords) with one asm statement:
//     buffer_store_dwordx{2,3,4} v[data], voff, desc, soff offen      (soff in an SGPR / soff = 0 as an inline constant)
//     [s_nop N]
//     v_mov_b32 data[R], 0xdeadbeef                                   (R = which data register is overwritten)
// and the host counts the dwords that read 0xdeadbeef.  NLOADS buffer loads in front of the store keep the wave's memory queue busy
// (the store's data read is what is late when the path is busy).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// W: dwords per store (2, 3, 4); R: data register overwritten; NOP: -1 none, else s_nop NOP; SOFF: 1 = soffset in an SGPR, 0 = constant 0
template <int W, int R, int NOP, int SOFF>
__global__ __launch_bounds__(256) void k(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int rounds, unsigned src_bytes, unsigned dst_bytes) {
  const unsigned gl = blockIdx.x * 256u + threadIdx.x;
  const unsigned long long sa = reinterpret_cast<unsigned long long>(src), da = reinterpret_cast<unsigned long long>(dst);
  const i32x4 sdesc = {(int)(unsigned)sa, (int)(unsigned)(sa >> 32) & 0xffff, (int)src_bytes, 0x00020000};
  const i32x4 ddesc = {(int)(unsigned)da, (int)(unsigned)(da >> 32) & 0xffff, (int)dst_bytes, 0x00020000};
  const int lane16 = (threadIdx.x & 63) * 16;
  const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * 256u + threadIdx.x) >> 6));
  u32x4 sink = {0u, 0u, 0u, 0u};
  for (int r = 0; r < rounds; ++r) {
    // traffic in front of the store: four 1-KB loads of a streaming region
    u32x4 ld[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int so = (int)(((unsigned)(wave * 4 + i) * 1024u + (unsigned)r * 4096u * 4096u) % (src_bytes - 1024u)) & ~1023;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ld[i]) : "v"(lane16), "s"(sdesc), "s"(so) : "memory");
    }
    const int soff = __builtin_amdgcn_readfirstlane((wave * rounds + r) * 1024);      // this wave's KB of this round
    const int voff = SOFF ? lane16 : lane16 + soff;
    // the store's four data registers are fixed (v100 .. v103) so that the overwrite can name one of them
    unsigned d0 = gl, d1 = gl, d2 = gl, d3 = gl;
    asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
#define BODY(STORE, NOPS, OVER)                                                                                                     \
    asm volatile("v_mov_b32 v100, %0\n\tv_mov_b32 v101, %1\n\tv_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\ts_nop 4\n\t"               \
                 STORE "\n\t" NOPS OVER "\n\ts_nop 4"                                                                               \
                 :: "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(voff), "s"(ddesc), "s"(soff) : "memory", "v100", "v101", "v102", "v103");
#define ST4S "buffer_store_dwordx4 v[100:103], %4, %5, %6 offen"
#define ST4C "buffer_store_dwordx4 v[100:103], %4, %5, 0 offen"
#define ST3S "buffer_store_dwordx3 v[100:102], %4, %5, %6 offen"
#define ST2S "buffer_store_dwordx2 v[100:101], %4, %5, %6 offen"
#define OV(r) "v_mov_b32 v10" #r ", 0xdeadbeef"
    if constexpr (W == 4 && SOFF == 1 && NOP < 0 && R == 0) { BODY(ST4S, "", OV(0)) }
    else if constexpr (W == 4 && SOFF == 1 && NOP < 0 && R == 1) { BODY(ST4S, "", OV(1)) }
    else if constexpr (W == 4 && SOFF == 1 && NOP < 0 && R == 2) { BODY(ST4S, "", OV(2)) }
    else if constexpr (W == 4 && SOFF == 1 && NOP < 0 && R == 3) { BODY(ST4S, "", OV(3)) }
    else if constexpr (W == 4 && SOFF == 1 && NOP == 0 && R == 0) { BODY(ST4S, "s_nop 0\n\t", OV(0)) }
    else if constexpr (W == 4 && SOFF == 1 && NOP == 1 && R == 0) { BODY(ST4S, "s_nop 1\n\t", OV(0)) }
    else if constexpr (W == 4 && SOFF == 0 && NOP < 0 && R == 0) { BODY(ST4C, "", OV(0)) }
    else if constexpr (W == 4 && SOFF == 0 && NOP == 0 && R == 0) { BODY(ST4C, "s_nop 0\n\t", OV(0)) }
    else if constexpr (W == 3 && SOFF == 1 && NOP < 0 && R == 0) { BODY(ST3S, "", OV(0)) }
    else if constexpr (W == 2 && SOFF == 1 && NOP < 0 && R == 0) { BODY(ST2S, "", OV(0)) }
#pragma unroll
    for (int i = 0; i < 4; ++i) sink += ld[i];
  }
  if (sink.x == 0x12345678u && sink.y == 0x9abcdef0u) dst[0] = sink.z + sink.w;      // keeps the loads alive, never true
}

template <int W, int R, int NOP, int SOFF>
static int run(const char* what, const unsigned* src, unsigned src_bytes, unsigned* dst, unsigned dst_bytes, int blocks, int rounds) {
  CK(hipMemset(dst, 0, dst_bytes));
  hipLaunchKernelGGL((k<W, R, NOP, SOFF>), dim3(blocks), dim3(256), 0, 0, src, dst, rounds, src_bytes, dst_bytes);
  CK(hipDeviceSynchronize());
  std::vector<unsigned> h(dst_bytes / 4);
  CK(hipMemcpy(h.data(), dst, dst_bytes, hipMemcpyDeviceToHost));
  long long bad = 0, wrong = 0, stores = (long long)blocks * 4 * rounds * 64;
  long long by_lane[64] = {0}, by_dword[4] = {0};
  for (long long s = 0; s < stores; ++s) {
    const int lane = (int)(s & 63);
    for (int q = 0; q < W; ++q) {
      const unsigned v = h[(size_t)s * 4 + q];
      if (v == 0xdeadbeefu) { ++bad; ++by_lane[lane]; ++by_dword[q]; }
      else if (v != (unsigned)((s >> 6) / rounds * 64 + lane)) ++wrong;
    }
  }
  printf("%-74s %10lld lane-stores: %8lld dwords read 0xdeadbeef (%.4f %%), %lld other mismatches", what, stores, bad, 100.0 * bad / stores, wrong);
  if (bad) {
    printf("; by dword %lld %lld %lld %lld; lanes:", by_dword[0], by_dword[1], by_dword[2], by_dword[3]);
    for (int l = 0; l < 64; ++l) if (by_lane[l]) printf(" %d", l);
  }
  printf("\n");
  return 0;
}

int main() {
  const int blocks = 256 * 8, rounds = 32;
  const unsigned src_bytes = 256u << 20, dst_bytes = (unsigned)((size_t)blocks * 4 * rounds * 1024);
  unsigned *src = nullptr, *dst = nullptr;
  CK(hipMalloc(reinterpret_cast<void**>(&src), src_bytes));
  CK(hipMalloc(reinterpret_cast<void**>(&dst), dst_bytes));
  CK(hipMemset(src, 1, src_bytes));
  printf("store-data hazard probe: %d blocks x 4 waves x %d rounds, one wide store per lane and round, the data register overwritten behind it\n", blocks, rounds);
  for (int rep = 0; rep < 2; ++rep) {
    if (run<4, 0, -1, 1>("dwordx4, soffset in an SGPR, v_mov to data[0] in the NEXT instruction", src, src_bytes, dst, dst_bytes, blocks, rounds)) return 1;
    if (run<4, 1, -1, 1>("dwordx4, SGPR soffset, v_mov to data[1] in the next instruction", src, src_bytes, dst, dst_bytes, blocks, rounds)) return 1;
    if (run<4, 2, -1, 1>("dwordx4, SGPR soffset, v_mov to data[2] in the next instruction", src, src_bytes, dst, dst_bytes, blocks, rounds)) return 1;
    if (run<4, 3, -1, 1>("dwordx4, SGPR soffset, v_mov to data[3] in the next instruction", src, src_bytes, dst, dst_bytes, blocks, rounds)) return 1;
    if (run<4, 0, 0, 1>("dwordx4, SGPR soffset, s_nop 0 in between", src, src_bytes, dst, dst_bytes, blocks, rounds)) return 1;
    if (run<4, 0, 1, 1>("dwordx4, SGPR soffset, s_nop 1 in between", src, src_bytes, dst, dst_bytes, blocks, rounds)) return 1;
    if (run<4, 0, -1, 0>("dwordx4, soffset = 0 (constant), v_mov to data[0] in the next instruction", src, src_bytes, dst, dst_bytes, blocks, rounds)) return 1;
    if (run<4, 0, 0, 0>("dwordx4, soffset = 0 (constant), s_nop 0 in between", src, src_bytes, dst, dst_bytes, blocks, rounds)) return 1;
    if (run<3, 0, -1, 1>("dwordx3, SGPR soffset, v_mov to data[0] in the next instruction", src, src_bytes, dst, dst_bytes, blocks, rounds)) return 1;
    if (run<2, 0, -1, 1>("dwordx2, SGPR soffset, v_mov to data[0] in the next instruction", src, src_bytes, dst, dst_bytes, blocks, rounds)) return 1;
  }
  (void)hipFree(src); (void)hipFree(dst);
  return 0;
}
