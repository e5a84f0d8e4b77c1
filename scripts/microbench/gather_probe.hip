// Micro-benchmark: how fast can a CU gather 1 KB rows (d = 256 fp32) of a table at random — the access pattern of the sampled losses
// (rt_loss.hip: 1.7 M row gathers per C2 step) — through (a) the register path the kernels use today (global_load_dwordx4, one row per
// quarter-wave, PF rows in flight per quarter) and (b) the LDS-DMA path (global_load_lds_dwordx4: one instruction moves one whole row
// into the LDS, the wave then reads its slice back with ds_read_b128)?  Tables of 27 MB (the C2 catalog: served by the Infinity Cache),
// 3.4 MB (an eighth: fits one XCD's L2) and 1 GB (HBM).
//   hipcc --offload-arch=gfx950 -O3 gather_probe.hip -o _bin/gather_probe && _bin/gather_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int D = 256;        // floats per row
constexpr int NC = 132;       // candidates per position (multiple of 4)

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x; }

// (a) register path: wave per position, quarter-wave per row, PF iterations (4 rows each) in flight
template <int PF>
__global__ __launch_bounds__(256) void gather_regs(const float* __restrict__ table, int V, int M, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & 15, grp = lane >> 4;
  const int m = blockIdx.x * 4 + wave;
  if (m >= M) return;
  f32x4 acc = {0, 0, 0, 0};
  f32x4 ev[PF][4];
  auto row_of = [&](int it) { return table + (size_t)(hash32((unsigned)(m * NC + it * 4 + grp)) % (unsigned)V) * D; };
  auto issue = [&](int it, f32x4 (&e)[4]) {
    const float* r = row_of(it);
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(e[i]) : "v"(r + (sub + 16 * i) * 4) : "memory");
  };
#pragma unroll
  for (int u = 0; u < PF - 1; ++u) issue(u, ev[u]);
  constexpr int NIT = NC / 4;
#pragma unroll 1
  for (int it0 = 0; it0 < NIT; it0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      issue(it0 + u + PF - 1, ev[(u + PF - 1) % PF]);
      asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ev[u][0]), "+v"(ev[u][1]), "+v"(ev[u][2]), "+v"(ev[u][3]) : "n"((PF - 1) * 4) : "memory");
#pragma unroll
      for (int i = 0; i < 4; ++i) acc += ev[u][i];
    }
  }
#pragma unroll
  for (int u = 0; u < PF; ++u) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ev[u][0]), "+v"(ev[u][1]), "+v"(ev[u][2]), "+v"(ev[u][3])::"memory");
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1;
}

// (b) LDS-DMA path: wave per position, one instruction per row (64 lanes x 16 B = the row), NS rows in flight per wave
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int NS>   // rows in flight per wave (ring slots of 1 KB)
__global__ __launch_bounds__(256) void gather_dma(const float* __restrict__ table, int V, int M, float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m = blockIdx.x * 4 + wave;
  if (m >= M) return;
  unsigned char* ring = smem + wave * NS * 1024;
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)ring);
  f32x4 acc = {0, 0, 0, 0};
  auto issue = [&](int j, int slot) {
    const float* r = table + (size_t)(hash32((unsigned)(m * NC + j)) % (unsigned)V) * D;
    dma16(r + lane * 4, base + (unsigned)(slot * 1024));
  };
#pragma unroll
  for (int u = 0; u < NS - 1; ++u) issue(u, u);
  int slot = 0;
#pragma unroll 1
  for (int j = 0; j < NC; ++j) {
    int nslot = slot + NS - 1; if (nslot >= NS) nslot -= NS;
    issue(j + NS - 1, nslot);                       // (reads a valid row past the end: weightless look-ahead)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS - 1) : "memory");
    acc += *reinterpret_cast<const f32x4*>(ring + slot * 1024 + lane * 16);
    if (++slot == NS) slot = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1;
}

int main() {
  const int M = 13312;
  float* out; CK(hipMalloc(&out, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int Vs[3] = {26744, 3343, 1000000};
  for (int vi = 0; vi < 3; ++vi) {
    const int V = Vs[vi];
    float* table; CK(hipMalloc(&table, (size_t)V * D * 4)); CK(hipMemset(table, 0, (size_t)V * D * 4));
    const double gb = (double)M * NC * D * 4 / 1e9;
    auto timeit = [&](const char* name, auto launch) {
      for (int i = 0; i < 3; ++i) launch();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < 10; ++i) launch();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
      printf("V=%8d (%7.1f MB)  %-28s %8.1f us  %7.2f TB/s  %5.1f B/clk/CU (2.4 GHz)\n", V, V * 1024.0 / 1e6, name, ms * 1e3, gb / ms, gb / ms * 1e12 / 256 / 2.4e9 / 1e3 * 1e3 / 1e3);
    };
    const int blocks = (M + 3) / 4;
    timeit("regs PF=2", [&] { gather_regs<2><<<blocks, 256>>>(table, V, M, out); });
    timeit("regs PF=3", [&] { gather_regs<3><<<blocks, 256>>>(table, V, M, out); });
    timeit("regs PF=4", [&] { gather_regs<4><<<blocks, 256>>>(table, V, M, out); });
    auto dma = [&](auto kern, int ns) {
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * ns * 1024));
      kern<<<blocks, 256, 4 * ns * 1024>>>(table, V, M, out);
    };
    timeit("lds-dma 4 rows in flight", [&] { dma(&gather_dma<4>, 4); });
    timeit("lds-dma 8 rows in flight", [&] { dma(&gather_dma<8>, 8); });
    timeit("lds-dma 12 rows in flight", [&] { dma(&gather_dma<12>, 12); });
    timeit("lds-dma 16 rows in flight", [&] { dma(&gather_dma<16>, 16); });
    CK(hipFree(table));
  }
  return 0;
}
