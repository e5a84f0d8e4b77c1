// Micro-benchmark: achievable HBM read bandwidth on MI355X for the catalog access patterns the top-k
// scorer can use (10 GB catalog, 2 KB rows).  Build: hipcc --offload-arch=gfx950 -O3 hbm_patterns.hip -o hbm_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// P0: contiguous grid-stride float4 read
__global__ void p0_linear(const f32x4* __restrict__ src, long long n4, float* out) {
  f32x4 acc = {0, 0, 0, 0};
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  long long stride = (long long)gridDim.x * blockDim.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    f32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    acc += a + b + c + d;
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1;
}

// P1: column slabs — block of ROWS rows x row_floats; per chunk each row contributes `cw` bytes
//     (cw = 128: today's kernel; 256, 512, 1024: wider chunks). 256 threads; lanes per row = cw/16.
template <int CW>
__global__ void p1_slab(const float* __restrict__ src, long long n_rows, int row_floats, int rotate, float* out) {
  constexpr int LPR = CW / 16;            // lanes per row piece
  constexpr int RPI = 256 / LPR;          // rows per 256-thread instruction
  constexpr int ROWS = 128;
  constexpr int NI = ROWS / RPI;          // instructions per chunk per thread
  const int n_chunks = row_floats * 4 / CW;
  const long long n_blocks = n_rows / ROWS;
  f32x4 acc = {0, 0, 0, 0};
  const int rot = rotate ? (blockIdx.x * 5) % n_chunks : 0;
  for (long long b = blockIdx.x; b < n_blocks; b += gridDim.x) {
    const float* base = src + b * ROWS * (long long)row_floats;
    for (int c = 0; c < n_chunks; ++c) {
      int cc = c + rot; if (cc >= n_chunks) cc -= n_chunks;
      f32x4 v[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        int r = threadIdx.x / LPR + RPI * j;
        v[j] = *reinterpret_cast<const f32x4*>(base + (long long)r * row_floats + cc * (CW / 4) + (threadIdx.x % LPR) * 4);
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) acc += v[j];
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1;
}

int main(int argc, char** argv) {
  const long long n_rows = 5000000; const int row_floats = 512;
  const size_t bytes = (size_t)n_rows * row_floats * 4;
  float* d; float* out;
  CK(hipMalloc(&d, bytes)); CK(hipMalloc(&out, 4));
  CK(hipMemset(d, 0, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int it = 0; it < 3; ++it) {
      CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-44s %8.3f ms  %7.1f GB/s\n", name, best, bytes / (best * 1e-3) / 1e9);
  };
  for (int g : {512, 1024, 2048}) {
    char nm[128];
    snprintf(nm, 128, "P0 linear float4, grid %d", g);
    timeit(nm, [&] { p0_linear<<<g, 256>>>((const f32x4*)d, (long long)(bytes / 16), out); });
  }
  for (int g : {512, 1024, 2048}) {
    for (int rot : {0, 1}) {
      char nm[128];
      snprintf(nm, 128, "P1 slab cw=128 rot=%d grid %d", rot, g);
      timeit(nm, [&] { p1_slab<128><<<g, 256>>>(d, n_rows, row_floats, rot, out); });
      snprintf(nm, 128, "P1 slab cw=256 rot=%d grid %d", rot, g);
      timeit(nm, [&] { p1_slab<256><<<g, 256>>>(d, n_rows, row_floats, rot, out); });
      snprintf(nm, 128, "P1 slab cw=512 rot=%d grid %d", rot, g);
      timeit(nm, [&] { p1_slab<512><<<g, 256>>>(d, n_rows, row_floats, rot, out); });
      snprintf(nm, 128, "P1 slab cw=1024 rot=%d grid %d", rot, g);
      timeit(nm, [&] { p1_slab<1024><<<g, 256>>>(d, n_rows, row_floats, rot, out); });
    }
  }
  return 0;
}
