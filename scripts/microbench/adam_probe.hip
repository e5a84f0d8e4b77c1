// Micro-benchmark: the streaming pattern of the Adam update (4 arrays read, 3 written, one pass) on arrays far larger than the Infinity
// Cache (the HSTU catalog of the bench: 55.8 M parameters = 223 MB per array, where adam_segs_kernel runs at 1.4 TB/s), in a few
// variants: plain / non-temporal loads and stores, 1 - 4 float4 per thread in flight, chunked blocks vs a grid-stride loop.
//   hipcc --offload-arch=gfx950 -O3 adam_probe.hip -o _bin/adam_probe && _bin/adam_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void upd(f32x4& p, f32x4 g, f32x4& m, f32x4& v) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    m[j] = 0.9f * m[j] + 0.1f * g[j];
    v[j] = 0.999f * v[j] + 0.001f * g[j] * g[j];
    p[j] -= 1e-3f * (m[j] / (sqrtf(v[j]) * 1.01f + 1e-8f));
  }
}
template <bool NT> __device__ __forceinline__ f32x4 ld(const f32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(f32x4* p, f32x4 x) { if (NT) __builtin_nontemporal_store(x, p); else *p = x; }

// U float4 per thread, all loads first; block = 256 threads x U consecutive float4 tiles
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void adam_chunk(f32x4* __restrict__ p, const f32x4* __restrict__ g, f32x4* __restrict__ m, f32x4* __restrict__ v, long long n4) {
  const long long base = (long long)blockIdx.x * (256 * U) + threadIdx.x;
  f32x4 pp[U], gg[U], mm[U], vv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) { const long long i = base + u * 256; if (i < n4) { gg[u] = ld<NTL>(g + i); mm[u] = ld<NTL>(m + i); vv[u] = ld<NTL>(v + i); pp[u] = ld<NTL>(p + i); } }
#pragma unroll
  for (int u = 0; u < U; ++u) { const long long i = base + u * 256; if (i < n4) { upd(pp[u], gg[u], mm[u], vv[u]); st<NTS>(m + i, mm[u]); st<NTS>(v + i, vv[u]); st<NTS>(p + i, pp[u]); } }
}
// the product kernel's shape: loads and stores interleaved per float4
__global__ __launch_bounds__(256) void adam_now(f32x4* __restrict__ p, const f32x4* __restrict__ g, f32x4* __restrict__ m, f32x4* __restrict__ v, long long n4) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long long i = (long long)blockIdx.x * 1024 + u * 256 + threadIdx.x;
    if (i >= n4) break;
    f32x4 gg = g[i], mm = m[i], vv = v[i], pp = p[i];
    upd(pp, gg, mm, vv);
    m[i] = mm; v[i] = vv; p[i] = pp;
  }
}
template <bool NTS> __global__ __launch_bounds__(256) void fill_k(f32x4* __restrict__ p, long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n4) st<NTS>(p + i, f32x4{1.f, 2.f, 3.f, 4.f});
}
__global__ __launch_bounds__(256) void read_k(const f32x4* __restrict__ p, long long n4, float* out) {
  const long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
  f32x4 a = {0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < 4; ++u) if (i + u * 256 < n4) a += p[i + u * 256];
  if (a[0] + a[1] + a[2] + a[3] == 1.2345e30f) out[0] = 1.f;
}
template <bool NTS> __global__ __launch_bounds__(256) void copy_k(const f32x4* __restrict__ s, f32x4* __restrict__ d, long long n4) {
  const long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
  f32x4 a[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) if (i + u * 256 < n4) a[u] = s[i + u * 256];
#pragma unroll
  for (int u = 0; u < 4; ++u) if (i + u * 256 < n4) st<NTS>(d + i + u * 256, a[u]);
}

int main(int argc, char** argv) {
  const long long n = argc > 1 ? atoll(argv[1]) : 55836672LL;   // floats per array
  const long long n4 = n / 4;
  f32x4 *p, *g, *m, *v; float* out;
  CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&m, n * 4)); CK(hipMalloc(&v, n * 4)); CK(hipMalloc(&out, 4));
  CK(hipMemset(p, 0, n * 4)); CK(hipMemset(g, 0, n * 4)); CK(hipMemset(m, 0, n * 4)); CK(hipMemset(v, 0, n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, double bytes, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-44s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / ms / 1e9);
  };
  const double B = (double)n * 4;
  printf("n = %lld floats per array (%.0f MB)\n", n, B / 1e6);
  timeit("read 1 array", B, [&] { read_k<<<(n4 + 1023) / 1024, 256>>>(p, n4, out); });
  timeit("fill 1 array (plain stores)", B, [&] { fill_k<false><<<(n4 + 255) / 256, 256>>>(p, n4); });
  timeit("fill 1 array (nt stores)", B, [&] { fill_k<true><<<(n4 + 255) / 256, 256>>>(p, n4); });
  timeit("hipMemsetAsync 1 array", B, [&] { CK(hipMemsetAsync(p, 0, n * 4, 0)); });
  timeit("copy (plain stores)", 2 * B, [&] { copy_k<false><<<(n4 + 1023) / 1024, 256>>>(g, p, n4); });
  timeit("copy (nt stores)", 2 * B, [&] { copy_k<true><<<(n4 + 1023) / 1024, 256>>>(g, p, n4); });
  timeit("adam: product shape (interleaved)", 7 * B, [&] { adam_now<<<(n4 + 1023) / 1024, 256>>>(p, g, m, v, n4); });
  timeit("adam: U=1 loads first", 7 * B, [&] { adam_chunk<1, false, false><<<(n4 + 255) / 256, 256>>>(p, g, m, v, n4); });
  timeit("adam: U=2 loads first", 7 * B, [&] { adam_chunk<2, false, false><<<(n4 + 511) / 512, 256>>>(p, g, m, v, n4); });
  timeit("adam: U=4 loads first", 7 * B, [&] { adam_chunk<4, false, false><<<(n4 + 1023) / 1024, 256>>>(p, g, m, v, n4); });
  timeit("adam: U=2 nt stores", 7 * B, [&] { adam_chunk<2, false, true><<<(n4 + 511) / 512, 256>>>(p, g, m, v, n4); });
  timeit("adam: U=2 nt loads + nt stores", 7 * B, [&] { adam_chunk<2, true, true><<<(n4 + 511) / 512, 256>>>(p, g, m, v, n4); });
  timeit("adam: U=4 nt loads + nt stores", 7 * B, [&] { adam_chunk<4, true, true><<<(n4 + 1023) / 1024, 256>>>(p, g, m, v, n4); });
  timeit("adam: U=1 nt loads + nt stores", 7 * B, [&] { adam_chunk<1, true, true><<<(n4 + 255) / 256, 256>>>(p, g, m, v, n4); });
  return 0;
}
