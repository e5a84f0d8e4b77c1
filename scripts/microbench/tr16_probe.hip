// Probe of ds_read_b64_tr_b16 on gfx950: which (source lane, source element) does result element j of lane l come from?
// Every lane supplies the address of its own 8-byte chunk (4 bf16); chunks are disjoint and hold their own element index, so the
// gather pattern can be read off the result.  Build: hipcc --offload-arch=gfx950 -O2 tr16_probe.hip -o _bin/tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const int* addr_elems, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  __attribute__((address_space(3))) s16x4* p = (__attribute__((address_space(3))) s16x4*)(lds + addr_elems[threadIdx.x]);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
  int* d_addr; unsigned short* d_out;
  hipMalloc(&d_addr, 64 * sizeof(int)); hipMalloc(&d_out, 256 * sizeof(unsigned short));
  for (int variant = 0; variant < 3; ++variant) {
    std::vector<int> addr(64);
    for (int l = 0; l < 64; ++l) {
      if (variant == 0) addr[l] = l * 8;                                         // disjoint chunks, lane-linear
      if (variant == 1) addr[l] = (l >> 4) * 64 + (l & 15) * 4;                   // the canonical contiguous [4][16] block per 16-lane group
      if (variant == 2) addr[l] = (l >> 4) * 1024 + ((l & 15) >> 2) * 72 + (l & 3) * 4;   // [4 rows][16 cols] with a row stride of 72 elements
    }
    hipMemcpy(d_addr, addr.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d_addr, d_out);
    std::vector<unsigned short> out(256);
    hipMemcpy(out.data(), d_out, 256 * sizeof(unsigned short), hipMemcpyDeviceToHost);
    printf("variant %d\n", variant);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d (addr %4d):", l, addr[l]);
      for (int j = 0; j < 4; ++j) {
        int v = out[l * 4 + j], src = -1, se = -1;
        for (int s = 0; s < 64; ++s) if (v >= addr[s] && v < addr[s] + 4) { src = s; se = v - addr[s]; }
        printf("  [%d]=%4d<-(lane %2d,e%d)", j, v, src, se);
      }
      printf("\n");
    }
  }
  return 0;
}
