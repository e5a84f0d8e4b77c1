// Measurement shim (NOT product code): LD_PRELOAD this library and every kernel launch / async memset of the process — this package's
// and torch's alike — can be switched into a no-op that only counts.  With the launches elided the GPU has nothing to do, so the wall
// time of `loop.step()` is the HOST's cost of issuing a step (Python + autograd + ctypes + the HIP runtime's argument marshalling up to
// the point where the packet would be written), which `host_issue_ms_per_step` could not tell: that counter follows the GPU whenever the
// launch queue exerts back-pressure.
//
//   g++ -O2 -shared -fPIC -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ dry_launch.cpp -o libdry_launch.so -ldl
//   LD_PRELOAD=.../libdry_launch.so python bench.py --workload train --host-only-child
//
// rt_dry_set(1) turns the elision on, rt_dry_set(0) off; rt_dry_count() returns the launches + memsets seen while it was on.
// Event records, stream waits, memcpys and allocations stay real (they are part of the host's cost and finish at once on an idle GPU).
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <atomic>

namespace {
std::atomic<int> g_on{0};
std::atomic<long long> g_launches{0}, g_memsets{0};
// The runtime is a dependency of torch's extension module, loaded into a LOCAL scope: RTLD_NEXT from a preloaded library does not
// see it.  Ask the already loaded copy by its soname instead (RTLD_NOLOAD: never load a second runtime).
void* resolve(const char* name) {
  void* p = dlsym(RTLD_NEXT, name);
  if (p != nullptr) return p;
  for (const char* so : {"libamdhip64.so.7", "libamdhip64.so", "libamdhip64.so.6"}) {
    void* h = dlopen(so, RTLD_NOLOAD | RTLD_LAZY);
    if (h != nullptr && (p = dlsym(h, name)) != nullptr) return p;
  }
  return nullptr;
}
template <typename F>
F next(const char* name) {
  return reinterpret_cast<F>(resolve(name));
}
}  // namespace

extern "C" {

void rt_dry_set(int on) { g_on.store(on); if (on) { g_launches.store(0); g_memsets.store(0); } }
int rt_dry_selftest() {   // 1 when the real entry points can be reached (call after the HIP runtime is loaded)
  return resolve("hipLaunchKernel") != nullptr && resolve("hipMemsetAsync") != nullptr && resolve("hipModuleLaunchKernel") != nullptr;
}
long long rt_dry_count(int what) { return what == 0 ? g_launches.load() : g_memsets.load(); }

hipError_t hipLaunchKernel(const void* f, dim3 grid, dim3 block, void** args, size_t shmem, hipStream_t stream) {
  if (g_on.load(std::memory_order_relaxed)) { g_launches.fetch_add(1, std::memory_order_relaxed); return hipSuccess; }
  static auto real = next<hipError_t (*)(const void*, dim3, dim3, void**, size_t, hipStream_t)>("hipLaunchKernel");
  return real(f, grid, block, args, shmem, stream);
}

hipError_t hipExtLaunchKernel(const void* f, dim3 grid, dim3 block, void** args, size_t shmem, hipStream_t stream, hipEvent_t e0,
                              hipEvent_t e1, int flags) {
  if (g_on.load(std::memory_order_relaxed)) { g_launches.fetch_add(1, std::memory_order_relaxed); return hipSuccess; }
  static auto real =
      next<hipError_t (*)(const void*, dim3, dim3, void**, size_t, hipStream_t, hipEvent_t, hipEvent_t, int)>("hipExtLaunchKernel");
  return real(f, grid, block, args, shmem, stream, e0, e1, flags);
}

hipError_t hipModuleLaunchKernel(hipFunction_t f, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz,
                                 unsigned shmem, hipStream_t stream, void** params, void** extra) {
  if (g_on.load(std::memory_order_relaxed)) { g_launches.fetch_add(1, std::memory_order_relaxed); return hipSuccess; }
  static auto real = next<hipError_t (*)(hipFunction_t, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, hipStream_t,
                                         void**, void**)>("hipModuleLaunchKernel");
  return real(f, gx, gy, gz, bx, by, bz, shmem, stream, params, extra);
}

hipError_t hipMemsetAsync(void* dst, int value, size_t bytes, hipStream_t stream) {
  if (g_on.load(std::memory_order_relaxed)) { g_memsets.fetch_add(1, std::memory_order_relaxed); return hipSuccess; }
  static auto real = next<hipError_t (*)(void*, int, size_t, hipStream_t)>("hipMemsetAsync");
  return real(dst, value, bytes, stream);
}

}  // extern "C"
