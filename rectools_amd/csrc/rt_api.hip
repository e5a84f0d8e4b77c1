// Library introspection entry points (include/rectools_hip.h).
#include <stdio.h>
#include <string.h>

#include "rt_common.h"

static thread_local char g_last_error[512] = "";

void rt_set_last_error(const char* file, int line, hipError_t e) {
  const char* base = strrchr(file, '/');
  snprintf(g_last_error, sizeof(g_last_error), "%s:%d: %s (%d)", base ? base + 1 : file, line,
           hipGetErrorString(e), (int)e);
}

extern "C" {
int rt_version(void) { return 100; }  // 0.1.0
int rt_device_cu_count(void) { return rt_num_cus(); }
const char* rt_last_error(void) { return g_last_error; }
}
