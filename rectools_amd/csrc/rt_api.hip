// Library introspection entry points (include/rectools_hip.h).
#include "rt_common.h"

extern "C" {
int rt_version(void) { return 100; }  // 0.1.0
int rt_device_cu_count(void) { return rt_num_cus(); }
}
