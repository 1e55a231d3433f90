// Error reporting, device check and version for libaether_hip.so (host-side only).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "../../include/aether_hip.h"

static thread_local char g_last_error[512] = "";

extern "C" int aether_set_error(int code, const char* msg) {
    snprintf(g_last_error, sizeof(g_last_error), "%s", msg ? msg : "");
    return code;
}

extern "C" int aether_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_last_error, sizeof(g_last_error), "%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return AETHER_ERR_LAUNCH;
    }
    return AETHER_OK;
}

extern "C" const char* aether_last_error(void) { return g_last_error; }

extern "C" int aether_version(void) { return 100; }

extern "C" int aether_check_device(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return aether_set_error(AETHER_ERR_ARCH, "no HIP device");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return aether_set_error(AETHER_ERR_ARCH, "no device properties");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_last_error, sizeof(g_last_error), "device arch %s is not gfx950", prop.gcnArchName);
        return AETHER_ERR_ARCH;
    }
    return AETHER_OK;
}
