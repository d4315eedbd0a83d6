// capi_core.cpp -- error plumbing + version/device queries of the C ABI (include/augcl.h).
#include <stdarg.h>

#include "augcl_common.h"

namespace augcl {

thread_local char g_err[512] = {0};

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(AUGCL_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return AUGCL_OK;
}

} // namespace augcl

extern "C" {

int augcl_abi_version(void) { return AUGCL_ABI_VERSION; }

const char *augcl_last_error(void) { return augcl::g_err; }

int augcl_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return augcl::fail(AUGCL_E_LAUNCH, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

} // extern "C"
