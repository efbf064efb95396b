// Library-level entry points: version, last-error string, device probe.
#include "ppk_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void ppk_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

const char* ppk_version(void) { return "ppk 0.1.0 (gfx950)"; }

const char* ppk_last_error(void) { return g_err; }

// Number of visible HIP devices, or a negative PPK_ERR_HIP.  Lets the host
// side fail loudly when the library is loaded on a box without a GPU.
int ppk_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ppk_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return PPK_ERR_HIP;
    }
    return n;
}

}  // extern "C"
