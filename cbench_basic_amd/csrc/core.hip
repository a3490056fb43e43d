// Error channel and device management of libbasic_hip.so.
#include "common.h"

#include <mutex>
#include <set>
#include <utility>

namespace basic {

static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    char buf[512];
    snprintf(buf, sizeof(buf), "HIP error %d (%s) at %s:%d in %s", static_cast<int>(e), hipGetErrorString(e), file, line, what);
    g_last_error = buf;
    (void)hipGetLastError();  // clear sticky state
    return BASIC_ERR_HIP;
}

hipError_t ensure_max_lds(const void *kernel)
{
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kernel, dev})) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done.insert({kernel, dev});
    return e;
}

int require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        set_error("no HIP device visible: libbasic_hip has no CPU fallback (gfx950 required)");
        return BASIC_ERR_NO_DEVICE;
    }
    return BASIC_OK;
}

}  // namespace basic

extern "C" const char *basic_last_error(void) { return basic::g_last_error.c_str(); }

extern "C" int basic_device_count(int *count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    if (count) *count = n;
    if (n <= 0) { basic::set_error("no HIP device visible"); return BASIC_ERR_NO_DEVICE; }
    return BASIC_OK;
}

extern "C" int basic_set_device(int ordinal)
{
    BASIC_HIP_TRY(hipSetDevice(ordinal));
    return BASIC_OK;
}

extern "C" int basic_stream_synchronize(void *hip_stream)
{
    BASIC_HIP_TRY(hipStreamSynchronize(basic::as_stream(hip_stream)));
    return BASIC_OK;
}
