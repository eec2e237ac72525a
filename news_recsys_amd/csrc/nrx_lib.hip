// Library-level entry points of libnrx_hip.so: ABI version, thread-local error text, device facts.
#include "nrx_common.h"

static thread_local char g_err[512] = "";

void nrx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int nrx_abi_version(void) { return NRX_ABI_VERSION; }

extern "C" const char* nrx_last_error(void) { return g_err; }

extern "C" int nrx_device_info(int device, int64_t info[6]) {
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) {
        nrx_set_error("nrx_device_info: %s", hipGetErrorString(e));
        return NRX_ERR_LAUNCH;
    }
    info[0] = p.multiProcessorCount;
    info[1] = p.warpSize;
    info[2] = p.clockRate;            // kHz
    info[3] = (int64_t)p.totalGlobalMem;
    info[4] = p.memoryClockRate;      // kHz
    info[5] = p.memoryBusWidth;       // bits
    return NRX_OK;
}
