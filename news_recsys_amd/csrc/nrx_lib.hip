// Library-level entry points of libnrx_hip.so: ABI version, thread-local error text, device facts.
#include "nrx_common.h"

static thread_local char g_err[512] = "";

void nrx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Zero-fill as a KERNEL.  hipMemsetAsync is avoided throughout the library: inside a captured HIP graph (news_recsys_amd/graph.py)
// a small memset node did not take effect on replay on this stack (ROCm 7.0 / gfx950) -- counters that a call clears kept their
// values from the previous replay.  A kernel node replays like any other launch.
namespace {
__global__ void nrx_zero_kernel(uint32_t* __restrict__ p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
}  // namespace

int nrx_zero_async(void* p, size_t bytes, hipStream_t st) {
    if (bytes == 0) return NRX_OK;
    if ((reinterpret_cast<uintptr_t>(p) & 3u) != 0 || (bytes & 3u) != 0) {
        nrx_set_error("nrx_zero_async: buffer not dword-aligned");
        return NRX_ERR_BAD_ARG;
    }
    const size_t n4 = bytes / 4;
    size_t grid = (n4 + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(nrx_zero_kernel, dim3((unsigned)grid), dim3(256), 0, st, reinterpret_cast<uint32_t*>(p), n4);
    return hipGetLastError() == hipSuccess ? NRX_OK : NRX_ERR_LAUNCH;
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
