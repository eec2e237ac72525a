// Library-level entry points of libnrx_hip.so: ABI version, thread-local error text, device facts.
#include "nrx_common.h"
#include <cstdlib>

static thread_local char g_err[512] = "";

void nrx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- roctx ranges (see NRX_TRACE in nrx_common.h)
#include <dlfcn.h>
int nrx_trace_on = -1;
namespace {
int (*g_roctx_push)(const char*) = nullptr;
int (*g_roctx_pop)() = nullptr;
}
void nrx_trace_push(const char* name) {
    if (nrx_trace_on < 0) {
        nrx_trace_on = 0;
        const char* e = getenv("NRX_ROCTX");
        if (e && e[0] == '1') {
            void* h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
            if (!h) h = dlopen("/opt/rocm/lib/libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
            if (h) {
                g_roctx_push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                g_roctx_pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (g_roctx_push && g_roctx_pop) nrx_trace_on = 1;
            }
        }
    }
    if (nrx_trace_on == 1) g_roctx_push(name);
}
void nrx_trace_pop() {
    if (nrx_trace_on == 1 && g_roctx_pop) g_roctx_pop();
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

// Two regions in one launch (a launch is ~4.7 us of dependent latency whatever it clears).
namespace {
__global__ void nrx_zero2_kernel(uint32_t* __restrict__ p, size_t n4, uint32_t* __restrict__ q, size_t m4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4 + m4; i += (size_t)gridDim.x * blockDim.x) {
        if (i < n4) p[i] = 0u;
        else q[i - n4] = 0u;
    }
}
}  // namespace

int nrx_zero2_async(void* p, size_t bytes_p, void* q, size_t bytes_q, hipStream_t st) {
    if (bytes_p == 0) return nrx_zero_async(q, bytes_q, st);
    if (bytes_q == 0) return nrx_zero_async(p, bytes_p, st);
    if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(q)) & 3u) != 0 || ((bytes_p | bytes_q) & 3u) != 0) {
        nrx_set_error("nrx_zero2_async: buffer not dword-aligned");
        return NRX_ERR_BAD_ARG;
    }
    const size_t n4 = bytes_p / 4, m4 = bytes_q / 4;
    size_t grid = (n4 + m4 + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(nrx_zero2_kernel, dim3((unsigned)grid), dim3(256), 0, st, reinterpret_cast<uint32_t*>(p), n4, reinterpret_cast<uint32_t*>(q), m4);
    return hipGetLastError() == hipSuccess ? NRX_OK : NRX_ERR_LAUNCH;
}

// Streaming copy (16 bytes per lane, grid-stride): the bench harness times it next to the gather so that a roofline fraction can be
// read against what THIS box's memory system sustains for a plain copy (guides/MI355X_MICROARCH.md: ~6.3 of the 8 TB/s spec).
namespace {
template <bool NT, int UNROLL>
__global__ __launch_bounds__(256) void nrx_copy_kernel(const nrx_f32x4* __restrict__ src, nrx_f32x4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        nrx_f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}
}  // namespace

extern "C" int nrx_stream_copy(void* dst, const void* src, int64_t bytes, void* stream) {
    NRX_TRACE();
    NRX_REQUIRE(dst != nullptr && src != nullptr && bytes >= 0 && (bytes & 15) == 0 && nrx_aligned16(dst) && nrx_aligned16(src),
                "nrx_stream_copy: buffers must be 16-byte aligned and a multiple of 16 bytes long");
    if (bytes == 0) return NRX_OK;
    const size_t n16 = (size_t)bytes / 16;
    // variant knob for the measurement itself (NRX_COPY_VARIANT = <nt 0|1><unroll 1|4><blocks per CU>, e.g. "1432"); default below
    const char* env = getenv("NRX_COPY_VARIANT");
    // measured (tools/probe_stream_copy.py, 1 GiB, GB/s read + written): nt 1 / unroll 1 / 32 blocks per CU 5145; plain 4615; unroll 4
    // 4720-5057; 8 blocks per CU 4790-5077; torch's copy_ 5113
    const bool nt = env ? env[0] == '1' : true;
    const int unroll = env && env[1] == '4' ? 4 : 1;
    const int bpc = env && env[2] ? atoi(env + 2) : 32;
    size_t grid = (n16 + 255) / 256 / unroll;
    if (grid > (size_t)256 * bpc) grid = (size_t)256 * bpc;
    if (grid < 1) grid = 1;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const nrx_f32x4* s4 = reinterpret_cast<const nrx_f32x4*>(src);
    nrx_f32x4* d4 = reinterpret_cast<nrx_f32x4*>(dst);
    if (nt && unroll == 4) hipLaunchKernelGGL((nrx_copy_kernel<true, 4>), dim3((unsigned)grid), dim3(256), 0, st, s4, d4, n16);
    else if (nt) hipLaunchKernelGGL((nrx_copy_kernel<true, 1>), dim3((unsigned)grid), dim3(256), 0, st, s4, d4, n16);
    else if (unroll == 4) hipLaunchKernelGGL((nrx_copy_kernel<false, 4>), dim3((unsigned)grid), dim3(256), 0, st, s4, d4, n16);
    else hipLaunchKernelGGL((nrx_copy_kernel<false, 1>), dim3((unsigned)grid), dim3(256), 0, st, s4, d4, n16);
    NRX_LAUNCH_CHECK("nrx_stream_copy");
    return NRX_OK;
}

extern "C" int nrx_abi_version(void) { return NRX_ABI_VERSION; }

extern "C" const char* nrx_last_error(void) { return g_err; }

extern "C" int nrx_device_info(int device, int64_t info[6]) {
    NRX_TRACE();
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
