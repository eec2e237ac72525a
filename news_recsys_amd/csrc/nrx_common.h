// Internal helpers shared by the HIP translation units of libnrx_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "nrx_embed.h"

#define NRX_WAVE 64
#define NRX_BLOCK 256

void nrx_set_error(const char* fmt, ...);
// roctx range around an entry point (NRX_ROCTX=1 in the environment; libroctx64 is loaded on first use, never linked): shows the
// library's calls as named ranges in rocprofv3 --marker-trace timelines.  Off: one predictable branch per call.
void nrx_trace_push(const char* name);
void nrx_trace_pop();
extern int nrx_trace_on;       // -1 unknown, 0 off, 1 on
struct NrxTrace {
    bool on;
    explicit NrxTrace(const char* name) : on(nrx_trace_on != 0) {
        if (on) nrx_trace_push(name);
    }
    ~NrxTrace() {
        if (on) nrx_trace_pop();
    }
};
#define NRX_TRACE() NrxTrace nrx_trace_scope__(__func__)
// zero-fill by a kernel launch (capture-safe replacement of hipMemsetAsync; see nrx_lib.hip); p and bytes dword-aligned
int nrx_zero_async(void* p, size_t bytes, hipStream_t st);
int nrx_zero2_async(void* p, size_t bytes_p, void* q, size_t bytes_q, hipStream_t st);

#define NRX_REQUIRE(cond, ...)                \
    do {                                      \
        if (!(cond)) {                        \
            nrx_set_error(__VA_ARGS__);       \
            return NRX_ERR_BAD_ARG;           \
        }                                     \
    } while (0)

#define NRX_LAUNCH_CHECK(name)                                                    \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            nrx_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return NRX_ERR_LAUNCH;                                                \
        }                                                                         \
    } while (0)

static inline bool nrx_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Device-side feature descriptor (48 B; 64 of them fit the 4 KiB kernarg segment).
struct FeatDev {
    const float* table;   // weight table, or grad table in the backward
    const void* index;
    const float* weight;
    int64_t rows;
    int32_t out_col;
    int32_t wide_col;
    int16_t dim;
    int16_t bag_len;
    uint8_t kind;
    uint8_t idx64;
    uint8_t fm;
    uint8_t flags;          // NRX_FEAT_* bits (ROW0_IS_DATA, BAG_CSR)
};
static_assert(sizeof(FeatDev) == 48, "FeatDev must stay 48 bytes");

// Address-space helpers.  Kernel arguments that are indexed dynamically are read through an
// explicit constant-address-space (4) pointer to the kernarg segment (scalar s_load with a uniform
// index; never copied to scratch), and data pointers fetched from them are cast to the global
// address space (1) so the compiler emits global_load/global_store rather than flat_*.
#define NRX_CONST __attribute__((address_space(4)))
#define NRX_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const NRX_CONST T* nrx_kernarg() {
    return (const NRX_CONST T*)(__builtin_amdgcn_kernarg_segment_ptr());
}
template <typename T>
__device__ __forceinline__ const NRX_GLOBAL T* nrx_gconst(const void* p) {
    return (const NRX_GLOBAL T*)(p);
}
template <typename T>
__device__ __forceinline__ NRX_GLOBAL T* nrx_gmut(void* p) {
    return (NRX_GLOBAL T*)(p);
}

// 16-byte global accesses through a native vector type (HIP's float4 class has no
// address-space-qualified operator=).
using nrx_f32x4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ float4 nrx_ldg4(const void* base, int64_t i) {
    const nrx_f32x4 t = ((const NRX_GLOBAL nrx_f32x4*)(base))[i];
    return make_float4(t.x, t.y, t.z, t.w);
}
// Non-temporal variant for rows of tables far larger than L2 + Infinity Cache: measured +8 % on
// 64 B rows (26 x 1M x 16) -- the streamed rows stop displacing the output lines and id blocks.
__device__ __forceinline__ float4 nrx_ldg4_nt(const void* base, int64_t i) {
    const nrx_f32x4 t = __builtin_nontemporal_load(((const NRX_GLOBAL nrx_f32x4*)(base)) + i);
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void nrx_stg4(void* base, int64_t i, float4 v) {
    nrx_f32x4 t;
    t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    ((NRX_GLOBAL nrx_f32x4*)(base))[i] = t;
}

// First offender wins the detail slots; the count is exact.
__device__ __forceinline__ void nrx_report_oob(int32_t* status, int feat, int64_t b, int64_t id) {
    if (status != nullptr) {
        int prev = atomicAdd(&status[0], 1);
        if (prev == 0) {
            status[1] = feat;
            status[2] = (int32_t)b;
            status[3] = (int32_t)id;
        }
    }
}

__device__ __forceinline__ int64_t nrx_load_id(const void* p, int64_t i, bool is64) {
    return is64 ? reinterpret_cast<const int64_t*>(p)[i] : (int64_t)reinterpret_cast<const int32_t*>(p)[i];
}

// Sum over the 64 lanes, result in every lane.  Data-parallel-primitive moves inside each row of 16 lanes (no
// LDS round trips: __shfl_xor compiles to ds_bpermute_b32, ~100 cycles of latency per step, six steps per
// reduction -- the DCN-v1 backward does five reductions per sample and was bound by them), then the four row
// sums are read as scalars.  Fixed summation tree: ((xor 1, xor 2), half-row mirror, row mirror), (r0+r1)+(r2+r3).
template <int CTRL>
__device__ __forceinline__ float nrx_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float nrx_wave_sum(float v) {
    v += nrx_dpp<0xB1>(v);      // quad_perm [1,0,3,2]: lane ^ 1
    v += nrx_dpp<0x4E>(v);      // quad_perm [2,3,0,1]: lane ^ 2
    v += nrx_dpp<0x141>(v);     // row_half_mirror: the other quad of each 8
    v += nrx_dpp<0x140>(v);     // row_mirror: the other half of each 16
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (r0 + r1) + (r2 + r3);
}
