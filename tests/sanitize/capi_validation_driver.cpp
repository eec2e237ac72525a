// ASan + UBSan run of the HOST half of the C-ABI (include/nrx_embed.h): argument validation, status codes and the thread-local
// error text of libnrx_hip's entry points, compiled from the library's own sources with host-side sanitizers (hipcc
// -fsanitize=address,undefined -fno-gpu-sanitize; the device code is not instrumented and never runs: every call below fails
// validation BEFORE any launch, or has nothing to do).  No GPU needed.  Built and run by tests/test_sanitizers.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "nrx_embed.h"

static int failures = 0;
#define EXPECT_BAD(call)                                                                                  \
    do {                                                                                                  \
        const int rc__ = (call);                                                                          \
        const char* e__ = nrx_last_error();                                                               \
        if (rc__ >= 0 || e__ == nullptr || std::strlen(e__) == 0) {                                       \
            std::fprintf(stderr, "expected a validation error from %s, got %d\n", #call, rc__);           \
            ++failures;                                                                                   \
        }                                                                                                 \
    } while (0)
#define EXPECT_OK(call)                                                                                   \
    do {                                                                                                  \
        const int rc__ = (call);                                                                          \
        if (rc__ != 0) { std::fprintf(stderr, "%s returned %d (%s)\n", #call, rc__, nrx_last_error()); ++failures; } \
    } while (0)

int main() {
    // "device" buffers: never dereferenced by the host code under test
    float* a = static_cast<float*>(std::aligned_alloc(64, 4096));
    float* b = static_cast<float*>(std::aligned_alloc(64, 4096));
    int64_t* i64 = static_cast<int64_t*>(std::aligned_alloc(64, 4096));
    if (nrx_abi_version() < 1) ++failures;
    // streaming copy
    EXPECT_BAD(nrx_stream_copy(nullptr, a, 64, nullptr));
    EXPECT_BAD(nrx_stream_copy(a, b, 60, nullptr));                       // not a multiple of 16
    EXPECT_BAD(nrx_stream_copy(reinterpret_cast<char*>(a) + 4, b, 64, nullptr));
    EXPECT_OK(nrx_stream_copy(a, b, 0, nullptr));
    // DCN-v2 forward
    EXPECT_BAD(nrx_dcn_v2_layer_fwd(nullptr, a, 16, 8, 16, b, b, 1, b, 16, nullptr, nullptr));
    EXPECT_BAD(nrx_dcn_v2_layer_fwd(a, a, 8, 8, 16, b, b, 1, b, 16, nullptr, nullptr));        // ld < dim
    EXPECT_BAD(nrx_dcn_v2_layer_fwd(a, a, 16, 8, 16, b, b, 3, a, 16, nullptr, nullptr));       // out aliases the input
    EXPECT_BAD(nrx_dcn_v2_layer_fwd(a, a, 16, -1, 16, b, b, 0, b, 16, nullptr, nullptr));
    EXPECT_OK(nrx_dcn_v2_layer_fwd(a, a, 16, 0, 16, b, b, 2, b, 16, nullptr, nullptr));        // empty batch: nothing to launch
    // top-k retrieval
    EXPECT_BAD(nrx_topk_ip(a, -1, 16, b, 4, 3, nullptr, nullptr, i64, a, a, nullptr));
    EXPECT_BAD(nrx_topk_ip(a, 8, 16, b, 4, 0, nullptr, nullptr, i64, a, a, nullptr));          // k < 1
    EXPECT_BAD(nrx_topk_ip(a, 8, 16, nullptr, 4, 3, nullptr, nullptr, i64, a, a, nullptr));
    EXPECT_BAD(nrx_topk_ip(a + 1, 8, 16, b, 4, 3, nullptr, nullptr, i64, a, a, nullptr));      // items not 16-byte aligned
    EXPECT_BAD(nrx_topk_ip(a, 8, 16, b, 4, 3, i64, nullptr, i64, a, a, nullptr));              // exclusion offsets without items
    // integer utilities
    EXPECT_BAD(nrx_bucketize_by_owner(i64, 64, 10, 0, i64, i64, i64, i64, nullptr));           // world < 1
    EXPECT_BAD(nrx_bucketize_by_owner(i64, 64, 10, 65, i64, i64, i64, i64, nullptr));
    EXPECT_BAD(nrx_bucketize_by_owner(i64, 16, 10, 2, i64, i64, i64, i64, nullptr));           // index_bits
    EXPECT_BAD(nrx_bucketize_by_owner(nullptr, 64, 10, 2, i64, i64, i64, i64, nullptr));
    EXPECT_BAD(nrx_csr_to_padded(i64, 48, i64, nullptr, 4, 3, i64, a, nullptr));               // value_bits
    EXPECT_BAD(nrx_csr_to_padded(i64, 64, i64, nullptr, 4, 0, i64, a, nullptr));               // bag_len < 1
    EXPECT_BAD(nrx_csr_to_padded(i64, 64, nullptr, nullptr, 4, 3, i64, a, nullptr));
    EXPECT_BAD(nrx_mask_lengths(nullptr, 4, 3, i64, nullptr));
    EXPECT_BAD(nrx_mask_lengths(a, -1, 3, i64, nullptr));
    {   // a formatted message with an integer argument: table %d is null
        const float* tabs[2] = {a, nullptr};
        int64_t rows[2] = {4, 4};
        int32_t seg_table[1] = {0};
        EXPECT_BAD(nrx_gather_rows_segmented(tabs, rows, 2, i64, seg_table, 1, 8, 16, i64, b, nullptr, nullptr));
        if (std::strstr(nrx_last_error(), "table 1") == nullptr) { std::fprintf(stderr, "error text: %s\n", nrx_last_error()); ++failures; }
    }
    std::free(a); std::free(b); std::free(i64);
    if (failures) { std::fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
    std::puts("C-ABI validation sanitize driver: OK");
    return 0;
}
