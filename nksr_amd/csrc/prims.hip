// Device-wide primitives: radix sort, unique, exclusive sum.  These are plain library
// operations (rocPRIM through the hipCUB front-end); every domain kernel is hand-written.
#include "common.h"
#include <hipcub/hipcub.hpp>
#include <stdarg.h>

thread_local char g_nksr_err[512] = "";

int nksr_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_nksr_err, sizeof(g_nksr_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* nksr_last_error(void) { return g_nksr_err; }
extern "C" int nksr_version(void) { return 100; }

extern "C" int nksr_sort_keys_u64(void* tmp, size_t* tmp_bytes, const uint64_t* in, uint64_t* out, int64_t n,
                                  int begin_bit, int end_bit, void* stream) {
    if (!tmp_bytes) return nksr_set_error(NKSR_ERR_ARG, "tmp_bytes is NULL");
    NKSR_CHECK_HIP(hipcub::DeviceRadixSort::SortKeys(tmp, *tmp_bytes, in, out, n, begin_bit, end_bit, (hipStream_t)stream));
    return NKSR_OK;
}

extern "C" int nksr_sort_pairs_u64_u32(void* tmp, size_t* tmp_bytes, const uint64_t* kin, uint64_t* kout,
                                       const uint32_t* vin, uint32_t* vout, int64_t n, int begin_bit, int end_bit,
                                       void* stream) {
    if (!tmp_bytes) return nksr_set_error(NKSR_ERR_ARG, "tmp_bytes is NULL");
    NKSR_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, *tmp_bytes, kin, kout, vin, vout, n, begin_bit, end_bit,
                                                      (hipStream_t)stream));
    return NKSR_OK;
}

extern "C" int nksr_unique_u64(void* tmp, size_t* tmp_bytes, const uint64_t* in, uint64_t* out, int64_t* d_count,
                               int64_t n, void* stream) {
    if (!tmp_bytes) return nksr_set_error(NKSR_ERR_ARG, "tmp_bytes is NULL");
    NKSR_CHECK_HIP(hipcub::DeviceSelect::Unique(tmp, *tmp_bytes, in, out, d_count, n, (hipStream_t)stream));
    return NKSR_OK;
}

extern "C" int nksr_exclusive_sum_i32(void* tmp, size_t* tmp_bytes, const int32_t* in, int32_t* out, int64_t n,
                                      void* stream) {
    if (!tmp_bytes) return nksr_set_error(NKSR_ERR_ARG, "tmp_bytes is NULL");
    NKSR_CHECK_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, *tmp_bytes, in, out, n, (hipStream_t)stream));
    return NKSR_OK;
}

extern "C" int nksr_exclusive_sum_i64(void* tmp, size_t* tmp_bytes, const int64_t* in, int64_t* out, int64_t n,
                                      void* stream) {
    if (!tmp_bytes) return nksr_set_error(NKSR_ERR_ARG, "tmp_bytes is NULL");
    NKSR_CHECK_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, *tmp_bytes, in, out, n, (hipStream_t)stream));
    return NKSR_OK;
}
