// Device-wide primitives: radix sort, unique, exclusive sum.  These are plain library
// operations (rocPRIM through the hipCUB front-end); every domain kernel is hand-written.
#include "common.h"
#include <hipcub/hipcub.hpp>
#include <rocprim/rocprim.hpp>

// rocPRIM sorts up to 2^20 items by merge sort (block sort + log2(n / 1024) merge passes of two launches each: ~17 launches of 6-7 us
// for the 10^5-key streams of the coarser hierarchy levels and the mesh keys); with the bit range of a cloud's Morton keys (one
// onesweep pass per 8 varying bits) the onesweep path is fewer launches and less time from ~16 k items on.
using nksr_sort_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 16384>;
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
    NKSR_CHECK_HIP(rocprim::radix_sort_keys<nksr_sort_config>(tmp, *tmp_bytes, in, out, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, (hipStream_t)stream));
    return NKSR_OK;
}

extern "C" int nksr_sort_pairs_u64_u32(void* tmp, size_t* tmp_bytes, const uint64_t* kin, uint64_t* kout,
                                       const uint32_t* vin, uint32_t* vout, int64_t n, int begin_bit, int end_bit,
                                       void* stream) {
    if (!tmp_bytes) return nksr_set_error(NKSR_ERR_ARG, "tmp_bytes is NULL");
    NKSR_CHECK_HIP(rocprim::radix_sort_pairs<nksr_sort_config>(tmp, *tmp_bytes, kin, kout, vin, vout, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit,
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
