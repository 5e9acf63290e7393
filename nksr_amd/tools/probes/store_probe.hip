// Store-pattern probe for the kernel-rows layout (rows of 27 floats = 108 bytes): how fast can 10-30 GB of rows be
// WRITTEN, by the shape of the stores?  hipcc --offload-arch=gfx950 -O3 store_probe.hip -o store_probe && ./store_probe [GB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// P1: contiguous float4 fill (grid-stride free: one float4 per thread)
__global__ void k_fill4(float4* p, int64_t n4, float v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) p[i] = make_float4(v, v + 1, v + 2, v + 3);
}
__global__ void k_fill4_nt(float4* p, int64_t n4, float v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 t = {v, v + 1, v + 2, v + 3};
    if (i < n4) __builtin_nontemporal_store(t, reinterpret_cast<f4*>(p) + i);
}
// P2: lane = row, NR consecutive rows per lane, 16-byte stores at 4-byte alignment; rows of consecutive lanes are STEP rows apart
template <int NR, bool NT>
__global__ void k_lane_rows(float* p, int64_t nsite, int step, int first, float v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsite) return;
    float* r = p + (i * step + first) * 27;
#pragma unroll
    for (int a = 0; a < NR; ++a) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            f32x4_u t = {v + q, v, v, v + a};
            if (NT) __builtin_nontemporal_store(t, reinterpret_cast<f32x4_u*>(r + a * 27 + 4 * q));
            else *reinterpret_cast<f32x4_u*>(r + a * 27 + 4 * q) = t;
        }
        r[a * 27 + 24] = v; r[a * 27 + 25] = v; r[a * 27 + 26] = v;
    }
}
// P3: lane = slot, a 32-lane half-wave per site, NR rows per site
template <int NR>
__global__ void k_slot_rows(float* p, int64_t nsite, int step, int first, float v) {
    const int lane = threadIdx.x & 31;
    int64_t hw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    for (int q = 0; q < 32; ++q) {
        int64_t i = hw * 32 + q;
        if (i >= nsite) return;
        float* r = p + (i * step + first) * 27;
        if (lane < 27) {
#pragma unroll
            for (int a = 0; a < NR; ++a) r[a * 27 + lane] = v + q;
        }
    }
}
// P4: a wavefront writes the rows of 64 consecutive sites as ONE contiguous image (64 * NR * 108 bytes), 16 bytes per lane and instruction
template <int NR>
__global__ void k_wave_image(float* p, int64_t nsite, float v) {
    const int lane = threadIdx.x & 63;
    int64_t wv = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wv * 64 >= nsite) return;
    float4* base = reinterpret_cast<float4*>(p + wv * 64 * NR * 27);        // 64 * 27 * 4 = 6912 bytes: 16-byte aligned
    constexpr int N4 = 64 * NR * 27 / 4;
#pragma unroll
    for (int j = 0; j < (N4 + 63) / 64; ++j)
        if (j * 64 + lane < N4) base[j * 64 + lane] = make_float4(v, v + j, v, v);
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 12.0;
    const int64_t rows = (int64_t)(gb * 1e9 / 108.0) / 256 * 256;
    float* buf;
    CK(hipMalloc(&buf, rows * 108 + 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double bytes, auto&& launch) {
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 3; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 3;
        printf("%-64s %8.3f ms  %6.2f TB/s\n", name, ms, bytes / ms / 1e9);
        fflush(stdout);
    };
    const double total = (double)rows * 108;
    const int64_t n4 = rows * 27 / 4;
    timeit("P1 contiguous float4 fill", total, [&] { k_fill4<<<(n4 + 255) / 256, 256>>>((float4*)buf, n4, 1.f); });
    timeit("P1 contiguous float4 fill, nontemporal", total, [&] { k_fill4_nt<<<(n4 + 255) / 256, 256>>>((float4*)buf, n4, 1.f); });
    timeit("P2 lane = row, all rows (1 row per lane)", total, [&] { k_lane_rows<1, false><<<(rows + 127) / 128, 128>>>(buf, rows, 1, 0, 1.f); });
    timeit("P2 lane = row, all rows, nontemporal", total, [&] { k_lane_rows<1, true><<<(rows + 127) / 128, 128>>>(buf, rows, 1, 0, 1.f); });
    timeit("P2 lane = 3 rows, all rows", total, [&] { k_lane_rows<3, false><<<(rows / 3 + 127) / 128, 128>>>(buf, rows / 3, 3, 0, 1.f); });
    timeit("P2 lane = 3 rows of every 4 (gradient rows of the scene)", total * 0.75, [&] { k_lane_rows<3, false><<<(rows / 4 + 127) / 128, 128>>>(buf, rows / 4, 4, 0, 1.f); });
    timeit("P2 lane = 3 rows of every 4, nontemporal", total * 0.75, [&] { k_lane_rows<3, true><<<(rows / 4 + 127) / 128, 128>>>(buf, rows / 4, 4, 0, 1.f); });
    timeit("P2 lane = 1 row of every 4 (position rows of the scene)", total * 0.25, [&] { k_lane_rows<1, false><<<(rows / 4 + 127) / 128, 128>>>(buf, rows / 4, 4, 3, 1.f); });
    timeit("P2 lane = 1 row of every 4, nontemporal", total * 0.25, [&] { k_lane_rows<1, true><<<(rows / 4 + 127) / 128, 128>>>(buf, rows / 4, 4, 3, 1.f); });
    timeit("P3 lane = slot, all rows", total, [&] { k_slot_rows<1><<<(rows + 255) / 256, 256>>>(buf, rows, 1, 0, 1.f); });
    timeit("P3 lane = slot, 3 rows of every 4", total * 0.75, [&] { k_slot_rows<3><<<(rows / 4 + 255) / 256, 256>>>(buf, rows / 4, 4, 0, 1.f); });
    timeit("P3 lane = slot, 1 row of every 4", total * 0.25, [&] { k_slot_rows<1><<<(rows / 4 + 255) / 256, 256>>>(buf, rows / 4, 4, 3, 1.f); });
    timeit("P4 wave image, 1 row per site (64 rows = 6912 B contiguous)", total, [&] { k_wave_image<1><<<(rows + 255) / 256, 256>>>(buf, rows, 1.f); });
    timeit("P4 wave image, 4 rows per site (27648 B contiguous)", total, [&] { k_wave_image<4><<<(rows / 4 + 255) / 256, 256>>>(buf, rows / 4, 1.f); });
    return 0;
}
