// Layout check of v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 blocks, K = 1): which lane supplies which A row / B column and
// where D[i][j] lands.  hipcc --offload-arch=gfx950 mfma4_probe.hip -o /tmp/mfma4_probe && /tmp/mfma4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int R) {
    const int l = threadIdx.x;
    const float a = (l % 4 == R) ? 1.f : 0.f;      // one-hot row R of every block
    const float b = 100.f + l;                     // column value = the lane's own number
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[l * 4 + i] = acc[i];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4 * sizeof(float));
    float h[256];
    int bad = 0;
    for (int R = 0; R < 4; ++R) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, R);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 4; ++i) {
                const float want = (i == R) ? 100.f + l : 0.f;     // D[i][j] in register i of lane 4 block + j
                if (h[l * 4 + i] != want) { if (bad < 8) printf("R=%d lane %d reg %d: got %g want %g\n", R, l, i, h[l * 4 + i], want); ++bad; }
            }
    }
    printf(bad ? "LAYOUT MISMATCH (%d)\n" : "layout OK: lane 4b+i supplies A[b][i] / B[b][i]; D[i][j] = register i of lane 4b+j (%d)\n", bad);
    return bad != 0;
}
