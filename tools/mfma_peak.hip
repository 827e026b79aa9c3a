// Sustained MFMA issue rate with nothing else in the way: 4 independent accumulators per wave, W waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_peak tools/mfma_peak.hip ; run on the MI355X box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void spin(float* out, int iters) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const float x = (float)threadIdx.x * 1e-6f, y = (float)blockIdx.x * 1e-6f;
    bf16x8 xb, yb;
    for (int i = 0; i < 8; ++i) { xb[i] = (__bf16)x; yb[i] = (__bf16)y; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if constexpr (MODE == 0) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
            else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, acc[a], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 123.456f) out[0] = s;
}

template <int MODE>
static void run(const char* name, double flop_per_mfma, int wg_per_cu) {
    float* out; hipMalloc(&out, 4);
    const int iters = 20000, blocks = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(spin<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(spin<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 /*waves*/ * iters * 4 /*mfma*/ * flop_per_mfma;
    printf("%-28s %d WG/CU: %.3f ms  %.1f TFLOP/s\n", name, wg_per_cu, ms, flop / ms * 1e-9);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 3; ++w) run<0>("v_mfma_f32_32x32x2_f32", 2.0 * 32 * 32 * 2, w);
    for (int w = 1; w <= 3; ++w) run<1>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, w);
    return 0;
}
