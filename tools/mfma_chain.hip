// MFMA 32x32x16 bf16 issue rate of ONE wave per SIMD as a function of the number of independent accumulators (dependent chains):
// what the v4 count kernel's two alternating accumulators can reach.  build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_chain tools/mfma_chain.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// DATA: 0 tiny constants (few bits toggle), 1 pseudo-random operands (what a real table looks like to the power budget)
// AREG: the A operand read from an AGPR (inline asm), as the v4 count kernel does
template <int NACC, int WAVES, int DATA, int AREG>
__global__ __launch_bounds__(64 * WAVES) void spin2(float* out, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 xb[4], yb[4];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u;
            const float x = DATA ? ((int)(h >> 8) % 2001 - 1000) * 1e-4f : (float)threadIdx.x * 1e-6f;
            h = h * 1664525u + 1013904223u;
            const float y = DATA ? ((int)(h >> 8) % 2001 - 1000) * 1e-4f : (float)blockIdx.x * 1e-6f;
            xb[j][i] = (__bf16)x; yb[j][i] = (__bf16)y;
        }
    if (AREG) for (int j = 0; j < 4; ++j) asm volatile("" : "+a"(xb[j]));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                if (AREG) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[a]) : "a"(xb[u & 3]), "v"(yb[(u + a) & 3]));
                else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb[u & 3], yb[(u + a) & 3], acc[a], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 123.456f) out[0] = s;
}
template <int NACC, int WAVES, int DATA, int AREG>
static void run2() {
    float* out; (void)hipMalloc(&out, 4);
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((spin2<NACC, WAVES, DATA, AREG>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, 100);
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((spin2<NACC, WAVES, DATA, AREG>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep == 4) best = ms;   // the LAST of five back-to-back launches: the clock has settled
    }
    const double n = (double)iters * 8 * NACC;
    const double flop = (double)blocks * WAVES * n * 2.0 * 32 * 32 * 16;
    printf("%d acc, %d waves/CU, %s data, A from %s: %.3f ms  %.1f TFLOP/s  %.1f ns per MFMA per wave\n", NACC, WAVES, DATA ? "random" : "constant",
           AREG ? "AGPR (asm)" : "VGPR (builtin)", best, flop / best * 1e-9, best * 1e6 / n);
    (void)hipFree(out);
}

// NV independent VALU instructions behind every MFMA of ONE wave per SIMD: do they issue in the MFMA's shadow?
template <int NV, int DATA>
__global__ __launch_bounds__(256) void spin3(float* out, int iters) {
    f32x16 acc[2];
    for (int a = 0; a < 2; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 xb, yb;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u;
        xb[i] = (__bf16)(DATA ? ((int)(h >> 8) % 2001 - 1000) * 1e-4f : 1e-6f * threadIdx.x);
        h = h * 1664525u + 1013904223u;
        yb[i] = (__bf16)(DATA ? ((int)(h >> 8) % 2001 - 1000) * 1e-4f : 1e-6f * blockIdx.x);
    }
    unsigned c[8];
    for (int i = 0; i < 8; ++i) c[i] = threadIdx.x + i;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[a]) : "v"(xb), "v"(yb));
#pragma unroll
                for (int v = 0; v < NV; ++v) asm volatile("v_add_u32 %0, %0, %1" : "+v"(c[v & 7]) : "v"(c[(v + 1) & 7]));
            }
    }
    float s = 0.f;
    for (int a = 0; a < 2; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    unsigned cs = 0;
    for (int i = 0; i < 8; ++i) cs += c[i];
    if (s == 123.456f || cs == 0x12345678u) out[0] = s;
}
template <int NV, int DATA>
static void run3() {
    float* out; (void)hipMalloc(&out, 4);
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((spin3<NV, DATA>), dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("1 wave/SIMD, %d VALU behind each MFMA, %s data: %.3f ms  %.1f ns per MFMA\n", NV, DATA ? "random" : "constant", ms, ms * 1e6 / (iters * 16.0));
    (void)hipFree(out);
}

template <int NACC, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void spin(float* out, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const float x = (float)threadIdx.x * 1e-6f, y = (float)blockIdx.x * 1e-6f;
    bf16x8 xb, yb;
    for (int i = 0; i < 8; ++i) { xb[i] = (__bf16)x; yb[i] = (__bf16)y; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 123.456f) out[0] = s;
}

template <int NACC, int WAVES>
static void run() {
    float* out; hipMalloc(&out, 4);
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((spin<NACC, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((spin<NACC, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 8 * NACC;   // MFMAs per wave
    const double flop = (double)blocks * WAVES * n * 2.0 * 32 * 32 * 16;
    printf("%d accumulators, %d waves/CU: %.3f ms  %.1f TFLOP/s  %.1f ns per MFMA per wave\n", NACC, WAVES, ms, flop / ms * 1e-9, ms * 1e6 / n);
    hipFree(out);
}

int main() {
    run<1, 4>(); run<2, 4>(); run<3, 4>(); run<4, 4>();
    run<1, 8>(); run<2, 8>(); run<4, 8>();
    run3<0, 0>(); run3<2, 0>(); run3<4, 0>(); run3<6, 0>(); run3<8, 0>(); run3<12, 0>();
    run3<0, 1>(); run3<4, 1>(); run3<8, 1>();
    run2<2, 4, 0, 0>(); run2<2, 4, 1, 0>(); run2<2, 4, 0, 1>(); run2<2, 4, 1, 1>(); run2<2, 8, 1, 0>();
    return 0;
}
