// Development probe: sustained rate of v_mfma_f32_16x16x4_f32 / 32x32x2_f32 with NACC independent accumulators,
// W waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o mfma_probe tools/mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (f32x4){0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) acc[i][j] = 0;
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4 * 4);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        int threads = 64 * 4 * wps;  // wps waves per SIMD
        float ms6 = timeit([&] { hipLaunchKernelGGL(k16<6>, dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 2.f); });
        float ms2 = timeit([&] { hipLaunchKernelGGL(k16<2>, dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 2.f); });
        float ms32 = timeit([&] { hipLaunchKernelGGL(k32<2>, dim3(256), dim3(threads), 0, 0, out, iters, 1.f, 2.f); });
        double f6 = 256.0 * 4 * wps * iters * 6 * 2048 / (ms6 * 1e-3) / 1e12;
        double f2 = 256.0 * 4 * wps * iters * 2 * 2048 / (ms2 * 1e-3) / 1e12;
        double f32 = 256.0 * 4 * wps * iters * 2 * 4096 / (ms32 * 1e-3) / 1e12;
        printf("waves/SIMD %d: 16x16x4 NACC6 %.1f TF (%.2f ms)  NACC2 %.1f TF  32x32x2 NACC2 %.1f TF\n", wps, f6, ms6, f2, f32);
    }
    return 0;
}
