// Development probe 2: the decode inner loop shape (6 accumulators, B fragments from LDS, A fragments from global)
// in isolation, to find which ingredient costs MFMA throughput.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: regs only; 1: + B from LDS; 2: + A from global (3 x dwordx4 per 4 k-steps, 1-deep prefetch)
template <int MODE>
__global__ __launch_bounds__(1024) void loop(const float4* __restrict__ A, float* out, int groups, int items) {
    extern __shared__ float smem[];
    for (int i = threadIdx.x; i < 15 * 256 * 2; i += blockDim.x) smem[i] = 0.001f * (i & 255);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 s[6];
    float tot = 0;
    for (int it = 0; it < items; it++) {
#pragma unroll
        for (int i = 0; i < 6; i++) s[i] = (f32x4){0, 0, 0, 0};
        const float4* Ap = A + ((size_t)(blockIdx.x * 16 + wave) * items + it) % 3000 * (size_t)groups * 192 + lane;
        float4 c0 = make_float4(1, 2, 3, 4), c1 = c0, c2 = c0;
        if (MODE >= 2) { c0 = Ap[0]; c1 = Ap[64]; c2 = Ap[128]; }
        for (int g = 0; g < groups; g++) {
            float4 n0 = c0, n1 = c1, n2 = c2;
            if (MODE >= 2) {
                int gn = g + 1 < groups ? g + 1 : g;
                n0 = Ap[(size_t)gn * 192]; n1 = Ap[(size_t)gn * 192 + 64]; n2 = Ap[(size_t)gn * 192 + 128];
            }
            const float* P = smem + (size_t)g * 512 + lane * 2;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                f32x2 bq = (MODE >= 1) ? *reinterpret_cast<const f32x2*>(P + j * 128) : (f32x2){1.f + j, 2.f};
                float a0 = j == 0 ? c0.x : j == 1 ? c0.y : j == 2 ? c0.z : c0.w;
                float a1 = j == 0 ? c1.x : j == 1 ? c1.y : j == 2 ? c1.z : c1.w;
                float a2 = j == 0 ? c2.x : j == 1 ? c2.y : j == 2 ? c2.z : c2.w;
                s[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bq[0], s[0], 0, 0, 0);
                s[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bq[1], s[1], 0, 0, 0);
                s[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bq[0], s[2], 0, 0, 0);
                s[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bq[1], s[3], 0, 0, 0);
                s[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, bq[0], s[4], 0, 0, 0);
                s[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, bq[1], s[5], 0, 0, 0);
            }
            c0 = n0; c1 = n1; c2 = n2;
        }
#pragma unroll
        for (int i = 0; i < 6; i++) tot += s[i][0] + s[i][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = tot;
}
template <typename F>
float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    const int groups = 15, items = 2;
    float4* A; hipMalloc(&A, (size_t)3000 * groups * 192 * 16 + 4096);
    hipMemset(A, 0, (size_t)3000 * groups * 192 * 16);
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    hipFuncSetAttribute((const void*)loop<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)loop<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)loop<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    double flop = 256.0 * 16 * items * groups * 24 * 2048;
    float m0 = timeit([&] { hipLaunchKernelGGL(loop<0>, dim3(256), dim3(1024), 65536, 0, A, out, groups, items); });
    float m1 = timeit([&] { hipLaunchKernelGGL(loop<1>, dim3(256), dim3(1024), 65536, 0, A, out, groups, items); });
    float m2 = timeit([&] { hipLaunchKernelGGL(loop<2>, dim3(256), dim3(1024), 65536, 0, A, out, groups, items); });
    printf("regs only %.1f us %.1f TF | +LDS B %.1f us %.1f TF | +global A %.1f us %.1f TF\n", m0 * 1e3, flop / m0 / 1e9,
           m1 * 1e3, flop / m1 / 1e9, m2 * 1e3, flop / m2 / 1e9);
    return 0;
}
