// Development probe: do exact-f32 MFMA waves and packed-f32 VALU FMA waves on the SAME SIMDs add their rates?
// A 16-wave workgroup per CU: waves [0, NM) run a v_mfma_f32_16x16x4_f32 loop (6 independent accumulators), waves
// [NM, 16) run a v_pk_fma_f32 loop (24 independent accumulator pairs, one operand broadcast through op_sel the way a
// VALU outer-product GEMM body would use it).  Reports TFLOP/s of each role alone and together.
//   hipcc --offload-arch=gfx950 -O3 -o coexec_probe tools/coexec_probe.hip && ./coexec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(1024) void k(float* out, int iters_m, int iters_v, int nm, float a0, float b0) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float s = 0;
    if (wave < nm) {
        f32x4 acc[6];
#pragma unroll
        for (int i = 0; i < 6; i++) acc[i] = (f32x4){0, 0, 0, 0};
        float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
        for (int it = 0; it < iters_m; it++) {
#pragma unroll
            for (int i = 0; i < 6; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 6; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f32x2 acc[24];
#pragma unroll
        for (int i = 0; i < 24; i++) acc[i] = (f32x2){0.f, 0.f};
        f32x2 p = {b0 + threadIdx.x, b0 - threadIdx.x};
        f32x2 av = {a0 * 1e-3f, a0 * 2e-3f};  // only the low half is used (op_sel_hi:[0,..]): the broadcast operand
        for (int it = 0; it < iters_v; it++) {
#pragma unroll
            for (int i = 0; i < 24; i++)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(av), "v"(p));
        }
#pragma unroll
        for (int i = 0; i < 24; i++) s += acc[i][0] + acc[i][1];
    }
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
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    const int IM = 4000, IV = 4000;
    for (int nm : {16, 8, 4, 12}) {
        const int nv = 16 - nm;
        // MFMA flops per wave-iteration: 6 * 2048 ; VALU: 24 pk_fma * 64 lanes * 2 * 2 = 6144
        auto tf_m = [&](float ms, int iters) { return 256.0 * nm * iters * 6 * 2048 / (ms * 1e-3) / 1e12; };
        auto tf_v = [&](float ms, int iters) { return 256.0 * nv * iters * 6144.0 / (ms * 1e-3) / 1e12; };
        float m_only = timeit([&] { hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, out, IM, 0, nm, 1.f, 2.f); });
        float v_only = nv ? timeit([&] { hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, out, 0, IV, nm, 1.f, 2.f); }) : 0.f;
        float both = timeit([&] { hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, out, IM, IV, nm, 1.f, 2.f); });
        printf("MFMA waves %2d (%.1f TF alone, %.2f ms) | VALU waves %2d (%.1f TF alone, %.2f ms) | together %.2f ms = %.1f TF "
               "combined\n", nm, tf_m(m_only, IM), m_only, nv, nv ? tf_v(v_only, IV) : 0.0, v_only, both,
               tf_m(both, IM) + (nv ? tf_v(both, IV) : 0.0));
    }
    // balanced: scale the VALU iterations so that both roles finish together (8 + 8 waves)
    for (int iv : {2000, 3000, 4000, 6000, 8000}) {
        float both = timeit([&] { hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, out, IM, iv, 8, 1.f, 2.f); });
        printf("8 MFMA waves x %d iters + 8 VALU waves x %d iters: %.2f ms = %.1f TF combined\n", IM, iv, both,
               (256.0 * 8 * IM * 6 * 2048 + 256.0 * 8 * iv * 6144.0) / (both * 1e-3) / 1e12);
    }
    return 0;
}
