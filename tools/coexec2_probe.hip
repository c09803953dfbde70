// Development probe: does the exact-f32 MFMA stream of the decode tolerate OTHER waves' vector work on the same SIMDs, and does
// it matter what that work is?  Round 2's coexec_probe paired it with v_pk_fma_f32 streams (they share the FMA lanes: no gain).
// The emit kernel's vector work is mostly NOT fused multiply-adds: min3 / max3 / ceil / floor / compares / selects / integer.
// A 16-wave workgroup per CU: waves [0, NM) run the MFMA loop, waves [NM, 16) one of three vector mixes:
//   0 = v_fma_f32 chain mix, 1 = emit-like float mix without FMA (min3, max3, ceil, floor, sub, cmp + cndmask), 2 = integer mix
// Reported: time of each role alone and together; "serial" = the sum, "free" = the max.
//   hipcc --offload-arch=gfx950 -O3 -o tools/coexec2_probe tools/coexec2_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MIX>
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
        float x[8];
        int n[8];
#pragma unroll
        for (int i = 0; i < 8; i++) { x[i] = a0 * (i + 1) + threadIdx.x * 0.37f; n[i] = (int)threadIdx.x * (i + 3); }
        for (int it = 0; it < iters_v; it++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int j = (i + 1) & 7, l = (i + 3) & 7;
                if (MIX == 0) {
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(x[j]), "v"(b0));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(x[l]), "v"(a0));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(x[j]), "v"(a0));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(x[l]), "v"(b0));
                } else if (MIX == 1) {
                    float t0, t1;
                    asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(x[i]), "v"(x[j]), "v"(x[l]));
                    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(t1) : "v"(x[i]), "v"(x[j]), "v"(x[l]));
                    asm volatile("v_ceil_f32 %0, %0" : "+v"(t0));
                    asm volatile("v_floor_f32 %0, %0" : "+v"(t1));
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(x[i]) : "v"(t1), "v"(t0));
                    asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "+v"(x[j]) : "v"(t0), "v"(t1) : "vcc");
                } else {
                    asm volatile("v_add_u32 %0, %1, %0" : "+v"(n[i]) : "v"(n[j]));
                    asm volatile("v_xor_b32 %0, %1, %0" : "+v"(n[i]) : "v"(n[l]));
                    asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(n[i]));
                    asm volatile("v_and_or_b32 %0, %1, %2, %0" : "+v"(n[i]) : "v"(n[j]), "v"(n[l]));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) s += x[i] + (float)n[i];
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
template <int MIX>
void run(float* out, const char* name) {
    const int IM = 4000;
    for (int nm : {12, 8}) {
        // scale the vector role so that alone it takes about as long as the MFMA role alone
        float m_only = timeit([&] { hipLaunchKernelGGL(k<MIX>, dim3(256), dim3(1024), 0, 0, out, IM, 0, nm, 1.f, 2.f); });
        float v_probe = timeit([&] { hipLaunchKernelGGL(k<MIX>, dim3(256), dim3(1024), 0, 0, out, 0, 1000, nm, 1.f, 2.f); });
        const int IV = (int)(1000.0 * m_only / v_probe);
        float v_only = timeit([&] { hipLaunchKernelGGL(k<MIX>, dim3(256), dim3(1024), 0, 0, out, 0, IV, nm, 1.f, 2.f); });
        float both = timeit([&] { hipLaunchKernelGGL(k<MIX>, dim3(256), dim3(1024), 0, 0, out, IM, IV, nm, 1.f, 2.f); });
        printf("{\"mix\": \"%s\", \"mfma_waves\": %d, \"vector_waves\": %d, \"mfma_alone_ms\": %.3f, \"vector_alone_ms\": %.3f, "
               "\"together_ms\": %.3f, \"if_serial_ms\": %.3f, \"if_free_ms\": %.3f}\n", name, nm, 16 - nm, m_only, v_only, both,
               m_only + v_only, m_only > v_only ? m_only : v_only);
    }
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    run<0>(out, "v_fma_f32");
    run<1>(out, "emit-like float mix without FMA");
    run<2>(out, "integer mix");
    return 0;
}
