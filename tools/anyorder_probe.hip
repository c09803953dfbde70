// Probe: does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) let a kernel start before its predecessor IN THE SAME STREAM has
// finished on gfx950?  Two independent spin kernels (each ~half the chip's slots), launched back to back K times:
//   ordered    : both with the default barrier bit          -> time ~ K * (tA + tB)
//   any-order B: B launched with hipExtAnyOrderLaunch       -> time ~ K * max(tA, tB) if the flag works
// build: hipcc --offload-arch=gfx950 -O3 -o tools/anyorder_probe tools/anyorder_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>

__global__ void spin(unsigned long long cycles, unsigned* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned x = 0;
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) x++;
    if (x == 0xFFFFFFFFu) *sink = x;
}

int main() {
    unsigned* sink;
    hipMalloc(&sink, 4);
    hipStream_t st;
    hipStreamCreate(&st);
    const int K = 200;
    const unsigned long long cyc = 40000;   // ~20 us
    auto run = [&](int flagsB) {
        for (int i = 0; i < K; i++) {
            hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, nullptr, nullptr, 0, cyc, sink);
            hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, nullptr, nullptr, flagsB, cyc, sink);
        }
    };
    for (int rep = 0; rep < 2; rep++) {
        for (int mode = 0; mode < 2; mode++) {
            run(mode ? hipExtAnyOrderLaunch : 0);
            hipStreamSynchronize(st);
            auto t0 = std::chrono::steady_clock::now();
            run(mode ? hipExtAnyOrderLaunch : 0);
            hipStreamSynchronize(st);
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K;
            printf("{\"mode\": \"%s\", \"us_per_pair\": %.2f}\n", mode ? "B any-order" : "ordered", us);
        }
    }
    hipError_t e = hipGetLastError();
    printf("{\"last_error\": \"%s\"}\n", hipGetErrorString(e));
    return 0;
}
