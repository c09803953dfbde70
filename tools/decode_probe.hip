// Development probe for the decode kernel: times fr_launch_decode on random data, built with -DFR_PROBE_DECODE=<bits>
// (1 = no MFMA, 2 = every A request hits the first 256 tiles (L2/MALL resident), 4 = no per-CU prologue) to separate the
// memory stream, the matrix pipe and the fixed cost.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17
//   -DFR_PROBE_DECODE=n -o decode_probe_n tools/decode_probe.hip
#include "../3dfacerecon_amd/csrc/fr_decode.hip"
#include <stdio.h>
#include <vector>

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, N = 53215, ns = 199, ne = 29, iters = 50;
    const size_t pb = fr_packed_basis_bytes(N, ns, ne);
    void *packed, *params, *out;
    (void)hipMalloc(&packed, pb);
    (void)hipMalloc(&params, (size_t)B * 235 * 4);
    (void)hipMalloc(&out, (size_t)B * 3 * N * 4);
    std::vector<float> h(pb / 4);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (float)(x >> 8) * (1.0f / 16777216.0f) - 0.5f; }
    (void)hipMemcpy(packed, h.data(), pb, hipMemcpyHostToDevice);
    (void)hipMemcpy(params, h.data(), (size_t)B * 235 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; i++) fr_launch_decode((float*)params, packed, nullptr, B, N, ns, ne, 200.f, (float*)out, 0);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < iters; i++) fr_launch_decode((float*)params, packed, nullptr, B, N, ns, ne, 200.f, (float*)out, 0);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("probe=%d B=%d decode %.1f us\n", FR_PROBE_DECODE, B, ms * 1e3 / iters);
    return 0;
}
