// Development probe: does the resolver's plane-store pattern (dword stores at a 12-byte stride for the two 3-channel
// planes) cost bandwidth against fully coalesced 16-byte stores?  Same bytes, same grid shape as the resolver at B = 64.
// hipcc --offload-arch=gfx950 -O3 -o tools/store_pattern_probe tools/store_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int NPIX = 2000;   // pixels per bin (10 rows x 200)
__global__ __launch_bounds__(256) void stride12(float* dep, float* tin, float* tex, float* nrm, float v) {
    const size_t base = (size_t)blockIdx.x * NPIX;
    for (int i = threadIdx.x; i < NPIX; i += 256) {
        dep[base + i] = v;
        tin[base + i] = v;
        float* t = tex + 3 * (base + i);
        t[0] = v; t[1] = v; t[2] = v;
        float* n = nrm + 3 * (base + i);
        n[0] = v; n[1] = v; n[2] = v;
    }
}
__global__ __launch_bounds__(256) void linear16(float* dep, float* tin, float* tex, float* nrm, float v) {
    const size_t base = (size_t)blockIdx.x * NPIX;
    const float4 q = make_float4(v, v, v, v);
    for (int i = threadIdx.x; i < NPIX / 4; i += 256) {
        reinterpret_cast<float4*>(dep + base)[i] = q;
        reinterpret_cast<float4*>(tin + base)[i] = q;
    }
    for (int i = threadIdx.x; i < 3 * NPIX / 4; i += 256) {
        reinterpret_cast<float4*>(tex + 3 * base)[i] = q;
        reinterpret_cast<float4*>(nrm + 3 * base)[i] = q;
    }
}
int main() {
    const int bins = 1280;
    const size_t npx = (size_t)bins * NPIX;
    float *dep, *tin, *tex, *nrm, *junk;
    hipMalloc(&dep, npx * 4); hipMalloc(&tin, npx * 4); hipMalloc(&tex, npx * 12); hipMalloc(&nrm, npx * 12);
    hipMalloc(&junk, 512u << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 2; pat++) {
        for (int rep = 0; rep < 3; rep++) {
            float tot = 0;
            const int n = 20;
            for (int it = 0; it < n; it++) {
                hipMemsetAsync(junk, it, 512u << 20, 0);   // push the planes out of the Infinity Cache, as a pipeline would
                hipEventRecord(e0, 0);
                if (pat == 0) hipLaunchKernelGGL(stride12, dim3(bins), dim3(256), 0, 0, dep, tin, tex, nrm, (float)it);
                else hipLaunchKernelGGL(linear16, dim3(bins), dim3(256), 0, 0, dep, tin, tex, nrm, (float)it);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                tot += ms;
            }
            printf("%s: %.1f us per 82 MB of planes (%.2f TB/s)\n", pat == 0 ? "dword stores, 12-byte stride" : "16-byte linear stores",
                   tot / n * 1e3, npx * 32.0 / (tot / n * 1e-3) / 1e12);
        }
    }
    return 0;
}
