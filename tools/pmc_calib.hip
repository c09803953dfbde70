// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of the three hot-path kernels -- kernels
// of KNOWN traffic on buffers larger than the 256 MiB Infinity Cache (so that every byte really comes from / goes to HBM):
//   calib_read16      every lane reads 16 B of a contiguous stream (the decode's basis fragments)
//   calib_gather4     every lane reads 4 B at base + 4 * (row-major id), ids shared by neighbouring lanes like the emit kernel's
//                     vertex gathers: 64 lanes touch a 132-byte window of one row and the same window of the next row
//   calib_write16     every lane writes 16 B of a contiguous stream
//   calib_write4_12   every lane writes 4 B and a 12-byte triple at a 12-byte stride (the resolver's plane writer)
// bytes moved are printed as JSON; tools/pmc_calib_report.py divides the counters by them.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/pmc_calib tools/pmc_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f3 __attribute__((ext_vector_type(3), aligned(4)));
__global__ void calib_read16(const float4* __restrict__ p, size_t n, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 1.2345e30f) *sink = acc;
}
__global__ void calib_gather4(const float* __restrict__ p, size_t rows, int rowlen, float* sink) {
    // one wave = 64 "triangles" of 32 cells in row r: lanes read p[r][c0 + (lane >> 1) + (lane & 1)] and p[r + 1][same]
    float acc = 0.f;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const size_t waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    const int per_row = rowlen / 32;
    for (size_t w = wave; w < rows * per_row; w += waves) {
        const size_t r = w / per_row, c0 = (w % per_row) * 32;
        const size_t a = r * rowlen + c0 + (lane >> 1) + (lane & 1);
        acc += p[a] + p[a + rowlen];
    }
    if (acc == 1.2345e30f) *sink = acc;
}
__global__ void calib_write16(float4* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void calib_write4_12(float* __restrict__ d, float* __restrict__ t, size_t npix) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        d[i] = (float)i;
        *reinterpret_cast<f3*>(t + 3 * i) = (f3){1.f, 2.f, (float)i};
    }
}
int main() {
    const size_t MB = 1 << 20, NB = 1024 * MB;   // 1 GiB buffers
    float *a, *b, *sink;
    hipMalloc(&a, NB); hipMalloc(&b, NB); hipMalloc(&sink, 4);
    hipMemset(a, 0, NB); hipMemset(b, 0, NB);
    hipDeviceSynchronize();
    const int rowlen = 53216;
    const size_t rows = NB / 4 / rowlen - 1;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(calib_read16, dim3(4096), dim3(256), 0, 0, (const float4*)a, NB / 16, sink);
        hipLaunchKernelGGL(calib_gather4, dim3(4096), dim3(256), 0, 0, (const float*)b, rows, rowlen, sink);
        hipLaunchKernelGGL(calib_write16, dim3(4096), dim3(256), 0, 0, (float4*)a, NB / 16);
        hipLaunchKernelGGL(calib_write4_12, dim3(4096), dim3(256), 0, 0, b, b + NB / 16, NB / 16);
    }
    hipDeviceSynchronize();
    // bytes the kernels NEED from / hand to HBM: gather4 touches every row's first 32 * (rowlen / 32) floats + 1 per window ~ the whole row, twice
    // (row r as "own" row and as "next" row: the second touch of a 213 KB row comes 1,663 waves later -- L2 / MALL serve it)
    printf("{\"calib_read16\": %zu, \"calib_gather4_unique\": %zu, \"calib_gather4_requested\": %zu, \"calib_write16\": %zu, \"calib_write4_12\": %zu}\n",
           NB, (rows + 1) * (size_t)rowlen * 4, rows * (size_t)(rowlen / 32) * 64 * 2 * 4, NB, (NB / 16) * 16);
    return 0;
}
