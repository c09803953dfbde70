// Probe: what does ONE scattered vector-memory instruction cost a CU's address path (TA) on gfx950, by addressing form, width
// and lane pattern?  The emit kernel's TA is 69 % busy on eighteen dword gathers per thread; this measures whether
// another encoding of the same gathers (64-bit vaddr / saddr + 32-bit voffset / buffer offen; dword / x2 / x3 / x4) or another
// lane pattern is cheaper per instruction when every line is already in the vector L1.
// Every wave runs ITER rounds of G independent loads (addresses from registers, no dependent chain between rounds except the
// accumulation), 8 workgroups of 256 threads per CU; reported: shader cycles per wave-instruction per CU.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/gather_rate_probe tools/gather_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f3v __attribute__((ext_vector_type(3)));
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int G = 8;       // independent loads per round
constexpr int ITER = 64;   // rounds per wave

// MODE 0: global_load v, v[addr:addr+1], off     1: global_load v, voff, s[base:base+1]     2: buffer_load v, voff, s[rsrc], 0 offen
// WIDTH 1, 2, 3, 4 dwords per lane
template <int MODE, int WIDTH>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ base, const uint32_t* __restrict__ offs,
                                                     uint32_t bytes, float* __restrict__ sink, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    uint32_t off[G];
#pragma unroll
    for (int g = 0; g < G; g++) off[g] = offs[(size_t)g * 64 + lane];   // the same pattern for every wave: L1-resident window
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
    for (int it = 0; it < ITER; it++) {
        float v[G][4];
#pragma unroll
        for (int g = 0; g < G; g++) {
            const uint32_t o = off[g];
            if constexpr (MODE == 0) {
                const char* p = reinterpret_cast<const char*>(base) + o;
                if constexpr (WIDTH == 1) asm volatile("global_load_dword %0, %1, off" : "=v"(v[g][0]) : "v"(p));
                else if constexpr (WIDTH == 2) { f2v t; asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(t) : "v"(p)); v[g][0] = t.x + t.y; }
                else if constexpr (WIDTH == 3) { f3v t; asm volatile("global_load_dwordx3 %0, %1, off" : "=v"(t) : "v"(p)); v[g][0] = t.x + t.y + t.z; }
                else { f4v t; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(p)); v[g][0] = t.x + t.y + t.z + t.w; }
            } else if constexpr (MODE == 1) {
                if constexpr (WIDTH == 1) asm volatile("global_load_dword %0, %1, %2" : "=v"(v[g][0]) : "v"(o), "s"(base));
                else if constexpr (WIDTH == 2) { f2v t; asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(t) : "v"(o), "s"(base)); v[g][0] = t.x + t.y; }
                else if constexpr (WIDTH == 3) { f3v t; asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(t) : "v"(o), "s"(base)); v[g][0] = t.x + t.y + t.z; }
                else { f4v t; asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(t) : "v"(o), "s"(base)); v[g][0] = t.x + t.y + t.z + t.w; }
            } else {
                if constexpr (WIDTH == 1) asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(v[g][0]) : "v"(o), "s"(rs));
                else if constexpr (WIDTH == 2) { f2v t; asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(t) : "v"(o), "s"(rs)); v[g][0] = t.x + t.y; }
                else if constexpr (WIDTH == 3) { f3v t; asm volatile("buffer_load_dwordx3 %0, %1, %2, 0 offen" : "=v"(t) : "v"(o), "s"(rs)); v[g][0] = t.x + t.y + t.z; }
                else { f4v t; asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(t) : "v"(o), "s"(rs)); v[g][0] = t.x + t.y + t.z + t.w; }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int g = 0; g < G; g++) {
            asm volatile("" : "+v"(v[g][0]));
            acc += v[g][0];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 1.2345e30f) sink[0] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

struct Pattern { const char* name; std::vector<uint32_t> offs; };

int main() {
    const uint32_t bytes = 1u << 20;
    float* base; uint32_t* doffs; float* sink; unsigned long long* cyc;
    hipMalloc(&base, bytes); hipMemset(base, 0, bytes);
    hipMalloc(&doffs, G * 64 * 4); hipMalloc(&sink, 4);
    const int grid = 256 * 8;
    hipMalloc(&cyc, grid * 8);
    std::vector<Pattern> pats;
    srand(7);
    {   // lanes read consecutive dwords (width-strided so that x4 stays in bounds and aligned): 2-8 lines per instruction
        Pattern p{"consecutive", {}};
        for (int g = 0; g < G; g++) for (int l = 0; l < 64; l++) p.offs.push_back((uint32_t)(g * 2048 + l * 16));
        pats.push_back(p);
    }
    {   // every lane its own 128-byte line inside a 16 KiB window (L1-resident): 64 lines per instruction
        Pattern p{"one_line_per_lane_16KiB", {}};
        for (int g = 0; g < G; g++) for (int l = 0; l < 64; l++) p.offs.push_back((uint32_t)(((l * 2 + (g & 1)) * 128) + ((rand() & 7) * 16)));
        pats.push_back(p);
    }
    {   // mesh-like: a wave's lanes fall into a ~1.3 KiB run (what 64 neighbouring triangles' vertices span in one row): ~10 lines
        Pattern p{"mesh_like_run", {}};
        for (int g = 0; g < G; g++) for (int l = 0; l < 64; l++) p.offs.push_back((uint32_t)(g * 2048 + (((l * 5 + (rand() % 24)) & 0x7f) * 16)));
        pats.push_back(p);
    }
    {   // all lanes the same address
        Pattern p{"broadcast", {}};
        for (int g = 0; g < G; g++) for (int l = 0; l < 64; l++) p.offs.push_back((uint32_t)(g * 128));
        pats.push_back(p);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, const char* mode, int width, const Pattern& p) {
        hipMemcpy(doffs, p.offs.data(), G * 64 * 4, hipMemcpyHostToDevice);
        for (int w = 0; w < 2; w++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, base, doffs, bytes, sink, cyc);
        hipEventRecord(e0);
        const int K = 5;
        for (int w = 0; w < K; w++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, base, doffs, bytes, sink, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per CU: 8 workgroups x 4 waves x ITER x G wave-instructions per launch
        printf("{\"mode\": \"%s\", \"dwords\": %d, \"pattern\": \"%s\", \"us\": %.2f, \"ns_per_wave_instr_per_cu\": %.2f}\n", mode, width,
               p.name, ms * 1e3 / K, ms * 1e6 / K / (8.0 * 4 * ITER * G));
    };
    for (auto& p : pats) {
        run(gather_kernel<0, 1>, "vaddr64", 1, p); run(gather_kernel<1, 1>, "saddr_voff", 1, p); run(gather_kernel<2, 1>, "buffer_offen", 1, p);
        run(gather_kernel<1, 2>, "saddr_voff", 2, p); run(gather_kernel<1, 3>, "saddr_voff", 3, p); run(gather_kernel<1, 4>, "saddr_voff", 4, p);
        run(gather_kernel<2, 2>, "buffer_offen", 2, p); run(gather_kernel<2, 4>, "buffer_offen", 4, p);
    }
    // dword-only lane patterns (4-byte granularity): what does the address path coalesce?
    std::vector<Pattern> dp;
    auto add = [&](const char* name, auto f) {
        Pattern p{name, {}};
        for (int g = 0; g < G; g++) for (int l = 0; l < 64; l++) p.offs.push_back((uint32_t)(g * 4096 + f(l)));
        dp.push_back(p);
    };
    const int row = 367 * 4;   // the synthetic mesh's vertex-row pitch in bytes
    add("stride4", [](int l) { return l * 4; });
    add("stride8", [](int l) { return l * 8; });
    add("stride16", [](int l) { return l * 16; });
    add("stride32", [](int l) { return l * 32; });
    add("stride64", [](int l) { return l * 64; });
    add("pairs (l/2)*4", [](int l) { return (l / 2) * 4; });
    add("quads (l/4)*4", [](int l) { return (l / 4) * 4; });
    add("mesh p1: v00, v01 of consecutive cells", [](int l) { return ((l + 1) / 2) * 4; });
    add("mesh p2: v10 twice per cell", [row](int l) { return row + (l / 2) * 4; });
    add("mesh p3: v01 / v11 alternating rows", [row](int l) { return (l & 1) * row + (l / 2 + 1) * 4; });
    add("reversed stride4", [](int l) { return (63 - l) * 4; });
    add("random in 256 B", [](int) { return (rand() & 63) * 4; });
    add("random in 1 KiB", [](int) { return (rand() & 255) * 4; });
    add("random in 4 KiB", [](int) { return (rand() & 1023) * 4; });
    add("two rows by half-wave", [row](int l) { return (l / 32) * row + (l & 31) * 4; });
    add("two rows by 16 lanes", [row](int l) { return ((l / 16) & 1) * row + ((l & 15) + (l / 32) * 16) * 4; });
    add("stride4 + 4 B phase", [](int l) { return 4 + l * 4; });
    add("stride4 + 32 B phase", [](int l) { return 32 + l * 4; });
    add("stride4 + 60 B phase", [](int l) { return 60 + l * 4; });
    add("pairs + 36 B phase", [](int l) { return 36 + (l / 2) * 4; });
    add("quads tight (16 B), quads 256 B apart", [](int l) { return (l / 4) * 256 + (l & 3) * 4; });
    add("quads in a 32 B sector (stride 8), quads 256 B apart", [](int l) { return (l / 4) * 256 + (l & 3) * 8; });
    add("8-lane runs (32 B), runs 256 B apart", [](int l) { return (l / 8) * 256 + (l & 7) * 4; });
    add("16-lane runs (64 B aligned), runs 256 B apart", [](int l) { return (l / 16) * 256 + (l & 15) * 4; });
    add("16-lane runs (64 B, phase 32), runs 256 B apart", [](int l) { return 32 + (l / 16) * 256 + (l & 15) * 4; });
    add("32-lane runs (128 B aligned), 2 rows", [row](int l) { return (l / 32) * 2048 + (l & 31) * 4; });
    add("even lanes row 0, odd lanes row 0 + 64 B", [](int l) { return (l & 1) * 64 + (l / 2) * 4; });
    add("even lanes row 0, odd lanes + 128 B", [](int l) { return (l & 1) * 128 + (l / 2) * 4; });
    add("lane pairs swap (l^1)*4", [](int l) { return (l ^ 1) * 4; });
    add("random permutation of 64 consecutive dwords", [](int l) { static int perm[64]; static bool init = false; if (!init) { for (int i = 0; i < 64; i++) perm[i] = i; for (int i = 63; i > 0; i--) { int j = rand() % (i + 1); int t = perm[i]; perm[i] = perm[j]; perm[j] = t; } init = true; } return perm[l] * 4; });
    for (auto& p : dp) run(gather_kernel<1, 1>, "saddr_voff", 1, p);
    printf("{\"last_error\": \"%s\"}\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
