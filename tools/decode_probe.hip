// Development probe for decode_ring_kernel (csrc/fr_decode.hip) -- NOT part of the product library.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -o tools/decode_probe tools/decode_probe.hip
//   tools/decode_probe [B] > profiles/round3_decode_breakdown.json
//
// Two kinds of measurement, all in ONE process on one device (interleaved rounds, median reported):
//  (1) ablation builds of the kernel through its probe policy (bits: 1 = no MFMA, 2 = every A request hits the first 256
//      tiles (cache resident), 4 = no per-CU prologue, 8 = no A requests, 16 = no LDS B reads, 32 = no stores), each with
//      the non-temporal and the default-policy basis stream, timed back to back (basis may stay in the Infinity Cache) and
//      behind a 512 MiB flush write (basis comes from HBM, the state inside the decode -> render pipeline);
//  (2) a stamped build: every wave records s_memtime / s_memrealtime at kernel entry, after the ring is primed, after the
//      prologue, around every item's MFMA stream and its stores, and at exit, plus HW_ID / XCC_ID -- from which the
//      account of the kernel's duration is computed per SIMD: launch ramp, prologue, matrix-pipe utilisation inside the
//      item window, tail, and the clock the chip actually held (memtime ticks per 10 ns realtime tick).
// The stamps leave the kernel through a buffer nothing else reads (guide: "In-kernel stamps"); the stamped build's own
// duration is not quoted, only its shares.
#include "../3dfacerecon_amd/csrc/fr_decode.hip"
#include <algorithm>
#include <map>
#include <stdio.h>
#include <type_traits>
#include <vector>

namespace fr { int opt(Opt) { return 0; } }   // the launcher of the included TU is not used here

__device__ unsigned long long* g_stamp_out;   // [waves][16]

template <int BITS>
struct AblateProbe : fr::NoProbe {
    static constexpr int bits = BITS;
};

__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
__device__ __forceinline__ unsigned long long real_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

template <int BITS>
struct StampProbe {
    static constexpr int bits = BITS;
    unsigned long long t_begin, r_begin, t_primed, t_prologue, t_first, mfma_sum, store_sum, t_item, t_mid, t_last_end, t_last_mid;
    unsigned items;
    __device__ __forceinline__ void begin() {
        t_begin = stamp_now();
        r_begin = real_now();
        t_primed = t_prologue = t_first = mfma_sum = store_sum = t_item = t_mid = t_last_end = t_last_mid = 0;
        items = 0;
    }
    template <int ID>
    __device__ __forceinline__ void stamp() {
        if (ID == 0) t_primed = stamp_now();
        else t_prologue = stamp_now();
    }
    __device__ __forceinline__ void item_begin() {
        t_item = stamp_now();
        if (items == 0) t_first = t_item;
    }
    __device__ __forceinline__ void item_mfma_done() {
        t_mid = stamp_now();
        mfma_sum += t_mid - t_item;
        t_last_mid = t_mid;
    }
    __device__ __forceinline__ void item_end() {
        t_last_end = stamp_now();
        store_sum += t_last_end - t_mid;
        items++;
    }
    __device__ __forceinline__ void finish(int block, int wave, int) {
        const unsigned long long t_end = stamp_now(), r_end = real_now();
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
        if ((threadIdx.x & 63) == 0) {
            unsigned long long* o = g_stamp_out + ((size_t)block * 16 + wave) * 16;
            o[0] = t_begin; o[1] = r_begin; o[2] = t_primed; o[3] = t_prologue; o[4] = t_first; o[5] = mfma_sum;
            o[6] = store_sum; o[7] = items; o[8] = t_last_end; o[9] = t_end; o[10] = r_end; o[11] = hw; o[12] = xcc;
            o[13] = t_last_mid; o[14] = 1; o[15] = 0;
        }
    }
};

struct Ctx {
    fr::DecodeArgs a;
    size_t lds;
    int grid;
    hipStream_t st;
    void* flush;
    size_t flush_bytes;
};

static int g_prio = 1;
static int g_ring = 8;   // fragment-ring depth (8 = the product; 6 and 4 only for the unablated, unstamped kernel with priorities)
template <bool NT, class PR, bool PRIO, int R = 8>
static void launch_t(const Ctx& c) {
    auto k = fr::decode_ring_kernel<13, 2, R, 2, 16, 64, 4, NT, PR, PRIO>;
    static bool once = false;
    if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
    hipLaunchKernelGGL(k, dim3(c.grid), dim3(1024), c.lds, c.st, c.a);
}
template <bool NT, class PR>
static void launch(const Ctx& c) {
    if constexpr (std::is_same<PR, AblateProbe<0>>::value) {
        if (g_prio && g_ring == 6) return launch_t<NT, PR, true, 6>(c);
        if (g_prio && g_ring == 4) return launch_t<NT, PR, true, 4>(c);
    }
    if (g_prio) launch_t<NT, PR, true>(c);
    else launch_t<NT, PR, false>(c);
}

template <bool NT, class PR>
static double time_us(const Ctx& c, bool flush, int rounds) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<float> v;
    for (int i = 0; i < 3; i++) launch<NT, PR>(c);
    for (int r = 0; r < rounds; r++) {
        if (flush) (void)hipMemsetAsync(c.flush, r & 255, c.flush_bytes, c.st);
        (void)hipEventRecord(e0, c.st);
        launch<NT, PR>(c);
        (void)hipEventRecord(e1, c.st);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        v.push_back(ms * 1e3f);
    }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

template <bool NT, class PR>
static double time_b2b_us(const Ctx& c, int iters) {  // K launches between one event pair (no per-launch event gap)
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 5; i++) launch<NT, PR>(c);
    (void)hipEventRecord(e0, c.st);
    for (int i = 0; i < iters; i++) launch<NT, PR>(c);
    (void)hipEventRecord(e1, c.st);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / iters;
}

static double med(std::vector<double> v) {
    if (v.empty()) return 0;
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}
static double vmax(const std::vector<double>& v) { return v.empty() ? 0 : *std::max_element(v.begin(), v.end()); }
static double vmin(const std::vector<double>& v) { return v.empty() ? 0 : *std::min_element(v.begin(), v.end()); }

template <bool NT, int BITS>
static void stamped(const Ctx& c, bool flush, const char* name, unsigned long long* dstamps, bool last) {
    const int nw = c.grid * 16;
    std::vector<unsigned long long> h((size_t)nw * 16);
    // warm: >= 0.4 s of back-to-back launches so the clock has settled where the load holds it
    for (int i = 0; i < 6000; i++) launch<NT, AblateProbe<BITS>>(c);
    (void)hipStreamSynchronize(c.st);
    std::vector<double> clk_all, dur_all, prol_all, prime_all, util_all, ramp_all, tail_all, win_all, item_all, store_all, lastmid_all;
    // by XCD (all repetitions pooled): items per SIMD, item window, utilisation, first item's start and the last wave's exit
    // after the kernel's first wave started (realtime stamps: comparable across XCDs, 10 ns resolution)
    std::vector<double> x_items[16], x_win[16], x_util[16], x_exit[16], x_first[16], x_clk[16];
    double items_total = 0;
    for (int rep = 0; rep < 5; rep++) {
        (void)hipMemsetAsync(dstamps, 0, h.size() * 8, c.st);
        for (int i = 0; i < 20; i++) launch<NT, AblateProbe<BITS>>(c);
        if (flush) (void)hipMemsetAsync(c.flush, rep, c.flush_bytes, c.st);
        launch<NT, StampProbe<BITS>>(c);
        (void)hipStreamSynchronize(c.st);
        (void)hipMemcpy(h.data(), dstamps, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long r0 = ~0ull, r1 = 0;
        for (int w = 0; w < nw; w++) {
            const unsigned long long* o = &h[(size_t)w * 16];
            if (!o[14]) continue;
            r0 = std::min(r0, o[1]); r1 = std::max(r1, o[10]);
        }
        const double dur_us = (double)(r1 - r0) * 0.01;
        dur_all.push_back(dur_us);
        std::map<unsigned long long, std::vector<int>> simd;  // (xcc, se/sh/cu, simd) -> waves
        std::vector<double> clk;
        items_total = 0;
        for (int w = 0; w < nw; w++) {
            const unsigned long long* o = &h[(size_t)w * 16];
            if (!o[14]) continue;
            const double ticks = (double)(o[9] - o[0]), rt = (double)(o[10] - o[1]);
            if (rt > 100) clk.push_back(ticks / rt * 0.1);  // GHz: memtime ticks per 10 ns
            if (rt > 100) x_clk[o[12] & 0xF].push_back(ticks / rt * 0.1);
            x_exit[o[12] & 0xF].push_back((double)(o[10] - r0) * 0.01);
            if (o[7] && rt > 100) x_first[o[12] & 0xF].push_back((double)(o[1] - r0) * 0.01 + (double)(o[4] - o[0]) / (ticks / rt * 0.1) * 1e-3);
            simd[((o[12] & 0xF) << 32) | ((o[11] >> 4) & 0xFFF3)].push_back(w);  // cu/sh/se bits 8.. + simd bits 5:4 -> after >>4: bits 0,1 = simd
            ramp_all.push_back((double)(o[1] - r0) * 0.01);
            tail_all.push_back((double)(r1 - o[10]) * 0.01);
            prime_all.push_back((double)(o[2] - o[0]));
            prol_all.push_back((double)(o[3] - o[2]));
            if (o[7]) {
                item_all.push_back((double)o[5] / (double)o[7]);
                store_all.push_back((double)o[6] / (double)o[7]);
            }
            items_total += (double)o[7];
        }
        clk_all.push_back(med(clk));
        for (auto& kv : simd) {
            unsigned long long first = ~0ull, lastmid = 0, t0 = ~0ull, t9 = 0;
            double items = 0;
            for (int w : kv.second) {
                const unsigned long long* o = &h[(size_t)w * 16];
                t0 = std::min(t0, o[0]); t9 = std::max(t9, o[9]);
                if (o[7]) { first = std::min(first, o[4]); lastmid = std::max(lastmid, o[13]); }
                items += (double)o[7];
            }
            if (items == 0) continue;
            const double window = (double)(lastmid - first);
            util_all.push_back(items * 348.0 * 32.0 / window);
            win_all.push_back(window);
            const int xc = (int)((kv.first >> 32) & 0xF);
            x_items[xc].push_back(items);
            x_win[xc].push_back(window);
            x_util[xc].push_back(items * 348.0 * 32.0 / window);
            lastmid_all.push_back((double)(t9 - lastmid));
        }
    }
    const double clk = med(clk_all);
    printf("    {\"name\": \"%s\", \"nt\": %s, \"bits\": %d, \"hbm_sourced\": %s,\n", name, NT ? "true" : "false", BITS, flush ? "true" : "false");
    printf("     \"note\": \"stamped build: shares only, its own duration is not the kernel's\",\n");
    printf("     \"kernel_span_us_realtime\": %.2f, \"clock_GHz_median\": %.3f, \"items_total\": %.0f, \"simds_seen\": %zu,\n", med(dur_all), clk, items_total, util_all.size() / 5);
    printf("     \"wave_start_after_kernel_start_us\": {\"median\": %.2f, \"max\": %.2f},\n", med(ramp_all), vmax(ramp_all));
    printf("     \"ring_prime_cycles\": {\"median\": %.0f, \"max\": %.0f}, \"prologue_cycles\": {\"median\": %.0f, \"max\": %.0f, \"median_us\": %.2f},\n",
           med(prime_all), vmax(prime_all), med(prol_all), vmax(prol_all), med(prol_all) / clk * 1e-3);
    printf("     \"item_mfma_stream_cycles_per_item\": {\"median\": %.0f, \"min\": %.0f, \"max\": %.0f, \"ideal_4_waves_sharing_one_pipe\": %d},\n",
           med(item_all), vmin(item_all), vmax(item_all), 4 * 348 * 32);
    printf("     \"item_store_epilogue_cycles_per_item\": {\"median\": %.0f, \"max\": %.0f},\n", med(store_all), vmax(store_all));
    printf("     \"simd_item_window_cycles\": {\"median\": %.0f, \"max\": %.0f, \"median_us\": %.2f, \"max_us\": %.2f},\n", med(win_all), vmax(win_all),
           med(win_all) / clk * 1e-3, vmax(win_all) / clk * 1e-3);
    printf("     \"matrix_pipe_utilisation_inside_window\": {\"median\": %.3f, \"min\": %.3f, \"max\": %.3f},\n", med(util_all), vmin(util_all), vmax(util_all));
    printf("     \"simd_last_mfma_to_last_wave_exit_cycles\": {\"median\": %.0f, \"max\": %.0f},\n", med(lastmid_all), vmax(lastmid_all));
    printf("     \"by_xcd\": [\n");
    {
        int lastx = -1;
        for (int xc = 0; xc < 16; xc++) if (!x_items[xc].empty()) lastx = xc;
        for (int xc = 0; xc <= lastx; xc++) {
            if (x_items[xc].empty()) continue;
            double isum = 0;
            for (double v : x_items[xc]) isum += v;
            const double ck = med(x_clk[xc]);
            printf("       {\"xcd\": %d, \"simds\": %zu, \"items_per_simd\": %.2f, \"clock_GHz\": %.3f, \"first_item_begin_us\": {\"median\": %.2f, \"max\": %.2f}, "
                   "\"window_us\": {\"median\": %.2f, \"max\": %.2f}, \"utilisation\": {\"median\": %.3f, \"min\": %.3f}, \"wave_exit_us\": {\"median\": %.2f, \"max\": %.2f}}%s\n",
                   xc, x_items[xc].size() / 5, isum / x_items[xc].size(), ck, med(x_first[xc]), vmax(x_first[xc]), med(x_win[xc]) / ck * 1e-3,
                   vmax(x_win[xc]) / ck * 1e-3, med(x_util[xc]), vmin(x_util[xc]), med(x_exit[xc]), vmax(x_exit[xc]), xc == lastx ? "" : ",");
        }
    }
    printf("     ],\n");
    printf("     \"wave_exit_before_kernel_end_us\": {\"median\": %.2f, \"max\": %.2f}}%s\n", med(tail_all), vmax(tail_all), last ? "" : ",");
}

int main(int argc, char** argv) {
    using namespace fr;
    // decode_probe [B] [N] [pitched rows 0|1] [quick 0|1|2 (2 = ring-depth / priority A/B + stamps by XCD)] [wave priorities 0|1] [pitch/prio A/B 0|1]
    const int B = argc > 1 ? atoi(argv[1]) : 64, N = argc > 2 ? atoi(argv[2]) : 53215, ns = 199, ne = 29;
    const int pitched = argc > 3 ? atoi(argv[3]) : 1;
    const bool quick = argc > 4 && atoi(argv[4]) == 1;
    const bool saw_ab = argc > 4 && atoi(argv[4]) == 2;
    g_prio = argc > 5 ? atoi(argv[5]) : 1;
    const size_t pb = fr_packed_basis_bytes(N, ns, ne);
    void *packed, *params, *out, *flush;
    unsigned long long* dstamps;
    const size_t flush_bytes = (size_t)512 << 20;
    (void)hipMalloc(&packed, pb);
    (void)hipMalloc(&params, (size_t)B * 235 * 4);
    (void)hipMalloc(&out, (size_t)B * 3 * (N + 64) * 4);
    (void)hipMalloc(&flush, flush_bytes);
    (void)hipMalloc(&dstamps, (size_t)256 * 16 * 16 * 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_out), &dstamps, sizeof(dstamps));
    std::vector<float> h(pb / 4);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = (float)(x >> 8) * (1.0f / 16777216.0f) - 0.5f; }
    (void)hipMemcpy(packed, h.data(), pb, hipMemcpyHostToDevice);
    (void)hipMemcpy(params, h.data(), (size_t)B * 235 * 4, hipMemcpyHostToDevice);

    Ctx c;
    const size_t tiles = (size_t)tiles_of(N), G = 15;
    c.a.params = (const float*)params;
    c.a.A = (const float4*)packed;
    c.a.mu_p = (const float*)(c.a.A + tiles * G * 3 * 64);
    c.a.R_override = nullptr;
    c.a.out = (float*)out;
    c.a.B = B; c.a.N = N; c.a.ns = ns; c.a.ne = ne; c.a.b0 = 0; c.a.halves = B > 32 ? 2 : 1; c.a.im_size = 200.f; c.a.pitch = pitched ? (N + 31) & ~31 : N;
    c.lds = G * KGROUP * 16 * sizeof(float4) + 64 * 12 * sizeof(float) + 64 * 3 * 2 * sizeof(double);
    c.grid = std::min(fr_device_cu_count(), (int)((tiles + 16 / c.a.halves - 1) / (16 / c.a.halves)));
    c.st = 0;
    c.flush = flush;
    c.flush_bytes = flush_bytes;

    if (argc > 6 && atoi(argv[6])) {   // interleaved A/B of (pitch, prio, late pose) in one process: median of 9 rounds each
        printf("{\"ab_after_512MiB_flush_us\": {\n");
        double res[4][9];
        for (int r = 0; r < 9; r++)
            for (int v = 0; v < 4; v++) {
                c.a.pitch = (v & 1) ? (N + 31) & ~31 : N; g_prio = (v >> 1) & 1;
                res[v][r] = time_us<true, AblateProbe<0>>(c, true, 7);
            }
        for (int v = 0; v < 4; v++) {
            std::sort(res[v], res[v] + 9);
            printf("  \"pitched=%d prio=%d\": {\"median\": %.2f, \"min\": %.2f, \"max\": %.2f}%s\n", v & 1, (v >> 1) & 1,
                   res[v][4], res[v][0], res[v][8], v == 3 ? "" : ",");
        }
        printf("}}\n");
        return 0;
    }
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    printf("{\"device\": \"%s\", \"cus\": %d, \"B\": %d, \"N\": %d, \"row_pitch\": %d, \"wave_priorities\": %d, \"items\": %zu, \"mfma_per_item\": 348,\n", prop.gcnArchName, c.grid, B, N, c.a.pitch, g_prio, tiles * 2);
    printf(" \"mfma_floor_us_at_2.4GHz\": {\"average_6.5_items_per_simd\": %.2f, \"worst_simd_7_items\": %.2f},\n",
           tiles * 2 * 348.0 * 32.0 / 1024.0 / 2.4e3, 7 * 348.0 * 32.0 / 2.4e3);
    if (saw_ab) {   // interleaved A/B in one process: ring depth and wave priorities; then the stamped kernel by XCD
        // (round 6, r6a / r6b: the store-aware waits and the balanced tile walk were A/B'd through this mode; both lost and left the tree)
        struct V { const char* name; int ring, prio; };
        const V vs[] = {{"ring=8 prio=1 (product)", 8, 1}, {"ring=6 prio=1", 6, 1}, {"ring=4 prio=1", 4, 1}, {"ring=8 prio=0", 8, 0}};
        const int nv = sizeof(vs) / sizeof(vs[0]);
        printf(" \"ab_us\": {\n");
        std::vector<double> r[8][4];
        for (int round = 0; round < 5; round++)
            for (int v = 0; v < nv; v++) {
                g_ring = vs[v].ring; g_prio = vs[v].prio;
                r[v][0].push_back(time_us<true, AblateProbe<0>>(c, true, 9));
                r[v][1].push_back(time_b2b_us<true, AblateProbe<0>>(c, 200));
                r[v][2].push_back(time_us<false, AblateProbe<0>>(c, true, 9));
                r[v][3].push_back(time_b2b_us<false, AblateProbe<0>>(c, 200));
            }
        for (int v = 0; v < nv; v++)
            printf("  \"%s\": {\"nt_after_512MiB_flush\": {\"median\": %.2f, \"min\": %.2f, \"max\": %.2f}, \"nt_back_to_back\": {\"median\": %.2f, \"min\": %.2f}, "
                   "\"cached_after_512MiB_flush\": {\"median\": %.2f, \"min\": %.2f}, \"cached_back_to_back\": {\"median\": %.2f, \"min\": %.2f}}%s\n", vs[v].name,
                   med(r[v][0]), vmin(r[v][0]), vmax(r[v][0]), med(r[v][1]), vmin(r[v][1]), med(r[v][2]), vmin(r[v][2]), med(r[v][3]), vmin(r[v][3]), v == nv - 1 ? "" : ",");
        printf(" },\n \"stamps\": [\n");
        g_ring = 8; g_prio = 1;
        stamped<true, 0>(c, true, "product configuration (nt basis), basis from HBM", dstamps, false);
        stamped<false, 0>(c, false, "default-policy basis back to back (Infinity-Cache resident)", dstamps, true);
        printf(" ]}\n");
        return 0;
    }
    printf(" \"timing_us\": {\n");
#define ROW(name, PR)                                                                                                   \
    printf("  \"%s\": {\"nt_back_to_back\": %.1f, \"nt_event_single\": %.1f, \"nt_after_512MiB_flush\": %.1f, "     \
           "\"cached_back_to_back\": %.1f, \"cached_event_single\": %.1f, \"cached_after_512MiB_flush\": %.1f}%s\n", \
           name, time_b2b_us<true, PR>(c, 200), time_us<true, PR>(c, false, 41), time_us<true, PR>(c, true, 41),        \
           time_b2b_us<false, PR>(c, 200), time_us<false, PR>(c, false, 41), time_us<false, PR>(c, true, 41),
    ROW("full", AblateProbe<0>) ",");
    if (quick) {
        ROW("no_mfma(1)", AblateProbe<1>) ",");
        ROW("no_stores(32)", AblateProbe<32>) ",");
        ROW("stores_as_whole_128B_lines_timing_only(64)", AblateProbe<64>) ",");
        ROW("no_mfma_stores_as_whole_lines(65)", AblateProbe<65>) ",");
        ROW("full_again", AblateProbe<0>) "");
        printf(" },\n \"stamps\": [\n");
        stamped<true, 0>(c, true, "product configuration (nt basis), basis from HBM", dstamps, true);
        printf(" ]}\n");
        return 0;
    }
    ROW("no_mfma(1)", AblateProbe<1>) ",");
    ROW("A_from_256_resident_tiles(2)", AblateProbe<2>) ",");
    ROW("no_prologue(4)", AblateProbe<4>) ",");
    ROW("no_A_requests(8)", AblateProbe<8>) ",");
    ROW("no_lds_B_reads(16)", AblateProbe<16>) ",");
    ROW("no_stores(32)", AblateProbe<32>) ",");
    ROW("no_stores_no_prologue(36)", AblateProbe<36>) ",");
    ROW("mfma_only: no A requests, no stores, no prologue, no LDS B (60)", AblateProbe<60>) ",");
    ROW("loads_only: no MFMA, no stores, no prologue (37)", AblateProbe<37>) ",");
    ROW("full_again", AblateProbe<0>) "");
    printf(" },\n \"stamps\": [\n");
    stamped<true, 0>(c, true, "product configuration (nt basis), basis from HBM", dstamps, false);
    stamped<true, 0>(c, false, "nt basis, back to back", dstamps, false);
    stamped<false, 0>(c, false, "default-policy basis, back to back (Infinity-Cache resident)", dstamps, false);
    stamped<false, 2>(c, false, "A requests confined to 256 resident tiles (L2)", dstamps, false);
    stamped<true, 32>(c, true, "no stores, basis from HBM", dstamps, false);
    stamped<true, 8>(c, false, "no A requests (matrix pipe + LDS + stores only)", dstamps, true);
    printf(" ]}\n");
    return 0;
}
