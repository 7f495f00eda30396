// bs_shape_probe.hip -- what can the SHAPE of k_bs_barcode issue?  (round 6)
//
// tools/valu_bank.hip launched 256-thread blocks of 60 VGPRs and no LDS: nothing pinned how many of them a CU got, and
// its figures at 3 / 4 / 8 waves per SIMD (2.77 / 2.52 / 2.28 cycles per v_bitop3_b32) are 2.04 x 4/3, 5/4, 9/8 -- the
// signature of one more block on some CU, not of an issue limit.  This probe pins the placement with LDS (a workgroup
// allocates 160 KB / workgroups-per-CU, so exactly that many fit), reads cycles from s_memtime and the clock from
// s_memrealtime (100 MHz), records where every workgroup ran (HW_ID / XCC_ID), and runs three things:
//   1. PEAK: eight independent v_bitop3_b32 chains with three VGPR sources per wave, for >= 20 ms, in the shapes
//      1 x 1024, 2 x 512, 4 x 256 (four waves per SIMD), 3 x 256 (three), 2 x 256 (two), 1 x 256 (one), 1 x 512 (two);
//   2. ROWS: the production row loops themselves -- QBS_2::rows of bs_static_generated.inc, i.e. the generated code of
//      PBC096's 5' family, 96 loop bodies -- on one 1024-thread workgroup per CU over LDS rows and a global plane buffer
//      filled with random letters, 6 barcodes per wave, 150 or 47 rows, exactly as a unit's barcode phase runs them, but
//      with nothing else in the kernel: no transposition, no shared passes, no barriers between units.
//      Variants: every wave its own barcodes (production: 16 different loop bodies hot per CU) / all waves the same
//      barcode (one loop body per CU: the instruction cache's share), with and without the priority rotation;
//   3. the same with only 8 or 12 of the 16 waves walking barcodes (two / three waves per SIMD).
// (-DQCAT_BS_NO_ROTATE: the row loops without their s_setprio rotation, the second binary of tools/bs_shape_probe.sh)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I qcat_amd/csrc tools/bs_shape_probe.hip -o tools/bs_shape_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <map>

#define QCAT_BS_PART 0
#include "rtc_prelude.inc"
#include "bs_static_generated.inc"

#define REP8(x) x x x x x x x x

// ---- 1. peak issue rate of v_bitop3_b32 in a pinned shape ------------------------------------------------------
// out[wg][wave][0..3] = s_memtime start, end, s_memrealtime start, end; out2[wg] = HW_ID | XCC_ID << 32
__global__ void k_peak(unsigned long long* out, unsigned long long* where, unsigned seed, int iters) {
    extern __shared__ unsigned pin[];
    if (threadIdx.x == 0) pin[0] = seed;                                  // (the allocation must be used to exist)
    __syncthreads();
    asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n v_mov_b32 v44, %0\n v_mov_b32 v45, %0\n v_mov_b32 v46, %0\n v_mov_b32 v47, %0\n"
                 "v_mov_b32 v48, %0\n v_mov_b32 v49, %0\n v_mov_b32 v50, %0\n v_mov_b32 v51, %0\n v_mov_b32 v52, %0\n v_mov_b32 v53, %0\n v_mov_b32 v54, %0\n v_mov_b32 v55, %0\n"
                 :: "v"(seed + threadIdx.x + pin[0]) : "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_bitop3_b32 v40, v40, v49, v50 bitop3:0x96\n v_bitop3_b32 v41, v41, v50, v51 bitop3:0x96\n v_bitop3_b32 v42, v42, v51, v48 bitop3:0x96\n v_bitop3_b32 v43, v43, v48, v49 bitop3:0x96\n"
                          "v_bitop3_b32 v44, v44, v53, v54 bitop3:0x96\n v_bitop3_b32 v45, v45, v54, v55 bitop3:0x96\n v_bitop3_b32 v46, v46, v55, v52 bitop3:0x96\n v_bitop3_b32 v47, v47, v52, v53 bitop3:0x96\n"
                          ::: "v40","v41","v42","v43","v44","v45","v46","v47");)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    unsigned r;
    asm volatile("v_xor_b32 %0, v40, v41\n v_xor_b32 %0, %0, v42\n v_xor_b32 %0, %0, v43\n v_xor_b32 %0, %0, v44\n v_xor_b32 %0, %0, v45\n v_xor_b32 %0, %0, v46\n v_xor_b32 %0, %0, v47" : "=v"(r));
    if (r == 0x12345678u) where[gridDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* o = out + ((size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 4;
        o[0] = t0; o[1] = t1; o[2] = r0; o[3] = r1;
    }
    if (threadIdx.x == 0) where[blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);
}

// ---- 2. the production row loops alone ---------------------------------------------------------------------------
namespace qk {
struct ProbeArgs {
    uint2* rplanes;                 // [workgroup][150][64]
    unsigned long long* out;        // [workgroup][16 waves][4]
    unsigned long long* where;
    int L, nb, reps, same, rotate, active;   // rows, barcodes per wave, repetitions, all waves one barcode, priority rotation, waves that work
    unsigned seed;
};
__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void __launch_bounds__(1024) k_rows(ProbeArgs a) {
    __shared__ uint4 s_rows[BS_MAX_ROWS * 64];                          // 153 600 B: one workgroup per CU, as in k_bs_barcode
    __shared__ u32 s_tail[3 * BS_NB * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
    uint2* __restrict__ rp = a.rplanes + (size_t)blockIdx.x * (BS_MAX_ROWS * 64);
    for (int q = tid; q < BS_MAX_ROWS * 64; q += 1024) {
        const unsigned h = mix(a.seed + blockIdx.x * 9973u + q);
        // dv + 1 in 0..3 with the planes' usual statistics does not matter for timing; any bits do
        s_rows[q] = make_uint4(h, mix(h), mix(h + 1), mix(h + 2));
        rp[q] = make_uint2(mix(h + 3), mix(h + 4));
    }
    for (int q = tid; q < 3 * BS_NB * 64; q += 1024) s_tail[q] = mix(q);
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0, r0 = 0, r1 = 0;
    u32 acc = 0;
    if (wave < a.active) {
        const BsRowArgs ra{s_rows, a.L, lane, true, nullptr, 4, rp, nullptr, 0, nullptr};
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int rep = 0; rep < a.reps; ++rep)
            for (int b = 0; b < a.nb; ++b) {
                u32 h1[24], h0[24], f[BS_ND];
#pragma unroll
                for (int j = 0; j < 24; ++j) { h1[j] = 0u; h0[j] = 0xFFFFFFFFu; }
#pragma unroll
                for (int q = 0; q < BS_ND; ++q) f[q] = 0u;
                const int kase = uni(a.same ? b : (b * 16 + wave) % 96);
                QBS_2::rows(kase, ra, h1, h0, f);
#pragma unroll
                for (int j = 0; j < 24; ++j) acc ^= h1[j] ^ h0[j];
#pragma unroll
                for (int q = 0; q < BS_ND; ++q) acc ^= f[q];
            }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
    }
    if (acc == 0x12345678u) a.where[gridDim.x + tid] = acc;             // (a sink the optimiser cannot see through)
    if (lane == 0) {
        unsigned long long* o = a.out + ((size_t)blockIdx.x * 16 + wave) * 4;
        o[0] = t0; o[1] = t1; o[2] = r0; o[3] = r1;
    }
    if (tid == 0) a.where[blockIdx.x] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32);
}
}  // namespace qk

#ifdef QCAT_BS_NO_ROTATE
#define ROT "no rotation"
#else
#define ROT "rotation"
#endif
static int g_cus = 256;

struct Stat { double cyc_mean, cyc_max, ghz, span_ms; int max_per_cu; };
static Stat reduce(const std::vector<unsigned long long>& o, const std::vector<unsigned long long>& where, int wgs, int waves, int active) {
    Stat s{0, 0, 0, 0, 0};
    double sum = 0, n = 0, ghz = 0;
    unsigned long long rmin = ~0ull, rmax = 0;
    for (int g = 0; g < wgs; ++g) {
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < waves && w < active; ++w) {
            const unsigned long long* e = &o[((size_t)g * waves + w) * 4];
            lo = std::min(lo, e[0]); hi = std::max(hi, e[1]);
            rmin = std::min(rmin, e[2]); rmax = std::max(rmax, e[3]);
            ghz += (double)(e[1] - e[0]) / ((double)(e[3] - e[2]) * 10.0);      // cycles per ns
            n += 1;
        }
        const double c = (double)(hi - lo);
        sum += c; s.cyc_max = std::max(s.cyc_max, c);
    }
    s.cyc_mean = sum / wgs; s.ghz = ghz / n; s.span_ms = (double)(rmax - rmin) * 1e-5;
    std::map<unsigned long long, int> per;
    for (int g = 0; g < wgs; ++g) {
        // HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 ...; CU = (xcc, se, sh, cu)
        const unsigned hw = (unsigned)where[g], xcc = (unsigned)(where[g] >> 32) & 0xF;
        per[((unsigned long long)xcc << 16) | ((hw >> 8) & 0xFF)]++;
    }
    for (auto& kv : per) s.max_per_cu = std::max(s.max_per_cu, kv.second);
    return s;
}

static void run_peak(const char* name, int threads, int per_cu, int iters) {
    const int wgs = g_cus * per_cu, waves = threads / 64;
    unsigned long long *d_out, *d_where;
    hipMalloc(&d_out, (size_t)wgs * waves * 32); hipMalloc(&d_where, (size_t)(wgs + 1024) * 8);
    const size_t lds = (size_t)(160 * 1024) / per_cu - 1024;                 // exactly per_cu workgroups fit
    hipFuncSetAttribute((const void*)k_peak, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int pass = 0; pass < 2; ++pass) { hipLaunchKernelGGL(k_peak, dim3(wgs), dim3(threads), lds, 0, d_out, d_where, 1u + pass, iters); hipDeviceSynchronize(); }
    std::vector<unsigned long long> o((size_t)wgs * waves * 4), w(wgs);
    hipMemcpy(o.data(), d_out, o.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(w.data(), d_where, w.size() * 8, hipMemcpyDeviceToHost);
    const Stat s = reduce(o, w, wgs, waves, waves);
    const double inst_per_simd = (double)iters * 64.0 * (waves * per_cu / 4.0);
    printf("PEAK %-26s %2d waves/SIMD  %7.2f ms  clock %.3f GHz  %.3f cycles/inst (mean wg)  %.3f (slowest wg)  max wgs on one CU %d\n", name, waves * per_cu / 4,
           s.span_ms, s.ghz, s.cyc_mean / inst_per_simd, s.cyc_max / inst_per_simd, s.max_per_cu);
    hipFree(d_out); hipFree(d_where);
    fflush(stdout);
}

static void run_rows(const char* name, int L, int nb, int reps, int same, int rotate, int active, uint2* d_rp) {
    const int wgs = g_cus;
    unsigned long long *d_out, *d_where;
    hipMalloc(&d_out, (size_t)wgs * 16 * 32); hipMalloc(&d_where, (size_t)(wgs + 1024) * 8);
    hipMemset(d_out, 0, (size_t)wgs * 16 * 32);
    qk::ProbeArgs a{d_rp, d_out, d_where, L, nb, reps, same, rotate, active, 12345u};
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(qk::k_rows, dim3(wgs), dim3(1024), 0, 0, a);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> o((size_t)wgs * 16 * 4), w(wgs);
    hipMemcpy(o.data(), d_out, o.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(w.data(), d_where, w.size() * 8, hipMemcpyDeviceToHost);
    const Stat s = reduce(o, w, wgs, 16, active);
    // instructions per row of a PBC096 barcode: 24 cells x 7 + the split deficit update 23 = 191 v_bitop3_b32 (+ the four masks of a row,
    // loop head, loads: counted by the disassembly, tools/isa_stats.py)
    const double rows_per_simd = (double)reps * nb * L * (active / 4.0);
    printf("ROWS %-44s L %3d  %2d waves  %7.2f ms  clock %.3f GHz  %.1f cycles/row/SIMD = %.3f cycles per 191-inst row inst (mean)  %.3f (slowest wg)  max wgs/CU %d\n",
           name, L, active, s.span_ms, s.ghz, s.cyc_mean / rows_per_simd, s.cyc_mean / rows_per_simd / 191.0, s.cyc_max / rows_per_simd / 191.0, s.max_per_cu);
    hipFree(d_out); hipFree(d_where);
    fflush(stdout);
}

int main(int argc, char** argv) {
    int dev = 0; hipGetDevice(&dev);
    hipDeviceGetAttribute(&g_cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int scale = argc > 1 ? atoi(argv[1]) : 1;
    printf("CUs %d\n", g_cus);
#ifndef QCAT_BS_NO_ROTATE
    const int it = 6000 * scale;                                             // 6000 x 64 inst per wave; 4 waves/SIMD at 2 cycles: 3.07 M cycles... x scale
    run_peak("1 x 1024 (production)", 1024, 1, it * 8);
    run_peak("2 x 512", 512, 2, it * 8);
    run_peak("4 x 256", 256, 4, it * 8);
    run_peak("3 x 256", 256, 3, it * 8);
    run_peak("2 x 256", 256, 2, it * 8);
    run_peak("1 x 512", 512, 1, it * 8);
    run_peak("1 x 256", 256, 1, it * 8);
    run_peak("1 x 768", 768, 1, it * 8);
    run_peak("2 x 1024 (8 waves/SIMD)", 1024, 2, it * 8);
#endif
    uint2* d_rp; hipMalloc(&d_rp, (size_t)g_cus * 150 * 64 * 8);
    for (int L : {150, 47}) {
        const int reps = (L == 150 ? 8 : 24) * scale;
        run_rows("own barcodes (production), " ROT, L, 6, reps, 0, 1, 16, d_rp);
        run_rows("one barcode for all waves, " ROT, L, 6, reps, 1, 1, 16, d_rp);
        run_rows("own barcodes, 12 waves, " ROT, L, 6, reps, 0, 1, 12, d_rp);
        run_rows("own barcodes, 8 waves, " ROT, L, 6, reps, 0, 1, 8, d_rp);
        run_rows("own barcodes, 4 waves, " ROT, L, 6, reps, 0, 1, 4, d_rp);
    }
    return 0;
}
