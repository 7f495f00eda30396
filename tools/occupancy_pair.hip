// occupancy_pair.hip -- the bounded occupancy experiment for the bit-sliced barcode row loops (round 4).
// Question: the nominal-region class (47 rows) needs 48 KB of letter planes, so TWO 16-wave workgroups fit a CU; would eight
// waves per SIMD (64 VGPRs: the 31 own columns of a barcode split over a wave PAIR, dv handed from the front wave to the back
// wave through an LDS ring) beat the production shape (one wave per barcode, 31 columns in 128 VGPRs, four waves per SIMD)?
//   k_whole: 16 waves per CU, every wave walks its barcodes over the unit's 47 rows: 31 cells of bs_cell + bs_deficit per row,
//            two rows per pass, priority rotation -- the production row loop (kernels_bitslice.inc: bs_rows).
//   k_pair : 32 waves per CU (two workgroups), wave 2k = columns 0..15, wave 2k + 1 = columns 16..30 + the deficit counter; the
//            front wave stores (a1, a0) of a row into a ring of RING rows and publishes its progress (release), the back wave
//            waits for it (acquire), and the front wave waits for the ring slot to be free again.
// Same cells per CU in both kernels; the letter planes are random; the result words keep the optimiser honest.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/occupancy_pair.hip -o /tmp/occupancy_pair && /tmp/occupancy_pair
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef uint32_t u32;
constexpr int L = 47, C = 31, CF = 16, CB = C - CF, NF = 8, RING = 4;
#define QB3(X, Y, Z, ...) __builtin_amdgcn_bitop3_b32((X), (Y), (Z), (unsigned)([](unsigned x, unsigned y, unsigned z) constexpr { return (__VA_ARGS__) & 0xFFu; }(0xF0u, 0xCCu, 0xAAu)))
#define QT3(X, Y, Z, TABLE) __builtin_amdgcn_bitop3_b32((X), (Y), (Z), (unsigned)(TABLE))
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__host__ __device__ constexpr int letter(int j) { return (j * 7 + (j >> 2) * 3 + 1) & 3; }

// the production cell and deficit update (kernels_bitslice.inc: bs_cell, bs_deficit)
__device__ __forceinline__ void cell(u32 neq, u32& a1, u32& a0, u32& b1, u32& b0) {
    const u32 n5 = QT3(a1, a0, neq, 0xd5);
    const u32 n6 = QT3(b1, a1, n5, 0x5b);
    const u32 n7 = QT3(a0, b1, b0, 0x73);
    const u32 p0 = QT3(b0, b1, n6, 0x16);
    const u32 q1 = QT3(n5, a1, n7, 0x31);
    const u32 p1 = QT3(n5, b0, n6, 0xa1);
    const u32 q0 = QT3(b0, p0, a0, 0x16);
    a1 = p1; a0 = p0; b1 = q1; b0 = q0;
}
__device__ __forceinline__ void deficit(u32 (&f)[NF], u32 a1, u32 a0) {
    const u32 t1 = QB3(f[2], f[3], f[4], x | y | z), t2 = QB3(f[5], f[6], f[7], x | y | z);
    const u32 g1 = QB3(t1, t2, f[1], x | y | z), g0 = QB3(t1, t2, f[0], x | y | z);
    const u32 e1 = a1 & g1;
    const u32 same = QB3(a1, g1, g1, ~(x ^ y));
    const u32 pick = QB3(a1, g0, a0, (x & y) | (~x & z));
    const u32 e0 = QB3(same, a0 & g0, pick, (x & y) | (~x & z));
    u32 c = QB3(f[0], e0, e0, x & ~y);
    f[0] = QB3(f[0], e0, e0, ~(x ^ y));
#pragma unroll
    for (int k = 1; k < NF; ++k) {
        const u32 s = QB3(f[k], e1, c, x ^ y ^ z);
        if (k + 1 < NF) c = QB3(f[k], e1, c, (x & y) | (x & z) | (y & z));
        f[k] = s;
    }
}
struct Masks { u32 e[4]; };
__device__ __forceinline__ Masks masks(u32 c1, u32 c0) {
    Masks m;
    m.e[0] = c1 | c0; m.e[1] = QB3(c1, c0, c0, x | ~y); m.e[2] = QB3(c1, c0, c0, ~x | y); m.e[3] = QB3(c1, c0, c0, ~(x & y));
    return m;
}
__device__ __forceinline__ void setprio(int p) {
    switch (p) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}

__device__ __forceinline__ void fill_rows(uint4* s_rows, u32 seed) {
    for (int q = (int)threadIdx.x; q < L * 64; q += (int)blockDim.x) {
        u32 s = seed + (u32)q * 2654435761u + blockIdx.x * 40503u;
        uint4 v;
        s = s * 1664525u + 1013904223u; v.x = s;
        s = s * 1664525u + 1013904223u; v.y = s;
        s = s * 1664525u + 1013904223u; v.z = s & (s >> 3);
        s = s * 1664525u + 1013904223u; v.w = s | v.z;
        s_rows[q] = v;
    }
    __syncthreads();
}

__global__ void __launch_bounds__(1024) k_whole(u32* out, int nbar, u32 seed) {
    extern __shared__ uint4 s_rows[];
    fill_rows(s_rows, seed);
    const int lane = threadIdx.x & 63;
    const int rank = (int)(threadIdx.x >> 8);
    u32 acc = 0;
    for (int b = 0; b < nbar; ++b) {
        u32 h1[C], h0[C], f[NF];
#pragma unroll
        for (int j = 0; j < C; ++j) { h1[j] = 0u; h0[j] = 0xFFFFFFFFu; }
#pragma unroll
        for (int k = 0; k < NF; ++k) f[k] = 0u;
        int i = 0;
        {   // 47 rows: the first one alone
            const uint4 v = s_rows[lane];
            const Masks m = masks(v.x, v.y);
            u32 a1 = v.z, a0 = v.w;
#pragma unroll
            for (int j = 0; j < C; ++j) cell(m.e[letter(j)], a1, a0, h1[j], h0[j]);
            deficit(f, a1, a0);
            i = 1;
        }
        uint4 v = s_rows[i * 64 + lane], w = s_rows[(i + 1) * 64 + lane];
        int pr = rank;
        for (; i < L; i += 2) {
            setprio(pr);
            pr = pr + 1 == 4 ? 0 : pr + 1;
            const int nx = min(i + 2, L - 2);
            const uint4 nv = s_rows[nx * 64 + lane], nw = s_rows[(nx + 1) * 64 + lane];
            const Masks mv = masks(v.x, v.y), mw = masks(w.x, w.y);
            u32 a1 = v.z, a0 = v.w, b1 = w.z, b0 = w.w;
            cell(mv.e[letter(0)], a1, a0, h1[0], h0[0]);
#pragma unroll
            for (int j = 1; j < C; ++j) {
                cell(mv.e[letter(j)], a1, a0, h1[j], h0[j]);
                cell(mw.e[letter(j - 1)], b1, b0, h1[j - 1], h0[j - 1]);
            }
            cell(mw.e[letter(C - 1)], b1, b0, h1[C - 1], h0[C - 1]);
            deficit(f, a1, a0);
            deficit(f, b1, b0);
            v = nv; w = nw;
        }
#pragma unroll
        for (int k = 0; k < NF; ++k) acc ^= f[k] << k;
#pragma unroll
        for (int j = 0; j < C; ++j) acc += h1[j] ^ (h0[j] >> 1);
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// the wave pair.  Rows are numbered through the barcodes (g = b * L + i) so that the ring and the progress words never reset.
template <int TWO_ROWS>
__global__ void __launch_bounds__(1024, 8) k_pair(u32* out, int nbar, u32 seed) {
    extern __shared__ uint4 s_rows[];                                   // L rows, then the rings, then the progress words
    uint2* const rings = reinterpret_cast<uint2*>(s_rows + L * 64);     // [pair][RING][64]
    u32* const prog = reinterpret_cast<u32*>(rings + 8 * RING * 64);    // [pair][2]: rows the front wave has stored, rows the back wave has read
    if (threadIdx.x < 16) prog[threadIdx.x] = 0u;
    fill_rows(s_rows, seed);
    const int lane = threadIdx.x & 63;
    const int wave = uni((int)(threadIdx.x >> 6)), pair = wave >> 1;
    uint2* const ring = rings + pair * RING * 64 + lane;
    u32* const front_done = prog + pair * 2, * const back_done = prog + pair * 2 + 1;
    u32 acc = 0;
    const int total = nbar * L;
    constexpr int STEP = TWO_ROWS ? 2 : 1;                              // rows per hand-over
    if ((wave & 1) == 0) {                                              // columns 0 .. CF - 1
        u32 h1[CF], h0[CF];
        int i = 0;
#pragma nounroll
        for (int g0 = 0; g0 < total; g0 += STEP) {
            // the ring slots of this step must have been read by the back wave
            while (uni((int)__hip_atomic_load(back_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) + RING < g0 + STEP) __builtin_amdgcn_s_sleep(1);
#pragma nounroll
            for (int g = g0; g < g0 + STEP; ++g) {
                if (i == 0) {
#pragma unroll
                    for (int j = 0; j < CF; ++j) { h1[j] = 0u; h0[j] = 0xFFFFFFFFu; }
                }
                const uint4 v = s_rows[i * 64 + lane];
                const Masks m = masks(v.x, v.y);
                u32 a1 = v.z, a0 = v.w;
#pragma unroll
                for (int j = 0; j < CF; ++j) cell(m.e[letter(j)], a1, a0, h1[j], h0[j]);
                ring[(g % RING) * 64] = uint2{a1, a0};
                if (++i == L) {
                    i = 0;
#pragma unroll
                    for (int j = 0; j < CF; ++j) acc += h1[j] ^ (h0[j] >> 1);
                }
            }
            __hip_atomic_store(front_done, (u32)(g0 + STEP), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {                                                            // columns CF .. C - 1 and the deficit counter
        u32 h1[CB], h0[CB], f[NF];
        int i = 0;
#pragma nounroll
        for (int g0 = 0; g0 < total; g0 += STEP) {
            while (uni((int)__hip_atomic_load(front_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < g0 + STEP) __builtin_amdgcn_s_sleep(1);
#pragma nounroll
            for (int g = g0; g < g0 + STEP; ++g) {
                if (i == 0) {
#pragma unroll
                    for (int j = 0; j < CB; ++j) { h1[j] = 0u; h0[j] = 0xFFFFFFFFu; }
#pragma unroll
                    for (int k = 0; k < NF; ++k) f[k] = 0u;
                }
                const uint4 v = s_rows[i * 64 + lane];
                const Masks m = masks(v.x, v.y);
                const uint2 in = ring[(g % RING) * 64];
                u32 a1 = in.x, a0 = in.y;
                if (g + 1 == g0 + STEP)                                 // (the step's last slot is in registers: the front wave may refill)
                    __hip_atomic_store(back_done, (u32)(g0 + STEP), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
                for (int j = 0; j < CB; ++j) cell(m.e[letter(CF + j)], a1, a0, h1[j], h0[j]);
                deficit(f, a1, a0);
                if (++i == L) {
                    i = 0;
#pragma unroll
                    for (int k = 0; k < NF; ++k) acc ^= f[k] << k;
#pragma unroll
                    for (int j = 0; j < CB; ++j) acc += h1[j] ^ (h0[j] >> 1);
                }
            }
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int nbar = 48;                                              // barcodes per wave (k_whole) / per wave pair (k_pair)
    u32* d; (void)hipMalloc(&d, (size_t)cus * 2 * 1024 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t lds_whole = 100 * 1024;                              // (one workgroup per CU, like the production kernel's 157 KB)
    const size_t lds_pair = (size_t)L * 64 * 16 + 8 * RING * 64 * 8 + 64;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_whole), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_whole);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pair);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pair);
    const double cells = (double)cus * 16 * nbar * L * C;             // wave-cells (2048 alignments each), the same for every kernel
    float ref = 0;
    for (int which = 0; which < 3; ++which) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_whole, dim3(cus), dim3(1024), lds_whole, 0, d, nbar, 12345u);
            else if (which == 1) hipLaunchKernelGGL(k_pair<0>, dim3(cus * 2), dim3(1024), lds_pair, 0, d, nbar, 12345u);
            else hipLaunchKernelGGL(k_pair<1>, dim3(cus * 2), dim3(1024), lds_pair, 0, d, nbar, 12345u);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        hipError_t err = hipGetLastError();
        if (which == 0) ref = best;
        const char* name = which == 0 ? "whole: 16 waves/CU, 31 columns per wave, 4 waves/SIMD       "
                         : which == 1 ? "pair : 32 waves/CU, 16 + 15 columns, hand-over every row    "
                                      : "pair : 32 waves/CU, 16 + 15 columns, hand-over every 2 rows ";
        printf("%s %.3f ms  %.3f SIMD-cycles per wave-cell at 2.4 GHz  (%+.1f %% against whole)  %s\n", name, best,
               best * 1e-3 * 2.4e9 * cus * 4 / cells, (ref / best - 1.0) * 100.0, err == hipSuccess ? "" : hipGetErrorString(err));
    }
    return 0;
}
