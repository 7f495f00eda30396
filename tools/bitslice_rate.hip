// bitslice_rate.hip -- microbenchmark behind DESIGN.md 3.2 "bit-sliced barcode DP": cost per DP cell of
//   (A) the packed-binary16 form (v_pk_add_f16 + v_pk_maximum3_f16 per two cells per lane) and
//   (B) the bit-sliced difference form (14 boolean 32-bit ops per 32 cells per lane)
// on a 42-column target with compile-time letters, one wave per workgroup, 16 workgroups per CU.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bitslice_rate.hip -o tools/bitslice_rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32;
constexpr int M = 42;
__device__ __forceinline__ h2 hmax3(h2 a, h2 b, h2 c) { return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c); }
__device__ __forceinline__ u32 bfi(u32 m, u32 a, u32 b) { return (m & a) | (~m & b); }

// letters of the target: a fixed pseudo-random pattern
__host__ __device__ constexpr int letter(int j) { return (j * 7 + (j >> 2) * 3 + 1) & 3; }

__global__ void __launch_bounds__(64, 4) k_f16(u32* out, int rows, u32 seed) {
    h2 h[M + 1];
#pragma unroll
    for (int j = 0; j <= M; ++j) h[j] = h2{(_Float16)(float)j, (_Float16)(float)j};
    u32 s = seed + threadIdx.x * 2654435761u + blockIdx.x;
    h2 colmax = h2{0, 0};
    for (int i = 1; i <= rows; ++i) {
        s = s * 1664525u + 1013904223u;
        h2 E[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) { const u32 v = ((s >> (x * 2)) & 1u) ? 0x42004200u : 0x3C003C00u; E[x] = __builtin_bit_cast(h2, v); }
        h2 left = h2{(_Float16)1.0f, (_Float16)1.0f}, diag = h[0];
#pragma unroll
        for (int j = 1; j <= M; ++j) {
            const h2 up = h[j];
            left = hmax3(diag + E[letter(j)], up, left);
            h[j] = left;
            diag = up;
        }
        colmax = __builtin_elementwise_maximum(colmax, left);
    }
    u32 acc = __builtin_bit_cast(u32, colmax);
#pragma unroll
    for (int j = 1; j <= M; ++j) acc ^= __builtin_bit_cast(u32, h[j]);
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

__global__ void __launch_bounds__(64, 4) k_bits(u32* out, int rows, u32 seed) {
    u32 h1[M + 1], h0[M + 1];                 // dh + 1 of every column, two bit planes, 32 alignments per lane
#pragma unroll
    for (int j = 0; j <= M; ++j) { h1[j] = 0u; h0[j] = 0xFFFFFFFFu; }
    u32 s = seed + threadIdx.x * 2654435761u + blockIdx.x;
    u32 acc = 0;
    for (int i = 1; i <= rows; ++i) {
        s = s * 1664525u + 1013904223u;
        const u32 c1 = s, c0 = s * 2246822519u;
        u32 E[4];
        E[0] = ~(c1 | c0); E[1] = ~c1 & c0; E[2] = c1 & ~c0; E[3] = c1 & c0;
        u32 a1 = 0u, a0 = 0xFFFFFFFFu;       // dv + 1 of column 0: 1
#pragma unroll
        for (int j = 1; j <= M; ++j) {
            const u32 eq = E[letter(j)], b1 = h1[j], b0 = h0[j];
            const u32 t2 = (b1 & b0) | (a1 & a0);
            const u32 n = a1 | b1;
            const u32 m1 = n | eq;
            const u32 m0 = eq | ~n | t2;
            const u32 p0 = m0 ^ b0, p1 = m1 ^ b1 ^ bfi(m0, 0u, b0);
            const u32 q0 = m0 ^ a0, q1 = m1 ^ a1 ^ bfi(m0, 0u, a0);
            h1[j] = q1; h0[j] = q0;
            a1 = p1; a0 = p0;
        }
        acc ^= a1 + a0;
    }
#pragma unroll
    for (int j = 1; j <= M; ++j) acc ^= h1[j] ^ (h0[j] >> 1);
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * 16, rows = 20000;
    u32* d; hipMalloc(&d, blocks * 64 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_f16, dim3(blocks), dim3(64), 0, 0, d, rows, 12345u);
            else hipLaunchKernelGGL(k_bits, dim3(blocks), dim3(64), 0, 0, d, rows, 12345u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        const double per_lane = which == 0 ? 2.0 : 32.0;
        const double cells = (double)blocks * 64 * per_lane * rows * M;
        const double simd_cycles = best * 1e-3 * 2.4e9 * cus * 4;           // SIMD-cycles available at 2.4 GHz
        printf("%s: %.3f ms, %.3e cell updates/s, %.4f SIMD-cycles per cell, %.2f cycles per wave-column\n",
               which == 0 ? "packed f16 (2 ops / 2 cells per lane)" : "bit-sliced (14 ops / 32 cells per lane)",
               best, cells / (best * 1e-3), simd_cycles / cells, simd_cycles / ((double)blocks * rows * M));
    }
    return 0;
}
