// mix_rate.hip -- what does one DP column cost?  Compares the candidate instruction mixes (A-H) for the
// score lookup of the packed kernels at 4 waves/SIMD:
//   A: v_perm_b32 + v_add_u32 + 2 x v_pk_max_u16                (all VALU)
//   B: ds_bpermute_b32 (LDS crossbar) + v_add_u32 + 2 x v_pk_max_u16
// Each "column" keeps the real dependency shape: w -> d = diag + w -> m = max(d, up) -> left = max(m, left).
#include <hip/hip_runtime.h>
#include <cstdio>
#define NCOL 44
#define ROWS 2048
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ inline us2 U(unsigned x) { return __builtin_bit_cast(us2, x); }
__device__ inline unsigned X(us2 x) { return __builtin_bit_cast(unsigned, x); }

template <int MODE>
__global__ void __launch_bounds__(64, 4) k(unsigned* out, const unsigned* in, unsigned seed) {
    unsigned tbl[NCOL]; us2 h[NCOL + 1];
    for (int j = 0; j < NCOL; ++j) { tbl[j] = in[(j * 64 + threadIdx.x) & 1023] | 0x00010001u; asm volatile("" : "+v"(tbl[j])); h[j] = U(j * 0x00010001u); }
    h[NCOL] = U(0);
    unsigned codes[NCOL];
    for (int j = 0; j < NCOL; ++j) codes[j] = __builtin_amdgcn_readfirstlane(in[j] & 7);
    unsigned zero = __builtin_amdgcn_readfirstlane(in[100] & 0);
    unsigned special = seed * 3 + 1;
    asm volatile("" : "+v"(special));
    for (int i = 0; i < ROWS; ++i) {
        unsigned q = (in[(i + threadIdx.x) & 1023] + i);
        unsigned sel = (q & 0x00030003u) | 0x0C000C00u;
        int addr = (int)((q & 63u) << 2);
        us2 left = U(i * 0x00010001u), carry = left;
        if (MODE == 3) asm volatile("s_set_gpr_idx_on 0, 0x1");
        unsigned P0 = q & 0x00030003u, P1 = P0 + 0x00010001u, P2 = P0 ^ 0x00020002u, P3 = P0 + 0x00030003u,
                 P4 = P0 + 0x00010001u * (i & 3), P5 = 0x00020002u, P6 = 0x00020002u, P7 = 0;
#pragma unroll
        for (int j = 0; j < NCOL; ++j) {
            unsigned w;
            us2 up = h[j + 1];
            us2 d;
            if (MODE == 2) {
                // C: VGPR-indexing mode: d = P[t_j] + diag in ONE full-rate add; P[0..7] pinned to v40..v47
                unsigned dd;
                const unsigned tj = codes[j];
                asm volatile("s_set_gpr_idx_on %[t], 0x1\n\tv_add_u32 %[d], v40, %[c]\n\ts_set_gpr_idx_off"
                             : [d] "=&v"(dd), "+{v40}"(P0), "+{v41}"(P1), "+{v42}"(P2), "+{v43}"(P3),
                               "+{v44}"(P4), "+{v45}"(P5), "+{v46}"(P6), "+{v47}"(P7)
                             : [t] "s"(tj), [c] "v"(X(carry)));
                d = U(dd);
            } else if (MODE == 4) {
                // F: static target letters (what per-kit code generation would give): d = diag + P[t_j]
                const unsigned pj = (j % 5 == 0) ? P0 : (j % 5 == 1) ? P1 : (j % 5 == 2) ? P2 : (j % 5 == 3) ? P3 : P4;
                d = U(X(carry) + pj);
            } else if (MODE == 5 || MODE == 6) {
                // G/H: fp16 lanes (biased scores are small exact integers): d = diag + w in v_pk_add_f16, then ONE
                // v_pk_maximum3_f16 for max(d, up, left).  G looks w up with v_perm, H has static letters.
                unsigned w5;
                if (MODE == 5) w5 = __builtin_amdgcn_perm(special, tbl[j], sel);
                else w5 = (j % 5 == 0) ? P0 : (j % 5 == 1) ? P1 : (j % 5 == 2) ? P2 : (j % 5 == 3) ? P3 : P4;
                const h2 dd = __builtin_bit_cast(h2, X(carry)) + __builtin_bit_cast(h2, w5);
                const h2 ll = __builtin_elementwise_maximum(__builtin_elementwise_maximum(dd, __builtin_bit_cast(h2, X(up))), __builtin_bit_cast(h2, X(left)));
                carry = up; left = U(__builtin_bit_cast(unsigned, ll)); h[j + 1] = left;
                continue;
            } else if (MODE == 3) {
                // E: index mode stays on for the whole row; only the index changes per column
                unsigned dd;
                const unsigned tj = codes[j];
                asm volatile("s_set_gpr_idx_idx %[t]\n\tv_add_u32 %[d], v40, %[c]\n\ts_set_gpr_idx_idx 0"
                             : [d] "=&v"(dd), "+{v40}"(P0), "+{v41}"(P1), "+{v42}"(P2), "+{v43}"(P3),
                               "+{v44}"(P4), "+{v45}"(P5), "+{v46}"(P6), "+{v47}"(P7)
                             : [t] "s"(tj), [c] "v"(X(carry)));
                d = U(dd);
            } else {
                unsigned w;
                if (MODE == 0) w = __builtin_amdgcn_perm(special, tbl[j], sel);
                else w = (unsigned)__builtin_amdgcn_ds_bpermute(addr, (int)tbl[j]);
                d = U(X(carry) + w);
            }
            carry = up;
            left = __builtin_elementwise_max(__builtin_elementwise_max(d, up), left);
            h[j + 1] = left;
        }
    }
    if (MODE == 3) asm volatile("s_set_gpr_idx_off");
    unsigned acc = 0;
    for (int j = 0; j <= NCOL; ++j) acc += X(h[j]);
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int MODE> void run(const char* name, unsigned* d, unsigned* in, int per_cu = 16) {
    const int blocks = 256 * per_cu;    // 16 one-wave blocks per CU = 4 waves per SIMD
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 64>>>(d, in, 1); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); k<MODE><<<blocks, 64>>>(d, in, 2); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double cols = (double)blocks * ROWS * NCOL;                 // wave-columns
    double per_simd = cols / (256 * 4);
    printf("%-44s %8.3f ms  %.2f cycles per column per SIMD @2.4GHz\n", name, ms, ms * 1e-3 * 2.4e9 / per_simd);
}
int main() {
    unsigned *d, *in; (void)hipMalloc(&d, 256 * 32 * 64 * 4); (void)hipMalloc(&in, 4096); (void)hipMemset(in, 5, 4096);
    run<0>("A: v_perm + v_add_u32 + 2 v_pk_max_u16", d, in);
    run<1>("B: ds_bpermute + v_add_u32 + 2 v_pk_max_u16", d, in);
    run<2>("C: gpr-idx v_add_u32 + 2 v_pk_max_u16", d, in);
    run<3>("E: gpr-idx kept on, s_set_gpr_idx_idx per column", d, in);
    run<4>("F: static letters: v_add_u32 + 2 v_pk_max_u16", d, in);
    run<5>("G: v_perm + v_pk_add_f16 + v_pk_maximum3_f16", d, in);
    run<6>("H: static letters + v_pk_add_f16 + v_pk_maximum3_f16", d, in);
    run<6>("H at 2 waves/SIMD", d, in, 8);
    run<6>("H at 1 wave/SIMD", d, in, 4);
    run<5>("G at 2 waves/SIMD", d, in, 8);
    run<5>("G at 1 wave/SIMD", d, in, 4);
    run<0>("A at 2 waves/SIMD", d, in, 8);
    return 0;
}
