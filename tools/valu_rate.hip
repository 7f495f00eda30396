// valu_rate.hip -- micro-benchmark: issue rate of the VALU ops the packed DP kernels are built
// from (wave64 on gfx950).  Prints wave-instructions per cycle per SIMD at the measured wall
// clock.  Build+run:  hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define ITERS 4096

template <int OP>
__global__ void __launch_bounds__(256) k(unsigned* out, unsigned seed) {
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    unsigned b = a0 ^ 0x5a5a5a5a, c = 0x0c020c00 | (threadIdx.x & 3);
    unsigned long long qa0 = a0, qa1 = a1, qa2 = a2, qa3 = a3, qa4 = a4, qa5 = a5, qa6 = a6, qa7 = a7, qb = b;
    for (int i = 0; i < ITERS; ++i) {
#define ONE(r) \
        if (OP == 0) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(r) : "v"(b), "v"(c)); \
        else if (OP == 1) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 2) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 4) asm volatile("v_max_u32 %0, %0, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 5) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c)); \
        else if (OP == 6) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c)); \
        else if (OP == 7) asm volatile("v_max3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c)); \
        else if (OP == 8) asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(r) : "v"(b)); \
        else if (OP == 9) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(r) : "v"(b)); \
        else if (OP == 10) asm volatile("v_bfe_u32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c)); \
        else if (OP == 11) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(b)); \
        else if (OP == 13) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c)); \
        else if (OP == 14) asm volatile("v_perm_b32 %0, %1, %0, %2" : "+v"(r) : "v"(b), "s"(seed)); \
        else if (OP == 15) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 16) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 17) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 18) asm volatile("v_pk_min_f16 %0, %0, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 19) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c)); \
        else if (OP == 20) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 21) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 22) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c)); \
        else if (OP == 23) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c)); \
        else if (OP == 24) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(r)); \
        else if (OP == 25) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 26) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 27) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 28) asm volatile("v_max_i32 %0, %0, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 29) asm volatile("v_max_f16 %0, %0, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 30) asm volatile("v_max3_f16 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c)); \
        else if (OP == 31) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c)); \
        else if (OP == 32) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(r)); \
        else if (OP == 33) asm volatile("v_mov_b32 %0, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 34) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q##r) : "v"(qb)); \
        else if (OP == 35) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(q##r) : "v"(qb)); \
        else if (OP == 36) asm volatile("v_min_u32 %0, %0, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 37) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 38) asm volatile("v_max_u16 %0, %0, %1" : "+v"(r) : "v"(b)); \
        else if (OP == 39) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c)); \
        else if (OP == 40) asm volatile("v_pk_maximum3_f16 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c)); \
        else if (OP == 41) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c)); \
        else if (OP == 42) asm volatile("v_maximum3_f32 %0, %0, %1, %2" : "+v"(r) : "v"(b), "v"(c)); \
        else if (OP == 43) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r) : "v"(b), "v"(c));
        REP8(ONE(a0) ONE(a1) ONE(a2) ONE(a3) ONE(a4) ONE(a5) ONE(a6) ONE(a7))
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (unsigned)(qa0 + qa1 + qa2 + qa3 + qa4 + qa5 + qa6 + qa7);
}

template <int OP> void run(const char* name, unsigned* d) {
    const int blocks = 256 * 8;     // 8 blocks of 4 waves per CU -> 8 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, 2);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * 4 * ITERS * 64.0;        // wave-instructions
    double per_simd_per_s = insts / (ms * 1e-3) / (256 * 4);
    printf("%-34s %8.3f ms   %6.3f G wave-inst/s/SIMD  (= %.2f cycles/inst @2.4GHz)\n", name, ms, per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
}

int main() {
    unsigned* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<3>("v_add_u32", d); run<4>("v_max_u32", d); run<0>("v_perm_b32 (vgpr sel)", d); run<14>("v_perm_b32 (sgpr sel)", d);
    run<1>("v_pk_add_u16", d); run<2>("v_pk_max_u16", d); run<11>("v_pk_max_i16", d); run<8>("v_pk_sub_u16 clamp", d);
    run<5>("v_pk_mad_u16", d); run<6>("v_and_or_b32", d); run<7>("v_max3_u32", d); run<9>("v_add_u32_sdwa byte", d);
    run<10>("v_bfe_u32", d); run<13>("v_mad_u32_u24", d); run<15>("v_lshl_add_u32", d);
    run<16>("v_pk_add_f16", d); run<17>("v_pk_max_f16", d); run<18>("v_pk_min_f16", d); run<19>("v_pk_fma_f16", d);
    run<20>("v_add_f32", d); run<21>("v_max_f32", d); run<22>("v_max3_f32", d); run<23>("v_fma_f32", d); run<39>("v_med3_f32", d);
    run<24>("v_cvt_f32_ubyte1", d); run<25>("v_sub_u32", d); run<26>("v_and_b32", d); run<27>("v_xor_b32", d); run<28>("v_max_i32", d);
    run<29>("v_max_f16", d); run<30>("v_max3_f16", d); run<31>("v_add3_u32", d); run<32>("v_lshlrev_b32", d); run<33>("v_mov_b32", d);
    run<34>("v_pk_add_f32", d); run<35>("v_pk_fma_f32", d); run<36>("v_min_u32", d); run<37>("v_pk_add_i16", d); run<38>("v_max_u16", d);
    run<40>("v_pk_maximum3_f16", d); run<41>("v_pk_minimum3_f16", d); run<42>("v_maximum3_f32", d); run<43>("v_bitop3_b32", d);
    return 0;
}
