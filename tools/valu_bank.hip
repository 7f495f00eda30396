// VGPR bank conflict probe: v_bitop3_b32 with its three sources in the same / different register banks (index mod 4)
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
#define REP8(x) x x x x x x x x
template <int MODE> __global__ void __launch_bounds__(256) k(unsigned* out, unsigned seed, int iters = ITERS) {
    // explicit registers: destinations v40..v47 (8 independent chains), sources chosen per mode
    asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n v_mov_b32 v44, %0\n v_mov_b32 v45, %0\n v_mov_b32 v46, %0\n v_mov_b32 v47, %0\n"
                 "v_mov_b32 v48, %0\n v_mov_b32 v49, %0\n v_mov_b32 v50, %0\n v_mov_b32 v51, %0\n v_mov_b32 v52, %0\n v_mov_b32 v53, %0\n v_mov_b32 v54, %0\n v_mov_b32 v55, %0\n v_mov_b32 v56, %0\n v_mov_b32 v57, %0\n v_mov_b32 v58, %0\n v_mov_b32 v59, %0\n"
                 :: "v"(seed + threadIdx.x) : "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59");
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {        // all three sources in different banks; dst = src0
            REP8(asm volatile("v_bitop3_b32 v40, v40, v49, v50 bitop3:0x96\n v_bitop3_b32 v41, v41, v50, v51 bitop3:0x96\n v_bitop3_b32 v42, v42, v51, v48 bitop3:0x96\n v_bitop3_b32 v43, v43, v48, v49 bitop3:0x96\n"
                              "v_bitop3_b32 v44, v44, v53, v54 bitop3:0x96\n v_bitop3_b32 v45, v45, v54, v55 bitop3:0x96\n v_bitop3_b32 v46, v46, v55, v52 bitop3:0x96\n v_bitop3_b32 v47, v47, v52, v53 bitop3:0x96\n"
                              ::: "v40","v41","v42","v43","v44","v45","v46","v47");)
        } else if (MODE == 1) { // two sources share a bank (src1, src2)
            REP8(asm volatile("v_bitop3_b32 v40, v40, v49, v53 bitop3:0x96\n v_bitop3_b32 v41, v41, v50, v54 bitop3:0x96\n v_bitop3_b32 v42, v42, v51, v55 bitop3:0x96\n v_bitop3_b32 v43, v43, v48, v52 bitop3:0x96\n"
                              "v_bitop3_b32 v44, v44, v53, v57 bitop3:0x96\n v_bitop3_b32 v45, v45, v54, v58 bitop3:0x96\n v_bitop3_b32 v46, v46, v55, v59 bitop3:0x96\n v_bitop3_b32 v47, v47, v52, v56 bitop3:0x96\n"
                              ::: "v40","v41","v42","v43","v44","v45","v46","v47");)
        } else if (MODE == 2) { // all three in the same bank
            REP8(asm volatile("v_bitop3_b32 v40, v40, v48, v52 bitop3:0x96\n v_bitop3_b32 v41, v41, v49, v53 bitop3:0x96\n v_bitop3_b32 v42, v42, v50, v54 bitop3:0x96\n v_bitop3_b32 v43, v43, v51, v55 bitop3:0x96\n"
                              "v_bitop3_b32 v44, v44, v52, v56 bitop3:0x96\n v_bitop3_b32 v45, v45, v53, v57 bitop3:0x96\n v_bitop3_b32 v46, v46, v54, v58 bitop3:0x96\n v_bitop3_b32 v47, v47, v55, v59 bitop3:0x96\n"
                              ::: "v40","v41","v42","v43","v44","v45","v46","v47");)
        } else if (MODE == 3) { // src0 and src1 share a bank, src2 different
            REP8(asm volatile("v_bitop3_b32 v40, v40, v48, v49 bitop3:0x96\n v_bitop3_b32 v41, v41, v49, v50 bitop3:0x96\n v_bitop3_b32 v42, v42, v50, v51 bitop3:0x96\n v_bitop3_b32 v43, v43, v51, v48 bitop3:0x96\n"
                              "v_bitop3_b32 v44, v44, v52, v53 bitop3:0x96\n v_bitop3_b32 v45, v45, v53, v54 bitop3:0x96\n v_bitop3_b32 v46, v46, v54, v55 bitop3:0x96\n v_bitop3_b32 v47, v47, v55, v52 bitop3:0x96\n"
                              ::: "v40","v41","v42","v43","v44","v45","v46","v47");)
        } else if (MODE == 4) { // dependent chain: one accumulator only (latency), different banks
            REP8(asm volatile("v_bitop3_b32 v40, v40, v49, v50 bitop3:0x96\n v_bitop3_b32 v40, v40, v49, v50 bitop3:0x96\n v_bitop3_b32 v40, v40, v49, v50 bitop3:0x96\n v_bitop3_b32 v40, v40, v49, v50 bitop3:0x96\n"
                              "v_bitop3_b32 v40, v40, v49, v50 bitop3:0x96\n v_bitop3_b32 v40, v40, v49, v50 bitop3:0x96\n v_bitop3_b32 v40, v40, v49, v50 bitop3:0x96\n v_bitop3_b32 v40, v40, v49, v50 bitop3:0x96\n"
                              ::: "v40");)
        } else if (MODE == 5) { // two VGPR sources + an SGPR
            REP8(asm volatile("v_bitop3_b32 v40, v40, v49, %0 bitop3:0x96\n v_bitop3_b32 v41, v41, v50, %0 bitop3:0x96\n v_bitop3_b32 v42, v42, v51, %0 bitop3:0x96\n v_bitop3_b32 v43, v43, v48, %0 bitop3:0x96\n"
                              "v_bitop3_b32 v44, v44, v53, %0 bitop3:0x96\n v_bitop3_b32 v45, v45, v54, %0 bitop3:0x96\n v_bitop3_b32 v46, v46, v55, %0 bitop3:0x96\n v_bitop3_b32 v47, v47, v52, %0 bitop3:0x96\n"
                              :: "s"(seed) : "v40","v41","v42","v43","v44","v45","v46","v47");)
        } else if (MODE == 6) { // dst different from all sources, different banks
            REP8(asm volatile("v_bitop3_b32 v40, v57, v49, v50 bitop3:0x96\n v_bitop3_b32 v41, v58, v50, v51 bitop3:0x96\n v_bitop3_b32 v42, v59, v51, v48 bitop3:0x96\n v_bitop3_b32 v43, v56, v48, v49 bitop3:0x96\n"
                              "v_bitop3_b32 v44, v57, v53, v54 bitop3:0x96\n v_bitop3_b32 v45, v58, v54, v55 bitop3:0x96\n v_bitop3_b32 v46, v59, v55, v52 bitop3:0x96\n v_bitop3_b32 v47, v56, v52, v53 bitop3:0x96\n"
                              ::: "v40","v41","v42","v43","v44","v45","v46","v47");)
        }
    }
    unsigned r;
    asm volatile("v_xor_b32 %0, v40, v41\n v_xor_b32 %0, %0, v42\n v_xor_b32 %0, %0, v43\n v_xor_b32 %0, %0, v44\n v_xor_b32 %0, %0, v45\n v_xor_b32 %0, %0, v46\n v_xor_b32 %0, %0, v47" : "=v"(r));
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> void run(const char* name, unsigned* d, int blocks_per_cu, int mult = 1) {
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 1, ITERS * mult); hipDeviceSynchronize();
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(d, 2, ITERS * mult); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)blocks * 4 * ITERS * 64.0 * mult;
    double per = insts / (ms * 1e-3) / (256 * 4);
    printf("%-52s %3d blocks/CU %8.3f ms  %.2f cycles/inst @2.4GHz\n", name, blocks_per_cu, ms, 2.4e9 / per);
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 256 * 256 * 4);
    // sustained rate: the same kernel for tens of milliseconds (8 waves per SIMD resident, 32x the blocks)
    run<0>("3 sources, 3 banks, long run", d, 8); run<0>("3 sources, 3 banks, long run", d, 256); run<0>("3 sources, 3 banks, long run", d, 256);
    // sustained rates at the occupancies of the bit-sliced kernel (tens of milliseconds per launch)
    for (int w : {1, 2, 3, 4, 8}) { run<0>("8 independent chains per wave, long", d, w, 32); run<4>("one dependent chain per wave, long", d, w, 32); }
    for (int w : {8, 2, 1}) {
        if (w == 8) { run<0>("3 sources, 3 banks", d, 8); run<1>("src1, src2 same bank", d, 8); run<3>("src0, src1 same bank", d, 8); run<2>("all same bank", d, 8); run<5>("2 VGPR + SGPR", d, 8); run<6>("dst != src0, 3 banks", d, 8); run<4>("one dependent chain", d, 8); }
        if (w == 2) { run<0>("3 sources, 3 banks", d, 2); run<1>("src1, src2 same bank", d, 2); run<3>("src0, src1 same bank", d, 2); run<2>("all same bank", d, 2); run<5>("2 VGPR + SGPR", d, 2); run<6>("dst != src0, 3 banks", d, 2); run<4>("one dependent chain", d, 2); }
        if (w == 1) { run<0>("3 sources, 3 banks", d, 1); run<2>("all same bank", d, 1); run<4>("one dependent chain", d, 1); }
    }
    return 0;
}
