// abs_kernels.hip -- the bit-sliced ADAPTER kernels (kernels_abs.inc, abs_core.h, abs_generated.inc) in a translation
// unit of their own, so that they compile beside qcat_hip.hip: __graft_entry__.build() compiles this file with
// -Dqk=qk_abs (the shared kernel headers in a namespace of their own, nothing is defined twice in the library) and
// qcat_hip.hip reaches the kernels through the two extern "C" launchers below (declared in packed_host.inc).
#include <hip/hip_runtime.h>

#include "rtc_prelude.inc"
#include "kernels_abs.inc"

// planes + eligibility of `n_tiles` tiles (k_abs_planes)
extern "C" void qcat_abs_launch_planes(unsigned n_tiles, void* stream, const uint32_t* win2, const int32_t* wlen, const uint8_t* wspec,
                                       uint32_t n_ends, int rows, void* planes, uint32_t* valid, uint8_t* need128, uint32_t* tile_any) {
    if (rows == 150)
        hipLaunchKernelGGL(qk::k_abs_planes<150>, dim3(n_tiles), dim3(qk::ABS_PLANE_WAVES * 64), 0, static_cast<hipStream_t>(stream), win2, wlen, wspec, n_ends, rows,
                           static_cast<uint2*>(planes), valid, need128, tile_any);
    else
        hipLaunchKernelGGL(qk::k_abs_planes<0>, dim3(n_tiles), dim3(qk::ABS_PLANE_WAVES * 64), 0, static_cast<hipStream_t>(stream), win2, wlen, wspec, n_ends, rows,
                           static_cast<uint2*>(planes), valid, need128, tile_any);
}

// the same with the windows packed in the same pass (k_pack_planes): reads -> win / wlen / wspec AND planes / valid / flags
extern "C" void qcat_abs_launch_pack_planes(unsigned n_tiles, void* stream, const uint8_t* bases, const uint64_t* offsets, uint32_t n_reads, int ends,
                                            uint8_t* win, int32_t* wlen, uint8_t* wspec, uint32_t n_ends, int rows, void* planes, uint32_t* valid,
                                            uint8_t* need128, uint32_t* tile_any) {
    const qk::AbsPackSrc ps{bases, offsets, n_reads, ends};
    if (rows == 150)
        hipLaunchKernelGGL(qk::k_pack_planes<150>, dim3(n_tiles), dim3(qk::ABS_PLANE_WAVES * 64), 0, static_cast<hipStream_t>(stream), ps, win, wlen, wspec, n_ends, rows,
                           static_cast<uint2*>(planes), valid, need128, tile_any);
    else
        hipLaunchKernelGGL(qk::k_pack_planes<0>, dim3(n_tiles), dim3(qk::ABS_PLANE_WAVES * 64), 0, static_cast<hipStream_t>(stream), ps, win, wlen, wspec, n_ends, rows,
                           static_cast<uint2*>(planes), valid, need128, tile_any);
}

// kind 0: fused two-template plan `id` (g_static_fused), kind 1: single-template plan `id` (g_static_templates); kind 2: the
// single-template plan of four stages (k_adapter_ms, four waves per tile: medium batches); kind 3: the four-stage plan of a
// template too long for two stages (k_adapter_mw, up to 26 columns per stage).  Returns 0 when the plan does not
// exist (the caller keeps the other form or the binary16 kernel); args null: only asks
extern "C" int qcat_abs_launch(int kind, int id, unsigned grid, void* stream, const void* args) {
    const qk::AbsArgs& a = *static_cast<const qk::AbsArgs*>(args);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (kind == 2) {
        switch (id) {
#define QCAT_ABS_CASE(N) case N: if (args) hipLaunchKernelGGL(qk::k_adapter_ms<qabs::QAM_T##N>, dim3(grid), dim3(256), 0, s, a); return 1;
            QCAT_ABS_FOR_EACH_TEMPLATE_MS(QCAT_ABS_CASE)
#undef QCAT_ABS_CASE
        default: return 0;
        }
    }
    if (kind == 3) {
        switch (id) {
#define QCAT_ABS_CASE(N) case N: if (args) hipLaunchKernelGGL(qk::k_adapter_mw<qabs::QAW_T##N>, dim3(grid), dim3(256), 0, s, a); return 1;
            QCAT_ABS_FOR_EACH_TEMPLATE_MW(QCAT_ABS_CASE)
#undef QCAT_ABS_CASE
        default: return 0;
        }
    }
    if (kind == 0) {
        switch (id) {
#define QCAT_ABS_CASE(N) case N: if (args) hipLaunchKernelGGL(qk::k_adapter_bs<qabs::QAB_F##N>, dim3(grid), dim3(128), 0, s, a); return 1;
            QCAT_ABS_FOR_EACH_FUSED(QCAT_ABS_CASE)
#undef QCAT_ABS_CASE
        default: return 0;
        }
    }
    switch (id) {
#define QCAT_ABS_CASE(N) case N: if (args) hipLaunchKernelGGL(qk::k_adapter_bs<qabs::QAB_T##N>, dim3(grid), dim3(128), 0, s, a); return 1;
        QCAT_ABS_FOR_EACH_TEMPLATE(QCAT_ABS_CASE)
#undef QCAT_ABS_CASE
    default: return 0;
    }
}
