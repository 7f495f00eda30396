// abs_mid_kernels.hip -- the bit-sliced interior adapter scan of --detect-middle (kernels_abs_mid.inc) in a translation unit
// of its own: __graft_entry__.build() compiles this file with -Dqk=qk_absmid -Dqabs=qabs_mid -DQCAT_ABS_NI=14 (row indices of
// interiors up to 16 384 rows; abs_kernels.hip keeps the eight index planes of the read ends' windows) and qcat_hip.hip
// reaches the kernels through the extern "C" launchers below (declared in qcat_hip.hip beside middle_packed).
#include <hip/hip_runtime.h>
#include <algorithm>

#include "rtc_prelude.inc"
#include "kernels_abs.inc"
#include "kernels_abs_mid.inc"

static_assert(qabs::ABS_NI >= 14, "compile with -DQCAT_ABS_NI=14: interiors of up to 16 384 rows");

// row counts, places and letter planes of the big tiles (k_absmid_codes, k_absmid_tiles, k_absmid_scan, k_absmid_planes); rspec zeroed by the caller
// win2 / wspec non-null: the M-ends' first windows and flags from the packed batch as well (k_absmid_windows instead of k_mid_windows)
// what & 1: the packed batch (k_absmid_codes -- needs the reads only, so a scan starts it on a stream of its own beside the read
// ends' kernels); what & 2: everything after it
extern "C" void qcat_absmid_prepare(void* stream, const void* args, uint32_t* win2, uint8_t* wspec, int what) {
    const qk::AbsMidArgs& a = *static_cast<const qk::AbsMidArgs*>(args);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const unsigned long long chunks = (a.n_bases + 15) / 16 + 1;
    if (what & 1) hipLaunchKernelGGL(qk::k_absmid_codes, dim3((unsigned)std::min<unsigned long long>((chunks + 255) / 256, 1u << 20)), dim3(256), 0, s, a);
    if (!(what & 2)) return;
    hipLaunchKernelGGL(qk::k_absmid_tiles, dim3(a.n_tiles), dim3(256), 0, s, a);
    if (win2) {
        const unsigned long long wthreads = (unsigned long long)a.slot_cap * 16;
        hipLaunchKernelGGL(qk::k_absmid_windows, dim3((unsigned)((wthreads + 255) / 256)), dim3(256), 0, s, a, a.slot_cap, win2, wspec);
    }
    hipLaunchKernelGGL(qk::k_absmid_scan, dim3(1), dim3(1024), 0, s, a);
    hipLaunchKernelGGL(qk::k_absmid_planes, dim3(a.n_tiles, qk::ABSM_GY), dim3(256), 0, s, a);
}

// the single-template plan of adapter template `id` (g_static_templates) over the big tiles of its kit.  waves 2: the two-stage
// pipeline (k_adapter_mid, workgroups of 128); waves 1: the whole row on one wave (k_adapter_mid1, workgroups of 64) -- only
// for templates of up to ABSM_ONE_WAVE_COLS columns.  Returns 0 when that form does not exist; args null: only asks
extern "C" int qcat_absmid_launch(int id, int waves, unsigned grid, void* stream, const void* args) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (id) {
#define QCAT_ABS_CASE(N) case N: \
        if (waves == 1) { \
            if (qabs::QAB_T##N::NC0 + qabs::QAB_T##N::NC1 > qk::ABSM_ONE_WAVE_COLS) return 0; \
            if (args) hipLaunchKernelGGL(qk::k_adapter_mid1<qabs::QAB_T##N>, dim3(grid), dim3(64), 0, s, *static_cast<const qk::AbsMidArgs*>(args)); \
            return 1; \
        } \
        if (args) hipLaunchKernelGGL(qk::k_adapter_mid<qabs::QAB_T##N>, dim3(grid), dim3(128), 0, s, *static_cast<const qk::AbsMidArgs*>(args)); \
        return 1;
        QCAT_ABS_FOR_EACH_TEMPLATE(QCAT_ABS_CASE)
#undef QCAT_ABS_CASE
    default: return 0;
    }
}
