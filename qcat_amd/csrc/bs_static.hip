// bs_static.hip -- the bit-sliced barcode kernels with the target letters of the built-in kits compiled in
// (bs_static_generated.inc, tools/gen_static_kernels.py), in translation units of their own so that they compile in
// parallel with qcat_hip.hip: __graft_entry__.build() compiles this file once per part (-DQCAT_BS_PART=<n>
// -Dqk=qk_bs<n>: every part sees the shared kernel headers in a namespace of its own, so nothing is defined twice in
// the library) and qcat_hip.hip reaches the kernels through qcat_bs_launch_part<n> (static_generated.inc).
#include <hip/hip_runtime.h>

#ifndef QCAT_BS_PART
#error "compile with -DQCAT_BS_PART=<part> -Dqk=qk_bs<part>"
#endif
#include "rtc_prelude.inc"
#include "bs_static_generated.inc"
