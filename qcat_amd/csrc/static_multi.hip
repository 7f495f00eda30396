// static_multi.hip -- the merged launches of small batches (k_adapter_multi, k_barcode_multi: static_generated.inc) in a
// translation unit of their own, so that they compile beside qcat_hip.hip (each holds every static-letter chain of the built-in
// kits once more): __graft_entry__.build() compiles this file with -Dqk=qk_sm (the shared kernel headers in a namespace of
// their own, nothing is defined twice in the library) and qcat_hip.hip reaches the kernels through qcat_static_multi_adapter /
// qcat_static_multi_barcode (declared in static_generated.inc).
#include <hip/hip_runtime.h>

#define QCAT_STATIC_MULTI_TU 1
#include "rtc_prelude.inc"
#include "static_generated.inc"
