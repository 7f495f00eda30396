#!/bin/bash
# the adapter chains of a small batch in one launch (k_adapter_multi): parity, interleaved A/B on the reference driver's call shape, timeline
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_multi; mkdir -p $out
timeout 1500 python -m pytest tests/test_batch_auto_gpu.py tests/test_scan_api_gpu.py tests/test_tiny_gpu.py -q -x 2>&1 | tail -4 > $out/tests.txt
cat $out/tests.txt
run() {
  name=$1; shift
  env "$@" python bench.py --workload $W --steps 20 --warmup 3 > $out/$W.$name.json 2> $out/$W.$name.err
  python - <<PY
import json
d = json.load(open("$out/$W.$name.json")); print("$W", "$name", d["ms_per_step"], d.get("split_ms_per_call", {}).get("native_call_ms"))
PY
}
W=api4000
for i in 1 2 3 4; do run own QCAT_HIP_NO_ADAPTER_MULTI=1 QCAT_HIP_NO_BARCODE_MULTI=1; run bown QCAT_HIP_NO_BARCODE_MULTI=1; run one A=1; done
W=api1
for i in 1 2; do run own QCAT_HIP_NO_ADAPTER_MULTI=1 QCAT_HIP_NO_BARCODE_MULTI=1; run one A=1; done
bash tools/api4000_trace.sh > $out/trace.log 2>&1
cp gpurun_out/api4000_trace/timeline.txt $out/timeline.txt
sed -n 1,40p $out/timeline.txt
