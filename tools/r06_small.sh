#!/bin/bash
# the small kernels between the adapter and the barcode phase (k_adapter_finish, k_pick_kit): parity of the kit-auto and named-kit
# calls, then the reference driver's call shape and its timeline
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_small; mkdir -p $out
timeout 1500 python -m pytest tests/test_batch_auto_gpu.py tests/test_scan_api_gpu.py tests/test_hip_parity.py -q -x 2>&1 | tail -4 > $out/tests.txt
cat $out/tests.txt
bash tools/api4000_trace.sh > $out/trace.log 2>&1
cp gpurun_out/api4000_trace/timeline.txt $out/timeline.txt
cp gpurun_out/api4000_trace/bench_api4000.json $out/bench_api4000.json
sed -n 1,48p $out/timeline.txt
python - <<PY
import json
d = json.load(open("$out/bench_api4000.json")); print("api4000", d["ms_per_step"], d["split_ms_per_call"])
PY
for w in config2 api1; do python bench.py --workload $w --no-cpu-baseline > $out/bench_$w.json 2> $out/bench_$w.err; python - <<PY
import json
d = json.load(open("$out/bench_$w.json")); print("$w", d["ms_per_step"], d["value"])
PY
done
