#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r05_check
timeout 1500 python -m pytest tests/test_tiny_gpu.py -x -q -m gpu > gpurun_out/r05_check/tests_tiny.log 2>&1; tail -3 gpurun_out/r05_check/tests_tiny.log
timeout 300 python bench.py --workload api1 > gpurun_out/r05_check/bench_api1.json 2>gpurun_out/r05_check/bench_api1.err; python -c "
import json; d=json.load(open('gpurun_out/r05_check/bench_api1.json')); print(d['legs'])"
bash tools/api1_trace.sh > gpurun_out/r05_check/api1_trace.log 2>&1; sed -n 8,14p gpurun_out/api1_trace/timeline.txt
cp qcat_amd/csrc/libqcat_hip.so /tmp/a.so
for r in 1 2; do for wl in dual dual96 config2; do for v in 0 1; do
QCAT_HIP_ABS_SERIAL=$v timeout 300 python bench.py --workload $wl --no-host-inclusive --no-cpu-baseline > gpurun_out/r05_check/serial_${wl}_$v.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r05_check/serial_${wl}_$v.json')); print('$wl serial=$v', d['ms_per_step'], {k:round(x,3) for k,x in d['roofline']['kernels_avg_ms'].items() if x>0.1})"
done; done; done
timeout 1200 python tools/bench_cli.py 1000000 100000 > gpurun_out/r05_check/bench_cli.json 2> gpurun_out/r05_check/bench_cli.err; tail -c 1500 gpurun_out/r05_check/bench_cli.json; tail -3 gpurun_out/r05_check/bench_cli.err
