#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c38
(timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_batch_auto_gpu.py tests/test_scan_api_gpu.py tests/test_cli_gpu.py tests/test_hip_fuzz.py -x -q -m gpu) > gpurun_out/c38/tests.log 2>&1; tail -3 gpurun_out/c38/tests.log
run() { label=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-host-inclusive --no-cpu-baseline "$@" > gpurun_out/c38/$label.json 2>/dev/null
  python - gpurun_out/c38/$label.json $label <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); k = d['roofline']['kernels_avg_ms']
print(sys.argv[2], round(d['value'] / 1e6, 2), d['ms_per_step'], {x: round(v, 3) for x, v in k.items() if v > 0.05})
PY
}
for n in 500000 250000 125000; do run c2_$n A=1 -- --workload config2 --reads $n; done
for n in 1000000 250000 125000 62500 31250; do run c3_$n A=1 -- --workload config3 --reads $n; done
for n in 250000 62500; do run dual_$n A=1 -- --workload dual --reads $n; done
timeout 600 python bench.py --steps 10 --no-cpu-baseline > gpurun_out/c38/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/c38/bench.json')); print(d['value'], d['ms_per_step'], d['host_inclusive']['value'], d['host_inclusive']['from_fastq'])"
timeout 900 python tools/bench_cli.py 6000000 20000 > gpurun_out/c38/bench_cli6m.json 2> gpurun_out/c38/bench_cli6m.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/c38/bench_cli6m.json'))
for k in ('outputs_identical','ingest','native_tsv','native_per_barcode_fastq','kit_auto_20000_reads','kit_PBC096_20000_reads'): print(k, d[k])
PY
