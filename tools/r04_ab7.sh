#!/bin/bash
# round 4, call 13: parity of the lazy byte windows / interior bit-sliced barcodes / device kit choice over the big parity files;
# which of config 2's two side-by-side adapter launches is the slow one
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_ab7; mkdir -p $out
timeout 2400 python -m pytest tests/test_batch_auto_gpu.py tests/test_hip_parity.py tests/test_hip_fuzz.py tests/test_static_kernels.py tests/test_cli_gpu.py tests/test_jit.py tests/test_simple_gpu.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
B="python bench.py --no-host-inclusive --no-cpu-baseline"
for i in 1 2; do
  $B --workload config2 --steps 20 --warmup 3 > $out/c2_new_$i.json 2>/dev/null
  QCAT_HIP_ADAPTER_REVERSE=1 $B --workload config2 --steps 20 --warmup 3 > $out/c2_rev_$i.json 2>/dev/null
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
QCAT_HIP_ADAPTER_REVERSE=1 rocprofv3 --kernel-trace -d /tmp/tl_rev -o t --output-format csv -- python $R/bench.py --workload config2 --steps 4 --warmup 2 --no-cpu-baseline --no-host-inclusive > $R/$out/bench_rev.log 2>&1
python $R/tools/step_timeline.py /tmp/tl_rev > $R/$out/timeline_rev.txt 2>&1; grep -E "adapter|bs_barcode|finalize|pack" $R/$out/timeline_rev.txt
cd $R
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_ab7/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e); continue
    k = (d.get('roofline') or {}).get('kernels_avg_ms', {})
    print(os.path.basename(f), round(d['value'] / 1e6, 2), d['ms_per_step'], {x: round(v, 3) for x, v in k.items()})
PY
