#!/bin/bash
# round 4: the bit-sliced kernels forced onto the 4000-read kit-auto batch (thresholds through the environment)
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_api4; mkdir -p $out
run() { tag=$1; shift; for i in 1 2; do env "$@" timeout 600 python bench.py --workload api4000 > $out/api_${tag}_$i.json 2>$out/api_${tag}_$i.err; done; }
run base QCAT_X=1
run abs QCAT_HIP_ADAPTER_BITSLICE_MIN=1
run bs QCAT_HIP_BITSLICE_MIN=2048
run both QCAT_HIP_ADAPTER_BITSLICE_MIN=1 QCAT_HIP_BITSLICE_MIN=2048
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_api4/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e); continue
    print(os.path.basename(f), round(d['value'] / 1e6, 3), d['ms_per_step'], d.get('split_ms_per_call'), d.get('other_python_ms_per_call'))
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
QCAT_HIP_ADAPTER_BITSLICE_MIN=1 QCAT_HIP_BITSLICE_MIN=2048 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl_b -o t --output-format csv -- python $R/bench.py --workload api4000 --steps 8 --warmup 2 > $R/$out/bench_both.log 2>&1
python $R/tools/api_timeline.py /tmp/tl_b > $R/$out/timeline_both.txt 2>&1; tail -50 $R/$out/timeline_both.txt
