#!/bin/bash
# A/B of library builds inside ONE gpurun call (same box, same clocks): tools/ab_run.sh <outdir> <rounds> <bench args...> -- <lib>...
# libs are paths under qcat_amd/csrc/build/ab/ (built by hand: copy libqcat_hip.so there after a build)
cd ${GRAFT_REPO_ROOT:-.}
out=$1; rounds=$2; shift 2
args=()
while [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
mkdir -p $out
for r in $(seq 1 $rounds); do
  for lib in "$@"; do
    name=$(basename $lib .so)
    QCAT_HIP_LIBRARY=$PWD/$lib timeout 300 python bench.py --no-host-inclusive --no-cpu-baseline "${args[@]}" > $out/${name}_$r.json 2>/dev/null
    python - $out/${name}_$r.json $name <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d['roofline']['kernels_avg_ms']
print(sys.argv[2], round(d['value'] / 1e6, 2), d['ms_per_step'], {x: round(v, 3) for x, v in k.items() if v > 0.3})
PY
  done
done
