#!/bin/bash
# round 5: regions a few bases short of nominal in front-padded bit-sliced units -- A/B against the class switched off, three rounds on one box
cd ${GRAFT_REPO_ROOT:-.}
for r in 1 2 3; do for wl in config3 dual; do for off in 1 0; do
QCAT_HIP_BS_NO_SHORT=$off python bench.py --workload $wl --steps 10 --no-cpu-baseline --no-host-inclusive 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); k=d['roofline']['kernels_avg_ms']; print('no_short $off $wl', round(d['value']/1e6,2), d['ms_per_step'], {x: round(v,3) for x,v in k.items() if v>0.05})"
done; done; done
