cd ${GRAFT_REPO_ROOT:-.}
B="timeout 600 python bench.py --no-host-inclusive --no-cpu-baseline --workload middle --steps 10 --warmup 2"
for p in 0 2 3 4 5; do QCAT_HIP_MIDDLE_ABS_PRIO=$p $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prio $p', d['ms_per_step'], d['roofline']['kernels_avg_ms']['k_middle_packed'], d['roofline']['kernels_avg_ms']['k_adapter_bitslice'])"; done
