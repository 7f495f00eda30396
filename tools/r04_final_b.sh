#!/bin/bash
# round 4, final measurements: rocprofv3 evidence (trace + counter passes) for config 3 / config 2 / the shipped dual kit / --detect-middle,
# one bench line per workload, the driver on files (named kit and kit auto), the kit-auto file loop by chunk size
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_all
bash tools/profile.sh r04_config3 config3 --steps 3 --warmup 1 > gpurun_out/r04_all/prof_config3.log 2>&1
bash tools/profile.sh r04_config2 config2 --steps 10 --warmup 2 > gpurun_out/r04_all/prof_config2.log 2>&1
bash tools/profile.sh r04_dual dual --steps 5 --warmup 1 > gpurun_out/r04_all/prof_dual.log 2>&1
bash tools/profile.sh r04_middle middle --steps 5 --warmup 1 > gpurun_out/r04_all/prof_middle.log 2>&1
cd $GRAFT_REPO_ROOT
for wl in config3 config2 dual config4 middle api4000 dual96; do
  timeout 900 python bench.py --workload $wl > gpurun_out/r04_all/bench_$wl.json 2> gpurun_out/r04_all/bench_$wl.err
  python -c "
import json; d=json.loads(open('gpurun_out/r04_all/bench_$wl.json').read().strip().splitlines()[-1]); print('$wl', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('kernels_avg_ms'), d.get('split_ms_per_call'))" 2>&1 | cut -c1-400
done
QCAT_BENCH_TMP=/dev/shm timeout 900 python tools/bench_auto_file.py 2000000 > gpurun_out/r04_all/auto_file.json 2>gpurun_out/r04_all/auto_file.err; cat gpurun_out/r04_all/auto_file.json
QCAT_BENCH_TMP=/dev/shm timeout 1500 python tools/bench_cli.py 3000000 20000 > gpurun_out/r04_all/bench_cli.json 2>gpurun_out/r04_all/bench_cli.err; cat gpurun_out/r04_all/bench_cli.json | cut -c1-1500
(QCAT_HIP_BITSLICE_MIN=2048 QCAT_HIP_ADAPTER_BITSLICE_MIN=1 timeout 1500 python tools/fuzz_bitslice.py 67 120) > gpurun_out/r04_all/fuzz_adapter_forced_67.txt 2>&1; tail -2 gpurun_out/r04_all/fuzz_adapter_forced_67.txt
