#!/bin/bash
# round 4: padded super-tiles for small batches of big barcode sets -- parity (tests + the fuzz sweep with the padding forced), api4000
cd ${GRAFT_REPO_ROOT:-.}
out=gpurun_out/r04_pad; mkdir -p $out
timeout 2400 python -m pytest tests/test_hip_parity.py tests/test_batch_auto_gpu.py tests/test_cli_gpu.py tests/test_static_kernels.py tests/test_jit.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $out/pytest.log
(QCAT_HIP_BITSLICE_MIN=2048 QCAT_HIP_BITSLICE_PAD=128 timeout 1200 python tools/fuzz_bitslice.py 0 60) > $out/fuzz_pad.txt 2>&1; tail -2 $out/fuzz_pad.txt
for i in 1 2; do
  timeout 600 python bench.py --workload api4000 > $out/api_pad_$i.json 2>$out/api_pad_$i.err
  QCAT_HIP_BITSLICE_PAD=0 timeout 600 python bench.py --workload api4000 > $out/api_nopad_$i.json 2>$out/api_nopad_$i.err
done
B="python bench.py --no-host-inclusive --no-cpu-baseline"
for r in 25000 100000; do
  $B --workload config3 --reads $r --steps 20 --warmup 3 > $out/c3_${r}_pad.json 2>/dev/null
  QCAT_HIP_BITSLICE_PAD=0 $B --workload config3 --reads $r --steps 20 --warmup 3 > $out/c3_${r}_nopad.json 2>/dev/null
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r04_pad/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'ERR', e); continue
    print(os.path.basename(f), round(d['value'] / 1e6, 3), d['ms_per_step'], d.get('split_ms_per_call') or {k: round(v, 3) for k, v in ((d.get('roofline') or {}).get('kernels_avg_ms') or {}).items()})
PY
