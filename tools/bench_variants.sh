#!/bin/bash
# usage: tools/bench_variants.sh  -- quick A/B of env-tunable launch parameters (on the GPU box)
show() { python -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['value'], d['ms_per_step'], d['roofline']['kernels_avg_ms'])"; }
for wl in config2 config3 dual; do
  for ch in 2 4 6 8 12 16; do
    QCAT_HIP_CHUNK_BARCODES=$ch python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-host-inclusive > /tmp/o.json 2>/dev/null
    show /tmp/o.json "$wl chunk=$ch"
  done
done
