#!/bin/bash
# config 3's throughput against the synthetic reads' error rate: a read end whose adapter scores below region_min_adapter_score
# has its barcode searched in the WHOLE 150-base window (qcat/scanner_epi2me.py:74-82) -- three times the rows of a nominal region
cd $GRAFT_REPO_ROOT
for e in 0.0 0.02 0.04 0.06 0.08 0.10 0.12; do
  QCAT_HIP_DEBUG_BINS=1 python bench.py --workload config3 --error-rate $e --steps 5 --warmup 1 --no-cpu-baseline 2> /tmp/err.txt | python -c "
import json,sys,re
d=json.loads(sys.stdin.read())
bins={}
for l in open('/tmp/err.txt'):
    m=re.search(r'group (\d+) class (\d+): (\d+) jobs', l)
    if m: bins[(int(m.group(1)),int(m.group(2)))]=int(m.group(3))
tot=sum(bins.values()); full=sum(v for (g,c),v in bins.items() if c==6); nom=sum(v for (g,c),v in bins.items() if c==5)
print('error rate $e: %.2f ms per step, %.1f M reads/s; jobs of the last scan: %d, nominal region %.1f %%, whole window %.1f %%' % (d['ms_per_step'], d['value']/1e6, tot, 100.0*nom/max(tot,1), 100.0*full/max(tot,1)))"
done
