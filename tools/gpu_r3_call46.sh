#!/bin/bash
cd $GRAFT_REPO_ROOT
for k in PBC096 NBD103/NBD104 DUAL; do timeout 300 python tools/region_len_hist.py $k 200000 2>&1 | tail -28; done
