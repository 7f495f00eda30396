#!/bin/bash
# does a Python process that used the library ever crash at exit?  30 short runs each of three kinds
cd ${GRAFT_REPO_ROOT:-.}
crash=0
for i in $(seq 1 30); do
  QCAT_R1_RULE=scalar timeout 300 python tools/fuzz_tiny.py 0 25 > /tmp/ft.txt 2>&1; rc=$?
  [ $rc -ne 0 ] && { crash=$((crash+1)); echo "fuzz_tiny run $i rc $rc: $(tail -1 /tmp/ft.txt)"; }
done
echo "fuzz_tiny: $crash of 30 runs ended abnormally"
crash=0
for i in $(seq 1 20); do
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > /tmp/sm.txt 2>&1; rc=$?
  [ $rc -ne 0 ] && { crash=$((crash+1)); echo "smoke run $i rc $rc: $(tail -1 /tmp/sm.txt)"; }
done
echo "smoke: $crash of 20 runs ended abnormally"
crash=0
for i in $(seq 1 10); do
  timeout 300 python -m pytest tests/test_scan_api_gpu.py -q -m gpu > /tmp/pt.txt 2>&1; rc=$?
  [ $rc -ne 0 ] && { crash=$((crash+1)); echo "pytest run $i rc $rc: $(tail -2 /tmp/pt.txt)"; }
done
echo "pytest: $crash of 10 runs ended abnormally"
