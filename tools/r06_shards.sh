#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout 900 python -m pytest tests/test_file_shards.py -x -q 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_comm_gpu.py -x -q -m gpu 2>&1 | tail -6
