#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out
python tools/tight_study.py --config 4 2>&1 | grep -v "^Exc\|^Trace\|File\|TypeError" | tail -3
python tools/tight_study.py --config 2 2>&1 | grep -v "^Exc\|^Trace\|File\|TypeError" | tail -3
