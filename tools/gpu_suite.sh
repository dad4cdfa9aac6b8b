#!/bin/bash
# the whole -m gpu suite + smoke, as the driver runs them at round end (one GPU call; ~8 min)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out
TAG=${1:-suite}
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest -m gpu: exit $?"; tail -25 gpurun_out/${TAG}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
