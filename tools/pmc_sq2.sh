#!/bin/bash
# Extra SQ counter passes (wait breakdown, FIFO stalls, instruction fetch).  usage: tools/pmc_sq2.sh <tag> [bench args]
TAG=${1:-sq2}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/sq2_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency $*"
i=0
for SET in "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -- $BENCH > $OUT/p$i.json 2> $OUT/p$i.err
done
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(kt_[a-z_]+)", r["Kernel_Name"])
        if m: acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as out:
    for k in sorted(acc):
        if k in ("kt_ingest_pods",): continue
        line = k + ": " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(acc[k].items()))
        print(line); out.write(line + "\n")
PY
