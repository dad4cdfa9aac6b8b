cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pf -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/pf/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("<")[0].split("(")[0]
        if "kt_check" in k or "kt_aggregate_bitmap" in k: acc[k.split("::")[-1]].append(float(r["Counter_Value"]))
for k,v in acc.items(): print(k, len(v), sum(v)/len(v))
PY
