cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo "== suite run $i"; timeout 600 python -m pytest tests -m gpu -x -q -k "not config4 and not rccl" 2>&1 | grep -v "^\.\.\.\|^$" | tail -40
done
timeout 300 python bench.py --config 2 --steps 200 --warmup 10 --no-cpu-baseline --no-latency --verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('cfg2 %.4f ms/step, finalize %.4f, step frac %.3f'%(d['ms_per_step'], r['per_kernel_ms']['finalize'], r['step']['frac']))"
