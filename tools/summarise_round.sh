#!/bin/bash
# After `gpurun -- 'bash tools/round_evidence.sh <tag>'`: copy what should be judged from gpurun_out/ into profiles/
# (kernel stats, PMC passes, SQ counters, bench lines) and rebuild profiles/pmc_summary.json for the library's source hash.
#   bash tools/summarise_round.sh <tag> <round>      e.g.  bash tools/summarise_round.sh r04e r04
set -u
TAG=$1; RND=$2
cd "$(dirname "$0")/.."
for c in cfg2 cfg4 cfg2_4M cfg1 cfg3; do
  d=gpurun_out/prof_${TAG}_$c
  [ -d $d ] || continue
  case $c in cfg2_4M) key=config2_indexed_4M;; *) key=config${c#cfg}_indexed;; esac
  sq=gpurun_out/sq_${TAG}_$c
  if [ -d $sq ]; then python tools/pmc_summary.py $d $RND $key $sq > /dev/null; else python tools/pmc_summary.py $d $RND $key > /dev/null; fi
  echo "$c -> profiles/${RND}_${TAG}_${c}_* ($key)"
done
for f in gpurun_out/${TAG}_bench_*.json; do
  [ -s $f ] && cp $f profiles/${RND}_$(basename $f | sed "s/^${TAG}_//")
done
ls profiles | grep "^${RND}_" | head -60
