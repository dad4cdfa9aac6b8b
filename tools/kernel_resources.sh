#!/bin/bash
# Static resource usage of every kernel instantiation (no GPU needed): VGPRs, SGPR spills, scratch bytes per lane,
# occupancy — from the compiler's own remarks.   usage: tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt
#   FLAGS="-DKT_..." FILES="kt_kernels_check.hip" tools/kernel_resources.sh     (one file under extra flags: A/B of a variant)
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
SRC=$REPO/kube_throttler_amd/csrc
TMP=$(mktemp -d)
printf "%-62s %6s %6s %8s %9s %5s\n" kernel VGPRs SGPRs scratchB sgprSpill occ
for f in ${FILES:-kt_kernels_check.hip kt_kernels_aggregate.hip kt_kernels_few.hip kt_kernels_admit.hip kt_kernels.hip kt_kernels_finalize.hip}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 ${FLAGS:-} -I$REPO/include -I$SRC -S --cuda-device-only \
      -Rpass-analysis=kernel-resource-usage $SRC/$f -o $TMP/out.s 2>&1 |
    awk '/Function Name:/ {name=$(NF-1)} /TotalSGPRs:/ {sg=$(NF-1)} / VGPRs:/ {vg=$(NF-1)} /ScratchSize/ {sc=$(NF-1)}
         /Occupancy/ {oc=$(NF-1)} /SGPRs Spill:/ {sp=$(NF-1)} /LDS Size/ {printf "%s %s %s %s %s %s\n", name, vg, sg, sc, sp, oc}' |
    while read name vg sg sc sp oc; do printf "%-62s %6s %6s %8s %9s %5s\n" "$(echo $name | c++filt | sed 's/(.*//; s/^void //')" $vg $sg $sc $sp $oc; done
done
rm -rf $TMP
