#!/bin/bash
# GPU box: the share of samples whose spectral radiance equals the oracle's BIT FOR BIT (tools/bit_equal_share.py) and the throughput
# (tools/ab.sh) of three builds made beforehand in the build container with tools/build_variant.sh:
#   bx0 ""                                          the shipped defaults
#   bx1 "-DPT_EXACT_DIV=1"                          + the reference's divisions in the light connection and the sensor
#   bx2 "-DPT_EXACT_DIV=1 -DPT_SIGMOID_EXACT=1"     + the reference's sigmoid (glibc's expf restated in double): the bit-exact build
R=${GRAFT_REPO_ROOT:-$PWD}
for v in bx0 bx1 bx2; do
  echo "== $v"
  MI355PT_LIB=$R/build_variants/libmi355pt_$v.so timeout -k 10 400 python3 $R/tools/bit_equal_share.py 0:mis 9:mis 8:mis 10:mis 7:mis 12:mis 17:nee 19:mis 3:mis 15:mis || exit 1
done
$R/tools/ab.sh "3 10 8 17" libmi355pt_bx0.so libmi355pt_bx1.so libmi355pt_bx2.so
