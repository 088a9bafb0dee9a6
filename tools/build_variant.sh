#!/bin/bash
# usage (build container): tools/build_variant.sh <name> "<extra hipcc flags>"  ->  build_variants/libmi355pt_<name>.so
# Builds a copy of csrc/ with EXTRA flags (experiment macros such as -DPT_MIN_WAVES=5) without touching the shipped library;
# tools/ab.sh times the variants against each other on one GPU box (MI355PT_LIB selects the library).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
D=/tmp/mi355pt_variant_$N
rm -rf $D && mkdir -p $D/toy-cpu-pathtracing_amd $D/include
cp -r $R/toy-cpu-pathtracing_amd/csrc $D/toy-cpu-pathtracing_amd/ && rm -rf $D/toy-cpu-pathtracing_amd/csrc/build $D/toy-cpu-pathtracing_amd/csrc/*.so
cp $R/include/*.h $D/include/
make -C $D/toy-cpu-pathtracing_amd/csrc -j${JOBS:-4} ARCH=gfx950 EXTRA="$*" 2>&1 | grep -E "error|Error" || true
mkdir -p $R/build_variants && cp $D/toy-cpu-pathtracing_amd/csrc/libmi355pt.so $R/build_variants/libmi355pt_$N.so
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -disable-machine-licm $* -Rpass-analysis=kernel-resource-usage -c $D/toy-cpu-pathtracing_amd/csrc/pt_kernels_mis.hip -o /dev/null 2>&1 \
  | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" | paste - - - - - | sed 's/remark: [^ ]* //g; s/\[-Rpass[^]]*\]//g; s/[^ ]*pt_kernel.hpp:[0-9]*:1://g' | sed -n 2p
echo "built build_variants/libmi355pt_$N.so"
