#!/bin/bash
# Builds the forensics artefacts of DESIGN 4.4 into tools/bin/ (git-ignored, shipped to the GPU box by gpurun):
#   libaggressors.so                 synthetic aggressor / victim kernels (tools/probes/aggressors.hip)
#   mfma_clock_probe                 dense-MFMA clock / delivered-peak probe (tools/probes/mfma_clock_probe.hip; bench.py reports it)
#   libppgs_amd_<variant>.so         the product library with ONLY ppg_frontend.hip rebuilt with other flags
# usage: tools/probes/build.sh [variant "flags" ...]      e.g.  timing "-DPPG_FE_TIMING"  o1 "-O1"
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
CSRC=$ROOT/ppgs_amd/csrc
BIN=$ROOT/tools/bin
mkdir -p "$BIN"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC --offload-arch=gfx950 -O3 -std=c++20 -fPIC -shared -Wno-inline-asm -o "$BIN/libaggressors.so" "$ROOT/tools/probes/aggressors.hip"
$HIPCC --offload-arch=gfx950 -O3 -std=c++20 -Wno-unused-result -o "$BIN/mfma_clock_probe" "$ROOT/tools/probes/mfma_clock_probe.hip"
make -C "$CSRC" -j8 >/dev/null
while [ $# -ge 2 ]; do
    variant=$1; flags=$2; shift 2
    obj=$BIN/fe_$variant.o
    # (same flags as the Makefile; the variant's flags last so that e.g. -O1 wins)
    $HIPCC --offload-arch=gfx950 -O3 -std=c++20 -fPIC -Wall -Wno-unused-function -Wno-inline-asm $flags \
        -c "$CSRC/ppg_frontend.hip" -o "$obj" -save-temps=obj 2>"$BIN/fe_$variant.log" || { cat "$BIN/fe_$variant.log"; exit 1; }
    others=$(ls "$CSRC"/obj/*.o | grep -v ppg_frontend.o)
    $HIPCC --offload-arch=gfx950 -shared -fPIC -o "$BIN/libppgs_amd_$variant.so" $others "$obj"
    echo "built $variant: $flags"
done
