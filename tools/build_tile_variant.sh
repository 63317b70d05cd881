#!/bin/bash
# Builds lib/libcascadeclassifier_amd_<ty>_<threads>.so: the library with another cascade-kernel tile shape
# (window rows per tile x threads per block). Select it with CCAMD_LIB=<path>. Tuning experiments only.
set -e
ty=$1; th=$2
cd "$(dirname "$0")/../cascadeclassifier_amd/csrc"
make -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -DCC_TILE_Y=$ty -DCC_EVAL_THREADS=$th -c cc_detect.hip -o build/cc_detect_${ty}_${th}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../lib/libcascadeclassifier_amd_${ty}_${th}.so build/cc_xml.o build/cc_cascade.o build/cc_host.o build/cc_eval.o build/cc_split.o build/cc_comm.o build/cc_detect_${ty}_${th}.o -ldl
echo built ../lib/libcascadeclassifier_amd_${ty}_${th}.so
