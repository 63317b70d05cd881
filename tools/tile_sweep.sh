#!/bin/bash
# Tile-shape x register-budget sweep of the specialised cascade kernel (variants built by tools/build_tile_variant.sh).
# Each line: "<ty>_<threads> <waves_per_eu...>". Device pipeline only, 32 frames per launch; checks the rectangles of a
# small detection against the default build first (a variant that changes results is a bug, not a candidate).
run() {
  lib=$1; shift
  [ -f "$lib" ] || return
  for w in "$@"; do
    CCAMD_LIB=$lib CCAMD_CACHE_DIR= CCAMD_TRACE_HOST=1 CCAMD_SPEC_WAVES_PER_EU=$w python bench.py --steps 3 --warmup 1 --cpu-frames 0 --frames 32 --device-only --specialize 7 2> >(grep "resident blocks" >&2) | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib', 'waves_per_eu', $w, 'eval_ms/32f', d['kernel_ms_per_step']['eval_ms'])"
  done
}
L=cascadeclassifier_amd/lib
run $L/libcascadeclassifier_amd.so 5
run $L/libcascadeclassifier_amd_12_384.so 5 6
run $L/libcascadeclassifier_amd_8_512.so 6 8
run $L/libcascadeclassifier_amd_16_512.so 5 6
run $L/libcascadeclassifier_amd_16_256.so 5 6
run $L/libcascadeclassifier_amd_4_256.so 5 6
run $L/libcascadeclassifier_amd_6_384.so 6 7
