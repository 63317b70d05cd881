#!/bin/bash
# Generates the run-time specialised cascade kernel of a cascade on the CPU (no device needed) and compiles it with hipcc
# to ISA, printing the kernel's resource usage: the quick look at registers / spills / code size before a GPU run.
#   tools/spec_isa.sh <cascade.xml> <stages> [out_prefix] [extra -D flags...]
set -e
xml=$1; k=$2; out=${3:-/tmp/spec/k}; shift; shift; shift || true
mkdir -p "$(dirname "$out")"
here=$(cd "$(dirname "$0")/.." && pwd)
CCAMD_CACHE_DIR= CCAMD_DUMP_SPEC_SOURCE=$out.raw.hip python3 - "$xml" "$k" <<PY
import sys, ctypes as C
sys.path.insert(0, "$here")
from cascadeclassifier_amd import _lib as L
c = C.c_void_p(); n = C.c_size_t()
L.check(L.lib().cc_cascade_load_xml(sys.argv[1].encode(), C.byref(c)))
L.check(L.lib().cc_cascade_compile_specialized(c, int(sys.argv[2]), b"gfx950", C.byref(n)))
print("hiprtc code object bytes:", n.value)
PY
grep -v '^typedef ' $out.raw.hip > $out.hip
lbp=""; grep -q '<featureType>LBP' "$xml" && lbp="-DCC_SPEC_LBP"
# the library compiles kernels with 16-bit STEP-2 tiles for 7 wavefronts per SIMD (their LDS footprint allows it) and defines CC_SPEC_TILE16
t16=""; dw=5
if [ "$(grep -c 'reinterpret_cast<const unsigned short\*>(b)' $out.hip)" -gt 1 ]; then t16="-DCC_SPEC_TILE16"; dw=7; fi
# pair tile (CCAMD_SPEC_PAIR16=1): the generated source specialises spec_stage_pair
if grep -q 'void spec_stage_pair<2>' $out.hip; then t16="-DCC_SPEC_PAIR16"; fi
# LBP kernels with 16-bit tiles are compiled for tiles of 16 window rows (spec_tile_rows in cc_detect.hip)
if [ -n "$lbp" ] && [ "$t16" = "-DCC_SPEC_TILE16" ] && [ -z "$CC_TILE_Y" ]; then CC_TILE_Y=${CCAMD_SPEC_TILE_Y:-20}; dw=6; fi
W=$(grep -o '<width>[0-9]*' "$xml" | head -1 | grep -o '[0-9]*'); H=$(grep -o '<height>[0-9]*' "$xml" | head -1 | grep -o '[0-9]*')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -include hip/hip_runtime.h -DCC_SPEC_STAGES=$k \
  -DCC_TILE_Y=${CC_TILE_Y:-8} -DCC_EVAL_THREADS=${CC_EVAL_THREADS:-256} -DCC_EVAL_MIN_WAVES_PER_EU=${WAVES:-$dw} -DCC_SPEC_W0=$W -DCC_SPEC_H0=$H $lbp $t16 "$@" \
  --cuda-device-only -S -o $out.s -x hip $out.hip -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "remark|error" | sed 's/.*remark: //' | head -40
