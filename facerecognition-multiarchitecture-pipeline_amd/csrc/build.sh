#!/bin/bash
# Build libfrmap_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libfrmap_hip.so
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value"
objs=()
pids=()
for f in conv_igemm.hip conv_pp.hip conv_small_cin.hip stem_pool.hip layout_pool.hip transformer.hip head_match.hip resize.hip c_api.cpp model_api.cpp; do
  o="build_${f%.*}.o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ frmap_common.h -nt "$o" ] || [ ../../include/frmap_hip.h -nt "$o" ]; then
    echo "hipcc $f"
    rm -f "$o"
    if [[ "$f" == *.cpp ]]; then
      $HIPCC $FLAGS -x hip -c "$f" -o "$o" &
    else
      $HIPCC $FLAGS -c "$f" -o "$o" &
    fi
    pids+=($!)
  fi
  objs+=("$o")
done
fail=0
for p in "${pids[@]:-}"; do
  if [ -n "$p" ] && ! wait "$p"; then fail=1; fi
done
if [ "$fail" != 0 ]; then
  echo "build failed" >&2
  exit 1
fi
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${objs[@]}"
echo "built $(realpath $OUT)"
