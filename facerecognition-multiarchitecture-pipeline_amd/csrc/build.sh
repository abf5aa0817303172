#!/bin/bash
# Build libfrmap_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libfrmap_hip.so
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Wno-unused-value"
objs=()
pids=()
for f in conv_igemm.hip conv_pp.hip conv_small_cin.hip stem_pool.hip stem_s2d.hip layout_pool.hip transformer.hip head_match.hip resize.hip c_api.cpp model_api.cpp model_families.cpp; do
  o="build_${f%.*}.o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ frmap_common.h -nt "$o" ] || [ model_api.h -nt "$o" ] || [ ../../include/frmap_hip.h -nt "$o" ]; then
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
# The second-generation kernels order their LDS-DMA by hand-counted `s_waitcnt vmcnt(n)`; a scratch spill inside such a loop
# is a vector-memory access the count does not know about (stale LDS could be read).  Every *_pp_kernel must therefore compile
# without scratch: check the resource-usage remarks of conv_pp.hip (one extra, parallel, compile; FRMAP_SKIP_SPILL_CHECK=1 skips it).
spill_pid=""
if [ "${FRMAP_SKIP_SPILL_CHECK:-0}" != "1" ] && { [ ! -f .pp_spill_ok ] || [ conv_pp.hip -nt .pp_spill_ok ] || [ frmap_common.h -nt .pp_spill_ok ]; }; then
  ( $HIPCC $FLAGS -Rpass-analysis=kernel-resource-usage -c conv_pp.hip -o /dev/null 2> .pp_remarks.txt || exit 1
    python3 - <<'PY' || exit 1
import re, sys
blocks = open(".pp_remarks.txt").read().split("remark: Function Name: ")[1:]
bad = []
for b in blocks:
    name = b.split(" ")[0]
    if "pp_kernel" not in name:
        continue
    sc = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
    sp = int(re.search(r"VGPRs Spill: (\d+)", b).group(1))
    if sc or sp:
        bad.append((name, sc, sp))
if not blocks:
    sys.exit("spill check: no resource-usage remarks parsed")
for name, sc, sp in bad:
    print(f"spill check: {name} uses {sc} B/lane of scratch ({sp} VGPRs spilled)", file=sys.stderr)
sys.exit(1 if bad else 0)
PY
    touch .pp_spill_ok ) &
  spill_pid=$!
fi
fail=0
for p in "${pids[@]:-}"; do
  if [ -n "$p" ] && ! wait "$p"; then fail=1; fi
done
if [ -n "$spill_pid" ] && ! wait "$spill_pid"; then
  echo "build failed: a *_pp_kernel spills registers (see above)" >&2
  exit 1
fi
if [ "$fail" != 0 ]; then
  echo "build failed" >&2
  exit 1
fi
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${objs[@]}"
echo "built $(realpath $OUT)"
# the plain-C host of the model-level ABI (examples/cabi_host.c): gcc + the HIP runtime, no Python
EX=../../examples
if [ ! -f $EX/cabi_host ] || [ $EX/cabi_host.c -nt $EX/cabi_host ] || [ ../../include/frmap_hip.h -nt $EX/cabi_host ]; then
  gcc $EX/cabi_host.c -I../../include -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -O2 -o $EX/cabi_host -L.. -l:libfrmap_hip.so \
      -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../facerecognition-multiarchitecture-pipeline_amd' -Wl,-rpath,/opt/rocm/lib
  echo "built $(realpath $EX/cabi_host)"
fi
