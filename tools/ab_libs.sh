#!/bin/bash
# same-box A/B of library builds on the whole forward: tools/ab_libs.sh libA.so libB.so ...  (two passes, interleaved)
cd "$(dirname "$0")/.."
for pass in 1 2; do
  for lib in "$@"; do
    echo -n "$lib: "
    RAJNI_HIP_LIB=$PWD/rajni-vit_amd/rajni_amd/lib/$lib timeout -k 10 200 python tools/fwd_time.py 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
