#!/bin/bash
# same-box A/B of library builds on the whole forward: [RAJNI_AB_FORMAT=fp8|fp8_mfma] tools/ab_libs.sh libA.so libB.so ...
# Two passes, the second in REVERSE order: successive processes on a box alternate between a slower and a faster state (~0.5 %,
# whatever they run - measured with two identical builds, profiles/r03_h_ab_gelu_two_chains.txt), so "A B A B" hands every
# second library that bonus; "A B B A" gives each library one run of either kind.  Compare class times, not only totals.
cd "$(dirname "$0")/.."
run() {
  echo -n "$1: "
  RAJNI_HIP_LIB=$PWD/rajni-vit_amd/rajni_amd/lib/$1 timeout -k 10 200 python tools/fwd_time.py $RAJNI_AB_FORMAT 2>&1 | grep -v amdgpu.ids | tail -1
}
for lib in "$@"; do run "$lib"; done
rev=()
for lib in "$@"; do rev=("$lib" "${rev[@]}"); done
for lib in "${rev[@]}"; do run "$lib"; done
