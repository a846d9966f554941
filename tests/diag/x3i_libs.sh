#!/bin/bash
# A/B of library builds of the three-term inference kernel: every build runs tests/diag/x3i_variants.py in its own process
# usage: x3i_libs.sh name1 name2 ...   (nerf-pytorch_amd/lib/x3i_<name>.so)
for rep in 1 2; do
for n in "$@"; do
  echo "== $n"
  NERF_HIP_LIB=$PWD/nerf-pytorch_amd/lib/x3i_$n.so VARIANTS=0 timeout -k 10 120 python tests/diag/x3i_variants.py ${MODEL_ARG} 2>&1 | grep -E "x3i v0 +[0-9]|again|x3i v0 +colors" || exit 1
done
done
