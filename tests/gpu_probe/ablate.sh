#!/bin/bash
# per-kernel inference times under the timing-only ablation builds of imk_conv.hip (-DIMK_ABL=<bits>; results are WRONG by design):
#   ablate.sh <lib> ...      ("-" = the in-tree build).  Bits: 1 loads from one cached address, 2 no final stores, 4 no staging transform,
#   8 no first (1x1) stage, 16 one 3x3 k-step instead of all
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "$@"; do
  d=$R/gpurun_out/_abl_$(basename $lib .so)
  rm -rf $d
  if [ "$lib" = "-" ]; then unset IMK_LIB_PATH; else export IMK_LIB_PATH=$R/$lib; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/tests/gpu_probe/infer_ab.py > $d.log 2>&1
  echo "== $lib: $(grep 'ms  probs' $d.log | sed 's/ env.*//')"
  python3 $R/tests/gpu_probe/kstats.py $d 8 | grep -E "conv_pipe|conv_mfma|head"
  rm -rf $d
done
