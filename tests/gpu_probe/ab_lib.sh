#!/bin/bash
# A/B of two BUILDS of the library on tests/gpu_probe/step_time.py:  ab_lib.sh "<configs>" <lib A> <lib B> ...   ("-" = the in-tree build)
cfgs=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  [ "$lib" = "-" ] && ev="IMK_AB_DEFAULT=1" || ev="IMK_LIB_PATH=$lib"
  for cfg in $cfgs; do
    echo "[$lib] $cfg: $(env $ev CONFIG=${cfg%%:*} ALPHA=${cfg##*:} python tests/gpu_probe/step_time.py 2>&1 | grep -E 'train step|inference' | sed 's/(.*//' | tr '\n' ' ')"
  done
done
done
