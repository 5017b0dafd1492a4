#!/bin/bash
# A/B of environment switches on tests/gpu_probe/step_time.py:  ab_env.sh "<configs>" "<env 1>" "<env 2>" ...
#   configs: space-separated "name:alpha" (e.g. "city:2 suim:1"); an env of "-" = defaults
cfgs=$1; shift
for e in "$@"; do
  [ "$e" = "-" ] && ev="IMK_AB_DEFAULT=1" || ev="$e"
  for cfg in $cfgs; do
    echo "[$e] $cfg: $(env $ev CONFIG=${cfg%%:*} ALPHA=${cfg##*:} python tests/gpu_probe/step_time.py 2>&1 | grep -E 'train step|inference' | sed 's/(.*//' | tr '\n' ' ')"
  done
done
