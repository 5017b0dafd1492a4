#!/bin/bash
# The whole end-of-round evidence in ONE gpurun call, on one box, in the order that makes every committed line self-consistent:
#   1. collect_round.sh    -- traces, counters, per-configuration tables + provenance (bench.py sha, kernel-source id)
#   2. install_round.py    -- ON THE BOX: profiles/ now holds the files a line may replay from (the same script is run again at home
#                             on the merged gpurun_out/: it is a pure function of that directory)
#   3. rebench.sh          -- the six plain commands again: their lines now carry the replayed values
#   4. final_check.sh      -- GPU suite, smoke(), the driver's command, 2- / 8-rank functional lines
#   /usr/local/graft/bin/gpurun --timeout 5000 -- 'RND=r06 bash profiles/round_end.sh'
RND=${RND:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
RND=$RND bash profiles/collect_round.sh > gpurun_out/${RND}_collect.log 2>&1; echo "collect rc=$?"
python3 profiles/install_round.py $RND > gpurun_out/${RND}_install.log 2>&1; python3 profiles/traffic_ratio.py $RND
RND=$RND bash profiles/rebench.sh > gpurun_out/${RND}_rebench.log 2>&1; echo "rebench rc=$?"
RND=$RND bash tests/gpu_probe/final_check.sh
