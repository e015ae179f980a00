#!/bin/bash
# round 4 evidence: kernel-trace stats + PMC passes of the bench's configurations (tools/prof_round.sh), one group per call
cd ${GRAFT_REPO_ROOT:-.}
G=${1:-a}
case $G in
  a) bash tools/prof_round.sh a "hmc|normal32|65536|fd_sparse|L16" "hmc|normal32|65536|fd_dense|L16" "hmc|normal32|8192|fd_sparse|L16" ;;
  b) bash tools/prof_round.sh b "mh|refmodel20|65536" "mh|refmodel20|8192" "mh|c5|262144" "smc|c4|1048576" ;;
  c) bash tools/prof_round.sh c "hmc|c3|65536|fd_sparse|L16" "hmc|c3|8192|fd_sparse|L16" ;;
esac
