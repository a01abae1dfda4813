#!/bin/bash
# usage (GPU box): bash tools/final_round.sh <a|b>   a: GPU tests + C3 profiles; b: C2 / C4 / C5 profiles + every bench line
R=$GRAFT_REPO_ROOT; cd $R
if [ "$1" = a ]; then
  timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/fin_pytest.log 2>&1; echo pytest rc $?; tail -3 gpurun_out/fin_pytest.log
  bash tools/make_profiles_r03.sh a
else
  bash tools/make_profiles_r03.sh b
fi
