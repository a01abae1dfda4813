#!/bin/bash
# final GPU pass 1: every GPU test, the optional terms' rates, then profile set a (C3: records kernel, producers' kernel, per-ray kernel)
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_final_pytest.log 2>&1; rc=$?; echo pytest rc $rc; tail -3 gpurun_out/r05_final_pytest.log
[ $rc -ne 0 ] && exit $rc
bash tools/aux_sparse.sh > gpurun_out/r05_aux_sparse.log 2>&1; tail -3 gpurun_out/r05_aux_sparse.log
timeout -k 10 300 python tools/aux_rate.py > gpurun_out/r05_aux_rate.txt 2>&1; cat gpurun_out/r05_aux_rate.txt
bash tools/make_profiles.sh a
