#!/bin/bash
# final GPU pass 1: every GPU test, sr_trace on host arrays at the default chunk size, the optional terms' rates, then profile set a
# (C3: records kernel, producers' kernel, per-ray kernel)
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_final_pytest.log 2>&1; rc=$?; echo pytest rc $rc; tail -3 gpurun_out/r05_final_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/pcie_chunks.py 2621440 > gpurun_out/r05_pcie_default.txt 2>&1; tail -2 gpurun_out/r05_pcie_default.txt
timeout -k 10 300 python tools/pcie_chunks.py 2621440 --rays 4e6 6e6 2e7 > gpurun_out/r05_pcie_sizes.txt 2>&1; tail -4 gpurun_out/r05_pcie_sizes.txt
bash tools/aux_sparse.sh > gpurun_out/r05_aux_sparse.log 2>&1; tail -3 gpurun_out/r05_aux_sparse.log
timeout -k 10 300 python tools/aux_rate.py > gpurun_out/r05_aux_rate.txt 2>&1; cat gpurun_out/r05_aux_rate.txt
bash tools/make_profiles.sh a
