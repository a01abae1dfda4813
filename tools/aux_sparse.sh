#!/bin/bash
# usage (GPU box): bash tools/aux_sparse.sh -> the optional terms on the per-ray kernel (SYNTHRAY_F64_TILE=0; what a bundle below 8 rays per cell takes):
# round 4's one-pass kernel (all five fields, one wavefront per SIMD: SYNTHRAY_AUX_ONE_PASS=1) against the selected-field kernels (two passes
# for both terms, two wavefronts per SIMD), 256^3, at 30 and at 4 rays per cell of the beam's box; then the optional terms' GPU tests
out=gpurun_out/r05_aux_sparse.txt; : > $out
for n in 1e6 1.7e5; do
  for one in 1 0; do
    echo "## rays $n, SYNTHRAY_AUX_ONE_PASS=$one (per-ray kernel forced)" >> $out
    SYNTHRAY_F64_TILE=0 SYNTHRAY_AUX_ONE_PASS=$one timeout -k 10 300 python tools/aux_rate.py 256 $n 2>&1 | grep -E "kappa|Faraday|both" >> $out
  done
done
cat $out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "aux or optional or coherent or gather" > gpurun_out/r05_aux_pytest.log 2>&1; echo pytest rc $?; tail -3 gpurun_out/r05_aux_pytest.log
