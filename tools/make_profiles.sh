# usage (GPU box): bash tools/make_profiles.sh    -> gpurun_out/: kernel-trace stats + PMC passes of the headline workload (C3) in
# both builds, the VALU issue table and its class calibration.  Then, in the build container: bash tools/collect_profiles.sh
set -e
R=$GRAFT_REPO_ROOT; cd $R
bash tools/profile_valu_issue.sh > gpurun_out/mp_valu.log 2>&1
for prec in f64 mixed; do
  bash tools/profile_r02.sh c3_$prec --precision $prec > gpurun_out/mp_$prec.log 2>&1
done
echo collected
