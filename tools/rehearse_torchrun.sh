# usage (ONE-GPU box): bash tools/rehearse_torchrun.sh -> the N > 1 job as the DRIVER starts it (python -m torch.distributed.run ... bench.py --gpus N),
# every rank on device 0: once with the image sum played by the host (--rehearse-shared-gpu), once asking RCCL itself (--shared-gpu-rccl: RCCL
# refuses two ranks on one device -> the marked host-sum line).  Prints how many lines each job put on stdout (must be 1) and check.multi_gpu.
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
run() {
  name=$1; shift
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 \
    bench.py --gpus 2 --steps 2 --warmup 1 --grid 256 --rays 1e6 --cpu-sample 0 --other-steps 0 "$@" > gpurun_out/tr_$name.json 2> gpurun_out/tr_$name.err
  echo "$name rc=$? stdout lines: $(wc -l < gpurun_out/tr_$name.json)"
  python - gpurun_out/tr_$name.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["n_gpus"], d["scaling"], "collective:", d.get("collective"), "| multi_gpu:", (d.get("check") or {}).get("multi_gpu"), "| timing:", d.get("timing", {}).get("per_rank_ms_for_the_steps"))
PY
}
run host --rehearse-shared-gpu && run rccl --shared-gpu-rccl
