for n in "$@"; do
  SYNTHRAY_LIB=ab/libsynthray_$n.so python bench.py --steps 5 --warmup 1 --cpu-sample 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n', 'deposit ms', round(d['roofline']['deposit_kernel_ms'],3), 'step', round(d['ms_per_step'],2))"
done
