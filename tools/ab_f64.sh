for n in "$@"; do
  SYNTHRAY_LIB=ab/libsynthray_$n.so python bench.py --precision f64 --steps 3 --warmup 1 --cpu-sample 20000 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n', round(d['ms_per_step'],2), round(d['roofline']['kernel_ms'],2), d['check']['max_dx_m'])"
done
