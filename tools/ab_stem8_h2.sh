for i in 1 2; do
  timeout 300 python bench.py --no-other-configs --no-cpu-baseline > gpurun_out/ab_h2_$i.json 2>/dev/null
  SAGEN_NO_STEM8_H2=1 timeout 300 python bench.py --no-other-configs --no-cpu-baseline > gpurun_out/ab_noh2_$i.json 2>/dev/null
done
python - <<'PY'
import json
for t in ('h2','noh2'):
    for i in (1,2):
        d=json.load(open('gpurun_out/ab_%s_%d.json'%(t,i)))
        print(t,i,d['value'],d['ms_per_step'],d['one_in_flight']['value'],d['roofline']['whole_step']['kernel_time_us_per_step'])
PY
