for i in 1 2; do
for v in base new; do
  if [ $v = base ]; then export HAK_LIB=$PWD/cuda-akaze_amd/libhipakaze_base.so; else unset HAK_LIB; fi
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-configs --no-upload --no-fast --no-verify > /tmp/ab_$v.json 2>/dev/null
  python -c "
import json; d=json.load(open('/tmp/ab_$v.json')); c={x['class']:x['ms'] for x in d['roofline']['classes']}
print('$v', d['value'], c)"
done; done
