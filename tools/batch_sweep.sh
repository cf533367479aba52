for sc in soup atrium; do for b in 32 64 128; do
  python bench.py --scene $sc --batch $b --steps $((2*b)) --warmup $b --repeats 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$sc batch $b', 'value', j['value'], 'ms/step', j['ms_per_step'], 'trace_us', j['roofline'].get('avg_launch_us'), 'frac', j['roofline']['frac'])"
done; done
