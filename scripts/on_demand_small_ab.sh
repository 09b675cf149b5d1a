# the product's batch size (1 M reads, classify.cpp kBatchReads) and half of it: units on demand against the static hand-out
for n in 1048576 524288 2097152; do
for ab in "" on_demand; do
  echo "### flat8g reads=$n ablate='$ab'"
  GANON_HIP_ABLATE=$ab timeout 200 python bench.py --reads $n --steps 30 --warmup 5 --no-extra --no-cpu-baseline --no-variants --no-every-row 2>/dev/null | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); ro=r['roofline']
print('value',r['value'],'ms',r['ms_per_step'],'launch_ms',ro['avg_launch_ms'])"
done
done
