for ab in "" on_demand; do
  echo "### flat8g ablate='$ab'"
  GANON_HIP_ABLATE=$ab timeout 600 python bench.py --steps 10 --warmup 2 --no-extra --no-cpu-baseline --no-variants 2>/dev/null | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); ro=r['roofline']
print('value',r['value'],'ms',r['ms_per_step'],'frac',ro['frac'],'frac_fetched',ro['frac_fetched'],'launch_ms',ro['avg_launch_ms'],'every_row_ms',ro['avg_launch_ms_every_row'])"
done
