# flat8g with the batch run as a minimiser || count pipeline of read ranges (switch chunk=N, two HIP streams of one gn_stream) against one range
for ab in "" "chunk=5000000" "chunk=2500000" "chunk=1250000"; do
  echo "### flat8g ablate='$ab'"
  GANON_HIP_ABLATE=$ab timeout 300 python bench.py --steps 10 --warmup 2 --no-extra --no-cpu-baseline --no-variants --no-every-row 2>/dev/null | tail -1 | python -c "
import sys,json
r=json.loads(sys.stdin.read()); ro=r['roofline']
print('value',r['value'],'ms',r['ms_per_step'],'launch_ms',ro['avg_launch_ms'],'launches',ro['launches_per_step'],'checksum',r['config']['match_checksum_all_ranks'])"
done
