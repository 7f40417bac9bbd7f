cd "${GRAFT_REPO_ROOT:-/root/repo}"
for s in 90 180 360 720 1440; do
  python3 bench.py --sectors $s --steps 40 --warmup 10 --no-extras --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('sectors %5d  ms/step %.4f  us/sector %.4f  frac %.4f' % ($s, d['ms_per_step'], d['ms_per_step']*1000/$s, r['frac']))"
done
