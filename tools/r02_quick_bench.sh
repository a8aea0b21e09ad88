cd "$GRAFT_REPO_ROOT"
for spec in "humanoid walk 1024" "humanoid walk 8192" "walker walk 8192" "hopper hop 4096"; do
  set -- $spec
  timeout -k 10 300 python bench.py --domain $1 --task $2 --batch $3 --no-cpu-baseline --steps 300 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 $2 $3', '%.4g env-steps/s' % d['value'], 'kernel %.4f ms' % d['roofline']['kernel_ms_avg'], d['config']['kernel_shape'])"
done
