# round 5: the persistent sweep beyond the on-chip capacity: config 3's users / items with 10 M (on chip), 14 M, 20 M rows
cd $GRAFT_REPO_ROOT
for rows in 10000000 14000000 20000000; do
  python bench.py --rows $rows --steps 100 --warmup 5 --cpu-seconds 0 --fit-iters 0 --no-other-configs --long-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('rows', c['rows'], 'it/s', d['value'], 'row-iterations/s %.3g' % c['row_iterations_per_s'], 'plan_flags', c['plan_flags'], 'kernel', (d.get('roofline') or {}).get('kernel'))"
done
MFM_RES_NO_OVERFLOW=1 python bench.py --rows 20000000 --steps 50 --warmup 5 --cpu-seconds 0 --fit-iters 0 --no-other-configs --long-seconds 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('(no overflow form) rows', c['rows'], 'it/s', d['value'], 'row-iterations/s %.3g' % c['row_iterations_per_s'], 'plan_flags', c['plan_flags'])"
