for rep in 1 2 3; do for L in 0 1; do BROV_DEV_WIN_LONG=$L python bench.py --config 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); l=o['configs']['config5_shard_sweep']['legs'] if 'configs' in o else o.get('legs',{}); print('long=$L rep $rep', {k:round(v['solves_per_s']/1e6,3) for k,v in l.items()} if l else round(o['value']/1e6,3))"; done; done
for L in 0 1; do BROV_DEV_WIN_LONG=$L python bench.py --config 5 --horizon 80 --force-ipm --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('forced ipm N80 long=$L', round(o['value']/1e6,3))"; done
