#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
( time python bench.py --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err ) 2>&1 | grep real
python -c "
import json; d=json.load(open('/tmp/b.json'))
for k in ('mixed_batch_25pct_saturated','mixed_batch_25pct_saturated_shuffled','mixed_batch_25pct_saturated_iter_max_10'):
    v=d[k]; print(k, round(v['value']/1e6,2), v['ms_per_step'], v['median_tick_kernel_ms'], v['max_tick_kernel_ms'], v['status_histogram'], v['max_qp_iter_last_tick'])"
tail -2 /tmp/b.err
