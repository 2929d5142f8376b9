#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5k; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ticks.py tests/test_gpu_windowed.py -m gpu -q --timeout 600 > $O/first.log 2>&1; echo "first rc=$?"; grep -n "^E  \|FAILED\|passed\|failed" $O/first.log | cut -c1-300 | tail -10
python - <<'PY' 2>&1 | tee $O/ticks_rate_windowed.txt
import time, numpy as np, torch, bluerov2_amd as ba, bench
def rate(B, N, sat, ticks=20, warm=5, reps=3):
    x0, circ = bench.synthetic_inputs(B, seed=4)
    if sat: x0 = bench.saturate(x0, sat, seed=7)
    out = {}
    for mode in ("per_step", "one_launch"):
        s = ba.BatchSolver(B, ba.SolverOptions(N, 1.0 / N)); s.set_x0(x0); s.set_params(ba.P_NOMINAL); s.set_trajectory(circ)
        best = []
        for r in range(reps):
            s.init_iterate_default()
            for k in range(warm): s.set_yref_from_trajectory(k, 16); s.solve()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            if mode == "per_step":
                for k in range(warm, warm + ticks): s.set_yref_from_trajectory(k, 16); s.solve()
            else:
                s.set_yref_from_trajectory(warm, 16); s.solve_ticks(ticks, 1)
            torch.cuda.synchronize(); best.append(B * ticks / (time.perf_counter() - t0))
        out[mode] = (float(np.median(best)), int((s.results()["status"] != 0).sum()), int(s.results()["qp_iter"].max()))
        s.close()
    print(f"B={B} N={N} saturated={sat}: per-step launches {out['per_step'][0]/1e6:.3f} M/s, one launch of {ticks} steps {out['one_launch'][0]/1e6:.3f} M/s  ({out})")
rate(4096, 40, 0.0); rate(4096, 40, 0.25); rate(4096, 80, 0.0); rate(4096, 80, 0.25)
PY
python bench.py --config 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print({k:(round(v['solves_per_s']/1e6,3), v['kernel_path']) for k,v in o['sweep'].items()})"
