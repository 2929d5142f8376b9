#!/usr/bin/env python3
"""Turn the raw output of scripts/profile_round.sh (gpurun_out/prof_<tag>/) into the committed, judged summaries under profiles/:
<round>_kernel_stats.csv / <round>_cfg5_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py` and `bench.py --config 5`),
<round>_bench.json (the bench lines of that session) and <round>_pmc_summary.json (scripts/pmc_summary.py: HBM bytes per launch
from FETCH_SIZE / WRITE_SIZE with the calibration of scripts/dev/pmc_calib.hip, SQ and LDS counters per launch).

    python scripts/collect_profiles.py r2 r2
"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag, rnd):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    for sub, name in (("stats", "kernel_stats"), ("stats_cfg5", "cfg5_kernel_stats"), ("stats_ekf", "ekf_kernel_stats"),
                      ("stats_res", "resident_kernel_stats"), ("stats_mid", "mid_batch_kernel_stats")):
        f = os.path.join(src, sub, "stats_kernel_stats.csv")
        if os.path.exists(f):
            shutil.copy(f, os.path.join(dst, f"{rnd}_{name}.csv"))
    bench = {}
    for p in sorted(glob.glob(os.path.join(src, "bench_*.json"))):
        lines = [ln for ln in open(p).read().splitlines() if ln.startswith("{")]
        if lines:
            bench[os.path.basename(p)[:-5]] = json.loads(lines[-1])
    json.dump(bench, open(os.path.join(dst, f"{rnd}_bench.json"), "w"), indent=1)
    runs = []
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if os.path.isdir(d):
            runs.append(f"{os.path.basename(d)[4:]}={d}")
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "pmc_summary.py"), src] + runs, text=True)
    o = json.loads(out)
    for r in o["runs"].values():
        r["source"] = os.path.relpath(r["source"], ROOT)
    json.dump(o, open(os.path.join(dst, f"{rnd}_pmc_summary.json"), "w"), indent=1)
    sweep = os.path.join(src, "batch_sweep.log")
    if os.path.exists(sweep):
        lines = [ln for ln in open(sweep).read().splitlines() if ln.startswith("{")]
        if lines:
            json.dump(json.loads(lines[-1]), open(os.path.join(dst, f"{rnd}_batch_sweep.json"), "w"), indent=1)
    # what the parity rules of the last full `pytest -m gpu` run checked and excused (tests/conftest.py session hook)
    pe = os.path.join(ROOT, "gpurun_out", "parity_excused.json")
    if os.path.exists(pe):
        d = json.load(open(pe))
        if "u0_abs" not in d.get("by_rule", {}) or "bvls_abs_1e-8" not in d.get("by_rule", {}):
            print("WARNING: gpurun_out/parity_excused.json is not from a full `pytest tests -m gpu` run (rules:", sorted(d.get("by_rule", {})), ")")
        d["entries_with_excused"] = [{k: v for k, v in e.items() if k != "disagreements" or v} for e in d.get("entries_with_excused", [])][:40]
        d["note"] = ("summary of gpurun_out/parity_excused.json: per rule, the number of rule calls, instances checked and instances excused in ONE run of "
                     "`pytest tests -m gpu`; entries_with_excused lists the calls that excused anything (first 40)")
        json.dump(d, open(os.path.join(dst, f"{rnd}_parity_excused.json"), "w"), indent=1)
    for txt in ("phase_stamps_N20.txt", "phase_stamps_N80.txt", "pit_stamps_N80.txt", "phase_stamps_N80_B1_sequential.txt", "shim_latency.txt", "split_tick_latency.txt"):
        if os.path.exists(os.path.join(src, txt)):
            shutil.copy(os.path.join(src, txt), os.path.join(dst, f"{rnd}_{txt}"))
    print("wrote", sorted(f for f in os.listdir(dst) if f.startswith(rnd + "_")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
