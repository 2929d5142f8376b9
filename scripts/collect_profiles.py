#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of scripts/profile_round.sh (gpurun_out/prof_<tag>/) into the committed, judged summaries
under profiles/:  <round>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the default bench command),
<round>_bench.json (the bench lines of that session) and <round>_pmc_summary.json (HBM bytes per launch from
FETCH_SIZE / WRITE_SIZE, corrected with the calibration kernels of scripts/dev/pmc_calib.hip as
/opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes: separate --pmc passes, unit = KiB... see below).

    python scripts/collect_profiles.py r1b r1
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(dirname, counter):
    f = glob.glob(os.path.join(dirname, "*counter_collection.csv"))
    if not f:
        return {}
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0].replace("brov::", "")].append(float(r["Counter_Value"]))
    return acc


def main(tag, rnd):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "stats", "stats_kernel_stats.csv"), os.path.join(dst, f"{rnd}_kernel_stats.csv"))
    bench = {}
    if os.path.exists(os.path.join(src, "stats_ekf", "stats_kernel_stats.csv")):
        shutil.copy(os.path.join(src, "stats_ekf", "stats_kernel_stats.csv"), os.path.join(dst, f"{rnd}_ekf_kernel_stats.csv"))
    for name in ("bench_plain", "bench_stats", "bench_forced_ipm", "bench_streaming", "bench_b16384", "bench_ekf"):
        p = os.path.join(src, name + ".json")
        if os.path.exists(p):
            lines = [ln for ln in open(p).read().splitlines() if ln.startswith("{")]
            if lines:
                bench[name] = json.loads(lines[-1])
    json.dump(bench, open(os.path.join(dst, f"{rnd}_bench.json"), "w"), indent=1)
    # calibration: rocprofv3 FETCH_SIZE / WRITE_SIZE are in KiB-like units of the TCC EA request counters; the guide says
    # FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 and that other widths must be calibrated.
    cal_bytes = float(1 << 30)
    cf = per_kernel(os.path.join(src, "cal_fetch"), "FETCH_SIZE")
    cw = per_kernel(os.path.join(src, "cal_write"), "WRITE_SIZE")
    f_read = cf.get("calib_read8", [0])[-1]
    w_write = cw.get("calib_write8", [0])[-1]
    fetch_unit = cal_bytes / f_read if f_read else None   # bytes per FETCH_SIZE count for 8 B/lane coalesced reads
    write_unit = cal_bytes / w_write if w_write else None
    out = {"source": f"gpurun_out/prof_{tag} (scripts/profile_round.sh)", "batch": 4096, "N": 20,
           "calibration": {"bytes_streamed": cal_bytes, "FETCH_SIZE_reading_1GiB": f_read, "WRITE_SIZE_writing_1GiB": w_write,
                           "bytes_per_FETCH_SIZE_count": fetch_unit, "bytes_per_WRITE_SIZE_count": write_unit,
                           "note": "nominal unit is 1024 B; ratio to it is the gfx950 correction for this access pattern"},
           "hbm_bytes_per_launch": {}, "raw": {}}
    pf = per_kernel(os.path.join(src, "pmc_fetch"), "FETCH_SIZE")
    pw = per_kernel(os.path.join(src, "pmc_write"), "WRITE_SIZE")
    for k in ("lin_wave_kernel", "qp_kernel", "rti_fused_kernel"):
        fv = pf.get(k, [])
        wv = pw.get(k, [])
        if fv and wv and fetch_unit and write_unit:
            fa, wa = sum(fv[5:]) / len(fv[5:]), sum(wv[5:]) / len(wv[5:])
            out["raw"][k] = {"FETCH_SIZE": fa, "WRITE_SIZE": wa, "launches": len(fv)}
            out["hbm_bytes_per_launch"][k] = fa * fetch_unit + wa * write_unit
    sq = {}
    for cname in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES",
                  "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
        for k, v in per_kernel(os.path.join(src, "pmc_sq"), cname).items():
            if k in ("lin_wave_kernel", "qp_kernel", "rti_fused_kernel"):
                sq.setdefault(k, {})[cname] = sum(v[5:]) / max(1, len(v[5:]))
    out["sq_counters_per_launch"] = sq
    lds = {}
    for cname in ("SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_SALU",
                  "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
        for k, v in per_kernel(os.path.join(src, "pmc_lds"), cname).items():
            if k in ("lin_wave_kernel", "qp_kernel", "rti_fused_kernel"):
                lds.setdefault(k, {})[cname] = sum(v[5:]) / max(1, len(v[5:]))
    out["lds_counters_per_launch"] = lds
    ekf = {}
    for cname in ("SQ_WAVE_CYCLES", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS",
                  "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_ANY"):
        for k, v in per_kernel(os.path.join(src, "pmc_ekf"), cname).items():
            if k.startswith("ekf_update"):
                ekf.setdefault(k, {})[cname] = sum(v[2:]) / max(1, len(v[2:]))
    out["ekf_counters_per_launch_B16384"] = ekf
    json.dump(out, open(os.path.join(dst, f"{rnd}_pmc_summary.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
