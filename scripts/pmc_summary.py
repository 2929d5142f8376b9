#!/usr/bin/env python3
"""Summarise the counter passes of scripts/pmc_pass.sh into per-kernel, per-launch numbers.

    python scripts/pmc_summary.py <calibration dir with cal_fetch/ cal_write/> <name>=<pmc dir> [...]  > profiles/<round>_pmc_summary.json

HBM bytes per launch = FETCH_SIZE x (bytes per count) + WRITE_SIZE x (bytes per count), the two units measured on the same
box with scripts/dev/pmc_calib.hip (1 GiB streamed with this project's 8-byte-per-lane pattern), as
/opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes (separate --pmc passes; the guide's factor 2 on FETCH_SIZE for gfx950
shows up as 2048 B per count).  Kernel durations come from the kernel trace of the same pass."""
import collections
import csv
import glob
import json
import os
import sys

KEEP = ("lin_wave_kernel", "qp_kernel", "rti_fused_kernel", "rti_fused_kernel_w2", "rti_fused_kernel_grid", "rti_window_kernel", "rti_window_kernel_grid", "rti_window_kernel_res", "plant_kernel", "candidates_kernel",
        "window_kernel", "ekf_update_kernel_dpp", "ekf_update_kernel_sp")


def kname(s):
    return s.split("(")[0].replace("brov::", "").replace("void ", "").strip()


def per_kernel(dirname, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(dirname, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def durations(dirname):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(dirname, "*kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            acc[kname(r["Kernel_Name"])].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3)
    return acc


def mean_tail(v, skip):
    v = v[skip:] if len(v) > skip else v
    return sum(v) / max(1, len(v))


def main():
    cal = sys.argv[1]
    cal_bytes = float(1 << 30)
    f_read = per_kernel(os.path.join(cal, "cal_fetch"), "FETCH_SIZE").get("calib_read8", [0])[-1]
    w_write = per_kernel(os.path.join(cal, "cal_write"), "WRITE_SIZE").get("calib_write8", [0])[-1]
    fu, wu = (cal_bytes / f_read if f_read else None), (cal_bytes / w_write if w_write else None)
    cal_src = "this session (scripts/dev/pmc_calib.hip)"
    if not (fu and wu):
        # no calibration pass in this session (the calibration binary is built on the GPU box by scripts/profile_round.sh; a session that
        # could not build it leaves none): the units of the last session that had one -- what bench.py's in-run traffic uses as well
        prev = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r5_pmc_summary.json")
        if os.path.exists(prev):
            c = json.load(open(prev))["calibration"]
            fu, wu = c["bytes_per_FETCH_SIZE_count"], c["bytes_per_WRITE_SIZE_count"]
            cal_src = "profiles/r5_pmc_summary.json (no calibration pass in this session)"
    out = {"calibration": {"bytes_streamed": cal_bytes, "FETCH_SIZE_reading_1GiB": f_read, "WRITE_SIZE_writing_1GiB": w_write,
                           "bytes_per_FETCH_SIZE_count": fu, "bytes_per_WRITE_SIZE_count": wu, "units_from": cal_src,
                           "note": "nominal unit is 1024 B; the ratio to it is the gfx950 correction for this access pattern"},
           "runs": {}}
    for spec in sys.argv[2:]:
        name, d = spec.split("=", 1)
        run = {"source": d, "hbm_bytes_per_launch": {}, "kernel_us": {}, "counters_per_launch": {}}
        skip = 10   # 2 timed passes x 5 warm-up launches precede / interleave; drop the first launches of every kernel
        pf, pw = per_kernel(os.path.join(d, "fetch"), "FETCH_SIZE"), per_kernel(os.path.join(d, "write"), "WRITE_SIZE")
        for k in KEEP:
            if pf.get(k) and pw.get(k) and fu and wu:
                fa, wa = mean_tail(pf[k], skip), mean_tail(pw[k], skip)
                run["hbm_bytes_per_launch"][k] = fa * fu + wa * wu
                run["counters_per_launch"].setdefault(k, {}).update(FETCH_SIZE=fa, WRITE_SIZE=wa, launches=len(pf[k]))
        for sub in ("sq", "lds"):
            f = glob.glob(os.path.join(d, sub, "*counter_collection.csv"))
            if not f:
                continue
            names = sorted(set(r["Counter_Name"] for r in csv.DictReader(open(f[0]))))
            for c in names:
                for k, v in per_kernel(os.path.join(d, sub), c).items():
                    if k in KEEP:
                        run["counters_per_launch"].setdefault(k, {})[c] = mean_tail(v, skip)
        for k, v in durations(os.path.join(d, "sq")).items():
            if k in KEEP:
                run["kernel_us"][k] = mean_tail(v, skip)
        for k, c in run["counters_per_launch"].items():
            if "SQ_WAVE_CYCLES" in c and c.get("SQ_WAVES"):
                # SQ_WAVE_CYCLES / SQ_WAIT_* tick once per 4 shader cycles (a 96 k-cycle fused solve reads 24.5 k)
                c["derived"] = {
                    "shader_cycles_per_wave": 4 * c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"],
                    "wait_any_frac": c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"],
                    "mfma_issue_frac_of_wave_time": (c.get("SQ_INSTS_MFMA", 0) / c["SQ_WAVES"] * 64) / (4 * c["SQ_WAVE_CYCLES"] / c["SQ_WAVES"]),
                    "valu_insts_per_wave": c.get("SQ_INSTS_VALU", 0) / c["SQ_WAVES"],
                    "mfma_insts_per_wave": c.get("SQ_INSTS_MFMA", 0) / c["SQ_WAVES"]}
            if "SQ_ACTIVE_INST_LDS" in c and c["SQ_ACTIVE_INST_LDS"]:
                c.setdefault("derived", {})["lds_bank_conflict_over_active"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_ACTIVE_INST_LDS"]
        out["runs"][name] = run
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
