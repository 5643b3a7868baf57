#!/usr/bin/env python3
"""Run on the GPU box (via gpurun): rocprofv3 evidence for one bench.py configuration, summarised into a directory
that can be copied to profiles/ as it is.

    python3 tools/profile_round.py --tag r02_c2 [--workload NAME] [--streams 3] [--trace-steps 1000] [--no-pmc] [--stamp]

  * kernel trace + stats of `python3 bench.py <args>` (the program itself behind `--`: no env/bash wrappers)
      -> <out>/trace_kernel_stats.csv, <out>/bench.json (the JSON line bench.py printed in that same run)
  * PMC passes, one rocprofv3 run per counter set, never combined with a trace (MI355X_MICROARCH.md: HBM section;
    gpurun refuses the combination) -> <out>/pmc_summary.json (mean per launch of the dominant render kernel)
  * --stamp: also writes traffic_<round>.json / valu_<round>.json next to <out>, carrying the SHA-256 of the
    libmi355rt.so that was measured; bench.py reports these constants only for that very build.
Output goes to gpurun_out/prof_<tag>/ (gpurun merges gpurun_out/ back)."""
import argparse
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC_SETS = {
    "sq1": "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY",
    "sq2": "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT",
    "sq3": "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT",
    "sq4": "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_IOPS",
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE",
    "grbm": "GRBM_GUI_ACTIVE",
}


def run(cmd, log):
    with open(log, "w") as f:
        rc = subprocess.call(cmd, stdout=f, stderr=subprocess.STDOUT, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    if rc != 0:
        print(f"FAILED ({rc}): {' '.join(cmd)}\n" + open(log).read()[-3000:], flush=True)
        sys.exit(1)


def bench_json(log):
    for line in reversed(open(log).read().splitlines()):
        if line.startswith("{") and '"metric"' in line:
            return json.loads(line)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--trace-steps", type=int, default=992)
    ap.add_argument("--trace-warmup", type=int, default=48)
    ap.add_argument("--pmc-steps", type=int, default=32)
    ap.add_argument("--sets", default="sq1,sq2,sq3,sq4,fetch,write")
    ap.add_argument("--no-pmc", action="store_true")
    ap.add_argument("--stamp", action="store_true")
    ap.add_argument("--round", default="r03")
    ap.add_argument("--flags", type=int, default=0)
    a = ap.parse_args()
    out = os.path.join(REPO, "gpurun_out", f"prof_{a.tag}")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    bench = os.path.join(REPO, "bench.py")
    common = ["--streams", str(a.streams), "--no-cpu-baseline", "--no-dynamic"] + (["--workload", a.workload] if a.workload else []) + (["--flags", str(a.flags)] if a.flags else [])

    # 1. kernel trace of the bench command, without its serial pass and host-path timing: every render launch in the
    #    trace is then of the kind the timed region times (pre-heat, warm-up and timed steps on --streams streams), so
    #    the stats' average duration is directly comparable with roofline.kernel_ms of the bench line printed in the run
    tdir = os.path.join(out, "trace")
    run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", tdir, "--", "python3", bench,
         "--steps", str(a.trace_steps), "--warmup", str(a.trace_warmup), "--no-serial", "--no-host-path"] + common, os.path.join(out, "trace.log"))
    ks = glob.glob(os.path.join(tdir, "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], os.path.join(out, "trace_kernel_stats.csv"))
    bj = bench_json(os.path.join(out, "trace.log"))
    if bj:
        json.dump(bj, open(os.path.join(out, "bench.json"), "w"), indent=1)
    shutil.rmtree(tdir, ignore_errors=True)
    print("trace done:", (bj or {}).get("ms_per_step"), "ms/step", flush=True)

    # 2. PMC passes: short runs, nothing but the timed frames
    summary = {}
    if not a.no_pmc:
        for name in a.sets.split(","):
            pdir = os.path.join(out, "pmc_" + name)
            run(["rocprofv3", "--pmc"] + PMC_SETS[name].split() + ["--output-format", "csv", "-d", pdir, "--", "python3", bench,
                 "--steps", str(a.pmc_steps), "--warmup", "3", "--preheat-ms", "0", "--no-serial", "--no-host-path"] + common,
                os.path.join(out, f"pmc_{name}.log"))
            f = glob.glob(os.path.join(pdir, "**", "*counter_collection.csv"), recursive=True)[0]
            # a launch renders one or several frames (rt_render_sequence): its grid is frames x workgroups per frame, and the
            # run always holds single-frame launches of the same kernel (the first launches of a geometry), so
            # frames = Grid_Size / smallest Grid_Size.  Counters are reported per FRAME: sum over launches / sum of frames.
            per = collections.defaultdict(lambda: collections.defaultdict(list))
            for row in csv.DictReader(open(f)):
                if "render_kernel" in row["Kernel_Name"]:
                    per[row["Kernel_Name"]][row["Counter_Name"]].append((float(row["Counter_Value"]), int(row.get("Grid_Size") or 1)))
            if per:
                kname = max(per, key=lambda k: sum(g for _, g in next(iter(per[k].values()))))     # the dominant instantiation (by work)
                for cn, vals in per[kname].items():
                    g1 = min(g for _, g in vals)
                    frames = sum(round(g / g1) for _, g in vals)
                    summary[cn] = {"launches": len(vals), "frames": frames, "mean_per_launch": round(sum(v for v, _ in vals) / frames, 2),
                                   "note": "mean_per_launch = per FRAME (a launch may render several)"}
                summary["_kernel"] = kname
            shutil.rmtree(pdir, ignore_errors=True)
            print("pmc", name, "done", flush=True)
        json.dump({"command": f"rocprofv3 --pmc <set> -- python3 bench.py --steps {a.pmc_steps} --warmup 3 --preheat-ms 0 --no-serial --no-host-path "
                              + " ".join(common) + " (one run per set)", "counters": summary},
                  open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)

    # 3. constants bench.py reports, stamped with the measured build
    if a.stamp and summary:
        sys.path.insert(0, REPO)
        from python_ray_tracer_amd import _lib
        sha = hashlib.sha256(open(_lib.SO_PATH, "rb").read()).hexdigest()
        g = lambda k: summary.get(k, {}).get("mean_per_launch")     # noqa: E731
        suffix = f"_{a.workload}" if a.workload else ""     # bench.py looks the constants up by round and workload
        if g("WRITE_SIZE") is not None and g("FETCH_SIZE") is not None:
            wr, fe = g("WRITE_SIZE") * 1024.0, g("FETCH_SIZE") * 1024.0
            json.dump({"so_sha256": sha, "source": f"gpurun_out/prof_{a.tag} (tools/profile_round.py): rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE, separate passes",
                       "WRITE_SIZE_KB_per_launch": g("WRITE_SIZE"), "FETCH_SIZE_KB_per_launch_raw": g("FETCH_SIZE"),
                       "correction": "FETCH_SIZE doubled (gfx950 reports half of streamed read bytes); WRITE_SIZE as is",
                       "hbm_bytes_per_launch": int(wr + 2 * fe)}, open(os.path.join(out, f"traffic_{a.round}{suffix}.json"), "w"), indent=1)
        if g("SQ_INSTS_VALU") is not None and g("SQ_INSTS_VALU_FMA_F64") is not None:
            add, mul, fma = g("SQ_INSTS_VALU_ADD_F64"), g("SQ_INSTS_VALU_MUL_F64"), g("SQ_INSTS_VALU_FMA_F64")
            json.dump({"so_sha256": sha, "source": f"gpurun_out/prof_{a.tag} (tools/profile_round.py): rocprofv3 --pmc SQ_INSTS_VALU_*",
                       "valu_wave_instructions_per_launch": int(g("SQ_INSTS_VALU")),
                       "salu_wave_instructions_per_launch": int(g("SQ_INSTS_SALU") or 0),
                       "lds_wave_instructions_per_launch": int(g("SQ_INSTS_LDS") or 0),
                       "kernel": summary.get("_kernel"),
                       # live lanes per VALU wave-instruction (SQ_THREAD_CYCLES_VALU counts lane-quad-cycles, SQ_ACTIVE_INST_VALU quad-cycles)
                       "lane_utilisation": round(g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU")), 4)
                                           if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU") else None,
                       "fp64_flop_wave_level_x64": int(64 * (g("SQ_INSTS_VALU_FLOPS_FP64") or (add + mul + 2 * fma))),
                       "op_mix_wave_instructions": {k[len("SQ_INSTS_VALU_"):].lower(): int(v["mean_per_launch"]) for k, v in summary.items()
                                                    if k.startswith("SQ_INSTS_VALU_")},
                       "note": "non-fusable float64 (the reference never fuses multiply-add); bench.py prices this op mix with the measured "
                               "cycles per wave-instruction of profiles/r03_valu_prices.json (tools/valu_prices.hip)"},
                      open(os.path.join(out, f"valu_{a.round}{suffix}.json"), "w"), indent=1)
    print("done", out, flush=True)


if __name__ == "__main__":
    main()
