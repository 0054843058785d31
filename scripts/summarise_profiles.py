#!/usr/bin/env python3
"""Summaries of a scripts/gpu_validate.sh run (gpurun_out/<tag>/) into profiles/: kernel-trace
stats, HBM traffic per launch (FETCH_SIZE x2 on gfx950: 128-byte requests are tallied at 64
bytes; both counters in units of 1024 bytes) and SQ counters per wave, each stamped with the
hash of the kernel sources it was recorded with.   usage: summarise_profiles.py <tag> [round]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r3"
OUT = os.path.join(ROOT, "gpurun_out", tag)
PROF = os.path.join(ROOT, "profiles")
sha = bench.kernel_source_hash()


def find(sub, name):
    hits = glob.glob(os.path.join(OUT, sub, "**", name), recursive=True)
    return hits[0] if hits else None


def per_dispatch(path, kernel, counter=None):
    """{counter: median over dispatches of the sum over the dispatch's rows}, and waves per dispatch."""
    acc = collections.defaultdict(float)
    grid = None
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"] and (counter is None or r["Counter_Name"] == counter):
            acc[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
            grid = int(r["Grid_Size"]) // 64
    per = collections.defaultdict(list)
    for (d, c), v in acc.items():
        per[c].append(v)
    return {c: sorted(v)[len(v) // 2] for c, v in per.items()}, grid, {c: len(v) for c, v in per.items()}


def kernel_stats(sub, prefix, dest):
    p = find(sub, prefix + "_kernel_stats.csv")
    if p:
        shutil.copy(p, os.path.join(PROF, dest))
        return {r["Name"]: r for r in csv.DictReader(open(p))}
    return {}


def traffic(fetch_sub, write_sub, kernel):
    pf, pw = find(fetch_sub, "p_counter_collection.csv"), find(write_sub, "p_counter_collection.csv")
    if not pf or not pw:
        return None
    f, _, nf = per_dispatch(pf, kernel, "FETCH_SIZE")
    w, _, nw = per_dispatch(pw, kernel, "WRITE_SIZE")
    return {"fetch_size_kb_raw": f["FETCH_SIZE"], "write_size_kb_raw": w["WRITE_SIZE"],
            "fetch_bytes": 2.0 * f["FETCH_SIZE"] * 1024.0, "write_bytes": w["WRITE_SIZE"] * 1024.0,
            "bytes_per_launch": 2.0 * f["FETCH_SIZE"] * 1024.0 + w["WRITE_SIZE"] * 1024.0,
            "dispatches": [nf["FETCH_SIZE"], nw["WRITE_SIZE"]]}


def sq(subs, kernel):
    per_wave = {}
    for sub in subs:
        p = find(sub, "p_counter_collection.csv")
        if not p:
            continue
        med, grid, _ = per_dispatch(p, kernel)
        for c, v in med.items():
            per_wave[c] = v / grid
    return per_wave


def avg_ns(stats, kernel):
    for name, r in stats.items():
        if kernel in name:
            return float(r["AverageNs"]), int(r["Calls"])
    return None, None


how = "rocprofv3 --pmc, one counter group per run (scripts/gpu_validate.sh); median over dispatches"
# headline kernel
st = kernel_stats("trace", "t", f"{rnd}_bench_kernel_stats.csv")
t = traffic("pmc_fetch", "pmc_write", "k_solve_pair5")
if t:
    t.update(kernel="k_solve_pair5", batch=65536, csrc_sha=sha, how=how + "; FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B)")
    ns, calls = avg_ns(st, "k_solve_pair5<1, false, 0>")
    t["kernel_trace_avg_ns"] = ns
    json.dump(t, open(os.path.join(PROF, "hbm_traffic.json"), "w"), indent=1)
    print("headline traffic", t["bytes_per_launch"], "avg ns", ns)
pw = sq(["sq5_a", "sq5_b"], "k_solve_pair5")
if pw:
    rec = {"kernel": "k_solve_pair5<dual-steer, single QP>", "batch": 65536, "csrc_sha": sha, "per_wave": pw,
           "valu_busy_frac": pw.get("SQ_ACTIVE_INST_VALU", 0.0) / pw["SQ_WAVE_CYCLES"] if pw.get("SQ_WAVE_CYCLES") else None,
           "how": how + "; counters summed over the launch and divided by its waves; SQ_*_CYCLES and SQ_ACTIVE_* tick once per 4 clocks"}
    json.dump(rec, open(os.path.join(PROF, "pair5_sq_counters.json"), "w"), indent=1)
    print("headline valu busy", rec["valu_busy_frac"], "VALU/wave", pw.get("SQ_INSTS_VALU"))
pw = sq(["sq5_f64"], "k_solve_pair5")
if pw and pw.get("SQ_INSTS_VALU_FMA_F64") is not None:
    waves = 65536 // 32
    per = {k: pw.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64")}
    flops_wave = 64.0 * (2.0 * per["SQ_INSTS_VALU_FMA_F64"] + per["SQ_INSTS_VALU_MUL_F64"] + per["SQ_INSTS_VALU_ADD_F64"] + per["SQ_INSTS_VALU_TRANS_F64"])
    rec = {"kernel": "k_solve_pair5<dual-steer, single QP>", "batch": 65536, "csrc_sha": sha, "per_wave": per, "waves": waves,
           "flops_per_launch": flops_wave * waves,
           "how": how + "; wave instructions per wave; flops = 64 lanes x (2 FMA + MUL + ADD + TRANS), masked-off lanes counted"}
    json.dump(rec, open(os.path.join(PROF, "pair5_f64_counters.json"), "w"), indent=1)
    print("headline f64 instr/wave", per, "flops/solve", flops_wave * waves / 65536)
# chained pair kernel (MPC.iterativeLMPC: five QPs per call)
st = kernel_stats("ch_trace", "p", f"{rnd}_chain_kernel_stats.csv")
t = traffic("ch_fetch", "ch_write", "k_solve_pair5")
pw = sq(["ch_sq_a", "ch_sq_b"], "k_solve_pair5")
if t or pw:
    ns, calls = avg_ns(st, "k_solve_pair5<1, true, 0>")
    rec = {"kernel": "k_solve_pair5<dual-steer, chain>", "variant": "sim_ackermann", "batch": 65536, "csrc_sha": sha, "avg_ns": ns,
           "launches": calls, "algorithmic_bytes": 65536 * 649, "traffic": t, "per_wave": pw,
           "valu_busy_frac": (pw.get("SQ_ACTIVE_INST_VALU", 0.0) / pw["SQ_WAVE_CYCLES"]) if pw and pw.get("SQ_WAVE_CYCLES") else None, "how": how}
    json.dump(rec, open(os.path.join(PROF, f"{rnd}_chain_counters.json"), "w"), indent=1)
    print("chain kernel", ns, "ns", t and t["bytes_per_launch"])
# wavefront-per-instance kernel, T = 30, the batch the chip holds at once
st = kernel_stats("wv_trace", "p", f"{rnd}_wave_T30_kernel_stats.csv")
kernel_stats("wd_trace", "p", f"{rnd}_wave_T30_dsr_diff_kernel_stats.csv")
pw = sq(["wv_sq_a", "wv_sq_b"], "k_solve_wave")
if pw:
    ns, calls = avg_ns(st, "k_solve_wave")
    rec = {"kernel": "k_solve_wave<dual-steer>", "variant": "dsr_ackermann", "T": 30, "batch": 1024, "csrc_sha": sha, "avg_ns": ns,
           "launches": calls, "per_wave": pw,
           "valu_busy_frac": (pw.get("SQ_ACTIVE_INST_VALU", 0.0) / pw["SQ_WAVE_CYCLES"]) if pw.get("SQ_WAVE_CYCLES") else None, "how": how}
    json.dump(rec, open(os.path.join(PROF, f"{rnd}_wave_T30_counters.json"), "w"), indent=1)
    print("wave kernel T30", ns, "ns")
# T = 10: four lanes per instance (mpc.py's controller)
st = kernel_stats("q10_trace", "p", f"{rnd}_quad_T10_65536_kernel_stats.csv")
st32 = kernel_stats("q10b_trace", "p", f"{rnd}_quad_T10_32768_kernel_stats.csv")
t = traffic("q10_fetch", "q10_write", "k_solve_quad")
pw = sq(["q10_sq_a", "q10_sq_b"], "k_solve_quad")
if t or pw:
    ns, calls = avg_ns(st, "k_solve_quad")
    ns32, calls32 = avg_ns(st32, "k_solve_quad")
    rec = {"kernel": "k_solve_quad<dual-steer, T=10>", "variant": "mpc_py", "T": 10, "batch": 32768, "csrc_sha": sha,
           "algorithmic_bytes": 32768 * 1209, "traffic": t, "per_wave": pw,
           "valu_busy_frac": (pw.get("SQ_ACTIVE_INST_VALU", 0.0) / pw["SQ_WAVE_CYCLES"]) if pw and pw.get("SQ_WAVE_CYCLES") else None,
           "avg_ns_32768": ns32, "launches_32768": calls32, "avg_ns_65536": ns, "launches_65536": calls,
           "note": "kernel-trace averages over 200 back-to-back warm launches (settled clock)", "how": how}
    json.dump(rec, open(os.path.join(PROF, f"{rnd}_quad10_counters.json"), "w"), indent=1)
    print("quad10 traffic", t and t["bytes_per_launch"], "algorithmic", 32768 * 1209, "avg ns 32768/65536", ns32, ns)
# stage-wise kernel, T = 30
st = kernel_stats("st_trace", "p", f"{rnd}_stage_T30_kernel_stats.csv")
kernel_stats("sd_trace", "p", f"{rnd}_stage_T30_dsr_diff_kernel_stats.csv")
t = traffic("st_fetch", "st_write", "k_solve_stage")
if t:
    ns, calls = avg_ns(st, "k_solve_stage")
    rec = {"kernel": "k_solve_stage<dual steer>", "variant": "dsr_ackermann", "T": 30, "batch": 65536, "csrc_sha": sha,
           "avg_ns": ns, "launches": calls, "hbm_read_bytes": t["fetch_bytes"], "hbm_write_bytes": t["write_bytes"],
           "hbm_GBps": (t["bytes_per_launch"]) / ns if ns else None, "algorithmic_bytes": 65536 * 3449, "how": how}
    json.dump(rec, open(os.path.join(PROF, f"{rnd}_stage_T30_traffic.json"), "w"), indent=1)
    print("stage T30", rec["avg_ns"], "ns", t["bytes_per_launch"] / 1e9, "GB per launch")
t = traffic("sd_fetch", "sd_write", "k_solve_stage")
if t:
    sd = {r["Name"]: r for r in csv.DictReader(open(os.path.join(PROF, f"{rnd}_stage_T30_dsr_diff_kernel_stats.csv")))} if os.path.exists(os.path.join(PROF, f"{rnd}_stage_T30_dsr_diff_kernel_stats.csv")) else {}
    ns, calls = avg_ns(sd, "k_solve_stage")
    rec = {"kernel": "k_solve_stage<dual steer>", "variant": "dsr_diff", "T": 30, "batch": 65536, "csrc_sha": sha,
           "avg_ns": ns, "launches": calls, "hbm_read_bytes": t["fetch_bytes"], "hbm_write_bytes": t["write_bytes"],
           "hbm_GBps": (t["bytes_per_launch"]) / ns if ns else None, "algorithmic_bytes": 65536 * 3449, "how": how}
    json.dump(rec, open(os.path.join(PROF, f"{rnd}_stage_T30_dsr_diff_traffic.json"), "w"), indent=1)
    print("stage T30 dsr_diff", ns, "ns", t["bytes_per_launch"] / 1e9, "GB per launch (median over cold and warm launches)")
pw = sq(["st_sq"], "k_solve_stage")
if pw:
    json.dump({"kernel": "k_solve_stage<dual steer>", "T": 30, "batch": 65536, "csrc_sha": sha, "per_wave": pw, "how": how},
              open(os.path.join(PROF, f"{rnd}_stage_T30_sq_counters.json"), "w"), indent=1)
# the kernels around the solve: reference window, advance, course builder
st = kernel_stats("lp_trace", "p", f"{rnd}_loop_kernel_stats.csv")
recs = {}
for kern, alg in (("k_nearest_ref", 65536 * (32 + 8 + 176 + 4 + 24 * 6)), ("k_advance", 65536 * (8 * 4 * 2 + 8 * 4 + 4 + 4 + 2 + 16 + 4 + 16)),
                  ("k_build_courses", None)):
    t = traffic("lp_fetch", "lp_write", kern)
    if t:
        ns, calls = avg_ns(st, kern)
        recs[kern] = {"avg_ns": ns, "launches": calls, "hbm_bytes_per_launch": t["bytes_per_launch"], "fetch_bytes": t["fetch_bytes"],
                      "write_bytes": t["write_bytes"], "algorithmic_bytes": alg,
                      "hbm_GBps": t["bytes_per_launch"] / ns if ns else None}
if recs:
    json.dump({"batch": 65536, "courses": 1024, "csrc_sha": sha, "kernels": recs, "how": how}, open(os.path.join(PROF, f"{rnd}_loop_kernels_traffic.json"), "w"), indent=1)
    print("loop kernels", {k: (round(v["avg_ns"] / 1e3, 1), round(v["hbm_bytes_per_launch"] / 1e6, 1)) for k, v in recs.items()})
for name in ("bench.json", "mixed4m.json"):
    p = os.path.join(OUT, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(PROF, f"{rnd}_{name.replace('.json', '_line.json')}"))
