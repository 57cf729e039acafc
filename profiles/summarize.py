#!/usr/bin/env python3
"""Condenses rocprofv3 output (kernel-trace --stats CSV + separate --pmc FETCH_SIZE / WRITE_SIZE /
SQ passes, scripts/profile_bench.sh) into profiles/<tag>_summary.json and .md.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are
in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced streaming reads, so the
read side is doubled (calibration: kw_prepass reads y once and writes N ring-score planes: 80 MB in,
320 MB out at 10 M samples, N = 4); WRITE_SIZE is exact.

_meta.kernel_sources = sha256 over hmmspikesorter.jl_amd/csrc/*.{hip,h,cpp}: bench.py only quotes
the counters of a summary whose hash equals that of the sources it runs.

usage: summarize.py <tag> <raw dir with trace/ pmc_fetch/ pmc_write/ [pmc_sq/]> [samples block halo [neurons states channels]]
"""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def kernel_sources_hash():
    h = hashlib.sha256()
    src = os.path.join(os.path.dirname(HERE), "hmmspikesorter.jl_amd", "csrc")
    for p in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h")) +
                    glob.glob(os.path.join(src, "*.cpp"))):
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def short(name):
    m = re.search(r"(kw_[a-z_0-9]+|k_[a-z_0-9]+|gen_[a-z_]+|upd_[a-z_]+)", name)
    return m.group(1) if m else name.split("(")[0][:48]


def pmc(path, counter):
    agg = collections.defaultdict(list)
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    tag, raw = sys.argv[1:3]
    rows = []
    for r in csv.DictReader(open(os.path.join(raw, "trace", "b_kernel_stats.csv"))):
        rows.append((short(r["Name"]), int(r["Calls"]), float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
    res = {}
    tr = os.path.join(raw, "trace", "b_kernel_trace.csv")
    if os.path.exists(tr):
        for r in csv.DictReader(open(tr)):
            res[short(r["Kernel_Name"])] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"),
                                            r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
    fe = pmc(os.path.join(raw, "pmc_fetch", "b_counter_collection.csv"), "FETCH_SIZE")
    wr = pmc(os.path.join(raw, "pmc_write", "b_counter_collection.csv"), "WRITE_SIZE")
    sqp = os.path.join(raw, "pmc_sq", "b_counter_collection.csv")
    sq = {c: pmc(sqp, c) for c in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                                   "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU")}
    out = {}
    for name, calls, avg_ms, pct in rows:
        if not name.startswith(("kw_", "k_", "gen_", "upd_")):
            continue
        rd = fe.get(name, 0.0) * 1024 * 2      # KiB -> B, gfx950 wide-read correction x2
        ww = wr.get(name, 0.0) * 1024
        d = {"calls": calls, "avg_ms": round(avg_ms, 4), "pct": pct,
             "hbm_read_bytes": rd, "hbm_write_bytes": ww,
             "hbm_GBps": round((rd + ww) / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else None}
        if name in res:
            d["vgpr"], d["agpr"], d["sgpr"], d["lds"], d["grid"], d["wg"] = res[name]
        wc = sq["SQ_WAVE_CYCLES"].get(name)
        if wc:
            d["wait_frac"] = round(sq["SQ_WAIT_ANY"].get(name, 0) / wc, 3)
            d["issue_stall_frac"] = round(sq["SQ_WAIT_INST_ANY"].get(name, 0) / wc, 3)
            d["active_frac"] = round(sq["SQ_ACTIVE_INST_ANY"].get(name, 0) / wc, 3)
            d["valu_insts_per_wave"] = round(sq["SQ_INSTS_VALU"].get(name, 0) / max(1.0, sq["SQ_WAVES"].get(name, 1)), 1)
        out[name] = d
    meta = {"kernel_sources": kernel_sources_hash(),
            "command": "scripts/profile_bench.sh %s (python3 bench.py --quick --steps 5 --warmup 2 ...)" % tag}
    if len(sys.argv) >= 6:
        meta.update(samples=int(sys.argv[3]), block=int(sys.argv[4]), halo=int(sys.argv[5]))
    if len(sys.argv) >= 9:   # model shape and channels per plan (default: the headline's 4 x 60, one channel)
        meta.update(neurons=int(sys.argv[6]), states=int(sys.argv[7]), channels=int(sys.argv[8]))
    out["_meta"] = meta
    json.dump(out, open(os.path.join(HERE, tag + "_summary.json"), "w"), indent=1)
    with open(os.path.join(HERE, tag + "_summary.md"), "w") as f:
        f.write("| kernel | launches | avg ms | %% of GPU time | HBM read MB (FETCH_SIZE x2) | HBM write MB | HBM GB/s | VGPR | LDS B | wait | active | VALU insts/wave |\n|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for k, v in sorted(((k, v) for k, v in out.items() if k != "_meta"),
                           key=lambda kv: -kv[1]["avg_ms"] * kv[1]["calls"]):
            f.write("| %s | %d | %.4f | %.2f | %.1f | %.1f | %s | %s | %s | %s | %s | %s |\n" % (
                k, v["calls"], v["avg_ms"], v["pct"], v["hbm_read_bytes"] / 1e6,
                v["hbm_write_bytes"] / 1e6, v["hbm_GBps"], v.get("vgpr"), v.get("lds"), v.get("wait_frac"),
                v.get("active_frac"), v.get("valu_insts_per_wave")))
    import shutil
    shutil.copy(os.path.join(raw, "trace", "b_kernel_stats.csv"), os.path.join(HERE, tag + "_kernel_stats.csv"))
    print(open(os.path.join(HERE, tag + "_summary.md")).read())


if __name__ == "__main__":
    main()
