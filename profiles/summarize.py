#!/usr/bin/env python3
"""Condenses rocprofv3 output (kernel-trace --stats CSV + separate --pmc FETCH_SIZE / WRITE_SIZE
passes) into profiles/<tag>_summary.json and .md.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are
in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced streaming reads, so the
read side is doubled (checked here against the kernels' own byte counts: the forward chain reads
9 doubles per lane-step over 2*T lane-steps = 1.44 GB, FETCH_SIZE*2 = 1.43 GB); WRITE_SIZE is exact.

usage: summarize.py <tag> <kernel_stats.csv> <fetch_counter_collection.csv> <write_counter_collection.csv>
                    [samples block halo]
"""
import collections
import csv
import json
import os
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z_0-9]+|gen_[a-z_]+|upd_[a-z_]+)", name)
    return m.group(1) if m else name.split("(")[0][:48]


def pmc(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    tag, stats, fetch, write = sys.argv[1:5]
    rows = []
    for r in csv.DictReader(open(stats)):
        rows.append((short(r["Name"]), int(r["Calls"]), float(r["AverageNs"]) / 1e6, float(r["Percentage"])))
    fe, wr = pmc(fetch, "FETCH_SIZE"), pmc(write, "WRITE_SIZE")
    out = {}
    for name, calls, avg_ms, pct in rows:
        if not name.startswith(("k_", "gen_", "upd_")):
            continue
        rd = fe.get(name, 0.0) * 1024 * 2      # KiB -> B, gfx950 wide-read correction x2
        ww = wr.get(name, 0.0) * 1024
        out[name] = {"calls": calls, "avg_ms": round(avg_ms, 4), "pct": pct,
                     "hbm_read_bytes": rd, "hbm_write_bytes": ww,
                     "hbm_GBps": round((rd + ww) / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else None}
    if len(sys.argv) >= 8:
        out["_meta"] = {"samples": int(sys.argv[5]), "block": int(sys.argv[6]), "halo": int(sys.argv[7]),
                        "command": "python3 bench.py (N=4, K=60 model, one channel)"}
    here = os.path.dirname(os.path.abspath(__file__))
    json.dump(out, open(os.path.join(here, tag + "_summary.json"), "w"), indent=1)
    with open(os.path.join(here, tag + "_summary.md"), "w") as f:
        f.write("| kernel | launches | avg ms | %% of GPU time | HBM read MB (FETCH_SIZE x2) | HBM write MB | HBM GB/s |\n|---|---|---|---|---|---|---|\n")
        for k, v in sorted(((k, v) for k, v in out.items() if k != "_meta"),
                           key=lambda kv: -kv[1]["avg_ms"] * kv[1]["calls"]):
            f.write("| %s | %d | %.4f | %.2f | %.1f | %.1f | %s |\n" % (
                k, v["calls"], v["avg_ms"], v["pct"], v["hbm_read_bytes"] / 1e6,
                v["hbm_write_bytes"] / 1e6, v["hbm_GBps"]))
    print(open(os.path.join(here, tag + "_summary.md")).read())


if __name__ == "__main__":
    main()
