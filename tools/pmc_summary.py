#!/usr/bin/env python
"""Summaries of the rocprofv3 passes tools/profile.sh collects, and the committed profiles/traffic.json.

  python tools/pmc_summary.py <out_dir> <tag> [--renders N]
      reads <out_dir>/trace/**/kernel_stats.csv and <out_dir>/pmc_*/**/counter_collection.csv, writes
      <out_dir>/pmc_summary.txt and the fragment <out_dir>/traffic.json (per kernel: counters per launch and, with
      --renders, per render; plus bench.py's source_hash() of the tree that ran)
  python tools/pmc_summary.py --merge <fragment.json> [--section staircase]
      copies a fragment into profiles/traffic.json: top level (config 2) or under a named section (config 5)
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def summarise(out, tag, renders):
    from bench import source_hash
    for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
        print("== kernel stats", f)
        print(open(f).read()[:3000])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    with open(out + "/pmc_summary.txt", "w") as fh:
        for k, v in agg.items():
            fh.write(k + "\n")
            for c, val in sorted(v.items()):
                n = cnt[(k, c)]
                fh.write(f"   {c:28s} total {val:.6g}  dispatches {n}  per-dispatch {val / n:.6g}\n")
    print(open(out + "/pmc_summary.txt").read())
    traffic = {"source_hash": source_hash(), "profile": tag}
    # instantiations of one kernel are summed under its name (k_wf_trace<..., false> = closest hits and <..., true> = occlusion since
    # round 4); the occlusion instantiation is also listed on its own as k_wf_trace_any
    by_name = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt2 = collections.Counter()
    for k, v in agg.items():
        m = re.search(r"k_(fused|wf_[a-z_]+|develop_[a-z]+)", k)
        if not m:
            continue
        names = [m.group(0)]
        if m.group(0) == "k_wf_trace" and re.search(r"k_wf_trace<[^>]*true>", k):
            names.append("k_wf_trace_any")
        for nm in names:
            for c, val in v.items():
                by_name[nm][c] += val
                cnt2[(nm, c)] += cnt[(k, c)]
    cnt = cnt2
    for k, v in by_name.items():
        m = re.match(r"k_[a-z_]+", k)
        if "FETCH_SIZE" not in v:
            continue
        nf = cnt[(k, "FETCH_SIZE")]
        nw = cnt.get((k, "WRITE_SIZE"), 1)
        fetch_kb = v["FETCH_SIZE"] / nf
        write_kb = v.get("WRITE_SIZE", 0.0) / max(1, nw)
        # rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> x2 (MI355X_MICROARCH.md, HBM section)
        e = {"fetch_size_kib_per_launch": fetch_kb, "write_size_kib_per_launch": write_kb,
             "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0, "dispatches_profiled": nf,
             "note": "FETCH_SIZE doubled (gfx950 wide-read correction); WRITE_SIZE uncalibrated"}
        if renders:
            e["hbm_bytes_per_render"] = (2.0 * v["FETCH_SIZE"] + v.get("WRITE_SIZE", 0.0)) * 1024.0 / renders
        if v.get("SQ_ACTIVE_INST_VALU"):
            nv = cnt[(k, "SQ_INSTS_VALU")]
            e.update({"valu_insts_per_launch": v["SQ_INSTS_VALU"] / nv,
                      "valu_lanes_per_inst": v["SQ_THREAD_CYCLES_VALU"] / v["SQ_ACTIVE_INST_VALU"],
                      "lds_insts_per_launch": v.get("SQ_INSTS_LDS", 0.0) / nv, "salu_insts_per_launch": v.get("SQ_INSTS_SALU", 0.0) / nv,
                      "vmem_insts_per_launch": v.get("SQ_INSTS_VMEM", 0.0) / nv,
                      "wait_any_frac": v.get("SQ_WAIT_ANY", 0.0) / max(1.0, v.get("SQ_WAVE_CYCLES", 1.0)),
                      "wait_inst_any_frac": v.get("SQ_WAIT_INST_ANY", 0.0) / max(1.0, v.get("SQ_WAVE_CYCLES", 1.0)),
                      "profile": tag})
            if renders:
                e["valu_insts_per_render"] = v["SQ_INSTS_VALU"] / renders
                e["vmem_insts_per_render"] = v.get("SQ_INSTS_VMEM", 0.0) / renders
                e["l2_requests_per_render"] = (v.get("TCC_HIT_sum", 0.0) + v.get("TCC_MISS_sum", 0.0)) / renders
                e["launches_per_render"] = nv / renders
        traffic[m.group(0)] = e
    json.dump(traffic, open(out + "/traffic.json", "w"), indent=1)
    print(json.dumps(traffic, indent=1))


def merge(fragment, section):
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        cur = json.load(open(path))
    except Exception:
        cur = {}
    frag = json.load(open(fragment))
    if section:
        cur[section] = frag
    else:
        keep = {k: v for k, v in cur.items() if isinstance(v, dict) and "source_hash" in v}      # named sections stay
        cur = dict(frag, **keep)
    json.dump(cur, open(path, "w"), indent=1)
    print("profiles/traffic.json:", "section " + section if section else "top level", "<-", fragment, "source_hash", frag.get("source_hash"))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out", nargs="?")
    ap.add_argument("tag", nargs="?")
    ap.add_argument("--renders", type=int, default=0)
    ap.add_argument("--merge", default=None)
    ap.add_argument("--section", default=None)
    a = ap.parse_args()
    if a.merge:
        merge(a.merge, a.section)
    else:
        summarise(a.out, a.tag, a.renders)
