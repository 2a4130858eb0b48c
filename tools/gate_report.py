#!/usr/bin/env python3
"""Counts of the learning sweeps, straight from their jsonl records (profiles/r05_sweep_*.jsonl, r06_gate_*.jsonl; rows as
tools/gate_sweep.py / tools/seed_sweep2.sh write them).  One line per arm = (precision, first order, late order, switch
iteration, iterations): runs, how many reach 0.98 held-out count accuracy at some evaluation, how many END below 0.9
("stuck": one count class never separates), the lowest final of the others, the median first crossing of 0.98.
  python tools/gate_report.py profiles/r05_sweep_*.jsonl profiles/r06_gate_*.jsonl [--seeds 0-47] [--markdown]"""
import json
import sys
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    md = "--markdown" in sys.argv
    seeds = None
    if "--seeds" in sys.argv:
        a, b = sys.argv[sys.argv.index("--seeds") + 1].split("-")
        seeds = set(range(int(a), int(b) + 1))
        args = [x for x in args if x != "%s-%s" % (a, b)]
    arms = defaultdict(dict)
    for f in args:
        for line in open(f):
            try:
                r = json.loads(line)
            except ValueError:
                continue
            if "final_accuracy" not in r:
                continue
            if seeds is not None and int(r["seed"]) not in seeds:
                continue
            key = (r["precision"], r["backward"], r.get("late_backward") or "-", int(r.get("switch_at") or 0), int(r["iterations"]),
                   r.get("extra", "") or "")
            arms[key][int(r["seed"])] = r                      # a seed that was run twice counts once (the later record)
    rows = []
    for key in sorted(arms):
        rs = list(arms[key].values())
        reach = [r for r in rs if r["best_accuracy"] >= 0.98]
        stuck = sorted(r["final_accuracy"] for r in rs if r["final_accuracy"] < 0.9)
        others = [r["final_accuracy"] for r in rs if r["final_accuracy"] >= 0.9]
        firsts = sorted(r["first_step_at_98pct"] for r in rs if r.get("first_step_at_98pct") is not None)
        rows.append((key, len(rs), len(reach), stuck, min(others) if others else None, firsts[len(firsts) // 2] if firsts else None,
                     sorted(arms[key])))
    if md:
        print("| precision | order | runs (seeds) | reach 0.98 | final < 0.9 | lowest other final | median first crossing |")
        print("|---|---|---|---|---|---|---|")
    for key, n, nreach, stuck, low, med, sd in rows:
        prec, first, late, at, iters, extra = key
        order = first if late == "-" else "%s -> %s @ %d" % (first, late, at)
        srange = "%d-%d" % (sd[0], sd[-1]) if sd == list(range(sd[0], sd[-1] + 1)) else ",".join(map(str, sd))
        if md:
            print("| %s | `%s`%s, %d k | %d (%s) | %d | %d%s | %s | %s |" % (
                prec, order, (" " + extra) if extra else "", iters // 1000, n, srange, nreach, len(stuck),
                (" (" + ", ".join("%.2f" % s for s in stuck) + ")") if stuck else "", "%.3f" % low if low is not None else "-", med))
        else:
            print("%-5s %-46s %6dk  runs %3d (%s)  reach0.98 %3d  final<0.9 %2d %s  lowest other %s  median first crossing %s" % (
                prec, order + ((" " + extra) if extra else ""), iters // 1000, n, srange, nreach, len(stuck),
                ["%.2f" % s for s in stuck], "%.3f" % low if low is not None else "-", med))


if __name__ == "__main__":
    main()
