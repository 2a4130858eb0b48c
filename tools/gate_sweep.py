#!/usr/bin/env python3
"""Learning sweeps of training.py, several runs side by side on the one GPU (run on the GPU box).

An arm is  precision:first_order:late_order:switch_at:iterations:seeds  (late_order '-' = no switch; seeds 'a-b' or
'a,b,c'), e.g.  fp32:reference:reference_carried:5000:60000:0-47 .  Every finished run appends one JSON row to --out
(held-out count accuracy along the run from summary/scalars.jsonl).  Runs already present in --done files (same arm
key + seed) are skipped, and no new run is started after --deadline-min minutes, so that a sweep can be spread over
several gpurun calls.  Extra training.py flags go after '--'.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tf-attend-infer-repeat_amd")


def parse_seeds(s):
    out = []
    for part in s.split(","):
        if "-" in part:
            a, b = part.split("-")
            out += list(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def key_of(row):
    return (row["precision"], row["backward"], row.get("late_backward") or "-", int(row.get("switch_at") or 0),
            int(row["iterations"]), row.get("extra", ""), int(row["seed"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--arm", action="append", required=True)
    ap.add_argument("--parallel", type=int, default=4)
    ap.add_argument("--deadline-min", type=float, default=1e9)
    ap.add_argument("--done", action="append", default=[])
    ap.add_argument("--tag", default="")
    ap.add_argument("--job-timeout-s", type=float, default=900.0)
    ap.add_argument("extra", nargs="*")
    args = ap.parse_args()
    extra = " ".join(args.extra)

    done = set()
    for f in args.done + [args.out]:
        if os.path.exists(f):
            for line in open(f):
                try:
                    done.add(key_of(json.loads(line)))
                except Exception:
                    pass
    jobs = []
    for arm in args.arm:
        prec, first, late, at, iters, seeds = arm.split(":")
        for s in parse_seeds(seeds):
            row = {"precision": prec, "backward": first, "late_backward": None if late == "-" else late,
                   "switch_at": int(at) if late != "-" else 0, "iterations": int(iters), "extra": extra, "seed": s}
            if key_of(row) not in done:
                jobs.append(row)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    lock = threading.Lock()
    t_start = time.time()
    it = iter(jobs)

    def worker(wid):
        while True:
            with lock:
                if (time.time() - t_start) / 60.0 > args.deadline_min:
                    return
                job = next(it, None)
            if job is None:
                return
            folder = "/tmp/gate_%s_%d_%d" % (args.tag, wid, job["seed"])
            cmd = [sys.executable, "training.py", "-r", folder, "-o", "1", "--iterations", str(job["iterations"]),
                   "--print-every", "0", "--precision", job["precision"], "--seed", str(job["seed"]),
                   "--backward", job["backward"]]
            if job["late_backward"]:
                cmd += ["--late-backward", job["late_backward"], "--late-backward-from", str(job["switch_at"])]
            cmd += args.extra
            t0 = time.time()
            try:
                p = subprocess.run(cmd, cwd=PKG, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=args.job_timeout_s)
            except subprocess.TimeoutExpired as e:                   # a hung run must not eat the whole call
                p = subprocess.CompletedProcess(cmd, -9, stdout=(e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes)
                                                else (e.stdout or ""))
            row = dict(job)
            row["process_s"] = round(time.time() - t0, 2)
            try:
                rows = [json.loads(l) for l in open(folder + "/summary/scalars.jsonl")]
                accs = [(r["step"], r["accuracy"]) for r in rows]
                row.update({
                    "final_accuracy": accs[-1][1], "best_accuracy": max(a for _, a in accs),
                    "first_step_at_98pct": next((s for s, a in accs if a >= 0.98), None),
                    "acc_at": {str(s): a for s, a in accs if s % 10000 == 0},
                    "wall_s": rows[-1]["wall_s"], "rc": p.returncode})
            except Exception as e:                                   # keep the failure visible in the record
                row.update({"rc": p.returncode, "error": repr(e), "tail": p.stdout[-400:]})
            shutil.rmtree(folder, ignore_errors=True)
            with lock:
                with open(args.out, "a") as f:
                    f.write(json.dumps(row) + "\n")

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(args.parallel)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    print("gate_sweep: %d jobs queued, %.1f min" % (len(jobs), (time.time() - t_start) / 60.0))


if __name__ == "__main__":
    main()
