#!/usr/bin/env python3
"""Same-box A/B of library variants (tools/ab_build.sh): runs bench.py once per variant per round, interleaved, and prints the
per-kernel averages side by side.  Boxes of the pool differ by up to ~10 % on identical code; only numbers from one call compare.

    python tools/ab_bench.py [--rounds 2] [--steps 10] [--config 1] [--env tag:KEY=VALUE] default base default+tag ...
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--bench-args", default="", help="extra arguments for bench.py, e.g. '--method graph'")
    ap.add_argument("--env", action="append", default=[], help="TAG:KEY=VALUE sets an environment variable for the variants named LIB+TAG")
    args = ap.parse_args()
    extra = {}
    for e in args.env:
        name, kv = e.split(":", 1)
        k, v = kv.split("=", 1)
        extra.setdefault(name, {})[k] = v
    rows = {v: [] for v in args.variants}
    for r in range(args.rounds):
        for v in args.variants:
            env = dict(os.environ)
            lib = v.split("+")[0]
            if lib != "default":
                env["BRX_LIB_PATH"] = os.path.join(ROOT, "br_amd", "lib", "ab", f"libbrx_{lib}.so")
            for tag in v.split("+")[1:]:  # "default+nowide" = the default library with the variables given as --env nowide:K=V
                env.update(extra[tag])
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(args.steps), "--warmup", "2",
                                  "--config", str(args.config), "--no-cpu-baseline", "--no-e2e"] + args.bench_args.split(), env=env, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(v, "FAILED", out.stderr[-400:], flush=True)
                continue
            d = json.loads(line[-1])
            rows[v].append(d)
            print(f"round {r} {v}: {d['value']} {d['unit']} {d['ms_per_step']} ms", flush=True)
    kernels = []
    for v in args.variants:
        for d in rows[v]:
            for k in d["kernels"]:
                if k not in kernels:
                    kernels.append(k)
    print(f"{'':28s}" + "".join(f"{v:>14s}" for v in args.variants))
    for key in ["ms_per_step"]:
        print(f"{key:28s}" + "".join(f"{min(d[key] for d in rows[v]):14.3f}" if rows[v] else f"{'-':>14s}" for v in args.variants))
    for ph in ["build_ms_per_step", "correct_ms_per_step"]:
        print(f"{ph:28s}" + "".join(f"{min(d['phases'][ph] for d in rows[v]):14.3f}" if rows[v] else f"{'-':>14s}" for v in args.variants))
    for k in kernels:
        print(f"{k:28s}" + "".join(
            f"{min([d['kernels'][k]['avg_ms'] * d['kernels'][k]['launches'] / d['steps'] for d in rows[v] if k in d['kernels']] or [float('nan')]):14.3f}"
            if rows[v] else f"{'-':>14s}" for v in args.variants))


if __name__ == "__main__":
    main()
