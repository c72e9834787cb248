#!/usr/bin/env python3
"""Host issue time of one group step — ma_group_enqueue_sum_i64 + ma_group_enqueue_sum_f64 + ma_group_exchange — as a
function of the member count, with the calling thread issuing for every member (round 2, MA_GROUP_ISSUE_CALLER) against
one issue thread per member (round 3). The strong-scaling form of the metric (a 10^9-row column partitioned over 8 GPUs)
has 0.14 ms of scan per GPU per column: a host that needs 0.2 ms to issue a step is the bottleneck there.

On a one-GPU box the members share device 0 (independent contexts and streams; host exchange — RCCL refuses two ranks on
one device), which is exactly what the issue path costs on the host: the same launches, the same streams, only the
devices coincide. Chunks are tiny (4096 rows) and steps are issued in bursts that the streams absorb, so the figure is
the calling thread's time inside the three calls, not the GPU's.

    python tools/bench_group_issue.py [--members 1,2,4,8] [--burst 64] [--bursts 20] > profiles/r03_group_issue.json
"""
import argparse
import ctypes as C
import json
import statistics
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from minarrow_amd import ffi  # noqa: E402
from minarrow_amd.host import Group  # noqa: E402


def measure(members: int, issue: str, rows: int, burst: int, bursts: int, devices=None):
    lib = ffi.load_library()
    devs = devices or [0] * members
    with Group(devs, exchange="host" if len(set(devs)) < len(devs) else "rccl-or-host", issue=issue) as g:
        ctxs = [g.member_ctx(i) for i in range(members)]
        ci = [c.alloc(rows * 8) for c in ctxs]
        cf = [c.alloc(rows * 8) for c in ctxs]
        for r, c in enumerate(ctxs):
            c.synth_iota("i64", ci[r], rows, r * rows)
            c.synth_iota("f64", cf[r], rows, r * rows)
        g.synchronize()
        # argument tables built ONCE: the figure is the library's host time, not ctypes marshalling
        pi = (C.c_void_p * members)(*[b.ptr for b in ci])
        pf = (C.c_void_p * members)(*[b.ptr for b in cf])
        ln = (C.c_size_t * members)(*([rows] * members))
        api, apf, aln = (C.cast(x, C.c_void_p) for x in (pi, pf, ln))
        h = g.handle
        e_i, e_f, ex, sync = lib.ma_group_enqueue_sum_i64, lib.ma_group_enqueue_sum_f64, lib.ma_group_exchange, lib.ma_group_synchronize

        def step():
            return e_i(h, 0, api, aln, None, None) | e_f(h, 0, apf, aln, None, None) | ex(h)

        for _ in range(burst):
            assert step() == 0
        assert sync(h) == 0
        per_step, wall = [], []
        for _ in range(bursts):
            t0 = time.perf_counter()
            for _ in range(burst):
                step()
            t1 = time.perf_counter()
            assert sync(h) == 0
            t2 = time.perf_counter()
            per_step.append((t1 - t0) / burst * 1e6)
            wall.append((t2 - t0) / burst * 1e6)
        total = members * rows
        want = total * (total - 1) // 2
        got = g.result(0)
        assert got[0] == want and got[1] == total and got[3] == total, (got, want)
        out = {"members": members, "devices": devs, "issue": g.issue_kind, "exchange": g.exchange_kind, "rows_per_chunk": rows,
               "launches_per_step": 2 * members + (2 * members if g.exchange_kind == "rccl" else 0),
               "host_issue_us_per_step": {"median": statistics.median(per_step), "min": min(per_step), "max": max(per_step)},
               "wall_us_per_step_incl_gpu": {"median": statistics.median(wall), "min": min(wall)},
               "steps": burst * bursts}
        for b in ci + cf:
            b.free()
    return out


def main():
    import os

    sys.stdout.flush()
    result_fd = os.dup(1)  # RCCL prints a banner to fd 1 from native code: keep the real stdout for the JSON
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--members", default="1,2,4,8")
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--burst", type=int, default=64)
    ap.add_argument("--bursts", type=int, default=20)
    ap.add_argument("--distinct", action="store_true", help="one member per visible GPU instead of sharing device 0")
    args = ap.parse_args()
    res = []
    for m in [int(x) for x in args.members.split(",")]:
        devices = list(range(m)) if args.distinct else None
        if args.distinct and m > ffi.device_count():
            continue
        for issue in ("caller", "threads"):
            r = measure(m, issue, args.rows, args.burst, args.bursts, devices)
            res.append(r)
            print(json.dumps(r), file=sys.stderr, flush=True)
    by = {(r["members"], r["issue"]): r["host_issue_us_per_step"]["median"] for r in res}
    summary = {f"{m}_members": {"caller_us": by[(m, "caller")], "threads_us": by[(m, "threads")],
                                "ratio": by[(m, "caller")] / by[(m, "threads")]}
               for m in sorted({r["members"] for r in res})}
    os.write(result_fd, (json.dumps({"tool": "tools/bench_group_issue.py",
                                     "what": "host time of one group step (2 scans + exchange) on the calling thread",
                                     "summary": summary, "runs": res}, indent=1) + "\n").encode())


if __name__ == "__main__":
    main()
