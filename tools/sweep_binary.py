#!/usr/bin/env python3
"""Tuning sweep for the elementwise kernels at BASELINE config 3 size (1 B-row f64 add/mul, array (+) array and
array (+) scalar): unroll x workgroups-per-CU, dense and masked, HIP-event timed, interleaved rounds."""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from minarrow_amd.host import Context  # noqa: E402

OPS = {"add": 0, "subtract": 1, "multiply": 2, "divide": 3, "remainder": 4, "power": 5, "floordiv": 6}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--variants", type=str, default="0,4,6")
    ap.add_argument("--bpcs", type=str, default="0,1,2,4,8")
    ap.add_argument("--types", type=str, default="f64")
    ap.add_argument("--ops", type=str, default="add,multiply")
    ap.add_argument("--kinds", type=str, default="aa,as")
    ap.add_argument("--masked", type=str, default="0,1")
    ap.add_argument("--out", type=str, default="")
    args = ap.parse_args()
    n = args.rows
    ctx = Context(0)
    mask = ctx.alloc(n // 8 + 64)
    out_mask = ctx.alloc(n // 8 + 64)
    ctx.synth_validity(mask, n, seed=0xC0FFEE, null_every=10)
    results = []
    for tag in args.types.split(","):
        esz = 8 if tag.endswith("64") else 4
        a, b, out = ctx.alloc(n * esz), ctx.alloc(n * esz), ctx.alloc(n * esz)
        ctx.synth_iota(tag, a, n, 1)
        ctx.synth_iota(tag, b, n, 7)
        ctx.set_async(True)
        configs = [(op, kind, m, v, bpc) for op in args.ops.split(",") for kind in args.kinds.split(",")
                   for m in map(int, args.masked.split(",")) for v in map(int, args.variants.split(","))
                   for bpc in map(int, args.bpcs.split(","))]
        best = {c: float("inf") for c in configs}

        def run(op, kind, m):
            kw = dict(mask=mask, mask_bit_offset=0, out_mask=out_mask) if m else {}
            if kind == "aa":
                ctx.apply(tag, a, b, OPS[op], out, n, n, **kw)
            else:
                ctx.apply_scalar(tag, "rhs", a, n, 2.5 if tag[0] == "f" else 3, OPS[op], out, **kw)

        for _ in range(args.rounds):
            for c in configs:
                op, kind, m, v, bpc = c
                ctx.set_variant(v)
                ctx.set_blocks_per_cu(bpc)
                run(op, kind, m)
                ctx.timer_start()
                for _ in range(args.reps):
                    run(op, kind, m)
                ctx.timer_stop()
                best[c] = min(best[c], ctx.timer_elapsed_ms() / args.reps)
        ctx.set_async(False)
        ctx.synchronize()
        for (op, kind, m, v, bpc), ms in sorted(best.items(), key=lambda kv: (kv[0][:3], kv[1])):
            bytes_per_row = (3 if kind == "aa" else 2) * esz + (0.25 if m else 0)
            row = {"type": tag, "op": op, "kind": kind, "masked": bool(m), "variant": v,
                   "unroll": {0: "auto", 2: 4, 3: 8}[(v >> 1) & 7], "blocks_per_cu": bpc or "auto", "ms": ms,
                   "gbps": n * bytes_per_row / ms / 1e6, "grows": n / ms / 1e6}
            results.append(row)
            print(f"{tag} {op:9s} {kind} masked={m} unroll={row['unroll']!s:4s} bpc={row['blocks_per_cu']!s:4s} "
                  f"{ms:8.4f} ms {row['gbps']:8.1f} GB/s {row['grows']:7.1f} Grows/s", flush=True)
        for buf in (a, b, out):
            buf.free()
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(json.dumps(results, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
