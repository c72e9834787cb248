#!/usr/bin/env python3
"""Tuning sweep for the sum kernels: variant (unroll / non-temporal) x workgroups-per-CU, dense and masked,
timed with HIP events on the context's stream (interleaved rounds in one process). Prints a table and writes
JSON to --out."""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from minarrow_amd.host import Context, PinnedBuffer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--variants", type=str, default="0,2,4,6,8,3,5,7")
    ap.add_argument("--bpcs", type=str, default="0,1,2,3,4,8")
    ap.add_argument("--types", type=str, default="i64,f64")
    ap.add_argument("--out", type=str, default="")
    args = ap.parse_args()
    n = args.rows
    ctx = Context(0)
    slot = PinnedBuffer(64)
    mask = ctx.alloc(n // 8 + 64)
    ctx.synth_validity(mask, n, seed=0xC0FFEE, null_every=10)
    results = []
    for tag in args.types.split(","):
        esz = 8 if tag.endswith("64") else 4
        buf = ctx.alloc(n * esz)
        ctx.synth_iota(tag, buf, n, 0)
        ctx.set_async(True)
        configs = [(v, b, m) for m in (False, True) for v in map(int, args.variants.split(","))
                   for b in map(int, args.bpcs.split(","))]
        best = {c: float("inf") for c in configs}
        for _ in range(args.rounds):
            for c in configs:
                v, b, m = c
                ctx.set_variant(v)
                ctx.set_blocks_per_cu(b)
                kw = dict(mask=mask, mask_bit_offset=0) if m else {}
                ctx.sum_into(tag, buf, n, out_sum=slot.ptr, out_count=slot.ptr + 8, **kw)  # warm
                ctx.timer_start()
                for _ in range(args.reps):
                    ctx.sum_into(tag, buf, n, out_sum=slot.ptr, out_count=slot.ptr + 8, **kw)
                ctx.timer_stop()
                ms = ctx.timer_elapsed_ms() / args.reps
                best[c] = min(best[c], ms)
        ctx.set_async(False)
        for (v, b, m), ms in sorted(best.items(), key=lambda kv: kv[1]):
            bytes_ = n * esz + (n / 8 if m else 0)
            row = {"type": tag, "masked": m, "variant": v, "unroll": {0: "auto", 1: 2, 2: 4, 3: 8, 4: 16}[(v >> 1) & 7], "nt": not (v & 1),
                   "interleaved": bool(v & 16), "pace": {0: "auto", 1: 0, 2: 16, 3: 20, 4: 24, 5: 32}.get((v >> 5) & 7, 0),
                   "blocks_per_cu": b, "ms": ms, "gbps": bytes_ / ms / 1e6, "grows": n / ms / 1e6}
            results.append(row)
            print(f"{tag} masked={int(m)} unroll={row['unroll']} nt={int(row['nt'])} il={int(row['interleaved'])} pace={row['pace']} bpc={b:2d}  "
                  f"{ms:8.4f} ms  {row['gbps']:8.1f} GB/s  {row['grows']:7.1f} Grows/s", flush=True)
        buf.free()
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(json.dumps(results, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
