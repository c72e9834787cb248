#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs (bench.py measures configs[1], the headline metric):

  config 1  1M-row IntegerArray<i64> sum as one GPU call (latency-bound; the host-side loop is timed by bench.py)
  config 3  1B-row FloatArray<f64> elementwise add / mul, array (+) array and array (+) scalar (fused broadcast)
  config 4  1B-row i64 sum with 10 % nulls via Bitmask, row-chunk partitioned across the ranks + scalar exchange
  config 5  SuperTable of 8 batches (i64 + f64 columns, 10 % nulls) dealt to the ranks in order: per-column reduce
            through one scalar exchange + consolidate of each rank's batches (--config5-rows 1000000000 = full size)

Run on 1 GPU directly, or under torch.distributed.run for N ranks (configs 4 and 5 shard across them).
Prints one JSON object per config on rank 0. HIP-event timing on the launch stream, HBM-resident inputs."""
import argparse
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

OPS = {"add": 0, "multiply": 2}


def timed(ctx, fn, reps, warm=2):
    for _ in range(warm):
        fn()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    ctx.timer_stop()
    return ctx.timer_elapsed_ms() / reps


def _timeit(fn, t):
    t0 = t.perf_counter()
    fn()
    return t.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000_000)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--configs", type=str, default="3,4,5")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the multi-rank logic with the ranks sharing the visible GPU(s)")
    ap.add_argument("--config5-rows", type=int, default=0, help="rows per SuperTable batch (default rows/8)")
    args = ap.parse_args()

    # JSON lines only on stdout: native libraries (RCCL's version banner, gloo's connection messages) write to fd 1
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    from minarrow_amd.host import Context
    from minarrow_amd.parallel import ScalarExchange, batch_ranges, row_chunks

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rehearsal = args.backend == "gloo"
    device_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    stream = torch.cuda.current_stream(dev)
    ctx = Context(device_index, stream=stream.cuda_stream)
    n = args.rows
    configs = args.configs.split(",")
    out = []

    if "1" in configs and rank == 0:
        # configs[0] is the reference's own CPU-runnable case — the scalar loop of benches/hotloop_benchmark_std.rs:49-57
        # over 10^6 rows; bench.py's cpu_baseline leg times it on the host ("config0_1m_rows"). Here: what the same
        # column costs as one GPU call (8 MB is L2/MALL-resident, so this is a latency measurement, not bandwidth).
        import time as _t

        m = 1_000_000
        col = ctx.alloc(m * 8)
        ctx.synth_iota("i64", col, m, 0)
        slot = torch.zeros(8, dtype=torch.int64, device=dev)
        s, c = ctx.sum("i64", col, m)
        sync_us = min(_timeit(lambda: ctx.sum("i64", col, m), _t) for _ in range(200)) * 1e6
        ctx.set_async(True)
        ms = timed(ctx, lambda: ctx.sum_into("i64", col, m, out_sum=slot.data_ptr(), out_count=slot.data_ptr() + 8), 200, 5)
        ctx.set_async(False)
        ctx.synchronize()
        out.append({"config": 1, "workload": f"{m}-row IntegerArray<i64> sum as one GPU call (benches/hotloop_benchmark_std.rs shape)",
                    "gpu_sync_call_us": sync_us, "gpu_enqueued_us": ms * 1e3, "gpu_enqueued_gbps": 8 * m / ms / 1e6,
                    "sum": s, "expected": m * (m - 1) // 2, "count": c})
        col.free()

    if "3" in configs and rank == 0:
        a, b, o = (ctx.alloc(n * 8) for _ in range(3))
        ctx.synth_iota("f64", a, n, 0)
        ctx.apply_scalar("f64", "lhs", a, n, float(n), 1, b)  # b[i] = n - i
        ctx.set_async(True)
        res = {}
        for op in ("add", "multiply"):
            ms = timed(ctx, lambda: ctx.apply("f64", a, b, OPS[op], o, n, n), args.reps)
            res[f"{op}_array_array"] = {"ms": ms, "grows_per_s": n / ms / 1e6, "gbps": 24 * n / ms / 1e6, "bytes_per_row": 24}
            ms = timed(ctx, lambda: ctx.apply_scalar("f64", "rhs", a, n, 2.5, OPS[op], o), args.reps)
            res[f"{op}_array_scalar"] = {"ms": ms, "grows_per_s": n / ms / 1e6, "gbps": 16 * n / ms / 1e6, "bytes_per_row": 16}
        ctx.set_async(False)
        ctx.synchronize()
        s, _ = ctx.sum("f64", o, n)  # last op: a * 2.5
        res["check_sum_a_times_2p5"] = s
        out.append({"config": 3, "workload": f"{n}-row f64 add/mul, array(+)array and array(+)scalar, 1 MI355X", **res})
        for buf in (a, b, o):
            buf.free()

    if "4" in configs:
        lo, hi = row_chunks(n, world)[rank]
        rows = hi - lo
        data = ctx.alloc(max(rows, 1) * 8)
        mask = ctx.alloc(max(rows, 64) // 8 + 64)
        ctx.synth_iota("i64", data, rows, lo)
        ctx.synth_validity(mask, rows, seed=0xC0FFEE, first_index=lo, null_every=10)
        ex = ScalarExchange(dev)
        ctx.set_async(True)

        def step():
            ctx.sum_into("i64", data, rows, out_sum=ex.slot_ptr(0), out_count=ex.slot_ptr(1), mask=mask)
            ex.exchange()
            ex.fold_on_device(ctx)

        for _ in range(3):
            step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(args.reps):
            step()
        e1.record(stream)
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1) / args.reps
        if world > 1:
            t = torch.tensor([ms], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ms = float(t.item())
        ctx.set_async(False)
        total, cnt, _, _ = ex.results()
        if rank == 0:
            out.append({"config": 4, "workload": f"{n}-row i64 sum, 10 % nulls via Bitmask, row chunks over {world} GPU(s)"
                        + ((" + gloo all-gather (REHEARSAL: ranks share a GPU)" if rehearsal else " + RCCL all-gather") if world > 1 else ""), "n_gpus": world, "ms_per_step": ms,
                        "grows_per_s": n / ms / 1e6, "gbps": 8.125 * n / ms / 1e6, "bytes_per_row": 8.125,
                        "sum_valid": total, "valid_count": cnt, "null_fraction": 1 - cnt / n})
        data.free()
        mask.free()

    if "5" in configs:
        # The SuperTable's k batches are dealt to the ranks in order (batch_ranges); one column at a time so that the
        # full-size case (--config5-rows 1000000000: 64 GB of chunks + 64 GB consolidated) fits one 288 GB MI355X.
        k = 8
        rows = args.config5_rows or n // k
        ranges = batch_ranges(k, world)
        lo, hi = ranges[rank]
        local = hi - lo
        slots = max(z - a for a, z in ranges)
        res = {}
        for tag in ("i64", "f64"):
            chunks = [ctx.alloc(rows * 8) for _ in range(local)]
            masks = [ctx.alloc(rows // 8 + 64) for _ in range(local)]
            for c in range(local):
                ctx.synth_iota(tag, chunks[c], rows, lo + c)  # v[i] = i + chunk (benches/consolidate.rs:37-58 pattern)
                ctx.synth_validity(masks[c], rows, seed=0xABC + lo + c, null_every=10)
            o = ctx.alloc(max(local * rows, 1) * 8)
            om = ctx.alloc(local * rows // 8 + 64)
            # (1) per-column reduce, logically consolidated: every local batch into its own record, one exchange,
            #     batch-ordered fold on the GPU — no column bytes move
            ex = ScalarExchange(dev, n_columns=1, slots_per_rank=slots)
            ctx.set_async(True)

            def reduce_step():
                for c in range(local):
                    if tag == "i64":
                        ctx.sum_into(tag, chunks[c], rows, out_sum=ex.slot_ptr(0, 0, c), out_count=ex.slot_ptr(1, 0, c), mask=masks[c])
                    else:
                        ctx.sum_into(tag, chunks[c], rows, out_sum=ex.slot_ptr(2, 0, c), dd_lo=ex.slot_ptr(3, 0, c),
                                     out_count=ex.slot_ptr(4, 0, c), mask=masks[c])
                ex.exchange()
                ex.fold_on_device(ctx)

            for _ in range(2):
                reduce_step()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(args.reps):
                reduce_step()
            e1.record(stream)
            torch.cuda.synchronize(dev)
            ms_reduce_logical = e0.elapsed_time(e1) / args.reps
            # (2) this rank's batches joined into one contiguous piece (+ validity): ma_consolidate_column
            ms = timed(ctx, lambda: ctx.consolidate_column(8, chunks, [rows] * local, o, masks, [0] * local, om),
                       max(2, args.reps // 2), 1) if local else 0.0
            ms_reduce_physical = timed(ctx, lambda: ctx.sum_into(tag, o, local * rows, out_sum=ex.slot_ptr(5), out_count=ex.slot_ptr(6),
                                                                 mask=om), args.reps) if local else 0.0
            if world > 1:
                t = torch.tensor([ms_reduce_logical, ms, ms_reduce_physical], dtype=torch.float64, device="cpu" if rehearsal else dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                ms_reduce_logical, ms, ms_reduce_physical = (float(v) for v in t.tolist())
            ctx.set_async(False)
            ctx.synchronize()
            isum, icnt, fsum, fcnt = ex.results()
            got_sum, got_cnt = (isum, icnt) if tag == "i64" else (fsum, fcnt)
            whole = ctx.sum(tag, o, local * rows, mask=om) if local else (0, 0)
            entry = {"consolidate_ms": ms, "consolidate_gbps": 16.25 * local * rows / ms / 1e6 if ms else None,
                     "consolidate_grows_per_s": local * rows / ms / 1e6 if ms else None,
                     "reduce_logical_ms": ms_reduce_logical, "reduce_logical_grows_per_s": k * rows / ms_reduce_logical / 1e6,
                     "reduce_logical_gbps": 8.125 * k * rows / ms_reduce_logical / 1e6,
                     "reduce_physical_ms": ms_reduce_physical,
                     "reduce_physical_grows_per_s": local * rows / ms_reduce_physical / 1e6 if ms_reduce_physical else None,
                     "sum": got_sum, "valid_count": got_cnt}
            if world == 1:
                # the same per-chunk reduce as ONE ma_sum_columns call (two launches, host wall clock incl. the call overhead)
                fmt = "l" if tag == "i64" else "g"
                f_cols, i_cols, c_cols = ctx.sum_columns(fmt, chunks, [rows] * k, masks, [0] * k)
                import time as _t
                t0 = _t.perf_counter()
                for _ in range(args.reps):
                    ctx.sum_columns(fmt, chunks, [rows] * k, masks, [0] * k)
                ms_reduce_columns = (_t.perf_counter() - t0) / args.reps * 1e3
                entry["reduce_logical_one_call_ms_wall"] = ms_reduce_columns
                entry["reduce_logical_one_call_grows_per_s"] = k * rows / ms_reduce_columns / 1e6
                entry["count_matches"] = whole[1] == got_cnt == int(sum(int(c) for c in c_cols))
                entry["sum_matches"] = (whole[0] & ((1 << 64) - 1)) == got_sum if tag == "i64" else \
                    abs(whole[0] - got_sum) <= abs(whole[0]) * 2.0 ** -52
            res[tag] = entry
            for buf in chunks + masks + [o, om]:
                buf.free()
        if rank == 0:
            out.append({"config": 5, "workload": f"SuperTable of {k} x {rows}-row batches, columns i64 + f64 with 10 % nulls: per-column "
                                                 f"reduce (batch-sharded over {world} GPU(s), one scalar exchange) + consolidate of each "
                                                 f"rank's batches", "n_gpus": world, **res})

    if "x" in configs and rank == 0:
        # extras for DESIGN.md's kernel table: FMA (32 B/row), masked elementwise, bitmask kernels on 8 Gbit windows
        res = {}
        a, b, c, o = (ctx.alloc(n * 8) for _ in range(4))
        for buf, start in ((a, 1), (b, 2), (c, 3)):
            ctx.synth_iota("f64", buf, n, start)
        mask, om = ctx.alloc(n // 8 + 64), ctx.alloc(n // 8 + 64)
        ctx.synth_validity(mask, n, seed=0xC0FFEE, null_every=10)
        ctx.set_async(True)
        ms = timed(ctx, lambda: ctx.apply_fma("f64", a, b, c, o, n, n, n), args.reps)
        res["fma_f64"] = {"ms": ms, "gbps": 32 * n / ms / 1e6, "grows_per_s": n / ms / 1e6, "bytes_per_row": 32}
        ms = timed(ctx, lambda: ctx.apply("f64", a, b, 0, o, n, n, mask=mask, out_mask=om), args.reps)
        res["add_f64_masked"] = {"ms": ms, "gbps": 24.25 * n / ms / 1e6, "grows_per_s": n / ms / 1e6, "bytes_per_row": 24.25}
        ia, ib, io = a, b, o  # reuse the buffers as i64
        ctx.synth_iota("i64", ia, n, 1)
        ctx.synth_iota("i64", ib, n, 1)
        ms = timed(ctx, lambda: ctx.apply("i64", ia, ib, 3, io, n, n, mask=mask, out_mask=om), max(2, args.reps // 3))
        res["div_i64_masked_row_kernel"] = {"ms": ms, "gbps": 24.25 * n / ms / 1e6, "grows_per_s": n / ms / 1e6}
        ms = timed(ctx, lambda: ctx.apply("i64", ia, ib, 3, io, n, n), max(2, args.reps // 3))
        res["div_i64_dense"] = {"ms": ms, "gbps": 24 * n / ms / 1e6, "grows_per_s": n / ms / 1e6}
        ctx.set_async(False)
        ctx.synchronize()
        bits = 8 * n * 8  # treat an 8 GB buffer as a 64 Gbit bitmap
        ctx.set_async(True)
        ms = timed(ctx, lambda: ctx.mask_words_op("and_masks", a, 0, b, 0, bits, o), args.reps)
        res["and_masks_64Gbit"] = {"ms": ms, "gbps": 3 * bits / 8 / ms / 1e6, "gbits_per_s": bits / ms / 1e6}
        ctx.set_async(False)
        ctx.synchronize()
        import time as _t
        t0 = _t.perf_counter()
        pop = ctx.popcount_mask(a, 0, bits)
        res["popcount_64Gbit"] = {"ms_host_sync": (_t.perf_counter() - t0) * 1e3, "gbps": bits / 8 / (_t.perf_counter() - t0) / 1e9, "popcount": pop}
        out.append({"config": "extras", "workload": f"{n}-row fma / masked add / i64 div; 64 Gbit bitmap and/popcount, 1 MI355X", **res})
        for buf in (a, b, c, o, mask, om):
            buf.free()

    if rank == 0:
        for o_ in out:
            os.write(result_fd, (json.dumps(o_) + "\n").encode())
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
