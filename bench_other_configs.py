"""The configs 3-5 legs of bench.py — BASELINE.json's other configurations, measured after and outside the timed headline region and
carried in the line's `other_configs` key: config 3 (1B-row f64 add / multiply, array (op) array and array (op) scalar), config 4
(1B-row i64 sum with 10 % nulls behind a Bitmask, row-chunk partitioned), config 5 (a SuperTable of 8 x 1B-row batches: per-batch
reduce, consolidate, reduce of the consolidated column) — at N = 1 (gpu_other_configs), over the members of a one-process group
(group_other_configs) and over rank processes (ranks_other_configs), plus the CPU side of the same legs (cpu_other_configs: the
oracle, timed as a reported baseline). bench.py owns the contract (the headline, `roofline`, `cpu_baseline`, the N > 1 ladders);
nothing here is part of `value`. Reference shapes: src/kernels/arithmetic/dispatch.rs:138-206 (elementwise),
benches/benchmark_parallel_simd.rs:81-98 (partitioned sums), src/structs/chunked/super_table.rs:657-687 (consolidate)."""
from __future__ import annotations

import json
import math
import os
import sys
import time

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters)
OPS = {"add": 0, "subtract": 1, "multiply": 2}


def two_sum(a: float, b: float):
    s = a + b
    bp = s - a
    return s, (a - (s - bp)) + (b - bp)


def fold_dd(pairs):
    """Sum of double-double (hi, lo) pairs in the given order; returns the rounded double."""
    hi, lo = 0.0, 0.0
    for h, l in pairs:
        hi, e = two_sum(hi, h)
        lo += e + l
    return hi + lo


def cpu_other_configs(oracle, np):
    """The reference's CPU cost of the other BASELINE configs on bounded samples, for tools/bench_configs.py's GPU
    figures to stand beside. The reference's elementwise kernels, its broadcast and its consolidate are
    single-threaded (src/kernels/arithmetic/mod.rs:29-31; "TODO: Parallelise", src/kernels/broadcast/super_array.rs:193),
    so one thread is the faithful baseline here. Outputs are allocated and touched beforehand (the reference pays
    first-touch page faults on its fresh Vec64 inside the call; leaving them out favours the CPU)."""
    import ctypes as C

    def best(fn, reps=5):
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            times.append(time.perf_counter() - t0)
        return min(times)

    l = oracle.klib()
    n = 1 << 25
    a, b, out = (oracle.aligned_empty(n, np.float64) for _ in range(3))
    a[:] = np.arange(n, dtype=np.float64)
    b[:] = n - a
    out[:] = 0
    used = C.c_int(0)
    res = {}
    for op_name, op in (("add", 0), ("multiply", 2)):
        call = lambda: l.mo_apply_float_f64(oracle._p(a), n, oracle._p(b), n, op, None, n, oracle._p(out), None, 8,
                                            C.addressof(used))
        t = best(call)
        assert used.value == 1 and out[12345] == (a[12345] + b[12345] if op == 0 else a[12345] * b[12345])
        res[f"config3_f64_{op_name}_array_array"] = {"rows": n, "best_ms": t * 1e3, "grows_per_s": n / t / 1e9,
                                                     "gbps": 24 * n / t / 1e9}

        def broadcast_then_apply():  # maybe_broadcast_scalar_array materialises vec64![x; n] (routing/broadcast.rs:30-45)
            b[:] = 2.5
            call()

        t = best(broadcast_then_apply)
        res[f"config3_f64_{op_name}_array_scalar"] = {"rows": n, "best_ms": t * 1e3, "grows_per_s": n / t / 1e9}
    m = 1 << 26
    ints = np.arange(m, dtype=np.int64)
    bits = np.random.default_rng(1).integers(0, 256, size=m // 8 + 16, dtype=np.uint8)
    t = best(lambda: oracle.masked_sum(ints, bits, 0))
    res["config4_i64_sum_bitmask_gated"] = {"rows": m, "best_ms": t * 1e3, "grows_per_s": m / t / 1e9,
                                            "note": "build-defined semantics (the reference has no masked sum): scalar loop"}
    k, rows = 8, 1 << 22
    chunks = [np.arange(rows, dtype=np.int64) + c for c in range(k)]
    masks = [np.random.default_rng(c).integers(0, 256, size=rows // 8 + 16, dtype=np.uint8) for c in range(k)]
    t = best(lambda: oracle.consolidate_column(chunks, masks, [0] * k), reps=3)
    res["config5_consolidate_i64_column"] = {"rows": k * rows, "best_ms": t * 1e3, "grows_per_s": k * rows / t / 1e9,
                                             "gbps": 16.25 * k * rows / t / 1e9}
    return res


def _timed(ctx, fn, reps, warm=2):
    """Mean milliseconds per call of `fn` (enqueue-only calls on ctx's stream), HIP events on that stream."""
    for _ in range(warm):
        fn()
    ctx.timer_start()
    for _ in range(reps):
        fn()
    ctx.timer_stop()
    return ctx.timer_elapsed_ms() / reps


def gpu_other_configs(ctx, n: int, reps: int):
    """BASELINE configs 3, 4 and 5 at `n` rows on ONE MI355X, after and outside the timed headline region: per kernel
    {ms, gbps, frac_of_peak, frac_of_copy, parity}. Algorithmic bytes per row are SURVEY.md §8(d)'s. Parity here is by
    closed forms of the synthetic inputs, numpy on downloaded windows, and partition identities (the CPU oracle stays
    with the test-suite and the cpu_baseline leg). `frac_of_copy` divides by the best same-process copy rate
    (this library's 16-byte load->store kernel, and the runtime's hipMemcpyAsync) so that box-to-box spread of the
    read+write mix does not hide in the fraction of the 8 TB/s spec."""
    import numpy as np

    from minarrow_amd.parallel import row_chunks

    M64 = (1 << 64) - 1
    res = {}

    def entry(ms, bytes_per_row, rows, parity=None, **extra):
        gbps = bytes_per_row * rows / ms / 1e6
        e = {"ms": ms, "gbps": gbps, "grows_per_s": rows / ms / 1e6, "bytes_per_row": bytes_per_row,
             "frac_of_peak": gbps / HBM_PEAK_GBPS}
        if parity is not None:
            e["parity"] = bool(parity)
        e.update(extra)
        return e

    # The OUTPUT buffer is a plain ma_dev_alloc block — what a caller who brings its own `out` has (the ABI's normal contract:
    # the reference allocates `out` per call, src/kernels/arithmetic/dispatch.rs:88-89), wherever the driver put it. The write
    # rate of a block depends on that placement, which is why every read+write figure also carries
    # `frac_of_copy`: the fraction of a plain copy into the SAME block, measured in this process. The placement search of
    # rounds 2-3 (ma_dev_alloc_output) is opt-in since round 4 and not used here.
    o = ctx.alloc(n * 8)
    res["output_block"] = "plain ma_dev_alloc block (no placement search)"
    a, b = (ctx.alloc(n * 8) for _ in range(2))
    mask_bytes = ((n + 511) // 512) * 64 + 64
    mask, om = ctx.alloc(mask_bytes), ctx.alloc(mask_bytes)
    slot = ctx.alloc(256)
    ctx.synth_iota("f64", a, n, 0)
    ctx.apply_scalar("f64", "lhs", a, n, float(n), 1, b)  # b[i] = n - i

    # ---- the same-process reference: a plain copy (8 B read + 8 B written per row) ------------------------------
    ctx.set_async(True)
    ms = _timed(ctx, lambda: ctx.consolidate_column(8, [a], [n], o), reps)
    res["copy_kernel_16B_per_lane"] = entry(ms, 16, n)
    ms = _timed(ctx, lambda: ctx.dev_copy(o, a, n * 8), reps)
    res["copy_hipMemcpyDtoD"] = entry(ms, 16, n)
    copy_gbps = max(res["copy_kernel_16B_per_lane"]["gbps"], res["copy_hipMemcpyDtoD"]["gbps"])

    # ---- config 3: f64 add / mul, array (+) array and array (+) scalar (fused broadcast) ---------------------------
    def windows_equal(buf, fn, starts, count=4096):
        ok = True
        for s0 in starts:
            i = np.arange(s0, s0 + count, dtype=np.float64)
            ok = ok and bool(np.array_equal(buf.download(np.float64, count, s0 * 8), fn(i)))
        return ok

    starts = (0, n // 2, n - 4096)
    fn_n = float(n)
    tri = n * (n - 1) // 2
    c3 = {}
    ms = _timed(ctx, lambda: ctx.apply("f64", a, b, OPS["add"], o, n, n), reps)
    ctx.set_async(False)
    s, c = ctx.sum("f64", o, n)  # a + b == n everywhere: the sum is n^2, exact
    ok = c == n and s == fn_n * fn_n and windows_equal(o, lambda i: i + (fn_n - i), starts)
    c3["add_array_array"] = entry(ms, 24, n, ok)
    ctx.set_async(True)
    ms = _timed(ctx, lambda: ctx.apply("f64", a, b, OPS["multiply"], o, n, n), reps)
    ctx.set_async(False)
    s, c = ctx.sum("f64", o, n)
    exact = (n ** 3 - n) // 6  # sum of i * (n - i); each product is rounded once, so the sum is within n * 2^-53 relative
    ok = c == n and abs(s - exact) <= abs(exact) * 2.0 ** -50 and windows_equal(o, lambda i: i * (fn_n - i), starts)
    c3["multiply_array_array"] = entry(ms, 24, n, ok)
    ctx.set_async(True)
    ms = _timed(ctx, lambda: ctx.apply_scalar("f64", "rhs", a, n, 2.5, OPS["add"], o), reps)
    ctx.set_async(False)
    s, c = ctx.sum("f64", o, n)
    exact = tri + 2.5 * n  # every i + 2.5 is exact
    ok = c == n and abs(s - exact) <= math.ulp(exact) and windows_equal(o, lambda i: i + 2.5, starts)
    c3["add_array_scalar"] = entry(ms, 16, n, ok)
    ctx.set_async(True)
    ms = _timed(ctx, lambda: ctx.apply_scalar("f64", "rhs", a, n, 2.5, OPS["multiply"], o), reps)
    ctx.set_async(False)
    s, c = ctx.sum("f64", o, n)
    exact = 2.5 * tri  # every i * 2.5 is exact
    ok = c == n and abs(s - exact) <= math.ulp(exact) and windows_equal(o, lambda i: i * 2.5, starts)
    c3["multiply_array_scalar"] = entry(ms, 16, n, ok)
    res["config3_f64_elementwise"] = c3

    # ---- extras named by the round-1 review: FMA, masked add, bitmap AND ------------------------------------------
    ex = {}
    acc = ctx.alloc(n * 8)
    ctx.synth_iota("f64", acc, n, 3)
    ctx.set_async(True)
    ms = _timed(ctx, lambda: ctx.apply_fma("f64", a, b, acc, o, n, n, n), reps)
    ctx.set_async(False)
    ok = True
    for s0 in starts:  # fused: ONE rounding of the exact i * (n - i) + (i + 3) — Python's int -> float is correctly rounded
        want = np.array([float(i * (n - i) + i + 3) for i in range(s0, s0 + 1024)])
        ok = ok and bool(np.array_equal(o.download(np.float64, 1024, s0 * 8), want))
    ex["fma_f64"] = entry(ms, 32, n, ok)
    acc.free()
    ctx.synth_validity(mask, n, seed=0xC0FFEE, null_every=10)
    ctx.set_async(True)
    ms = _timed(ctx, lambda: ctx.apply("f64", a, b, OPS["add"], o, n, n, mask=mask, out_mask=om), reps)
    ctx.set_async(False)
    s, c = ctx.sum("f64", o, n, mask=om)
    pop = ctx.popcount_mask(mask, 0, n)
    ok = c == pop and s == fn_n * pop and \
        bool(np.array_equal(om.download(np.uint8, 4096), mask.download(np.uint8, 4096)))
    ex["add_f64_masked"] = entry(ms, 24.25, n, ok)
    bits = 8 * n * 8  # an 8-GB buffer as a 64-Gbit bitmap
    ctx.set_async(True)
    ms = _timed(ctx, lambda: ctx.mask_words_op("and_masks", a, 0, b, 0, bits, o), reps)
    ctx.set_async(False)
    wa, wb, wo = (x.download(np.uint64, 4096, (n // 2) * 8) for x in (a, b, o))
    ex["and_masks_64Gbit"] = entry(ms, 24, n, bool(np.array_equal(wa & wb, wo)), note="bytes_per_row counts 8-byte words")
    res["extras"] = ex

    # ---- config 4: i64 sum with 10 % nulls via Bitmask -----------------------------------------------------------
    ctx.synth_iota("i64", a, n, 0)
    ctx.set_async(True)
    ms = _timed(ctx, lambda: ctx.sum_into("i64", a, n, out_sum=slot.ptr, out_count=slot.ptr + 8, mask=mask), reps)
    ctx.set_async(False)
    whole, cnt = ctx.sum("i64", a, n, mask=mask)
    part_s, part_c = 0, 0
    for lo, hi in row_chunks(n, 8):  # the 8-way row-chunk partition of the multi-GPU config: checksum of checksums
        ps, pc = ctx.sum("i64", a.offset(lo * 8), hi - lo, mask=mask, mask_bit_offset=lo)
        part_s, part_c = (part_s + ps) & M64, part_c + pc
    w0, wn = (n // 2 // 64) * 64 + 192, 1 << 20
    wd = a.download(np.int64, wn, w0 * 8)
    wm = np.unpackbits(mask.download(np.uint8, wn // 8, w0 // 8), bitorder="little").astype(bool)
    ws, wc = ctx.sum("i64", a.offset(w0 * 8), wn, mask=mask, mask_bit_offset=w0)
    ok = part_s == (whole & M64) and part_c == cnt == pop and ws == int(wd[wm].sum()) and wc == int(wm.sum())
    res["config4_i64_sum_10pct_nulls"] = entry(ms, 8.125, n, ok, sum_valid=whole, valid_count=cnt, null_fraction=1 - cnt / n)

    # ---- config 5: SuperTable of 8 batches, consolidate + per-column reduce ---------------------------------------
    k = 8
    rows = n // k
    total = rows * k
    mstride = ((rows + 511) // 512) * 64
    chunk_masks = ctx.alloc(k * mstride + 64)
    recs = ctx.alloc(k * 64)
    fin = ctx.alloc(64)
    ctx.dev_memset(recs, 0, k * 64)
    c5 = {}
    for tag in ("i64", "f64"):
        chunks = [a.offset(c * rows * 8) for c in range(k)]
        masks = [chunk_masks.offset(c * mstride) for c in range(k)]
        for c in range(k):
            ctx.synth_iota(tag, chunks[c], rows, c)  # v[i] = i + batch (benches/consolidate.rs:37-58 pattern)
            ctx.synth_validity(masks[c], rows, seed=0xABC + c, null_every=10)
        ctx.set_async(True)

        def reduce_logical():  # per-batch sums into records + the batch-ordered fold: no column bytes move
            for c in range(k):
                r = recs.ptr + 64 * c
                if tag == "i64":
                    ctx.sum_into(tag, chunks[c], rows, out_sum=r, out_count=r + 8, mask=masks[c])
                else:
                    ctx.sum_into(tag, chunks[c], rows, out_sum=r + 16, dd_lo=r + 24, out_count=r + 32, mask=masks[c])
            ctx.fold_sum_records(recs.ptr, k, 8, fin.ptr)

        ms_red = _timed(ctx, reduce_logical, reps)
        ms_con = _timed(ctx, lambda: ctx.consolidate_column(8, chunks, [rows] * k, o, masks, [0] * k, om), max(2, reps // 2), 1)
        ms_phys = _timed(ctx, lambda: ctx.sum_into(tag, o, total, out_sum=slot.ptr, out_count=slot.ptr + 8, mask=om), reps)
        ctx.set_async(False)
        ctx.synchronize()
        f = fin.download(np.uint64, 4)
        phys_s, phys_c = ctx.sum(tag, o, total, mask=om)
        if tag == "i64":
            same = int(f[0]) == (phys_s & M64) and int(f[1]) == phys_c
        else:
            log_s = float(f[2:3].view(np.float64)[0])
            same = int(f[3]) == phys_c and abs(log_s - phys_s) <= 2 * math.ulp(phys_s)
        seam = rows - 2048  # a window across the join of batches 0 and 1 must equal the source bytes
        got = o.download(np.uint64, 4096, seam * 8)
        want = np.concatenate([a.download(np.uint64, 2048, seam * 8), a.download(np.uint64, 2048, rows * 8)])
        gm = om.download(np.uint8, 512, seam // 8)
        wmk = np.concatenate([chunk_masks.download(np.uint8, 256, seam // 8), chunk_masks.download(np.uint8, 256, mstride)])
        same = same and bool(np.array_equal(got, want)) and (rows % 8 != 0 or bool(np.array_equal(gm, wmk)))
        c5[tag] = {"reduce_per_batch": entry(ms_red, 8.125, total, same),
                   "consolidate": entry(ms_con, 16.25, total, same),
                   "reduce_consolidated": entry(ms_phys, 8.125, total, same)}
    res["config5_supertable_8_batches"] = {"batches": k, "rows_per_batch": rows, **c5}

    # ---- config 5 at the reference's OWN batch size: RechunkStrategy::Auto = 8192 rows (src/structs/chunked/super_array.rs:
    # 51-59) — the column is ~122 000 chunks per 10^9 rows, not 8. One call each: the total of the chunked column with 10 %
    # nulls (ma_sum_chunks: a wave per chunk on in-place descriptors, double-double fold of all partials) and its
    # consolidation incl. validity (chunk-per-workgroup form). The pointer tables are built once, as a host holding a
    # SuperTable would; parity against the same rows scanned / copied as ONE array.
    try:
        import ctypes as C

        per = 8192
        kc = min(n // per, 122_000)
        if kc >= 1024:
            rows_c = kc * per
            ctx.set_async(False)
            ctx.synth_iota("i64", a, rows_c, 3)
            ctx.synth_validity(mask, rows_c, seed=0xC5, first_index=0, null_every=10)
            tab = lambda xs: C.cast((C.c_void_p * kc)(*xs), C.c_void_p)  # noqa: E731
            t_d = tab([a.ptr + i * per * 8 for i in range(kc)])
            t_m = tab([mask.ptr + i * (per // 8) for i in range(kc)])
            t_n = C.cast((C.c_size_t * kc)(*([per] * kc)), C.c_void_p)
            has = C.c_int32()

            def sum_chunks():
                st = ctx.lib.ma_sum_chunks(ctx.handle, ord("l"), kc, t_d, t_n, t_m, None, None, slot.ptr, slot.ptr + 8)
                assert st == 0, st

            def consolidate_chunks():
                st = ctx.lib.ma_consolidate_column(ctx.handle, 8, kc, t_d, t_n, t_m, None, o.ptr, om.ptr, C.addressof(has))
                assert st == 0, st

            ctx.set_async(True)
            # five un-timed calls first: each of the context's four pinned staging buffers grows to the 3.9-MB table on its
            # first use (~1 ms of hipHostMalloc on the host each) — with two, two of those fell inside the events and the
            # 1.16-ms call read 1.35 ms (tools/probe_host_cost.py: the first calls of a process, 1.3-1.5 ms on the host)
            ms_s = _timed(ctx, sum_chunks, reps, 5)
            ms_k = _timed(ctx, consolidate_chunks, max(2, reps // 2), 2)
            ctx.set_async(False)
            ctx.synchronize()
            got = slot.download(np.int64, 2)
            one_s, one_c = ctx.sum("i64", a, rows_c, mask=mask)
            ok_s = (int(got[0]) & M64) == (one_s & M64) and int(got[1]) == one_c
            join_s, join_c = ctx.sum("i64", o, rows_c, mask=om)
            ok_k = (join_s & M64) == (one_s & M64) and join_c == one_c and bool(np.array_equal(
                o.download(np.int64, 4096, (rows_c // 2 - 2048) * 8), a.download(np.int64, 4096, (rows_c // 2 - 2048) * 8)))
            res["config5_supertable_8192_row_batches"] = {
                "batches": kc, "rows_per_batch": per,
                "i64_total_of_the_chunked_column_10pct_nulls": entry(ms_s, 8.125, rows_c, ok_s),
                "i64_consolidate_with_validity": entry(ms_k, 16.25, rows_c, ok_k)}
    except Exception as e:  # noqa: BLE001 — an extra leg must not cost the line
        res["config5_supertable_8192_row_batches"] = {"error": f"{type(e).__name__}: {e}"}

    def add_frac(d):
        for v in d.values():
            if isinstance(v, dict):
                if "gbps" in v and v.get("bytes_per_row", 0) > 8.2:  # read+write kernels only
                    v["frac_of_copy"] = v["gbps"] / copy_gbps
                add_frac(v)

    add_frac(res)
    res["rows"] = n
    res["copy_reference_gbps"] = copy_gbps
    for buf in (a, b, o, mask, om, slot, chunk_masks, recs, fin):
        buf.free()
    return res


def group_other_configs(group, ctxs, cols_i, cols_f, rows: int, reps: int, wait_ms: float = 0.0):
    """BASELINE configs 3, 4 and 5 of the multi-GPU kind, one process over N GPUs (after and outside the timed headline):
    config 3 = a SuperArray of N chunk pairs, one per GPU, added chunk by chunk with no exchange;
    config 4 = a 10^9-row i64 column with 10 % nulls, row-chunk partitioned over the N GPUs (strong scaling: N x fewer rows
    per GPU), masked scans + ONE exchange per step; config 5 = a SuperTable of N batches of `rows` rows (one per GPU),
    columns i64 + f64 with 10 % nulls, per-column reduce of BOTH columns with ONE exchange per step — a batch-sharded
    table consolidates logically, no column bytes move. Parity: the job's finals against the sum of the
    members' own synchronous scans of their chunks (checksum of checksums; f64 within 1 ULP-scale of the ordered fold)."""
    import numpy as np

    from minarrow_amd.parallel import fold_dd, row_chunks

    world = len(ctxs)
    M64 = (1 << 64) - 1
    res = {}

    def timed_steps(step):
        for _ in range(4):  # first touches of every GPU's buffers, communicator and fold kernel stay outside the clock
            step()
        group.synchronize_for(wait_ms)
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        group.synchronize_for(wait_ms)
        return (time.perf_counter() - t0) / reps * 1e3

    # ---- config 3: a SuperArray of N chunks of `rows` rows, one per GPU: out_i = a_i + b_i (i64; b = the f64 column's bit
    # patterns), fanned out by ma_group_route_super_array_broadcast — one launch per GPU, no exchange, output stays chunked.
    # Parity: linearity of the wrapping sum, sum(out) == sum(a) + sum(b) mod 2^64 over all GPUs, plus sampled windows.
    lens = [rows] * world
    outs = [c.alloc(rows * 8) for c in ctxs]
    ms = timed_steps(lambda: group.route_super_array_broadcast("l", 0, cols_i, cols_f, lens, lens, outs))
    sums = []
    for col in (cols_i, cols_f, outs):
        group.enqueue_sum("i64", 5, col, lens)
        group.exchange()
        group.synchronize_for(wait_ms)
        sums.append(group.result(5)[0] & M64)
    ok3 = sums[2] == (sums[0] + sums[1]) & M64
    for r in (0, world - 1):
        for start in (0, rows // 2 + 13, max(rows - 4096, 0)):
            k = min(4096, rows - start)
            a = cols_i[r].download(np.int64, k, start * 8)
            b = cols_f[r].download(np.int64, k, start * 8)
            ok3 = ok3 and bool(np.array_equal(outs[r].download(np.int64, k, start * 8), a + b))
    res["config3_i64_add_one_chunk_per_gpu"] = {
        "n_gpus": world, "rows_per_chunk": rows, "ms_per_step": ms, "grows_per_s": rows * world / ms / 1e6,
        "gbps": 24.0 * rows * world / ms / 1e6, "frac_of_peak_per_gpu": 24.0 * rows / ms / 1e6 / HBM_PEAK_GBPS,
        "parity": bool(ok3),
        "note": "route_super_array_broadcast over the group: chunk i on GPU i, no exchange, the result stays chunked"}
    for o in outs:
        o.free()

    # ---- config 4: 10^9 rows over N GPUs
    n = min(1_000_000_000, rows * world)
    chunks = row_chunks(n, world)
    lens = [hi - lo for lo, hi in chunks]
    masks = []
    for r, c in enumerate(ctxs):
        lo, hi = chunks[r]
        c.synth_iota("i64", cols_i[r], lens[r], lo)  # the chunk's slice of the global iota column
        m = c.alloc(lens[r] // 8 + 128)
        c.synth_validity(m, lens[r], seed=0xC0FFEE, first_index=lo, null_every=10)
        masks.append(m)
    zeros = [0] * world
    ms = timed_steps(lambda: (group.enqueue_sum("i64", 2, cols_i, lens, masks, zeros), group.exchange()))
    total, cnt, _, _ = group.result(2)
    part_s = part_c = 0
    for r, c in enumerate(ctxs):  # each member's own scan of its chunk, one at a time
        c.set_async(False)
        s_r, c_r = c.sum("i64", cols_i[r], lens[r], mask=masks[r])
        c.set_async(True)
        part_s, part_c = (part_s + s_r) & M64, part_c + c_r
    ok = (total & M64) == part_s and cnt == part_c and 0.09 < 1 - cnt / n < 0.11
    res["config4_i64_sum_10pct_nulls_row_chunks"] = {
        "n_gpus": world, "rows_total": n, "ms_per_step": ms, "grows_per_s": n / ms / 1e6, "gbps": 8.125 * n / ms / 1e6,
        "frac_of_peak_per_gpu": 8.125 * n / ms / 1e6 / world / HBM_PEAK_GBPS, "valid_count": cnt, "parity": bool(ok)}
    for m in masks:
        m.free()

    # ---- config 5: SuperTable of N batches of `rows` rows, one per GPU; columns i64 + f64, 10 % nulls
    lens = [rows] * world
    masks = []
    for r, c in enumerate(ctxs):
        c.synth_iota("i64", cols_i[r], rows, r)  # v[i] = i + batch (benches/consolidate.rs:37-58 pattern)
        c.synth_iota("f64", cols_f[r], rows, r)
        m = c.alloc(rows // 8 + 128)
        c.synth_validity(m, rows, seed=0xABC + r, null_every=10)
        masks.append(m)

    def table_step():  # both columns of every member's batch in ONE fused launch, then ONE exchange
        group.enqueue_sum_table([("l", 3, cols_i, lens, masks, zeros), ("g", 4, cols_f, lens, masks, zeros)])
        group.exchange()

    ms = timed_steps(table_step)
    isum, icnt, _, _ = group.result(3)
    _, _, fsum, fcnt = group.result(4)
    part_s = part_c = 0
    pairs = []
    for r, c in enumerate(ctxs):
        c.set_async(False)
        s_r, c_r = c.sum("i64", cols_i[r], rows, mask=masks[r])
        hi, lo, fc = c.sum_dd("f64", cols_f[r], rows, mask=masks[r])
        c.set_async(True)
        part_s, part_c = (part_s + s_r) & M64, part_c + c_r
        pairs.append((hi, lo))
        ok = ok and fc == c_r
    want_f = fold_dd(pairs)
    ok5 = (isum & M64) == part_s and icnt == part_c == fcnt and fsum == want_f
    res["config5_supertable_one_batch_per_gpu"] = {
        "n_gpus": world, "batches": world, "rows_per_batch": rows, "columns": ["i64", "f64"], "ms_per_step": ms,
        "grows_per_s": 2 * rows * world / ms / 1e6, "gbps": 2 * 8.125 * rows * world / ms / 1e6,
        "frac_of_peak_per_gpu": 2 * 8.125 * rows / ms / 1e6 / HBM_PEAK_GBPS, "valid_count": icnt, "parity": bool(ok5),
        "note": "per-column reduce of both columns with ONE exchange; the batch-sharded table is consolidated logically"}
    # ---- config 5, physically: the i64 column's N batches gathered onto GPU 0 (peer copies over xGMI on the owners' streams,
    # validity joined at bit granularity) — what a host asks for only when it needs one contiguous column. Its own
    # try: a failure here is reported in its entry and leaves the other legs standing.
    try:
        c0 = ctxs[0]
        whole = c0.alloc(rows * world * 8)
        wmask = c0.alloc(rows * world // 8 + 128)
        steps = max(1, min(reps, 3))
        for _ in range(1):
            group.consolidate_column(0, 8, cols_i, lens, whole, masks, zeros, wmask)
        group.synchronize_for(wait_ms)
        t0 = time.perf_counter()
        for _ in range(steps):
            group.consolidate_column(0, 8, cols_i, lens, whole, masks, zeros, wmask)
        group.synchronize_for(wait_ms)
        ms = (time.perf_counter() - t0) / steps * 1e3
        c0.set_async(False)
        s_w, c_w = c0.sum("i64", whole, rows * world, mask=wmask)
        c0.set_async(True)
        ok_g = (s_w & M64) == part_s and c_w == part_c  # the consolidated column reduces to the per-batch totals
        moved = 8.125 * rows * world
        res["config5_physical_consolidate_onto_gpu0"] = {
            "n_gpus": world, "rows_total": rows * world, "ms_per_step": ms, "gbps_into_gpu0": moved / ms / 1e6,
            "bytes_over_xgmi": 8.125 * rows * (world - 1), "parity": bool(ok_g),
            "note": "ma_group_consolidate_column: hipMemcpyPeerAsync per batch on its owner's stream + bit-granular validity join"}
        ok5 = ok5 and ok_g
        whole.free()
        wmask.free()
    except Exception as e:  # noqa: BLE001
        res["config5_physical_consolidate_onto_gpu0"] = {"error": f"{type(e).__name__}: {e}"}
    for m in masks:
        m.free()
    res["parity_ok"] = bool(ok3 and ok and ok5)
    return res


def ranks_other_configs(env, col_i, col_f, rows: int, reps: int):
    """The multi-GPU legs of configs 3-5 with one process per GPU: every rank scans its chunk / its batch, ONE exchange per
    step. `env` carries the run's plumbing — ctx, rank, world, make_ex(n_columns) -> records object, exchange(ex) (all-gather +
    device fold through whichever exchange the headline used), fence() (barrier + device drain), max_over_ranks(x),
    gather(obj), alloc(bytes)."""
    from minarrow_amd.parallel import fold_dd, row_chunks

    ctx, rank, world = env.ctx, env.rank, env.world
    M64 = (1 << 64) - 1

    def timed_steps(step):
        for _ in range(2):
            step()
        env.fence()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        env.fence()
        return env.max_over_ranks((time.perf_counter() - t0) / reps * 1e3)

    res = {}
    # ---- config 3: every rank adds its own chunk pair (i64; b = the f64 column's bit patterns); no exchange at all
    out3 = ctx.alloc(rows * 8)
    ms = timed_steps(lambda: ctx.apply("i64", col_i, col_f, 0, out3, rows, rows))
    ctx.set_async(False)
    s_a, s_b, s_o = (ctx.sum("i64", x, rows)[0] & M64 for x in (col_i, col_f, out3))
    ctx.set_async(True)
    ok3 = all(env.gather(s_o == (s_a + s_b) & M64))
    res["config3_i64_add_one_chunk_per_gpu"] = {
        "n_gpus": world, "rows_per_chunk": rows, "ms_per_step": ms, "grows_per_s": rows * world / ms / 1e6,
        "gbps": 24.0 * rows * world / ms / 1e6, "frac_of_peak_per_gpu": 24.0 * rows / ms / 1e6 / HBM_PEAK_GBPS,
        "parity": bool(ok3), "note": "one chunk pair per rank, no exchange, the result stays chunked"}
    out3.free()
    # ---- config 4
    n = min(1_000_000_000, rows * world)
    lo, hi = row_chunks(n, world)[rank]
    mine = hi - lo
    ctx.synth_iota("i64", col_i, mine, lo)
    mask = env.alloc(mine // 8 + 128)
    ctx.synth_validity(mask, mine, seed=0xC0FFEE, first_index=lo, null_every=10)
    ex = env.make_ex(1)

    def step4():
        ctx.sum_into("i64", col_i, mine, out_sum=ex.slot_ptr(0), out_count=ex.slot_ptr(1), mask=mask)
        env.exchange(ex)

    ms = timed_steps(step4)
    total, cnt, _, _ = ex.results()
    ctx.set_async(False)
    own = ctx.sum("i64", col_i, mine, mask=mask)
    ctx.set_async(True)
    parts = env.gather(own)
    ok = (total & M64) == (sum(p[0] for p in parts) & M64) and cnt == sum(p[1] for p in parts) and 0.09 < 1 - cnt / n < 0.11
    res["config4_i64_sum_10pct_nulls_row_chunks"] = {
        "n_gpus": world, "rows_total": n, "ms_per_step": ms, "grows_per_s": n / ms / 1e6, "gbps": 8.125 * n / ms / 1e6,
        "frac_of_peak_per_gpu": 8.125 * n / ms / 1e6 / world / HBM_PEAK_GBPS, "valid_count": cnt, "parity": bool(ok)}
    # ---- config 5: one batch of `rows` rows per rank, columns i64 + f64, BOTH scanned by one fused launch per step
    ctx.synth_iota("i64", col_i, rows, rank)
    ctx.synth_iota("f64", col_f, rows, rank)
    mask5 = env.alloc(rows // 8 + 128)
    ctx.synth_validity(mask5, rows, seed=0xABC + rank, null_every=10)
    ex2 = env.make_ex(2)

    def step5():
        ctx.sum_fused([("l", col_i, rows, ex2.slot_ptr(0, 0), mask5), ("g", col_f, rows, ex2.slot_ptr(2, 1), mask5)])
        env.exchange(ex2)

    ms = timed_steps(step5)
    (isum, icnt, _, _), (_, _, fsum, fcnt) = ex2.column_results()
    ctx.set_async(False)
    own_i = ctx.sum("i64", col_i, rows, mask=mask5)
    own_f = ctx.sum_dd("f64", col_f, rows, mask=mask5)
    ctx.set_async(True)
    parts = env.gather((own_i, own_f))
    want_f = fold_dd([(p[1][0], p[1][1]) for p in parts])
    ok5 = (isum & M64) == (sum(p[0][0] for p in parts) & M64) and icnt == sum(p[0][1] for p in parts) == fcnt and fsum == want_f
    res["config5_supertable_one_batch_per_gpu"] = {
        "n_gpus": world, "batches": world, "rows_per_batch": rows, "columns": ["i64", "f64"], "ms_per_step": ms,
        "grows_per_s": 2 * rows * world / ms / 1e6, "gbps": 2 * 8.125 * rows * world / ms / 1e6,
        "frac_of_peak_per_gpu": 2 * 8.125 * rows / ms / 1e6 / HBM_PEAK_GBPS, "valid_count": icnt, "parity": bool(ok5),
        "note": "both columns of the batch in ONE fused launch per rank, ONE exchange; the batch-sharded table is consolidated "
                "logically"}
    res["parity_ok"] = bool(ok3 and ok and ok5)
    return res
