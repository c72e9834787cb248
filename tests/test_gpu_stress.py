"""The cross-workgroup hand-off of the single-launch reductions under load, inside the suite the driver runs.

ma_reduce.hip publishes one partial per workgroup with write-through stores and lets the LAST arrival fold them; grids
above 96 workgroups arrive on eight sharded counters whose last arrivals arrive at a top counter (ma_reduce.hip, the
"sharded two-level arrival ticket"). That transitive last-arriver detection is not one of the hand-off forms the MI355X
guide lists as measured, so it is exercised here for ~6 s on every GPU test run: random grids across 97..2048 (and a few
below 96: the single-counter form), four private contexts issuing asynchronously back to back from four host threads, a
fifth context streaming large elementwise kernels on the same device the whole time (uneven, shifting load on the CUs and
the L2s), dense and Bitmask-gated, i64 and f64, both publish forms (fence-free and MINARROW_HIP_FENCED_REDUCE's) — every
single result checked: against prefix sums computed on the host (exact: the f64 data are integers, so the double-double sum
has one right answer), a sample of them against the CPU oracle as well. A stale partial read by the folding workgroup shows
up as a wrong sum; a lost arrival as a launch that never completes (the test's timeout)."""
import os
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SECONDS = float(os.environ.get("MA_STRESS_SECONDS", "4"))  # a longer soak: MA_STRESS_SECONDS=300 pytest -s tests/test_gpu_stress.py
N_MAX = 3_000_000


def test_sharded_ticket_handoff_under_uneven_load(oracle):
    from minarrow_amd.host import Context, PinnedBuffer

    errors, counts, oracle_checks = [], [], []
    stop = threading.Event()

    def background():
        """Large a + b launches back to back on the same device: the reductions' workgroups share CUs and L2s with them."""
        c = Context(0)
        n = 1 << 25
        a, b, o = c.alloc(n * 8), c.alloc(n * 8), c.alloc(n * 8)
        c.synth_iota("i64", a, n, 0)
        c.synth_iota("i64", b, n, 7)
        c.set_async(True)
        k = 0
        while not stop.is_set():
            for _ in range(8):
                c.apply("i64", a, b, 0, o, n, n)
            c.synchronize()
            k += 8
            if k % 64 == 0:
                time.sleep(0.002)  # uneven on purpose: bursts and gaps
        c.set_async(False)
        got = o.download(np.int64, 4096, (n - 4096) * 8)
        want = np.arange(n - 4096, n, dtype=np.int64) * 2 + 7
        if not np.array_equal(got, want):
            errors.append("background a + b produced wrong rows")
        counts.append(0)
        c.close()

    def worker(seed):
        ctx = Context(0)
        rng = np.random.default_rng(seed)
        a = rng.integers(-(1 << 40), 1 << 40, size=N_MAX, dtype=np.int64)
        f = rng.integers(-(1 << 30), 1 << 30, size=N_MAX).astype(np.float64)  # integers: every partial sum is exact
        bits = rng.integers(0, 256, size=N_MAX // 8 + 64, dtype=np.uint8)
        valid = np.unpackbits(bits, bitorder="little")[:N_MAX].astype(bool)
        d, df, m = ctx.to_device(a, 64), ctx.to_device(f, 64), ctx.to_device(bits, 16)
        fi = f.astype(np.int64)
        pre = {("i64", False): np.concatenate([[0], np.cumsum(a)]), ("i64", True): np.concatenate([[0], np.cumsum(np.where(valid, a, 0))]),
               ("f64", False): np.concatenate([[0], np.cumsum(fi)]), ("f64", True): np.concatenate([[0], np.cumsum(np.where(valid, fi, 0))])}
        cnt = np.concatenate([[0], np.cumsum(valid)])
        slots = PinnedBuffer(64 * 64)
        vi, vf = slots.view(np.int64, 512), slots.view(np.float64, 512)
        ctx.set_async(True)
        ctx.set_variant(256 if seed % 4 == 3 else 0)  # one of the four contexts forces the chunk forms (and, on the tuning build, the fenced publish)
        t_end = time.time() + SECONDS
        done = 0
        while time.time() < t_end and not errors:
            batch = []
            for k in range(64):
                n = int(rng.integers(1, N_MAX))
                off = int(rng.integers(0, N_MAX - n + 1)) & ~1  # keeps the window's 16-byte alignment
                # the sharded form is taken above 96 workgroups: most grids there, a few on the single counter
                grid = int(rng.integers(97, 2049)) if rng.random() < 0.85 else int(rng.choice([1, 2, 7, 64, 95, 96]))
                tag = "i64" if rng.random() < 0.5 else "f64"
                masked = rng.random() < 0.5
                ctx.set_grid(grid)
                base = slots.ptr + k * 64
                if tag == "i64":
                    ctx.sum_into("i64", d.offset(off * 8), n, out_sum=base, out_count=base + 8,
                                 mask=m if masked else None, mask_bit_offset=off)
                else:
                    ctx.sum_into("f64", df.offset(off * 8), n, out_sum=base, out_count=base + 8,
                                 mask=m if masked else None, mask_bit_offset=off)
                batch.append((k, n, off, grid, tag, masked))
            ctx.synchronize()
            for k, n, off, grid, tag, masked in batch:
                want = int(pre[(tag, masked)][off + n] - pre[(tag, masked)][off])
                want_n = int(cnt[off + n] - cnt[off]) if masked else n
                got = int(vi[k * 8]) if tag == "i64" else vf[k * 8]
                if got != want or int(vi[k * 8 + 1]) != want_n:
                    errors.append(f"seed {seed}: {tag} masked={masked} n={n} off={off} grid={grid}: got {got}, {int(vi[k * 8 + 1])} "
                                  f"want {want}, {want_n}")
            k, n, off, grid, tag, masked = batch[0]  # one per batch against the CPU oracle too
            if tag == "i64":
                ref = oracle.masked_sum(a[off:off + n], np.packbits(valid[off:off + n], bitorder="little"), 0) if masked \
                    else (oracle.sum_scalar(a[off:off + n]), n)
                if (int(vi[k * 8]), int(vi[k * 8 + 1])) != ref:
                    errors.append(f"seed {seed}: oracle disagrees: n={n} off={off} grid={grid} masked={masked}")
                oracle_checks.append(1)
            done += len(batch)
        ctx.set_async(False)
        ctx.set_grid(0)
        ctx.set_variant(0)
        counts.append(done)
        ctx.close()

    bg = threading.Thread(target=background)
    threads = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
    bg.start()
    [t.start() for t in threads]
    deadline = time.time() + SECONDS + 120
    while any(t.is_alive() for t in threads) and time.time() < deadline:
        threads[0].join(timeout=60)  # a soak (MA_STRESS_SECONDS >> 60) says it is alive once a minute (visible with -s)
        if SECONDS > 60:
            print(f"[stress] {int(time.time() - (deadline - SECONDS - 120))} s, no errors so far" if not errors else f"[stress] errors: {errors[:2]}",
                  flush=True)
    [t.join(timeout=1) for t in threads]
    stop.set()
    bg.join(timeout=60)
    assert not any(t.is_alive() for t in threads) and not bg.is_alive(), "a launch never completed (lost arrival?)"
    assert not errors, errors[:5]
    assert sum(counts) > 2000 and len(oracle_checks) > 10, (counts, len(oracle_checks))
    print(f"{sum(counts)} reductions checked ({len(oracle_checks)} also against the oracle), 0 errors")


def test_kernels_that_share_the_contexts_ticket_alternate(ctx, oracle):
    """Sums, fused sums and bitmap scans all end with the same partial + ticket hand-off on the context's one set of arrival
    counters (round 4: the scans joined). Interleaved on one stream in every order, small and large grids (one ticket / the
    sharded form), each launch must find the counters re-armed by its predecessor: 400 rounds, every result checked."""
    rng = np.random.default_rng(77)
    n = 3_000_017
    a = rng.integers(-(1 << 40), 1 << 40, size=n, dtype=np.int64)
    f = rng.standard_normal(n)
    bits = rng.integers(0, 256, size=n // 8 + 64, dtype=np.uint8)
    da, df, dm = ctx.to_device(a, 64), ctx.to_device(f, 64), ctx.to_device(bits, 16)
    dm2 = ctx.to_device(bits, 16)
    rec = ctx.alloc(256)
    valid = np.unpackbits(bits, bitorder="little")
    csum = np.concatenate([[0], np.cumsum(a, dtype=np.int64)])
    cpop = np.concatenate([[0], np.cumsum(valid, dtype=np.int64)])
    for it in range(400):
        k = int(rng.integers(1, n))           # rows of this round: from one workgroup's worth to the whole column
        if it % 5 == 0:
            k = int(rng.integers(1, 5000))
        order = rng.permutation(4)
        for op in order:
            if op == 0:
                s, c = ctx.sum("i64", da, k)
                assert (s, c) == (int(csum[k]), k)
            elif op == 1:
                assert ctx.popcount_mask(dm, 0, k) == int(cpop[k])
            elif op == 2:
                ctx.sum_fused([("l", da, k, rec.ptr), ("g", df, k, rec.ptr + 16)])
                w = rec.download(np.uint64, 8)
                assert int(w[0]) == int(csum[k]) & ((1 << 64) - 1) and int(w[1]) == k and int(w[4]) == k
            else:
                k8 = k & ~7  # all_eq addresses word-aligned windows
                assert ctx.mask_all("all_eq", dm, 0, dm2, 0, max(k8, 64))
                s, c = ctx.sum("i64", da, k, mask=dm)
                assert c == int(cpop[k])
