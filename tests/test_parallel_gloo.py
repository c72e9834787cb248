"""CPU tests of the N > 1 path: row-chunk partition + the scalar exchange, with world_size 2 and 3 over gloo.

The per-rank "kernel" here is the CPU oracle (this is a test: it stands in for the HIP sum so the host logic
that surrounds it — partition boundaries, Bitmask window offsets, the all-gather, the rank-ordered fold — runs
without a GPU). bench.py drives exactly the same ScalarExchange with the HIP kernels writing the local record.
"""
import math
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from minarrow_amd.parallel import fold_dd, fold_int, row_chunks, to_signed, two_sum  # noqa: E402


def test_row_chunks_cover_and_align():
    for n in (0, 1, 63, 64, 65, 1000, 1_000_000_000, 999_999_937):
        for world in (1, 2, 3, 4, 8):
            chunks = row_chunks(n, world)
            assert len(chunks) == world and chunks[0][0] == 0 and chunks[-1][1] == n
            for (a, b), (c, d) in zip(chunks, chunks[1:]):
                assert b == c and a <= b
            for a, b in chunks[:-1]:
                assert b % 64 == 0 or b == n  # interior boundaries never split a validity word
            sizes = [b - a for a, b in chunks]
            if n >= world * 64:
                assert max(sizes) - min(sizes) <= 64
    assert row_chunks(1_000_000_000, 8) == [(i * 125_000_000, (i + 1) * 125_000_000) for i in range(8)]


def test_folds():
    assert fold_int([2 ** 63, 2 ** 63, 5]) == 5
    assert to_signed(fold_int([2 ** 63 - 1, 1])) == -(2 ** 63)
    s, e = two_sum(1e16, 1.0)
    assert s == 1e16 and e == 1.0
    assert fold_dd([(1e16, 1.0), (-1e16, 0.25)]) == 1.25
    assert fold_dd([(float("inf"), float("nan")), (1.0, 0.0)]) == float("inf")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, seed, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist

    from minarrow_amd.parallel import ScalarExchange, row_chunks
    from oracle import oracle

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(seed)  # same data on every rank; each scans only its chunk
        ints = rng.integers(-(1 << 62), 1 << 62, size=n, dtype=np.int64)
        flts = rng.standard_normal(n) * 10.0 ** rng.integers(0, 12, size=n)
        bits = rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8)
        a, b = row_chunks(n, world)[rank]
        valid = np.unpackbits(bits, bitorder="little")[a:b].astype(bool)
        # the rank's own kernel call: data window + un-windowed validity buffer + bit offset = first row
        isum, icnt = oracle.masked_sum(np.ascontiguousarray(ints[a:b]), bits, a)
        sel = flts[a:b][valid]
        hi = math.fsum(sel.tolist())
        lo = math.fsum(sel.tolist() + [-hi])
        ex = ScalarExchange("cpu")
        ex.set_local(isum, icnt, hi, lo, int(valid.sum()))
        ex.exchange()
        q.put((rank, ex.results()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_chunk_sum_over_gloo(world):
    import torch.multiprocessing as mp

    n, seed = 200_003, 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(seed)
    ints = rng.integers(-(1 << 62), 1 << 62, size=n, dtype=np.int64)
    flts = rng.standard_normal(n) * 10.0 ** rng.integers(0, 12, size=n)
    bits = rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8)
    valid = np.unpackbits(bits, bitorder="little")[:n].astype(bool)
    want_int = int(ints[valid].astype(object).sum()) & ((1 << 64) - 1)
    want_f = math.fsum(flts[valid].tolist())
    assert len({r[1] for r in results}) == 1  # every rank folds to the same bits
    int_sum, int_cnt, f_sum, f_cnt = results[0][1]
    assert int_sum == want_int and int_cnt == int(valid.sum()) == f_cnt
    assert abs(f_sum - want_f) <= math.ulp(want_f)


# ---------------------------------------------------------------------------------------------------------------
# Rank-sharded SuperTable (BASELINE config 5): batches dealt to ranks in order, per-column reduce through one
# exchange, physical consolidation only on request.
# ---------------------------------------------------------------------------------------------------------------

def test_batch_ranges_are_contiguous_and_ordered():
    from minarrow_amd.parallel import batch_ranges

    for n in (0, 1, 5, 8, 17):
        for world in (1, 2, 3, 8):
            r = batch_ranges(n, world)
            assert r[0][0] == 0 and r[-1][1] == n and all(b == c for (_, b), (c, _) in zip(r, r[1:]))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    assert batch_ranges(8, 8) == [(i, i + 1) for i in range(8)]


def _table(seed, n_batches, ragged):
    """SuperTable stand-in: per batch an i64 column, an f64 column and their validity bitmaps (None = no mask)."""
    rng = np.random.default_rng(seed)
    batches = []
    for b in range(n_batches):
        rows = int(rng.integers(1, 40)) * 8 + (3 if ragged and b == n_batches - 1 else 0)
        ints = rng.integers(-(1 << 62), 1 << 62, size=rows, dtype=np.int64)
        flts = rng.standard_normal(rows) * 10.0 ** rng.integers(0, 12, size=rows)
        mask = None if b % 3 == 1 else np.packbits(rng.random(rows) > 0.1, bitorder="little")
        batches.append((ints, flts, mask))
    return batches


def _valid(mask, rows):
    return np.ones(rows, dtype=bool) if mask is None else np.unpackbits(mask, bitorder="little")[:rows].astype(bool)


def _table_worker(rank, world, port, seed, n_batches, ragged, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist

    from minarrow_amd.parallel import ScalarExchange, batch_ranges, gather_consolidated

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batches = _table(seed, n_batches, ragged)
        ranges = batch_ranges(n_batches, world)
        lo, hi = ranges[rank]
        slots = max(b - a for a, b in ranges)
        ex = ScalarExchange("cpu", n_columns=2, slots_per_rank=slots)
        for slot, (ints, flts, mask) in enumerate(batches[lo:hi]):
            v = _valid(mask, len(ints))
            sel = flts[v]
            h = math.fsum(sel.tolist())
            ex.set_local(int_sum=int(ints[v].astype(object).sum()), int_count=int(v.sum()), column=0, slot=slot)
            ex.set_local(hi=h, lo=math.fsum(sel.tolist() + [-h]), f_count=int(v.sum()), column=1, slot=slot)
        ex.exchange()
        # physical consolidation of the i64 column: every rank first joins its own batches (the local
        # ma_consolidate_column step; numpy stands in for it here), then the pieces are gathered in rank order
        rows = [sum(len(b[0]) for b in batches[a:z]) for a, z in ranges]
        own = batches[lo:hi]
        piece = np.concatenate([b[0] for b in own]) if own else np.zeros(0, dtype=np.int64)
        any_mask = any(b[2] is not None for b in own)
        piece_bits = np.packbits(np.concatenate([_valid(b[2], len(b[0])) for b in own]), bitorder="little") \
            if any_mask else None
        out = torch.zeros(sum(rows), dtype=torch.int64)
        out_bits = torch.zeros((sum(rows) + 63) // 64 * 8, dtype=torch.uint8)
        wrote = gather_consolidated(torch.from_numpy(piece), rows, out,
                                    torch.from_numpy(piece_bits) if piece_bits is not None else None, out_bits)
        q.put((rank, ex.column_results(), out.numpy().tobytes(), out_bits.numpy().tobytes(), wrote))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_batches,ragged", [(2, 5, False), (3, 8, True), (2, 1, True)])
def test_sharded_super_table_over_gloo(world, n_batches, ragged):
    import torch.multiprocessing as mp

    seed = 23
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_table_worker, args=(r, world, port, seed, n_batches, ragged, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    batches = _table(seed, n_batches, ragged)
    ints = np.concatenate([b[0] for b in batches])
    flts = np.concatenate([b[1] for b in batches])
    valid = np.concatenate([_valid(b[2], len(b[0])) for b in batches])
    want_bits = np.zeros((len(ints) + 63) // 64 * 8, dtype=np.uint8)
    packed = np.packbits(valid, bitorder="little")
    want_bits[:len(packed)] = packed
    for _, cols, data, bits, wrote in results:
        (isum, icnt, _, _), (_, _, fsum, fcnt) = cols
        assert isum == int(ints[valid].astype(object).sum()) & ((1 << 64) - 1) and icnt == int(valid.sum()) == fcnt
        exact = math.fsum(flts[valid].tolist())
        assert abs(fsum - exact) <= math.ulp(exact)
        assert data == ints.tobytes()  # == consolidate_concat of the batches in order
        assert wrote and bits == want_bits.tobytes()
    assert len({r[1][1][2] for r in results}) == 1  # bit-identical f64 finals on every rank


def test_gather_consolidated_without_a_process_group():
    """One rank, no torch.distributed: the piece is the column; several advertised ranks are an error, not a hang."""
    import torch

    from minarrow_amd.parallel import gather_consolidated

    vals = torch.arange(37, dtype=torch.int64)
    bits = torch.from_numpy(np.packbits(np.arange(37) % 2 == 0, bitorder="little"))
    out, out_bits = torch.zeros(37, dtype=torch.int64), torch.zeros(8, dtype=torch.uint8)
    assert gather_consolidated(vals, [37], out, bits, out_bits)
    assert torch.equal(out, vals) and torch.equal(out_bits[:5], bits)
    assert not gather_consolidated(vals, [37], out)  # no validity anywhere
    with pytest.raises(ValueError):
        gather_consolidated(vals, [20, 17], out)
