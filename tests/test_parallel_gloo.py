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
