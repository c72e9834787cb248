"""Row-chunk partition across the GPUs of one node + the scalar exchange that ends a reduction.

This is what the reference's `parallel_proc` / Rayon path becomes (benches/benchmark_parallel_simd.rs:81-98:
`slice.par_chunks(1 << 20).map(simd_sum).sum()`): one process per GPU, every rank scans its own contiguous row
range with the HIP kernels, and the per-rank scalars {sum, valid_count} (+ the double-double low word for
floats) are exchanged with ONE collective over RCCL/xGMI (`torch.distributed`, backend "nccl" on GPUs, "gloo" in
the CPU tests). The payload is 40 bytes per rank, so the step is latency-bound; xGMI bandwidth is irrelevant.

Elementwise / broadcast / bitmask ops need no collective at all: rows are independent, the output stays sharded
(a SuperArray of `world` chunks in the reference's terms, src/kernels/broadcast/super_array.rs:180-251).

The fold of the gathered partials is done in rank order on every rank, so all ranks hold bit-identical results
and an f64 sum stays within 1 ULP of the exactly rounded total.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

_MASK64 = (1 << 64) - 1


def row_chunks(n_rows: int, world: int, align: int = 64) -> List[Tuple[int, int]]:
    """Contiguous [start, stop) row ranges, one per rank. Interior boundaries are multiples of `align` rows
    (64 = one validity word), so a chunk's Bitmask window never splits a u64 word between two GPUs.
    Sizes differ by at most `align` rows; trailing ranks may be empty when n_rows < world * align."""
    if world < 1:
        raise ValueError("world must be >= 1")
    if n_rows < 0:
        raise ValueError("n_rows must be >= 0")
    units = (n_rows + align - 1) // align  # number of `align`-row units, the last possibly partial
    bounds = [min(n_rows, ((units * r) // world) * align) for r in range(world)] + [n_rows]
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def two_sum(a: float, b: float) -> Tuple[float, float]:
    """Knuth's error-free transformation: s + e == a + b exactly."""
    s = a + b
    bp = s - a
    return s, (a - (s - bp)) + (b - bp)


def fold_int(partials: Sequence[int]) -> int:
    """Wrapping (two's complement, 64-bit) sum of per-rank integer sums, returned as an unsigned pattern."""
    total = 0
    for p in partials:
        total = (total + (int(p) & _MASK64)) & _MASK64
    return total


def to_signed(x: int) -> int:
    x &= _MASK64
    return x - (1 << 64) if x >> 63 else x


def fold_dd(pairs: Sequence[Tuple[float, float]]) -> float:
    """Rounded sum of per-rank double-double partials (hi, lo), folded in the given (rank) order."""
    hi, lo = 0.0, 0.0
    for h, l in pairs:
        hi, e = two_sum(hi, h)
        lo += e + l
    if np.isfinite(hi) and np.isfinite(lo):
        return hi + lo
    return hi


class ScalarExchange:
    """All-gather of each rank's reduction scalars. Layout of a record (5 x int64 bit patterns):
    [0] integer sum   [1] integer valid count   [2] f64 hi bits   [3] f64 lo bits   [4] float valid count."""

    RECORD = 8  # int64 slots per rank (5 used; padded to 64 bytes)

    def __init__(self, device, group=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.group = group
        self.device = device
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.local = torch.zeros(self.RECORD, dtype=torch.int64, device=device)
        self.gathered = torch.zeros(self.RECORD * self.world, dtype=torch.int64, device=device)
        self.final = torch.zeros(4, dtype=torch.int64, device=device)  # written by fold_on_device
        self._folded_on_device = False

    def slot_ptr(self, index: int) -> int:
        """Device address of slot `index` of this rank's record (what the kernels write into)."""
        return int(self.local.data_ptr()) + 8 * index

    def exchange(self) -> None:
        """Enqueue the collective on the current stream (async w.r.t. the host)."""
        if self.dist.is_initialized():
            self.dist.all_gather_into_tensor(self.gathered, self.local, group=self.group)
        else:
            self.gathered.copy_(self.local)

    def fold_on_device(self, ctx) -> None:
        """Enqueue the rank-ordered fold of the gathered records on `ctx`'s stream (ma_fold_sum_records): the job's
        final scalars are then produced on the GPU, inside the step, identically on every rank."""
        ctx.fold_sum_records(int(self.gathered.data_ptr()), self.world, self.RECORD, int(self.final.data_ptr()))
        self._folded_on_device = True

    def set_local(self, int_sum: int = 0, int_count: int = 0, hi: float = 0.0, lo: float = 0.0, f_count: int = 0) -> None:
        """Host-side fill of the local record (used by the CPU tests; on GPUs the kernels write it)."""
        rec = np.zeros(self.RECORD, dtype=np.int64)
        rec[0] = to_signed(int_sum)
        rec[1] = int_count
        rec[2:4] = np.array([hi, lo], dtype=np.float64).view(np.int64)
        rec[4] = f_count
        self.local.copy_(self.torch.from_numpy(rec))

    def results(self):
        """(int_sum as unsigned 64-bit pattern, int_count, f64_sum, f64_count): the device-folded finals when
        `fold_on_device` ran, else a host fold of every rank's record in rank order (the gloo / CPU path)."""
        if self._folded_on_device:
            f = self.final.cpu().numpy()
            return int(f[0]) & _MASK64, int(f[1]), float(f[2:3].view(np.float64)[0]), int(f[3])
        g = self.gathered.cpu().numpy().reshape(self.world, self.RECORD)
        int_sum = fold_int([int(v) for v in g[:, 0]])
        pairs = [tuple(g[r, 2:4].view(np.float64).tolist()) for r in range(self.world)]
        return int_sum, int(g[:, 1].sum()), fold_dd(pairs), int(g[:, 4].sum())


def mean_from(sum_value, count: int) -> float:
    return float(sum_value) / count if count else float("nan")


def partition_description(n_rows: int, world: int) -> Optional[str]:
    chunks = row_chunks(n_rows, world)
    return ", ".join(f"rank {r}: [{a}, {b})" for r, (a, b) in enumerate(chunks))
