"""Row-chunk partition across the GPUs of one node + the scalar exchange that ends a reduction.

This is what the reference's `parallel_proc` / Rayon path becomes (benches/benchmark_parallel_simd.rs:81-98:
`slice.par_chunks(1 << 20).map(simd_sum).sum()`): one process per GPU, every rank scans its own contiguous row
range with the HIP kernels, and the per-rank scalars {sum, valid_count} (+ the double-double low word for
floats) are exchanged with ONE collective over RCCL/xGMI (`torch.distributed`, backend "nccl" on GPUs, "gloo" in
the CPU tests). The payload is 64 bytes per rank and column, so the step is latency-bound; xGMI bandwidth is
irrelevant. A SuperTable is sharded by batch (`batch_ranges`): its per-column reduce is the same exchange with one
record per (local batch, column); only an explicitly requested contiguous copy moves column bytes between GPUs
(`gather_consolidated`).

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
    [0] integer sum   [1] integer valid count   [2] f64 hi bits   [3] f64 lo bits   [4] float valid count.

    One record per rank serves a single column (BASELINE configs 2 and 4). A rank-sharded SuperTable (config 5:
    `batch_ranges` gives every rank a contiguous run of batches) reduces `n_columns` columns at once: a rank holds
    `slots_per_rank` x `n_columns` records — one per (local batch, column), unused slots stay zero and fold as
    nothing — and ONE collective still moves them all. Column c is folded over (rank, slot) in that order, i.e. in
    batch order."""

    RECORD = 8  # int64 slots per record (5 used; padded to 64 bytes)

    def __init__(self, device, group=None, n_columns: int = 1, slots_per_rank: int = 1):
        import torch
        import torch.distributed as dist

        self.torch, self.dist = torch, dist
        self.group = group
        self.device = device
        self.n_columns, self.slots = int(n_columns), int(slots_per_rank)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        per_rank = self.RECORD * self.n_columns * self.slots
        self.local = torch.zeros(per_rank, dtype=torch.int64, device=device)
        self.gathered = torch.zeros(per_rank * self.world, dtype=torch.int64, device=device)
        self.final = torch.zeros(4 * self.n_columns, dtype=torch.int64, device=device)  # written by fold_on_device
        self._folded_on_device = False

    def slot_ptr(self, index: int, column: int = 0, slot: int = 0) -> int:
        """Device address of word `index` of this rank's record for (local batch `slot`, `column`) — what the
        kernels write into."""
        if not (0 <= column < self.n_columns and 0 <= slot < self.slots and 0 <= index < self.RECORD):
            raise IndexError((index, column, slot))
        return int(self.local.data_ptr()) + 8 * ((slot * self.n_columns + column) * self.RECORD + index)

    def exchange(self) -> None:
        """Enqueue the collective on the current stream (async w.r.t. the host)."""
        if not self.dist.is_initialized():
            self.gathered = self.local  # one rank, no process group: the local records ARE the gathered ones
        elif self.local.is_cuda and self.dist.get_backend(self.group) == "gloo":
            # rehearsal of the multi-rank path on a box without RCCL peers (several ranks sharing one GPU): the records
            # take the detour over host memory; everything around the collective is the production path
            mine = self.local.cpu()
            everyone = self.torch.empty(self.gathered.numel(), dtype=self.torch.int64)
            self.dist.all_gather_into_tensor(everyone, mine, group=self.group)
            self.gathered.copy_(everyone)
        else:
            self.dist.all_gather_into_tensor(self.gathered, self.local, group=self.group)

    def fold_on_device(self, ctx) -> None:
        """Enqueue the rank-ordered fold of the gathered records on `ctx`'s stream (ma_fold_sum_records): the job's
        final scalars are then produced on the GPU, inside the step, identically on every rank."""
        stride = self.n_columns * self.RECORD
        for c in range(self.n_columns):
            ctx.fold_sum_records(int(self.gathered.data_ptr()) + 8 * c * self.RECORD, self.world * self.slots, stride,
                                 int(self.final.data_ptr()) + 32 * c)
        self._folded_on_device = True

    def set_local(self, int_sum: int = 0, int_count: int = 0, hi: float = 0.0, lo: float = 0.0, f_count: int = 0,
                  column: int = 0, slot: int = 0) -> None:
        """Host-side fill of one local record (used by the CPU tests; on GPUs the kernels write it)."""
        rec = np.zeros(self.RECORD, dtype=np.int64)
        rec[0] = to_signed(int_sum)
        rec[1] = int_count
        rec[2:4] = np.array([hi, lo], dtype=np.float64).view(np.int64)
        rec[4] = f_count
        at = (slot * self.n_columns + column) * self.RECORD
        self.local[at:at + self.RECORD].copy_(self.torch.from_numpy(rec))

    def column_results(self):
        """Per column (int_sum as unsigned 64-bit pattern, int_count, f64_sum, f64_count): the device-folded finals
        when `fold_on_device` ran, else a host fold of every record in (rank, slot) order (the gloo / CPU path)."""
        if self._folded_on_device:
            f = self.final.cpu().numpy().reshape(self.n_columns, 4)
            return [(int(r[0]) & _MASK64, int(r[1]), float(r[2:3].view(np.float64)[0]), int(r[3])) for r in f]
        g = self.gathered.cpu().numpy().reshape(self.world * self.slots, self.n_columns, self.RECORD)
        out = []
        for c in range(self.n_columns):
            col = g[:, c, :]
            pairs = [tuple(col[r, 2:4].view(np.float64).tolist()) for r in range(col.shape[0])]
            out.append((fold_int([int(v) for v in col[:, 0]]), int(col[:, 1].sum()), fold_dd(pairs), int(col[:, 4].sum())))
        return out

    def results(self):
        """`column_results()` of the first (or only) column."""
        return self.column_results()[0]


def batch_ranges(n_batches: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous [first, last) batch-index ranges, one per rank, sizes differing by at most one. Rank order is
    batch order, so concatenating the ranks' pieces in rank order is the reference's batch-ordered
    `consolidate_concat` (src/structs/chunked/super_table.rs:692-722)."""
    if world < 1 or n_batches < 0:
        raise ValueError((n_batches, world))
    return [((n_batches * r) // world, (n_batches * (r + 1)) // world) for r in range(world)]


def _gather_pieces(dist, local, out, lens: Sequence[int], group) -> None:
    """out = the ranks' 1-D pieces back to back (piece r has lens[r] elements). Equal pieces: one in-place
    all-gather; ragged pieces: each rank's view of `out` is broadcast from its owner (valid for RCCL and gloo)."""
    world = len(lens)
    if local.numel() != lens[dist.get_rank(group)]:
        raise ValueError("local piece does not have the advertised length")
    if out.numel() != sum(lens):
        raise ValueError("output does not hold the sum of the pieces")
    if len(set(lens)) == 1:
        dist.all_gather_into_tensor(out, local, group=group)
        return
    at = 0
    for r in range(world):
        view = out[at:at + lens[r]]
        at += lens[r]
        if lens[r] == 0:
            continue
        if r == dist.get_rank(group):
            view.copy_(local)
        dist.broadcast(view, src=dist.get_global_rank(group, r) if group is not None else r, group=group)


def gather_consolidated(local_values, rows_per_rank: Sequence[int], out_values, local_validity=None,
                        out_validity=None, ctx=None, group=None) -> bool:
    """Physical consolidation of a rank-sharded column onto EVERY rank: the all-gather over xGMI that SURVEY.md
    §8(e) prices at 7 x 8 GB per 8-byte column of config 5 — only for callers that need one contiguous buffer; a
    per-column reduce never does (ScalarExchange).

    local_values: this rank's already locally consolidated piece (1-D tensor, rows_per_rank[rank] elements);
    out_values: 1-D tensor of sum(rows_per_rank) elements of the same dtype, filled on every rank.
    local_validity / out_validity: uint8 tensors holding Arrow bitmaps (bit 0 = the piece's row 0), or None. A rank
    whose piece has no mask contributes all-valid bits when any other rank has one (`extend_null_mask`,
    src/traits/consolidate.rs:80-105). Pieces whose interior row counts are multiples of 8 join on byte
    boundaries and are gathered in place; any other split is joined at bit granularity on the GPU
    (ma_consolidate_boolean_column), which needs `ctx`. out_validity must hold 8 * ceil(total rows / 64) bytes.
    Returns True when out_validity was written."""
    import torch
    import torch.distributed as dist

    rows = [int(r) for r in rows_per_rank]
    if not dist.is_initialized():
        if len(rows) != 1:
            raise ValueError("rows_per_rank names several ranks but torch.distributed is not initialised")
        out_values.copy_(local_values)
        if local_validity is None or out_validity is None:
            return False
        nbytes = (rows[0] + 7) // 8
        out_validity[:nbytes].copy_(local_validity[:nbytes])
        return True
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if len(rows) != world:
        raise ValueError(f"rows_per_rank has {len(rows)} entries for {world} ranks")
    _gather_pieces(dist, local_values, out_values, rows, group)

    flag = torch.tensor([1 if local_validity is not None else 0], dtype=torch.int32, device=out_values.device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if int(flag.item()) == 0:
        return False
    if out_validity is None:
        raise ValueError("a rank holds a validity bitmap but out_validity is None")
    total = sum(rows)
    piece_bytes = [(r + 7) // 8 for r in rows]
    mine = local_validity[:piece_bytes[rank]] if local_validity is not None else \
        torch.full((piece_bytes[rank],), 0xFF, dtype=torch.uint8, device=out_values.device)
    byte_joined = all(r % 8 == 0 for r in rows[:-1])
    if byte_joined:
        # trailing bits of the last byte: zero, as Bitmask keeps them (src/structs/bitmask.rs:83-90)
        if rank == world - 1 and rows[-1] % 8 and local_validity is None:
            mine[-1] = (1 << (rows[-1] % 8)) - 1
        _gather_pieces(dist, mine, out_validity[:(total + 7) // 8], piece_bytes, group)
        return True
    if ctx is None:
        raise ValueError("pieces that do not end on byte boundaries are joined on the GPU: pass ctx")
    width = max(piece_bytes) + 8  # every staged piece is padded so that whole words can be read
    staged = torch.zeros(world * width, dtype=torch.uint8, device=out_values.device)
    padded = torch.zeros(width, dtype=torch.uint8, device=out_values.device)
    padded[:piece_bytes[rank]].copy_(mine)
    dist.all_gather_into_tensor(staged, padded, group=group)
    join_bit_pieces(ctx, staged, width, rows, out_validity)
    return True


def join_bit_pieces(ctx, staged, width: int, rows: Sequence[int], out_validity) -> None:
    """Joins the gathered validity pieces (piece r = rows[r] bits from byte r * width of `staged`) into one bitmap at
    bit granularity on the GPU — `Bitmask::extend_from_bitmask_range` (src/structs/bitmask.rs:520-592) for all pieces
    in one launch. out_validity must hold 8 * ceil(sum(rows) / 64) bytes."""
    pieces = [(int(staged.data_ptr()) + r * width, 0, int(rows[r])) for r in range(len(rows))]
    ctx.consolidate_boolean_column(pieces, out_validity)


def mean_from(sum_value, count: int) -> float:
    return float(sum_value) / count if count else float("nan")


def partition_description(n_rows: int, world: int) -> Optional[str]:
    chunks = row_chunks(n_rows, world)
    return ", ".join(f"rank {r}: [{a}, {b})" for r, (a, b) in enumerate(chunks))
