"""bench.py is what the driver runs: its launch modes and its configs-3-5 leg are exercised end to end at a small size
(2^24 rows), through the same code paths as the 10^9-row run. One JSON line on stdout, parity checked inside."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu
SMALL = ["--rows", str(1 << 24), "--steps", "3", "--warmup", "1"]


def run(cmd, extra_env=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]  # the contract: ONE JSON line
    return json.loads(lines[0])


def test_single_gpu_line_with_other_configs_and_cpu_baseline():
    out = run([sys.executable, "bench.py", *SMALL, "--cpu-rows", str(1 << 22), "--cpu-seconds", "1", "--other-reps", "2"])
    assert out["parity_ok"] and out["n_gpus"] == 1 and out["unit"] == "Grows/s" and out["scaling"] == "strong"
    assert out["config"]["rows_total_per_column"] == out["config"]["rows_per_gpu_per_column"] == 1 << 24
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(out["roofline"])
    assert 0 < out["roofline"]["frac"] <= 1.0
    oc = out["other_configs"]
    assert oc["parity_ok"] is True and oc["rows"] == 1 << 24
    for key in ("add_array_array", "multiply_array_array", "add_array_scalar", "multiply_array_scalar"):
        k = oc["config3_f64_elementwise"][key]
        assert k["parity"] is True and k["frac_of_peak"] <= 1.0 and k["frac_of_copy"] > 0
    assert oc["config4_i64_sum_10pct_nulls"]["parity"] is True
    assert 0.09 < oc["config4_i64_sum_10pct_nulls"]["null_fraction"] < 0.11
    for tag in ("i64", "f64"):
        for leg in ("reduce_per_batch", "consolidate", "reduce_consolidated"):
            assert oc["config5_supertable_8_batches"][tag][leg]["parity"] is True
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    # `value` is what the box sustains (the median of the pool sized to the CPU quota, `cores` threads); the un-throttled burst
    # of a larger pool rides along under its own name
    assert cb["cores"] == cb["pool_threads"] and 0 < cb["value"] <= cb["value_burst"] * 1.001 and cb["cores_burst"] >= cb["cores"]
    assert cb["cgroup_cpu_quota"] is None or cb["cores"] <= cb["cgroup_cpu_quota"]
    # the headline is the product's own form: no torch in the process, /opt/rocm's HIP runtime, timing marks for the kernels
    assert out["config"]["host"] == "torch-free" and "torch" not in out["config"]["hip_runtime"]
    assert set(out["kernels"]) == {"sum_i64", "sum_f64"} and out["roofline"]["kernel"].startswith("ma::sum_kernel")
    assert "torch_hosted" not in out  # retired in round 6: the torch-free host had matched it for three rounds
    pl = out["pipelined"]  # the same job as a pipeline of fused steps on two streams (ma_scan_lanes_*), labelled, never `value`
    assert pl["parity_ok"] and pl["value"] > 0 and 0 < pl["frac_of_peak"] <= 1.0 and "ma_scan_lanes" in pl["step"]


def test_single_gpu_fused_step():
    fused = run([sys.executable, "bench.py", *SMALL, "--no-cpu-baseline", "--no-other-configs", "--step", "fused"])
    assert fused["parity_ok"] and set(fused["kernels"]) == {"sum_fused"}
    assert fused["kernels"]["sum_fused"]["bytes_per_launch"] == 2 * 8 * (1 << 24) and "sum_fused" in fused["roofline"]["kernel"]
    n = 1 << 24
    assert fused["result"]["i64_sum"] == n * (n - 1) // 2 and fused["result"]["f64_ulps_from_exact"] <= 1.0


def test_one_process_group_mode_with_rccl():
    """`bench.py --gpus N` started directly = one process over N GPUs through ma_group_* (--force-group takes that path
    with the GPUs this box has: one on the test pool, where ncclCommInitAll runs with one rank)."""
    out = run([sys.executable, "bench.py", *SMALL, "--gpus", "1", "--force-group", "--no-cpu-baseline", "--other-reps", "2"])
    assert out["parity_ok"] and out["config"]["launch"] == "single process"
    # the N > 1 headline is the PARTITIONED column (strong scaling, the default); the same job on GPU 0 alone rides along
    assert out["scaling"] == "strong" and out["config"]["rows_total_per_column"] == 1 << 24
    assert "issue: threads" in out["config"]["parallelism"] and out["config"]["host_issue_us_per_step"] > 0
    assert out["n1_same_process"]["value"] > 0 and 0.2 < out["efficiency_vs_n1"] < 2.0
    piped = out["n1_same_process"]["pipelined"]  # the like-for-like denominator: the same job on one GPU as a pipeline of fused steps
    assert piped["parity_ok"] and piped["value"] > 0 and 0.2 < out["efficiency_vs_n1_pipelined"] < 2.0
    assert out["config"]["rccl_ranks"] == 1 and "RCCL all-gather (ncclCommInitAll" in out["config"]["exchange"]
    # the line explains itself: where the exchange's time goes (HIP events on member 0's exchange stream, every 4th step),
    # what RCCL says its communicator's size is, the members' scan times, and the step's form
    cfg = out["config"]
    assert cfg["exchange_samples"] >= 1 and cfg["exchange_us"] > 0 and cfg["fold_us"] > 0
    assert 0 < cfg["scan_ms_per_step_min_over_members"] <= cfg["scan_ms_per_step_max_over_members"]
    assert "fused" in cfg["step"] and set(out["kernels"]) == {"sum_fused"} and cfg["host"] == "torch-free"
    sep = run([sys.executable, "bench.py", *SMALL, "--gpus", "1", "--force-group", "--no-cpu-baseline", "--no-other-configs",
               "--step", "separate", "--exchange", "host"])  # two launches per member AND the host fold, in one run
    assert sep["parity_ok"] and set(sep["kernels"]) == {"sum_i64", "sum_f64"}
    assert sep["result"]["i64_sum"] == out["result"]["i64_sum"] and sep["result"]["f64_sum"] == out["result"]["f64_sum"]
    assert sep["config"]["rccl_ranks"] == 0 and "host fold" in sep["config"]["exchange"]
    oc = out["other_configs"]  # the multi-GPU legs of configs 4 and 5 (one GPU here)
    assert oc["parity_ok"] is True
    assert oc["config4_i64_sum_10pct_nulls_row_chunks"]["parity"] and oc["config5_supertable_one_batch_per_gpu"]["parity"]
    assert oc["config3_i64_add_one_chunk_per_gpu"]["parity"] is True  # chunk fan-out over the group, no exchange
    assert oc["config5_physical_consolidate_onto_gpu0"]["parity"] is True  # the gather onto one member
    assert oc["config4_i64_sum_10pct_nulls_row_chunks"]["rows_total"] == 1 << 24
    weak = run([sys.executable, "bench.py", *SMALL, "--gpus", "1", "--force-group", "--no-cpu-baseline", "--scaling", "weak",
                "--group-issue", "caller", "--no-other-configs", "--overlap", "on"])
    assert weak["parity_ok"] and weak["scaling"] == "weak" and "issue: caller" in weak["config"]["parallelism"]
    assert "on side streams" in weak["config"]["exchange"]
    assert weak["result"]["i64_sum"] == out["result"]["i64_sum"]


@pytest.mark.parametrize("exchange, extra", [("native", ["--overlap"])])
def test_launcher_mode_one_rank(exchange, extra):
    """One process per GPU under torch.distributed.run: the GPU path is torch-free (gloo carries the rendezvous only) and the
    exchange is the library's own communicator (ma_comm_*) or, as its fall-back, the records over host memory."""
    out = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
               "127.0.0.1", "--master-port", "29641", "bench.py", *SMALL, "--gpus", "1", "--force-dist", "--no-cpu-baseline",
               "--other-reps", "2", "--exchange", exchange, *extra])
    assert out["parity_ok"] and out["n_gpus"] == 1 and out["scaling"] == "strong"
    assert out["n1_same_process"]["value"] > 0 and out["efficiency_vs_n1"] > 0
    assert out["n1_same_process"]["pipelined"]["parity_ok"] and out["efficiency_vs_n1_pipelined"] > 0
    oc = out["other_configs"]  # configs 4 and 5 through the same exchange; config 3 needs none
    assert oc["config3_i64_add_one_chunk_per_gpu"]["parity"] is True
    assert oc["parity_ok"] is True, oc
    assert oc["config4_i64_sum_10pct_nulls_row_chunks"]["parity"] and oc["config5_supertable_one_batch_per_gpu"]["parity"]
    want = {"native": "ma_comm_*", "host": "device fold"}[exchange]
    assert want in out["config"]["exchange"]
    if "--overlap" in extra:
        assert "side stream" in out["config"]["exchange"]
    cfg = out["config"]
    assert "torch-free" in cfg["host"] and "torch" not in cfg["hip_runtime"] and "fused" in cfg["step"] and cfg["rehearsal"] is False
    assert {"exchange_us", "fold_us", "exchange_samples", "rccl_ranks", "scan_ms_per_step_min_over_ranks"} <= set(cfg)
    if exchange == "native":
        assert cfg["rccl_ranks"] == 1 and cfg["exchange_samples"] >= 1 and cfg["exchange_us"] > 0 and cfg["fold_us"] > 0


LAUNCH = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1"]
QUICK = ["--no-cpu-baseline", "--no-other-configs", "--wait-seconds", "0.6"]


def test_launcher_mode_falls_back_when_the_native_communicator_fails_its_check(monkeypatch):
    """bench.py checks the set-up step's finals on every rank before timing anything; a communicator that folds wrongly
    (forced here for every ma_comm form) is abandoned on all ranks, one notch at a time — stamp, event, in-stream — down to the
    records over host memory; the line says so."""
    monkeypatch.setenv("MA_BENCH_DISTRUST_NATIVE_COMM", "1")
    base = [*LAUNCH, "--master-port", "29642", "bench.py", *SMALL, "--gpus", "1", "--force-dist", "--no-cpu-baseline",
            "--no-other-configs", "--overlap", "on", "--scan-lanes", "on"]
    out = run(base)
    cfg = out["config"]
    assert out["parity_ok"] and [d["abandoned"] for d in cfg["downgrades"]] == [
        "ma_comm, overlapped, hand-off by stamp, two scan lanes", "ma_comm, overlapped, hand-off by stamp",
        "ma_comm, overlapped, hand-off by event", "ma_comm, in-stream"]
    assert all("set-up check" in d["why"] for d in cfg["downgrades"]) and "4 abandoned" in cfg["exchange"]
    assert cfg["exchange_form"].startswith("none (one rank)") and cfg["rccl_ranks"] == 0


G = ["rccl, overlapped, hand-off by stamp, two scan lanes, issue threads", "rccl, overlapped, hand-off by stamp, issue threads",
     "rccl, overlapped, hand-off by event, issue threads", "rccl, in-stream, issue threads", "rccl, in-stream, issue caller (grouped)",
     "host fold, issue threads"]  # the one-process ladder, best first
L = ["ma_comm, overlapped, hand-off by stamp, two scan lanes", "ma_comm, overlapped, hand-off by stamp",
     "ma_comm, overlapped, hand-off by event", "ma_comm, in-stream", "none (one rank): device fold on the scan stream"]


@pytest.mark.parametrize("fault, notches_down", [("stall@preflight,stall@setup", 2),
                                                 ("stall@setup,stall@setup,corrupt@setup,stall@timed,stall@preflight", 5)])
def test_group_mode_goes_down_its_ladder_instead_of_hanging(fault, notches_down):
    """`bench.py --gpus N` in one process, on first contact with an exchange that never completes (or folds wrongly): the
    line still comes, rc 0, parity ok, and names the form that ran and what was abandoned on the way (the faults are the
    library's own test hooks, armed through MA_BENCH_FAULT) — all the way down to the host fold."""
    out = run([sys.executable, "bench.py", *SMALL, "--gpus", "1", "--force-group", "--overlap", "on", "--scan-lanes", "on", *QUICK],
              {"MA_BENCH_FAULT": fault}, timeout=240)
    cfg = out["config"]
    assert out["parity_ok"] and cfg["faults_injected"] == fault.split(",")
    assert [d["abandoned"] for d in cfg["downgrades"]] == G[:notches_down] and cfg["exchange_form"] == G[notches_down], cfg["downgrades"]
    if notches_down == 5:
        assert "host fold" in cfg["exchange"] and cfg["rccl_ranks"] == 0
    for d in cfg["downgrades"]:
        assert ("did not finish within" in d["why"]) or ("finals are wrong" in d["why"]) or ("self-test failed" in d["why"]), d
    assert cfg["preflight"]["ok"] and cfg["attempts"] == notches_down + 1


@pytest.mark.parametrize("fault, notches_down", [("corrupt@setup,stall@timed", 2)])
def test_launcher_mode_goes_down_its_ladder_instead_of_hanging(fault, notches_down):
    """The same under torch.distributed.run (one process per GPU, ma_comm_*): the ranks agree over gloo after every bounded
    wait; the communicator is aborted and a new one made from a fresh id for the next form down."""
    out = run([*LAUNCH, "--master-port", "29643", "bench.py", *SMALL, "--gpus", "1", "--force-dist", "--overlap", "on", "--scan-lanes", "on",
               *QUICK], {"MA_BENCH_FAULT": fault}, timeout=240)
    cfg = out["config"]
    assert out["parity_ok"] and cfg["faults_injected"] == fault.split(",")
    assert [d["abandoned"] for d in cfg["downgrades"]] == L[:notches_down] and cfg["exchange_form"] == L[notches_down]
    assert cfg["rccl_ranks"] == 1 and cfg["preflight"]["ok"]


def test_the_two_n_gt_1_modes_measure_alike():
    """One process over the GPUs (ma_group_*) and one process per GPU (ma_comm_*) report the same keys, taken the same way —
    timing marks around the scan inside the timed steps, every 4th exchange sampled — and agree within noise on one GPU."""
    common = ["--rows", str(1 << 26), "--steps", "12", "--warmup", "2", "--no-cpu-baseline", "--no-other-configs", "--overlap", "on",
              "--scan-lanes", "on"]
    g = run([sys.executable, "bench.py", *common, "--gpus", "1", "--force-group"])
    d = run([*LAUNCH, "--master-port", "29644", "bench.py", *common, "--gpus", "1", "--force-dist"])
    shared = {"exchange_us", "fold_us", "exchange_samples", "host_issue_us_per_step", "scan_ms_per_step_min", "scan_ms_per_step_max",
              "rccl_ranks", "exchange_form", "downgrades", "preflight", "attempts", "step", "clock_ramp"}
    assert shared <= set(g["config"]) and shared <= set(d["config"])
    assert g["parity_ok"] and d["parity_ok"] and g["result"] == d["result"]
    assert set(g["kernels"]) == set(d["kernels"]) == {"sum_fused"}
    assert g["kernels"]["sum_fused"]["timed_steps"] == d["kernels"]["sum_fused"]["timed_steps"] >= 3
    for a, b in ((g["kernels"]["sum_fused"]["avg_ms"], d["kernels"]["sum_fused"]["avg_ms"]),
                 (g["config"]["scan_ms_per_step_max"], d["config"]["scan_ms_per_step_max"]), (g["ms_per_step"], d["ms_per_step"])):
        assert 0.8 < a / b < 1.25, (a, b)
    for key in ("exchange_us", "fold_us"):  # a 1-rank all-gather and a 1-record fold: microseconds either way
        assert 0 < g["config"][key] < 500 and 0 < d["config"][key] < 500
    assert g["config"]["downgrades"] == d["config"]["downgrades"] == []
    assert "two scan lanes" in g["config"]["exchange_form"] and "two scan lanes" in d["config"]["exchange_form"]
    assert "two scan lanes" in g["config"]["exchange"] and "two scan contexts" in d["config"]["exchange"]


@pytest.mark.parametrize("mode", ["group", "ranks"])
def test_the_scan_lanes_are_kept_only_if_the_untimed_trial_measures_them_faster(mode):
    """--scan-lanes auto (the default at N > 1): what two scan lanes gain depends on how the runtime mapped this process's streams
    onto hardware queues, so the run measures a few un-timed steps each way in front of the warm-up and keeps the lanes only if
    they are faster here; otherwise it goes one notch down and says why. Either way the line carries the trial's figures."""
    args = ["--rows", str(1 << 26), "--steps", "8", "--warmup", "1", "--no-cpu-baseline", "--no-other-configs", "--overlap", "on"]
    if mode == "group":
        out = run([sys.executable, "bench.py", *args, "--gpus", "1", "--force-group"])
    else:
        out = run([*LAUNCH, "--master-port", "29645", "bench.py", *args, "--gpus", "1", "--force-dist"])
    cfg = out["config"]
    trial = cfg["scan_lanes_trial"]
    assert out["parity_ok"] and trial["two_scan_lanes_ms_per_step"] > 0 and trial["one_scan_stream_ms_per_step"] > 0 and 1 <= trial["tries"] <= 3
    kept = trial["two_scan_lanes_ms_per_step"] <= trial["one_scan_stream_ms_per_step"] * 0.993
    assert ("two scan lanes" in cfg["exchange_form"]) == kept
    if kept:
        assert cfg["downgrades"] == []
    else:
        assert len(cfg["downgrades"]) == 1 and "measured no faster" in cfg["downgrades"][0]["why"]


@pytest.mark.parametrize("ranks", [3])
def test_launcher_rehearsal_partitions_the_column_over_several_ranks(ranks):
    """The N > 1 headline's own code path with more than one rank on a one-GPU box: `--backend gloo` lets the ranks share
    the visible GPU (RCCL refuses that) and carries the 64-byte records over host memory — never a reported number, but
    the strong-scaling partition (64-row-aligned row chunks of ONE 2^24 + 5-row column pair, ragged last chunk), the
    per-rank generation at the chunk's offset, the rank-ordered fold and the same-process N = 1 leg all run for real."""
    rows = (1 << 24) + 5
    out = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
               "127.0.0.1", "--master-port", str(29650 + ranks), "bench.py", "--rows", str(rows), "--steps", "3", "--warmup", "1",
               "--gpus", str(ranks), "--backend", "gloo", "--no-cpu-baseline"])
    assert out["parity_ok"] and out["n_gpus"] == ranks and out["scaling"] == "strong"
    assert out["config"]["rows_total_per_column"] == rows and out["result"]["rows"] == rows
    assert out["config"]["rows_per_gpu_per_column"] == ((rows // ranks) // 64) * 64  # rank 0's 64-row-aligned chunk
    assert out["result"]["i64_sum"] == rows * (rows - 1) // 2 and out["result"]["f64_ulps_from_exact"] <= 1.0
    assert "REHEARSAL" in out["config"]["parallelism"] and out["config"]["rccl_ranks"] == 0
    assert out["n1_same_process"]["rows_per_column"] == rows and out["efficiency_vs_n1"] > 0



# ---- REHEARSAL: both N > 1 modes with PEERS, on this box's one GPU, through the loopback collective double ---------------------
# (tests/test_gpu_rehearsal.py starts the session these run in: MINARROW_HIP_RCCL_PATH, GPU_MAX_HW_QUEUES are in the environment)


@pytest.mark.rehearsal
@pytest.mark.parametrize("members", [8])
def test_rehearsal_group_mode_with_eight_members_keeps_the_top_notch(members):
    """`bench.py --gpus 8` as the driver starts it on an 8-GPU node, here with the eight members on one GPU: ncclCommInitAll over
    8 ranks, eight issue threads, the overlapped exchange waiting on the scans' stamps, two scan lanes per member — the line
    says REHEARSAL, nothing was abandoned on the way, every member holds the same finals."""
    out = run([sys.executable, "bench.py", "--rows", str(1 << 26), "--steps", "6", "--warmup", "2", "--gpus", str(members), "--no-cpu-baseline",
               "--other-rows", str(1 << 22), "--other-reps", "2", "--scan-lanes", "on"], timeout=400)
    cfg = out["config"]
    assert out["parity_ok"] and out["n_gpus"] == members and cfg["rehearsal"] is True
    assert "REHEARSAL" in cfg["parallelism"] and cfg["exchange"].startswith("REHEARSAL") and "loopback" in cfg["exchange"]
    assert cfg["downgrades"] == [] and cfg["exchange_form"] == G[0] and cfg["attempts"] == 1, cfg["downgrades"]
    assert cfg["rccl_ranks"] == members and cfg["preflight"]["ok"] and cfg["preflight"]["members"] == members
    assert cfg["rows_per_gpu_per_column"] == (1 << 26) // members
    oc = out["other_configs"]  # configs 3-5's multi-GPU legs: one chunk / one batch per member, ONE exchange per step
    assert oc["parity_ok"] is True, oc
    assert oc["config4_i64_sum_10pct_nulls_row_chunks"]["parity"] and oc["config5_supertable_one_batch_per_gpu"]["parity"]
    assert oc["config5_physical_consolidate_onto_gpu0"]["parity"] is True


@pytest.mark.rehearsal
def test_rehearsal_group_mode_goes_down_its_ladder_with_peers():
    """A stalled LAST member of four during the set-up step and wrong finals on it one notch further down: two forms abandoned,
    the third does the job — with peers whose collectives really were waiting."""
    out = run([sys.executable, "bench.py", *SMALL, "--gpus", "4", "--scan-lanes", "on", *QUICK], {"MA_BENCH_FAULT": "stall@setup,corrupt@setup"},
              timeout=300)
    cfg = out["config"]
    assert out["parity_ok"] and cfg["rehearsal"] is True and cfg["faults_injected"] == ["stall@setup", "corrupt@setup"]
    assert [d["abandoned"] for d in cfg["downgrades"]] == G[:2] and cfg["exchange_form"] == G[2] and cfg["rccl_ranks"] == 4
    assert "member 3" in cfg["downgrades"][0]["why"] and "finals are wrong" in cfg["downgrades"][1]["why"]


@pytest.mark.rehearsal
@pytest.mark.parametrize("ranks", [2])
def test_rehearsal_launcher_mode_with_two_rank_processes_keeps_the_top_notch(ranks):
    """The driver's N > 1 command — torch.distributed.run, one process per rank, ma_comm_* — with two rank processes on one GPU:
    ncclCommInitRank across processes, the overlapped exchange on the scans' stamps, two scan contexts per rank."""
    out = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
               "--master-port", "29671", "bench.py", "--rows", str(1 << 26), "--steps", "6", "--warmup", "2", "--gpus", str(ranks),
               "--no-cpu-baseline", "--other-rows", str(1 << 22), "--other-reps", "2", "--scan-lanes", "on"], {"GPU_MAX_HW_QUEUES": "4"}, timeout=400)
    cfg = out["config"]
    assert out["parity_ok"] and out["n_gpus"] == ranks and cfg["rehearsal"] is True
    assert "REHEARSAL" in cfg["parallelism"] and cfg["exchange"].startswith("REHEARSAL")
    assert cfg["downgrades"] == [] and cfg["exchange_form"] == L[0] and cfg["rccl_ranks"] == ranks and cfg["preflight"]["ok"]
    assert out["other_configs"]["parity_ok"] is True, out["other_configs"]


@pytest.mark.rehearsal
def test_rehearsal_launcher_mode_goes_down_its_ladder_with_a_peer():
    out = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29672", "bench.py", *SMALL, "--gpus", "2", "--scan-lanes", "on", *QUICK],
              {"MA_BENCH_FAULT": "stall@setup,corrupt@setup", "GPU_MAX_HW_QUEUES": "4"}, timeout=300)
    cfg = out["config"]
    assert out["parity_ok"] and cfg["rehearsal"] is True
    assert [d["abandoned"] for d in cfg["downgrades"]] == L[:2] and cfg["exchange_form"] == L[2] and cfg["rccl_ranks"] == 2
