#!/usr/bin/env python3
"""Measured accuracy of the device's Power series against tests/golden/pow_series_kat.npz (the numbers quoted in
tests/test_gpu_pow_series.py and ma_binary.hpp):  python tools/pow_series_report.py > profiles/r03_pow_series_accuracy.json"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from minarrow_amd import ffi  # noqa: E402
from minarrow_amd.host import Context  # noqa: E402

z = np.load(ROOT / "tests" / "golden" / "pow_series_kat.npz")
ctx = Context(0)


def series(which, x):
    x = np.ascontiguousarray(x)
    out = np.empty(x.size, dtype=np.float64)
    ffi.check(ctx.lib.ma_test_pow_series(ctx.handle, which, x.ctypes.data, out.ctypes.data, x.size))
    return out


def apply(tag, a, b):
    dt = a.dtype
    da, db, do = ctx.to_device(a, 64), ctx.to_device(b, 64), ctx.alloc(a.nbytes + 64)
    ctx.apply(tag, da, db, 5, do, a.size, b.size)
    return do.download(dt, a.size)


res = {}
x, hi, lo = z["ln64_x"], z["ln64_hi"], z["ln64_lo"].astype(np.float64)
got = series(0, x)
nz = hi != 0
err = np.abs((got - hi) - lo)[nz] / np.spacing(np.abs(hi[nz]))
host = np.abs((np.log(x) - hi) - lo)[nz] / np.spacing(np.abs(hi[nz]))
res["pow_f64_ln"] = {"samples": int(x.size), "max_ulp": float(err.max()), "mean_ulp": float(err.mean()),
                     "correctly_rounded_frac": float(np.mean(got == hi)), "worst_input": float(x[nz][err.argmax()]),
                     "host_glibc_log": {"max_ulp": float(host.max()), "mean_ulp": float(host.mean()),
                                        "correctly_rounded_frac": float(np.mean(np.log(x) == hi))}}
for name, which, xs, want in (("pow_f32_ln", 1, z["ln32_x"], z["ln32_want"]), ("pow_f32_exp", 2, z["exp32_y"], z["exp32_want"])):
    g64 = series(which, xs)
    with np.errstate(over="ignore"):
        g = g64.astype(np.float32)
    fin = np.isfinite(want) & (want != 0)
    u = np.abs(g[fin].astype(np.float64) - want[fin].astype(np.float64)) / np.spacing(np.abs(want[fin])).astype(np.float64)
    normal = fin & (np.abs(want) >= np.finfo(np.float32).tiny)
    rel = np.abs(g64[normal] - want[normal].astype(np.float64)) / np.abs(want[normal].astype(np.float64))
    res[name] = {"samples": int(xs.size), "rounded_equals_correctly_rounded_f32_frac": float(np.mean(u == 0)),
                 "max_ulp_f32": float(u.max()), "max_rel_err_of_f64_value_vs_rn32_log2": float(np.log2(rel.max()))}
for tag in ("32", "64"):
    a, b, want = z[f"pow{tag}_a"], z[f"pow{tag}_b"], z[f"pow{tag}_want"]
    g = apply("f" + tag, a, b)
    fin = np.isfinite(want) & (want != 0)
    xx = np.abs(b.astype(np.float64) * np.log(a.astype(np.float64)))
    with np.errstate(invalid="ignore"):
        u = np.abs(g.astype(np.float64) - want.astype(np.float64)) / np.spacing(np.abs(want)).astype(np.float64)
    res[f"power_f{tag}_vs_exact_three_rounding"] = {
        "samples": int(a.size), "identical_frac": float(np.mean(u[fin] == 0)), "max_ulp": float(u[fin].max()),
        "max_of_ulp_minus_|b ln a|": float((u[fin] - xx[fin]).max()), "max_of_ulp_over_(1+|b ln a|)": float((u[fin] / (1 + xx[fin])).max()),
        "non_finite_or_zero_identical": bool(np.array_equal(g[~fin], want[~fin]))}
print(json.dumps(res, indent=1))
