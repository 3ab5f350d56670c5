#!/usr/bin/env python3
"""Soak of the LDS-staged recurrence kernel against the oracle: many panel widths x dtypes x orders on
Hilbert-ordered sensor graphs of several sizes, repeated; the pipelined host-pointer path against its one-shot
form (threads + three streams: looks for hand-off races); plus the randomised differential test of
tests/test_gpu_9_fuzz.py on extra seeds.  Looks for rare events (races between waves / threads), not for speed.
GPU box only.  usage: tools/soak.py [rounds]      (tests/test_gpu_9_fuzz.py runs a two-minute slice of it)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import cheby_oracle as orc  # noqa: E402
from pygsp_amd import engine, filters, graphs  # noqa: E402

FULL = dict(sizes=(40000, 200000, 700000), widths=(1, 2, 3, 4, 5, 6, 8, 12, 16, 20, 32, 36, 64, 72, 96, 128, 160),
            orders=(1, 2, 3, 5, 11, 30))
SLICE = dict(sizes=(40000, 200000), widths=(2, 5, 8, 16, 64, 128), orders=(2, 5, 30))  # (2- / 4- / 8- / 16-lane builds, a padded one)


def soak_kernel(rounds=3, sizes=FULL["sizes"], widths=FULL["widths"], orders=FULL["orders"], log=print):
    """k_step_tile (device-resident work: host_pipeline off) against oracle columns.  Returns (runs, bad)."""
    ctx = engine.default_context(0)
    ctx.set_option("host_pipeline", 0)
    rng = np.random.default_rng(0)
    bad = runs = 0
    try:
        for N in sizes:
            W, coords = graphs.sensor_weights(N, k=8, seed=N)
            L = orc.laplacian(W)
            lmax = 2.0 * float(np.ravel(W.sum(0)).max())
            for dtype in (np.float64, np.float32):
                dev = engine.DeviceGraph.from_w(W, dtype=dtype, perm=engine.locality_order(W, coords), ctx=ctx)
                assert dev.build_gather_tiles()["slow_blocks"] == 0
                tol = 1e-11 if dtype == np.float64 else 3e-5
                for nsig in widths:
                    x = rng.standard_normal((N, nsig))
                    for order in orders:
                        c = orc.compute_cheby_coeff(orc.heat_kernel(10, lmax), lmax, order)
                        cols = [0, nsig // 2, nsig - 1]
                        ref = orc.cheby_op(L, lmax, c, x[:, cols].astype(dtype).astype(np.float64))
                        for _ in range(rounds):
                            y, _ms = dev.cheby_filter(c, x, lmax)
                            err = float(np.max(np.abs(y[0][:, cols] - ref)) / np.max(np.abs(ref)))
                            runs += 1
                            if not err < tol:
                                bad += 1
                                log("BAD", N, np.dtype(dtype).name, nsig, order, err)
                dev.destroy()
            log("N", N, "done; runs", runs, "bad", bad)
    finally:
        ctx.set_option("host_pipeline", 1)
    return runs, bad


def soak_host_pipeline(calls=60, N=150000, log=print):
    """The pipelined gspx_cheby_filter (pack threads | H2D | kernels | D2H | unpack threads) against the one-shot
    form of the same call, bit for bit, over random batch widths / thread counts / panel widths / modes."""
    ctx = engine.default_context(0)
    rng = np.random.default_rng(1)
    W, coords = graphs.sensor_weights(N, k=8, seed=7)
    lmax = 2.0 * float(np.ravel(W.sum(0)).max())
    bad = runs = 0
    try:
        for dtype in (np.float64, np.float32):
            dev = engine.DeviceGraph.from_w(W, dtype=dtype, perm=engine.locality_order(W, coords), ctx=ctx)
            dev.build_gather_tiles()
            wmin = 32 // np.dtype(dtype).itemsize  # batches of at least 32-byte rows: same kernel family
            for _ in range(calls // 2):
                nf = int(rng.choice([1, 1, 3]))
                synth = nf > 1 and rng.random() < 0.4
                nsig = int(rng.integers(2, 9)) * 8
                order = int(rng.choice([3, 9, 20]))
                kern = [orc.heat_kernel(10, lmax)] if nf == 1 else orc.mexican_hat_kernels(lmax, nf)
                c = np.stack([orc.compute_cheby_coeff(k, lmax, order) for k in kern])
                x = rng.standard_normal((nf, N, nsig) if synth else (N, nsig)).astype(dtype)
                mode = 1 if synth else 0
                ctx.set_option("host_pipeline", 0)
                y0, _ = dev.cheby_filter(c, x, lmax, mode)
                ctx.set_option("host_pipeline", 2)
                ctx.set_option("host_batch", int(rng.choice([wmin, 8, 16, 24])))
                ctx.set_option("host_threads", int(rng.integers(1, 9)))
                y1, _ = dev.cheby_filter(c, x, lmax, mode)
                runs += 1
                if not np.array_equal(y0, y1):
                    bad += 1
                    log("PIPE BAD", np.dtype(dtype).name, nf, synth, nsig, order, float(np.max(np.abs(y0 - y1))))
            dev.destroy()
    finally:
        for k, v in (("host_pipeline", 1), ("host_batch", 0), ("host_threads", 0)):
            ctx.set_option(k, v)
    return runs, bad


def soak_programs(rounds=3, sizes=(40000, 200000), widths=(1, 3, 8, 16, 20, 64, 96), orders=(5, 30, 60), log=print):
    """The polynomial programs of evaluation='auto' (round 6: Newton form with o_s = x, product form written in place)
    on host arrays - one shot and pipelined - against oracle columns, and call against call bit for bit."""
    ctx = engine.default_context(0)
    rng = np.random.default_rng(2)
    bad = runs = 0
    try:
        for N in sizes:
            W, coords = graphs.sensor_weights(N, k=8, seed=N + 1)
            L = orc.laplacian(W)
            lmax = 2.0 * float(np.ravel(W.sum(0)).max())
            for dtype in (np.float64, np.float32):
                dev = engine.DeviceGraph.from_w(W, dtype=dtype, perm=engine.locality_order(W, coords), ctx=ctx)
                dev.build_gather_tiles()
                tol = 1e-10 if dtype == np.float64 else 1e-4
                for order in orders:
                    for scale in (10, 50):
                        c = orc.compute_cheby_coeff(orc.heat_kernel(scale, lmax), lmax, order)
                        forms = {}
                        if filters.product_guard(c, dtype)[0]:
                            forms["product"] = (filters.cheb_to_product(c, dtype), False)
                        ce = filters.effective_coefficients(c, dtype)
                        if filters.newton_guard(ce, dtype)[0]:
                            forms["newton"] = (filters.newton_program(*filters.cheb_to_newton(ce)), True)
                        for nsig in widths:
                            x = rng.standard_normal((N, nsig)).astype(dtype)
                            cols = [0, nsig // 2, nsig - 1]
                            ref = orc.cheby_op(L, lmax, c, x[:, cols].astype(np.float64))
                            for name, (prog, old_is_x) in forms.items():
                                first = None
                                for r in range(rounds):
                                    ctx.set_option("host_pipeline", 2 if r % 2 else 0)
                                    ctx.set_option("host_batch", int(rng.choice([8, 16, 24])) if r % 2 else 0)
                                    y, _ms = dev.program_filter(prog, x, lmax, old_is_x)
                                    err = float(np.max(np.abs(y[:, cols] - ref)) / np.max(np.abs(ref)))
                                    runs += 1
                                    if not err < tol:
                                        bad += 1
                                        log("PROGRAM BAD", name, N, np.dtype(dtype).name, nsig, order, scale, err)
                                    if r % 2 == 0:  # (batches of other widths run other builds: equal to rounding only)
                                        if first is None:
                                            first = y
                                        elif not np.array_equal(first, y):
                                            bad += 1
                                            log("PROGRAM NOT REPEATABLE", name, N, np.dtype(dtype).name, nsig, order)
                dev.destroy()
            log("programs N", N, "done; runs", runs, "bad", bad)
    finally:
        for k, v in (("host_pipeline", 1), ("host_batch", 0)):
            ctx.set_option(k, v)
    return runs, bad


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    log = lambda *a: print(*a, flush=True)  # noqa: E731
    runs, bad = soak_kernel(rounds, log=log)
    r2, b2 = soak_host_pipeline(200, log=log)
    r3, b3 = soak_programs(rounds, log=log)
    runs, bad = runs + r3, bad + b3
    import test_gpu_9_fuzz as fuzz
    for seed in range(4, 16):
        try:
            fuzz.test_fuzz_against_oracle(seed)
        except AssertionError as e:
            bad += 1
            log("FUZZ BAD seed", seed, str(e)[:300])
    log("soak finished: runs", runs, "+", r2, "pipelined calls + 12 fuzz seeds, bad", bad + b2)
    sys.exit(1 if bad + b2 else 0)


if __name__ == "__main__":
    main()
